"""Functional rehearsal of bench.py's N>1 paths on ONE GPU: N ranks share GPU 0 and talk through gloo.
Numbers printed by it are meaningless; it only proves the multi-rank code path runs and stays finite.
usage: python tools/rehearse_bench.py <cfg_p>x<seq_p> [extra bench.py args]"""
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, layout, extra):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": "29541", "LX2V_SHARE_GPU": "1", "LX2V_DIST_BACKEND": "gloo"})
    os.environ.setdefault("LX2V_BENCH_WATCHDOG", "150")
    sys.path.insert(0, ROOT)
    sys.argv = ["bench.py", "--gpus", str(world), "--layout", layout] + extra
    import bench

    bench.main()


if __name__ == "__main__":
    layout = sys.argv[1]
    a, b = (int(v) for v in layout.split("x"))
    mp.spawn(_rank, args=(a * b, layout, sys.argv[2:]), nprocs=a * b, join=True)
