"""Where does a two-rank exchange over gloo stall when both ranks share ONE GPU?

Runs scenarios, each in a fresh pair of processes under a deadline, and stops at the first one that
stalls (stack dumps of both ranks are left in the log directory):

  H  all_to_all_single on HOST tensors, one at a time (the GPU is not involved)
  D  the product's exchange (HipUlyssesAttnWeight._exchange_attend with the HIP pack and attention
     kernels, chunks = 2, several asynchronous collectives outstanding) over the host-staged
     transport (lightx2v_amd/comm.py), repeated
  E  D with the fp8 payload (twice as many collectives in flight)
  A  all_to_all_single on whole CUDA tensors handed to gloo directly, one at a time, waited for at
     once -- nothing of this package is involved.  Measured: both ranks park inside the 9th call.
  B  two to four asynchronous all_to_all_single on CUDA tensors outstanding, outputs are offset views
  C  B with GPU work queued on the stream before every collective

Default order H D E A: the scenario known to stall runs last.

usage: python tools/probe_gloo_shared_gpu.py [--iters N] [--deadline SECONDS] [--out DIR] [--only HDEA]
"""
import argparse
import faulthandler
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, scenario, iters, deadline, out_dir):
    log = open(os.path.join(out_dir, f"{scenario}_rank{rank}.log"), "w")
    faulthandler.dump_traceback_later(max(5, deadline - 5), exit=True, file=log)

    def say(msg):
        log.write(f"{time.time():.3f} {msg}\n")
        log.flush()

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(rank)
    n = 1 << 20
    if scenario == "H":
        send = torch.randn(world, n // world)
        for it in range(iters):
            out = torch.empty(world, n // world)
            dist.all_to_all_single(out, send)
            say(f"it {it} done")
    elif scenario in "ABC":
        send = torch.randn(4, world, n // world, generator=g, device=dev)
        recv = torch.empty(4, world * (n // world) + 64, device=dev)
        a = torch.randn(2048, 2048, device=dev)
        for it in range(iters):
            if scenario == "A":
                for c in range(2):
                    out = torch.empty(world, n // world, device=dev)
                    dist.all_to_all_single(out, send[c].contiguous())
            else:
                if scenario == "C":
                    for _ in range(20):
                        a = (a @ a).clamp_(-1, 1)
                works = [dist.all_to_all_single(recv[c, :n].view(world, n // world), send[c], async_op=True) for c in range(2 + it % 3)]
                for i, w in enumerate(works):
                    say(f"it {it} wait {i}")
                    w.wait()
                torch.cuda.synchronize()
            say(f"it {it} done")
    else:
        import lightx2v_amd as L
        from lightx2v_amd import comm

        comm.set_transport(comm.HostStagedTransport())
        os.environ["LX2V_ULYSSES_CHUNKS"] = "2"
        uly = L.ATTN_WEIGHT_REGISTER["hip_ulysses"]()
        attn = L.ATTN_WEIGHT_REGISTER["hip_flash_attn"]()
        sl, heads = 1024, 8
        for it in range(iters):
            q, k, v = (torch.randn(sl, heads, 128, generator=g, device=dev).to(torch.bfloat16) for _ in range(3))
            say(f"it {it} exchange")
            op, _ = uly._exchange_attend(q, k, v, attn, None, fp8=(scenario == "E"))
            torch.cuda.synchronize()
            say(f"it {it} done {float(op.flat.float().abs().mean()):.4f}")
    faulthandler.cancel_dump_traceback_later()
    dist.barrier()
    dist.destroy_process_group()
    say("finished")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--deadline", type=int, default=90)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "probe_gloo"))
    ap.add_argument("--only", default="HDEA")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    world = 2
    for scenario in args.only:
        t0 = time.time()
        ctx = mp.spawn(_worker, args=(world, _free_port(), scenario, args.iters, args.deadline, args.out), nprocs=world, join=False)
        ok = False
        while time.time() - t0 < args.deadline + 10:
            try:
                if ctx.join(timeout=1.0):
                    ok = True
                    break
            except Exception as e:  # a worker exited non-zero (the watchdog's exit included)
                print(f"scenario {scenario}: worker failed: {str(e).splitlines()[0]}", flush=True)
                break
        for pr in ctx.processes:
            if pr.is_alive():
                pr.kill()
        print(f"scenario {scenario}: {'completed' if ok else 'STALLED or failed'} after {time.time() - t0:.1f} s ({args.iters} iterations)", flush=True)
        if not ok:
            print(f"stopping; see {args.out}/{scenario}_rank*.log", flush=True)
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
