"""Time of the fp8 attention (exp formulation / exp-free P) against the spread of the scores: the rare path of either
re-bases a wave's rows when a tile's scores run away from the running maximum, which N(0, 1) operands almost never do.
   python tools/probe_fp8_attn_spread.py [S] [H]"""
import sys

import torch

sys.path.insert(0, ".")
from lightx2v_amd import hip  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 75600
H = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
base = [torch.randn(S, H, 128, generator=g, device=dev) for _ in range(3)]
for spread in (1.0, 2.0, 3.0, 4.0):
    q, k, v = ((t * (spread if i < 2 else 1.0)).to(torch.bfloat16) for i, t in enumerate(base))
    line = f"spread {spread}: score std {spread * spread:.1f} nat;"
    for mode in (0, 1):
        hip.set_option("attn_fp8_p", mode)
        out = hip.attn_fwd_fp8(q, k, v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            out = hip.attn_fwd_fp8(q, k, v)
        e1.record()
        torch.cuda.synchronize()
        line += f"  {'exp' if mode == 0 else 'exp-free'} {e0.elapsed_time(e1) / 3:.2f} ms (finite: {bool(torch.isfinite(out).all())})"
    hip.set_option("attn_fp8_p", 0)
    print(line, flush=True)
