"""GPU box: this library's bf16 GEMM against torch.matmul (hipBLASLt / rocBLAS) at the 14B/720p Linear
shapes, HIP events, alternating, to know the practical ceiling on the box at hand.
usage: python tools/gemm_vs_blas.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from lightx2v_amd import hip

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
S = 75600
shapes = [("qkv   [S,5120]x[15360,5120]", S, 15360, 5120), ("o     [S,5120]x[5120,5120]", S, 5120, 5120),
          ("ffn0  [S,5120]x[13824,5120]", S, 13824, 5120), ("ffn2  [S,13824]x[5120,13824]", S, 5120, 13824)]
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, M, N, K in shapes:
    a = (torch.randn(M, K, generator=g, device=dev) * 1.0).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    res = {}
    for rep in range(2):
        res.setdefault("lx2v", []).append(timed(lambda: hip.gemm(a, w, b, out=out)))
        ws = hip.gemm_workspace_for(a, N, K)   # None where the last round of tiles is not worth a K split
        if ws is not None:
            res.setdefault("lx2v + K split of the last round", []).append(timed(lambda: hip.gemm(a, w, b, out=out, workspace=ws)))
        res.setdefault("blas", []).append(timed(lambda: torch.matmul(a, w.t(), out=out)))
    line = f"{name}: "
    for k, v in res.items():
        ms = min(v)
        line += f"{k} {ms:.3f} ms = {fl / ms / 1e9:.0f} TFLOP/s   "
    print(line, flush=True)
    del a, w, out
