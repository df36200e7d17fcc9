"""GPU box: the W8A8 GEMM at the 14B/720p Linear shapes with and without a workspace (K split of a sparsely filled
last round of tiles), alternating.  usage: python tools/gemm8_tail_ab.py [int8]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from lightx2v_amd import hip

int8 = len(sys.argv) > 1 and sys.argv[1] == "int8"
dev = torch.device("cuda:0")
S = 75600
shapes = [("qkv", S, 15360, 5120, hip.EPI_NONE), ("o", S, 5120, 5120, hip.EPI_RESIDUAL), ("ffn0", S, 13824, 5120, hip.EPI_GELU_TANH), ("ffn2", S, 5120, 13824, hip.EPI_RESIDUAL)]
g = torch.Generator(device=dev).manual_seed(0)
quant = hip.quant_int8_rowwise if int8 else hip.quant_fp8_rowwise
gemm = hip.gemm_int8 if int8 else hip.gemm_fp8


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    aq, a_s = quant(a)
    wq, w_s = quant(w)
    del a, w
    b = torch.zeros(N, device=dev)
    out = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
    kw = dict(epilogue=epi, resid=out if epi == hip.EPI_RESIDUAL else None, out=out)
    ws = hip.gemm_w8a8_workspace_for(aq, N, K)
    fl = 2.0 * M * N * K
    line = f"{name} [{M},{K}]x[{N},{K}] epi {epi}: "
    for rep in range(2):
        t0 = timed(lambda: gemm(aq, a_s, wq, w_s, b, **kw))
        line += f"plain {t0:.3f} ms = {fl / t0 / 1e9:.0f} TF  "
        if ws is not None:
            t1 = timed(lambda: gemm(aq, a_s, wq, w_s, b, workspace=ws, **kw))
            line += f"K split {t1:.3f} ms = {fl / t1 / 1e9:.0f} TF  "
    print(line, flush=True)
    del aq, wq, out
