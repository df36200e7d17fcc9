"""GPU box: the 14B/720p self-attention call (40 heads x 75 600 tokens) with and without a workspace, alternating,
i.e. the last head's sparse last round as 64-row workgroups against its key split.  usage: python tools/attn_tail_ab.py [heads] [S]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from lightx2v_amd import hip

H = int(sys.argv[1]) if len(sys.argv) > 1 else 40
S = int(sys.argv[2]) if len(sys.argv) > 2 else 75600
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(S, 3 * H * 128, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
q, k, v = (qkv[:, i * H * 128:(i + 1) * H * 128].unflatten(1, (H, 128)) for i in range(3))
need = hip.attn_workspace_bytes(S, S, H)
ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
print(f"workspace {need / 1e6:.1f} MB")


def timed(w, iters=5):
    hip.attn_fwd(q, k, v, workspace=w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hip.attn_fwd(q, k, v, workspace=w)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for r in range(4):
    a, b = timed(None), timed(ws)
    fl = 4 * S * S * H * 128 / 1e9
    print(f"round {r}: without workspace {a:.3f} ms ({fl / a:.0f} TFLOP/s), with {b:.3f} ms ({fl / b:.0f} TFLOP/s)", flush=True)
