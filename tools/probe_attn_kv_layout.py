"""GPU box: does the K/V row stride matter to the self-attention kernel?  8 heads x 75 600 tokens with K/V rows 256 B apart
(a per-head contiguous image would look like this for one head) against rows 30 KiB apart (the fused q|k|v buffer of the 14B
model).  Heads are laid out the way the kernel expects (contiguous 128-channel blocks inside a token row)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lightx2v_amd import hip

dev = torch.device("cuda:0")
S, H = 75600, 8
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for width in (H * 128, 3 * 5120, 3 * H * 128):
    # q, k, v as column blocks of three separate [S, width] buffers (row stride = width elements)
    bufs = [torch.randn(S, width, generator=g, device=dev).to(torch.bfloat16) for _ in range(3)]
    q, k, v = (b[:, :H * 128].unflatten(1, (H, 128)) for b in bufs)
    for r in range(2):
        ms = timed(lambda: hip.attn_fwd(q, k, v))
        print(f"row stride {width * 2} B: {ms:.3f} ms = {4 * S * S * H * 128 / ms / 1e9:.0f} TFLOP/s", flush=True)
    del bufs, q, k, v
