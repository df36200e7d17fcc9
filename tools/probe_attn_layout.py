#!/usr/bin/env python3
"""Self-attention launch time against the token stride of q / k / v and the heads per launch
(14B at 720p: 75 600 tokens, 40 heads of 128).  Prints ms per 40 heads for each variant, variants
alternating on one box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import hip  # noqa: E402

S, H, d = int(os.environ.get("S", 75600)), 40, 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(S, 3 * H * d, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
out = torch.empty(S, H * d, device=dev, dtype=torch.bfloat16)


def cols(t, which, h0, h1, heads_total):
    D = heads_total * d
    return t[:, which * D + h0 * d: which * D + h1 * d].view(S, h1 - h0, d)


def split_launch(n):
    """n launches of H/n heads each on column ranges of the fused [S, 3D] buffer."""
    w = H // n
    for i in range(n):
        hip.attn_fwd(cols(qkv, 0, i * w, (i + 1) * w, H), cols(qkv, 1, i * w, (i + 1) * w, H), cols(qkv, 2, i * w, (i + 1) * w, H),
                     out=None)


compact = {}
for n in (1, 2, 4):
    w = H // n
    bufs = []
    for i in range(n):
        b = torch.empty(S, 3, w * d, device=dev, dtype=torch.bfloat16)
        for which in range(3):
            b[:, which] = qkv[:, which * H * d + i * w * d: which * H * d + (i + 1) * w * d]
        bufs.append(b)
    compact[n] = bufs
sep = [qkv[:, i * H * d:(i + 1) * H * d].contiguous() for i in range(3)]


def compact_launch(n):
    w = H // n
    for b in compact[n]:
        r = b.view(S, 3, w, d)
        hip.attn_fwd(r[:, 0], r[:, 1], r[:, 2])


variants = {
    "fused [S,3D] 1x40 heads": lambda: hip.attn_fwd(cols(qkv, 0, 0, H, H), cols(qkv, 1, 0, H, H), cols(qkv, 2, 0, H, H)),
    "fused [S,3D] 2x20 heads": lambda: split_launch(2),
    "fused [S,3D] 4x10 heads": lambda: split_launch(4),
    "separate q,k,v [S,D] 1x40": lambda: hip.attn_fwd(sep[0].view(S, H, d), sep[1].view(S, H, d), sep[2].view(S, H, d)),
    "compact [S,3,20*128] 2x20": lambda: compact_launch(2),
    "compact [S,3,10*128] 4x10": lambda: compact_launch(4),
}
for f in variants.values():
    f()
torch.cuda.synchronize()
res = {k: [] for k in variants}
for rep in range(int(os.environ.get("REPS", 4))):
    for name, f in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        f()
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 2)
for name, v in res.items():
    print(f"{name:32s} " + " ".join(f"{x:7.2f}" for x in v) + f"   min {min(v):.2f} ms = {4.0 * S * S * H * d / min(v) / 1e9:.0f} TFLOP/s", flush=True)


# ---- launches with few heads (head chunks of a sequence-parallel rank): one launch against the two-launch plan ----
for heads in (2, 3, 5, 10, 20, 40):
    qs, ks, vs = (cols(qkv, i, 0, heads, H) for i in range(3))
    line = f"{heads:2d} heads:"
    for tail in (0, 1):
        hip.set_option("attn_tail", tail)
        hip.attn_fwd(qs, ks, vs)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.attn_fwd(qs, ks, vs)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        line += f"   attn_tail={tail}: {min(ts):7.3f} ms ({4.0 * S * S * heads * d / min(ts) / 1e9:5.0f} TFLOP/s)"
    print(line, flush=True)
hip.set_option("attn_tail", 1)
