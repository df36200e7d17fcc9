"""Generate tests/golden/int8_weight_quant.npz by running the REFERENCE's own per-channel int8 weight
quantiser (lightx2v/utils/quant_utils.py IntegerQuantizer, as mm_weight.py:322-330 calls it) on seeded
bf16-valued weights, and assert that oracle/wan_block_ref.quant_int8_weight reproduces it bit for bit.

Usage: python oracle/gen_golden_int8.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import gen_golden as G  # noqa: E402
from oracle import wan_block_ref as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    a = ap.parse_args()
    G._install_reference(a.ref)
    from lightx2v.utils.quant_utils import IntegerQuantizer

    g = torch.Generator().manual_seed(31)
    w = (0.02 * torch.randn(96, 256, generator=g)).to(torch.bfloat16)
    w[3] = 0                                   # an all-zero output channel: the scale floor
    w[5] *= 1e-4                               # a channel below the floor's neighbourhood
    w[7, 11] = 3.0                             # an outlier: everything else rounds to small integers
    w[9] = torch.linspace(-1, 1, 256).to(torch.bfloat16)   # exact halves after scaling are likely: round half to even
    q, scales, zeros = IntegerQuantizer(8, True, "per_channel").real_quant_tensor(w.to(torch.float32))
    assert zeros is None
    q, scales = q.to(torch.int8), scales.to(torch.float32).reshape(-1)
    mq, ms = R.quant_int8_weight(w)
    assert torch.equal(mq, q) and torch.equal(ms, scales)
    # fp32 checkpoint weights that no bf16 value represents: quantised from their own values (mm_weight.py:324-327
    # hands the quantiser `weight.to(torch.float32)`); a bf16 rounding in front would move maxima, scales and codes
    w32 = 0.02 * torch.randn(64, 256, generator=g)
    w32[2] = 0
    w32[4, 17] = 1.2345678
    assert not torch.equal(w32.to(torch.bfloat16).float(), w32)
    q32, s32, z32 = IntegerQuantizer(8, True, "per_channel").real_quant_tensor(w32.clone())
    assert z32 is None
    q32, s32 = q32.to(torch.int8), s32.to(torch.float32).reshape(-1)
    mq32, ms32 = R.quant_int8_weight(w32)
    assert torch.equal(mq32, q32) and torch.equal(ms32, s32)
    via_bf16 = R.quant_int8_weight(w32.to(torch.bfloat16))
    assert not torch.equal(via_bf16[0], q32)   # the case is live: a bf16 detour gives other codes
    np.savez_compressed(os.path.join(a.out, "int8_weight_quant.npz"), w_bf16_bits=w.view(torch.int16).numpy(), q=q.numpy(), scale=scales.numpy(),
                        w_f32=w32.numpy(), q_f32=q32.numpy(), scale_f32=s32.numpy())
    print(f"int8_weight_quant: {tuple(w.shape)} bf16-valued and {tuple(w32.shape)} fp32 weights, restatement == reference (bit for bit); |q| max {int(q.abs().max())}; "
          f"{int((via_bf16[0] != q32).sum())} of {q32.numel()} codes differ when the fp32 weight is first rounded to bf16")


if __name__ == "__main__":
    main()
