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
    np.savez_compressed(os.path.join(a.out, "int8_weight_quant.npz"), w_bf16_bits=w.view(torch.int16).numpy(), q=q.numpy(), scale=scales.numpy())
    print(f"int8_weight_quant: {tuple(w.shape)} weight, restatement == reference (bit for bit); |q| max {int(q.abs().max())}")


if __name__ == "__main__":
    main()
