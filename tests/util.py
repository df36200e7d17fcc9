"""Shared helpers for the parity tests: golden fixtures + input re-generation from their recipe."""
import os

import numpy as np
import torch

from oracle import wan_block_ref as R

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLOCK_CASES = ["tiny_d256_g2x4x4", "tiny_d256_g3x6x8", "small_d512_g3x5x7", "wan1p3b_g3x6x8", "wan1p3b_g3x30x52",
               "wan14b_g3x30x52",     # Wan2.1-14B block dims (D=5120, H=40, F=13824) at 4 680 tokens
               "wan1p3b_g21x30x52"]   # the full 32 760-token 1.3B block: many-round launches, round alignment


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


_CASE_CACHE = {}


def cached_case(name):
    """(fixture, regenerated inputs) of a named golden; the large cases take seconds to regenerate,
    so one copy is shared by the tests of a session (treat the tensors as read-only)."""
    if name not in _CASE_CACHE:
        if len(_CASE_CACHE) >= 2:  # the 14B-dim case holds 1.4 GB of fp32 weights: keep few
            _CASE_CACHE.pop(next(iter(_CASE_CACHE)))
        rec = load_golden(name)
        _CASE_CACHE[name] = (rec, case_inputs(rec))
    return _CASE_CACHE[name]


def case_inputs(rec):
    """Regenerate (x, context, embed0, embed, block weights, head weights, cos_sin complex) from a fixture's recipe."""
    grid = tuple(int(v) for v in rec["grid"])
    dim, heads, ffn, text_len, seed = (int(rec[k]) for k in ("dim", "heads", "ffn", "text_len", "seed"))
    S = grid[0] * grid[1] * grid[2]
    w = R.make_block_weights(dim, ffn, seed=seed)
    hw = R.make_head_weights(dim, 64, seed=seed)
    x, context, embed0, embed = R.make_inputs(S, dim, text_len, seed=seed)
    cos_sin = R.build_cos_sin(grid, dim // heads)
    return dict(grid=grid, dim=dim, heads=heads, ffn=ffn, S=S, w=w, hw=hw, x=x, context=context, embed0=embed0, embed=embed, cos_sin=cos_sin)


def golden_rows(rec, S):
    return torch.from_numpy(rec["rows"]).long() if "rows" in rec.files else torch.arange(S)


def cos(a, b):
    return R.cosine(a.float().cpu(), b.float().cpu())


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
