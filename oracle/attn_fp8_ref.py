"""CPU emulation of the block-scaled fp8 attention (lightx2v_amd/csrc/attn_fp8.hip).

TEST INFRASTRUCTURE ONLY.  Parity unpinned: the reference's quantised attention backends
(sageattention 2.2.0, lightx2v/common/ops/attn/sage_attn.py:36-64) live in an absent wheel, use a
different quantisation scheme (int8 QK^T with per-block scales, fp8 PV) and the reference holds no
fixture for them.  What is restated here is this build's own scheme, in plain torch, so the kernel
can be checked (a) bit for bit on the quantisation pre-pass and (b) within fp8 rounding noise on the
attention output, next to the fp32 softmax of wan_block_ref.sdpa.
"""

from __future__ import annotations

import math

import torch

FP8_MAX = 448.0


def e8m0_for(amax: torch.Tensor) -> torch.Tensor:
    """Biased exponent byte of the smallest power of two 2^e with amax / 2^e <= 448, from the fp32
    bits of amax (448 = 1.75 * 2^8): exponent - 8, one more when the mantissa exceeds 1.75; clamped
    to [1, 254]."""
    b = amax.float().contiguous().view(torch.int32)
    e = (b >> 23) - 8 + ((b & 0x7FFFFF) > 0x600000).to(torch.int32)
    return e.clamp(1, 254)


def mx_quant(x: torch.Tensor, dim: int = -1):
    """Quantise blocks of 32 consecutive elements along `dim` to e4m3 with one power-of-two scale
    each.  Returns (q float8_e4m3fn same shape, e8 int32 with `dim` reduced by 32)."""
    x = x.float().movedim(dim, -1)
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // 32, 32)
    e8 = e8m0_for(xb.abs().amax(-1))
    inv = torch.pow(2.0, (127 - e8).float())
    q = (xb * inv[..., None]).to(torch.float8_e4m3fn).reshape(shp).movedim(-1, dim)
    return q, e8.movedim(-1, dim) if e8.dim() else e8


def mx_dequant(q: torch.Tensor, e8: torch.Tensor, dim: int = -1) -> torch.Tensor:
    x = q.float().movedim(dim, -1)
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // 32, 32) * torch.pow(2.0, (e8.movedim(dim, -1) - 127).float())[..., None]
    return xb.reshape(shp).movedim(-1, dim)


def linear_p(x: torch.Tensor) -> torch.Tensor:
    """2^x with the mantissa taken linearly, as the e4m3 value whose code is 8 (x + 7) rounded to nearest
    even and clamped to a byte (a code is 8 (exponent + 7) + mantissa; codes below 8 are subnormals and
    continue the line): the kernel's P under `attn_fp8_p = 1`, for x <= 0."""
    code = torch.clamp(torch.round(8.0 * x + 56.0), 0, 255).to(torch.uint8)
    return code.view(torch.float8_e4m3fn).float()


def attention_fp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float | None = None, p_mode: str = "exact") -> torch.Tensor:
    """q [Sq,H,d], k/v [Skv,H,d] (bf16 values) -> [Sq, H*d] fp32 with the kernel's quantisation:
    q*scale*log2(e) and k block-quantised over channels, v block-quantised over groups of 32
    consecutive keys per channel, P = exp2(s - rowmax) rounded to e4m3 for the numerator, the row sum taken over
    the unrounded values.  (The kernel's running maximum lags the true one by up to 2^6, so its P is this P times a
    per-row factor before rounding: same distribution of rounding errors, not the same bits.)
    p_mode "linear": P = linear_p(s - rowmax) in the numerator and in the row sum."""
    sq, H, d = q.shape
    skv = k.shape[0]
    if scale is None:
        scale = d ** -0.5
    qd = mx_dequant(*mx_quant(q.float() * (scale * math.log2(math.e))))
    kd = mx_dequant(*mx_quant(k.float()))
    groups = (skv + 31) // 32
    vp = torch.zeros(groups * 32, H, d)
    vp[:skv] = v.float()
    vd = mx_dequant(*mx_quant(vp, dim=0), dim=0)[:skv]
    s = torch.einsum("qhd,khd->hqk", qd, kd)
    if p_mode == "linear":
        p = linear_p(s - s.amax(-1, keepdim=True))
        o = torch.einsum("hqk,khd->qhd", p, vd) / p.sum(-1).transpose(0, 1)[..., None]
        return o.reshape(sq, H * d)
    p = torch.exp2(s - s.amax(-1, keepdim=True))
    o = torch.einsum("hqk,khd->qhd", p.to(torch.float8_e4m3fn).float(), vd) / p.sum(-1).transpose(0, 1)[..., None]
    return o.reshape(sq, H * d)
