"""CPU restatement (fp32, plain torch) of the HunyuanVideo-1.5 double-stream block and final layer.

TEST INFRASTRUCTURE ONLY: imported by tests/, the smoke check and the golden generator; the product
path (lightx2v_amd/) never imports it.

Follows lightx2v/models/networks/hunyuan_video/infer/transformer_infer.py:98-270 with the torch
choices of every switch (`modulate_type="torch"`, `rope_type="torch"`, `attn_type="torch_sdpa"`,
`ln_type="Default"`, `rms_type="sgl-kernel"` in its torch fallback), which is the only combination
whose arithmetic is fully inside the reference checkout.  Pinned by tests/golden/hy_*.npz, which
oracle/gen_golden_hy.py produced by running the reference's own classes.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import wan_block_ref as R

EPS = 1e-6
ROPE_DIM_LIST = (16, 56, 56)  # HunyuanVideo-1.5 transformer config: head_dim 128 split over (t, h, w)
ROPE_THETA = 256.0


# --------------------------------------------------------------------------- RoPE table
def build_cos_sin(grid: Tuple[int, int, int], head_dim: int = 128, rope_dim_list: Optional[Sequence[int]] = ROPE_DIM_LIST,
                  theta: float = ROPE_THETA) -> torch.Tensor:
    """`HunyuanVideo15Scheduler.prepare_cos_sin` (lightx2v/models/schedulers/hunyuan_video/scheduler.py:123-142)
    over `get_nd_rotary_pos_embed(use_real=True)` (posemb_layers.py:169-232, 235-283, 23-57):
    fp32 [t*h*w, head_dim] = cos || sin, each half the concatenation over the axes (t, h, w) of
    cos/sin(pos_axis * theta^(-2j/dim_axis)), j < dim_axis/2; tokens ordered t-major, then h, then w."""
    if rope_dim_list is None:
        rope_dim_list = [head_dim // 3] * 3
    assert sum(rope_dim_list) == head_dim
    axes = [torch.linspace(0, n, n + 1, dtype=torch.float32)[:n] for n in grid]
    mesh = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=0)
    cos, sin = [], []
    for i, dim in enumerate(rope_dim_list):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        ang = torch.outer(mesh[i].reshape(-1), freqs)
        cos.append(ang.cos())
        sin.append(ang.sin())
    return torch.cat([torch.cat(cos, dim=1), torch.cat(sin, dim=1)], dim=-1)


def rope_apply(x: torch.Tensor, cos_sin: torch.Tensor) -> torch.Tensor:
    """`apply_hunyuan_rope_with_torch` (transformer_infer.py:67-95): x [L, H, d], interleaved pairs
    (x[2i], x[2i+1]) rotated by angle i of the token's table row; same table for every head."""
    d = x.shape[-1]
    cos, sin = cos_sin[:, : d // 2].unsqueeze(1), cos_sin[:, d // 2:].unsqueeze(1)
    x1, x2 = x[..., ::2].float(), x[..., 1::2].float()
    out = torch.empty_like(x)
    out[..., ::2] = x1 * cos - x2 * sin
    out[..., 1::2] = x2 * cos + x1 * sin
    return out


# --------------------------------------------------------------------------- weights / inputs
BRANCH_LINEARS = ("mod.linear", "attn_q", "attn_k", "attn_v", "attn_proj", "mlp.fc1", "mlp.fc2")


def make_double_block_weights(dim: int, heads: int, mlp_dim: int, seed: int = 0, block_index: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Synthetic checkpoint of one double-stream block under the reference's key names
    (lightx2v/models/networks/hunyuan_video/weights/transformer_weights.py:91-318).  Fixed generator
    call order: img branch then txt branch, Linears in BRANCH_LINEARS order (weight, bias), then the
    q/k per-head RMS weights."""
    g = torch.Generator().manual_seed(seed)
    d = dim // heads
    w: Dict[str, torch.Tensor] = {}
    p = f"double_blocks.{block_index}."
    for br in ("img", "txt"):
        for name in BRANCH_LINEARS:
            out_f, in_f = {"mod.linear": (6 * dim, dim), "mlp.fc1": (mlp_dim, dim), "mlp.fc2": (dim, mlp_dim)}.get(name, (dim, dim))
            std = 0.02 if name != "mod.linear" else 1.0 / math.sqrt(dim)
            w[f"{p}{br}_{name}.weight"] = (torch.randn(out_f, in_f, generator=g) * std).to(dtype)
            w[f"{p}{br}_{name}.bias"] = (torch.randn(out_f, generator=g) * 0.02).to(dtype)
        for n in ("q", "k"):
            w[f"{p}{br}_attn_{n}_norm.weight"] = (1.0 + 0.1 * torch.randn(d, generator=g)).to(dtype)
    return w


def make_final_weights(dim: int, out_dim: int, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """`FinalLayerWeights` keys (transformer_weights.py:321-357)."""
    g = torch.Generator().manual_seed(seed + 7919)
    return {
        "final_layer.adaLN_modulation.1.weight": (torch.randn(2 * dim, dim, generator=g) / math.sqrt(dim)).to(dtype),
        "final_layer.adaLN_modulation.1.bias": (torch.randn(2 * dim, generator=g) * 0.02).to(dtype),
        "final_layer.linear.weight": (torch.randn(out_dim, dim, generator=g) * 0.02).to(dtype),
        "final_layer.linear.bias": (torch.randn(out_dim, generator=g) * 0.02).to(dtype),
    }


def make_inputs(img_len: int, txt_len: int, dim: int, seed: int = 0, dtype=torch.float32):
    """img [1, L, D] ~ N(0,1), txt [1, Lt, D] ~ N(0,1), vec [1, D] = silu(0.5 N(0,1)) (the
    pre-infer hands the blocks silu(time embedding), pre_infer.py:165)."""
    g = torch.Generator().manual_seed(seed + 104729)
    img = torch.randn(1, img_len, dim, generator=g).to(dtype)
    txt = torch.randn(1, txt_len, dim, generator=g).to(dtype)
    vec = torch.nn.functional.silu(0.5 * torch.randn(1, dim, generator=g)).to(dtype)
    return img, txt, vec


# --------------------------------------------------------------------------- block
@dataclass
class DoubleBlockTrace:
    stages: Dict[str, torch.Tensor] = field(default_factory=dict)

    def put(self, name, t):
        self.stages[name] = t.detach().clone()


def _rms_per_head(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """RMSNorm over the last dim of [L, H, d] with a [d] weight (per head;
    lightx2v/common/ops/norm/rms_norm_weight.py:131-146, torch fallback)."""
    x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + EPS)
    return x * weight


def _branch_before_attn(x, vec, w, p, br, heads, cos_sin, trace):
    """`_infer_img_branch_before_attn` / `_infer_txt_branch_before_attn` (transformer_infer.py:153-216)."""
    mod = R.linear(vec, w[f"{p}{br}_mod.linear.weight"], w[f"{p}{br}_mod.linear.bias"])
    m1_shift, m1_scale, m1_gate, m2_shift, m2_scale, m2_gate = mod.chunk(6, dim=-1)
    y = R.layer_norm(x) * (1 + m1_scale) + m1_shift
    if trace is not None:
        trace.put(f"{br}_modulated", y)
    q, k, v = (R.linear(y, w[f"{p}{br}_attn_{n}.weight"], w[f"{p}{br}_attn_{n}.bias"]).unflatten(1, (heads, -1)) for n in "qkv")
    q = _rms_per_head(q, w[f"{p}{br}_attn_q_norm.weight"])
    k = _rms_per_head(k, w[f"{p}{br}_attn_k_norm.weight"])
    if cos_sin is not None:
        q, k = rope_apply(q, cos_sin), rope_apply(k, cos_sin)
    if trace is not None:
        trace.put(f"{br}_q", q)
        trace.put(f"{br}_k", k)
    return q, k, v, (m1_gate, m2_shift, m2_scale, m2_gate)


def _branch_after_attn(x, attn, w, p, br, mods, trace):
    """`_infer_*_branch_after_attn` (transformer_infer.py:252-270)."""
    m1_gate, m2_shift, m2_scale, m2_gate = mods
    x = x + R.linear(attn, w[f"{p}{br}_attn_proj.weight"], w[f"{p}{br}_attn_proj.bias"]) * m1_gate
    if trace is not None:
        trace.put(f"{br}_after_attn", x)
    h = R.layer_norm(x) * (1 + m2_scale) + m2_shift
    h = R.linear(h, w[f"{p}{br}_mlp.fc1.weight"], w[f"{p}{br}_mlp.fc1.bias"])
    h = R.linear(torch.nn.functional.gelu(h, approximate="tanh"), w[f"{p}{br}_mlp.fc2.weight"], w[f"{p}{br}_mlp.fc2.bias"])
    return x + h * m2_gate


def double_block_forward(img: torch.Tensor, txt: torch.Tensor, vec: torch.Tensor, w: Dict[str, torch.Tensor], cos_sin: torch.Tensor,
                         heads: int, block_index: int = 0, trace: Optional[DoubleBlockTrace] = None, attn_fn=R.sdpa):
    """`infer_double_block` (transformer_infer.py:144-151): img [L, D], txt [Lt, D], vec [1, D] ->
    (img, txt).  Joint attention over img tokens followed by txt tokens; RoPE on the img part only."""
    p = f"double_blocks.{block_index}."
    li = img.shape[0]
    iq, ik, iv, imods = _branch_before_attn(img, vec, w, p, "img", heads, cos_sin, trace)
    tq, tk, tv, tmods = _branch_before_attn(txt, vec, w, p, "txt", heads, None, trace)
    attn = attn_fn(torch.cat([iq, tq]), torch.cat([ik, tk]), torch.cat([iv, tv]))
    if trace is not None:
        trace.put("attn", attn)
    img = _branch_after_attn(img, attn[:li], w, p, "img", imods, trace)
    txt = _branch_after_attn(txt, attn[li:], w, p, "txt", tmods, trace)
    return img, txt


def final_layer_forward(img: torch.Tensor, vec: torch.Tensor, fw: Dict[str, torch.Tensor]) -> torch.Tensor:
    """`infer_final_layer` (transformer_infer.py:135-142): adaLN (shift, scale) from vec, LN,
    modulate, Linear D -> out."""
    shift, scale = R.linear(vec, fw["final_layer.adaLN_modulation.1.weight"], fw["final_layer.adaLN_modulation.1.bias"]).chunk(2, dim=1)
    y = R.layer_norm(img) * (1 + scale) + shift
    return R.linear(y, fw["final_layer.linear.weight"], fw["final_layer.linear.bias"])


def forward(img, txt, vec, blocks: List[Dict[str, torch.Tensor]], fw, cos_sin, heads):
    """`HunyuanVideo15TransformerInfer.infer` (transformer_infer.py:124-133)."""
    for i, w in enumerate(blocks):
        img, txt = double_block_forward(img, txt, vec, w, cos_sin, heads, block_index=i)
    return final_layer_forward(img, vec, fw)


# --------------------------------------------------------------------------- Ulysses with a text part
def ulysses_joint_attention(q_shards, k_shards, v_shards, img_len_local: int, attn_fn=R.sdpa) -> List[torch.Tensor]:
    """What `UlyssesAttnWeight.apply` computes when text tokens follow the image tokens
    (lightx2v/common/ops/attn/ulysses_attn.py:41-118): every rank holds its image shard
    [S/N, H, d] followed by the WHOLE text [Lt, H, d] (replicated); rank r attends the full image
    sequence plus the text for its H/N heads; image rows return to their rank with all heads, text
    rows are gathered along the head axis.  Returns each rank's [S/N + Lt, H*d]."""
    n = len(q_shards)
    h = q_shards[0].shape[1]
    hs = h // n
    outs = []
    for r in range(n):
        sl = slice(r * hs, (r + 1) * hs)
        q, k, v = (torch.cat([s[:img_len_local, sl] for s in sh] + [sh[r][img_len_local:, sl]]) for sh in (q_shards, k_shards, v_shards))
        outs.append(attn_fn(q, k, v))  # [S + Lt, hs*d]
    res = []
    for r in range(n):
        img = torch.cat([o[r * img_len_local:(r + 1) * img_len_local] for o in outs], dim=1)
        txt = torch.cat([o[n * img_len_local:] for o in outs], dim=1)
        res.append(torch.cat([img, txt]))
    return res
