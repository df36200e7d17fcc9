"""CPU oracle for the Wan DiT transformer-block forward (TEST INFRASTRUCTURE ONLY).

This file is a plain-torch fp32 restatement of the reference's block forward with
the `torch_sdpa` attention plugin, the `Default` matmul plugin and the torch norm
fallbacks.  It is the checker that parity tests, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` compare the HIP path against.  Nothing under
`lightx2v_amd/` may import it: the product path runs on the HIP extension only.

Pinning: `oracle/gen_golden.py` imports the reference's own modules (CPU, fp32) and
checks every stage of this restatement against them; the vectors it wrote live in
`tests/golden/` and `tests/test_oracle_golden.py` re-checks this file against them
wherever the reference itself is not present.

Each function cites the reference code it restates (paths relative to the reference
repository root).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

EPS = 1e-6


# --------------------------------------------------------------------------- RoPE table
def rope_params(max_seq_len: int, dim: int, theta: float = 10000.0) -> torch.Tensor:
    """lightx2v/models/schedulers/wan/scheduler.py:41-48 -> complex64 [max_seq_len, dim/2]."""
    assert dim % 2 == 0
    freqs = torch.outer(
        torch.arange(max_seq_len),
        1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float32).div(dim)),
    )
    return torch.polar(torch.ones_like(freqs), freqs)


def rope_freqs(head_dim: int) -> torch.Tensor:
    """scheduler.py:31-39: three tables (temporal | height | width) concatenated on dim 1."""
    d = head_dim
    return torch.cat(
        [
            rope_params(1024, d - 4 * (d // 6)),
            rope_params(1024, 2 * (d // 6)),
            rope_params(1024, 2 * (d // 6)),
        ],
        dim=1,
    )


def build_cos_sin(grid: Tuple[int, int, int], head_dim: int) -> torch.Tensor:
    """scheduler.py:77-89,105 (`rope_type != "flashinfer"` branch): complex64 [S, 1, head_dim/2].

    Tokens are ordered f-major, then h, then w; slot split is c-2(c//3), c//3, c//3.
    """
    c = head_dim // 2
    freqs = rope_freqs(head_dim).split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    f, h, w = grid
    cos_sin = torch.cat(
        [
            freqs[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
            freqs[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
            freqs[2][:w].view(1, 1, w, -1).expand(f, h, w, -1),
        ],
        dim=-1,
    )
    return cos_sin.reshape(f * h * w, 1, -1)


def cos_sin_to_real(cos_sin: torch.Tensor) -> torch.Tensor:
    """scheduler.py:90-95 (flashinfer layout): fp32 [S, head_dim] = cos(64) || sin(64)."""
    cs = cos_sin.reshape(cos_sin.shape[0], -1)
    return torch.cat([cs.real.contiguous(), cs.imag.contiguous()], dim=-1).to(torch.float32)


def pad_and_shard_cos_sin(cos_sin: torch.Tensor, world_size: int, rank: int) -> torch.Tensor:
    """scheduler.py:106-113: zero-pad S to a multiple of world_size, take this rank's chunk."""
    s = cos_sin.shape[0]
    pad = (world_size - (s % world_size)) % world_size
    if pad > 0:
        cos_sin = F.pad(cos_sin, (0, 0, 0, 0, 0, pad))
    return torch.chunk(cos_sin, world_size, dim=0)[rank]


# --------------------------------------------------------------------------- element ops
def layer_norm(x: torch.Tensor, weight=None, bias=None, eps: float = EPS) -> torch.Tensor:
    """lightx2v/common/ops/norm/layer_norm_weight.py:157-169 (`Default`)."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def modulate(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """lightx2v/models/networks/wan/infer/transformer_infer.py:12-13."""
    return x * (1 + scale.squeeze()) + shift.squeeze()


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float = EPS) -> torch.Tensor:
    """lightx2v/common/ops/norm/rms_norm_weight.py:131-146 (`sgl-kernel` class, torch fallback
    branch taken when sgl_kernel is absent): RMS over the WHOLE last dim (all heads jointly)."""
    x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return x * weight


def rope_apply(xq: torch.Tensor, xk: torch.Tensor, cos_sin: torch.Tensor):
    """lightx2v/models/networks/wan/infer/utils.py:12-28 (`apply_wan_rope_with_torch`):
    interleaved pairs (x[2i], x[2i+1]) times e^{j theta_i}, same table for every head."""
    n = xq.size(1)
    s = cos_sin.size(0)
    q = torch.view_as_complex(xq[:s].to(torch.float32).reshape(s, n, -1, 2))
    k = torch.view_as_complex(xk[:s].to(torch.float32).reshape(s, n, -1, 2))
    q = torch.view_as_real(q * cos_sin).flatten(2)
    k = torch.view_as_real(k * cos_sin).flatten(2)
    return q, k


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """lightx2v/common/ops/attn/torch_sdpa.py:14-39: q [Sq,H,d], k/v [Skv,H,d] -> [Sq, H*d];
    non-causal, no mask, scale 1/sqrt(d)."""
    q4, k4, v4 = (t.unsqueeze(0).transpose(1, 2) for t in (q, k, v))
    x = F.scaled_dot_product_attention(q4, k4, v4, attn_mask=None, dropout_p=0.0, is_causal=False)
    x = x.transpose(1, 2)
    b, s, a, d = x.shape
    return x.reshape(b, s, -1).squeeze(0)


def sdpa_naive(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Materialised-softmax form of `sdpa` in float64 for small cases (independent cross-check)."""
    q, k, v = (t.to(torch.float64) for t in (q, k, v))
    s = torch.einsum("qhd,khd->hqk", q, k) / math.sqrt(q.shape[-1])
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hqk,khd->qhd", p, v).reshape(q.shape[0], -1)


def linear(x: torch.Tensor, w_out_in: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """lightx2v/common/ops/mm/mm_weight.py:163-170: addmm(bias, x, W^T) with checkpoint W [out,in]."""
    if b is None:
        return torch.mm(x, w_out_in.t())
    return torch.addmm(b, x, w_out_in.t())


# --------------------------------------------------------------------------- weights / inputs
BLOCK_LINEARS = (
    "self_attn.q", "self_attn.k", "self_attn.v", "self_attn.o",
    "cross_attn.q", "cross_attn.k", "cross_attn.v", "cross_attn.o",
    "ffn.0", "ffn.2",
)


def make_block_weights(dim: int, ffn_dim: int, seed: int = 0, block_index: int = 0,
                       affine_jitter: bool = True, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Synthetic checkpoint for one block, keyed like the reference's safetensors
    (lightx2v/models/networks/wan/weights/transformer_weights.py:170-470).  Fixed generator-call
    order: linears in BLOCK_LINEARS order (weight then bias), norm_q/k (self, cross), norm3 w/b,
    modulation."""
    g = torch.Generator().manual_seed(seed * 1000003 + block_index)
    p = f"blocks.{block_index}."
    w: Dict[str, torch.Tensor] = {}
    for name in BLOCK_LINEARS:
        out_f, in_f = dim, dim
        if name == "ffn.0":
            out_f = ffn_dim
        if name == "ffn.2":
            in_f = ffn_dim
        w[p + name + ".weight"] = 0.02 * torch.randn(out_f, in_f, generator=g)
        w[p + name + ".bias"] = 0.02 * torch.randn(out_f, generator=g)
    for name in ("self_attn.norm_q", "self_attn.norm_k", "cross_attn.norm_q", "cross_attn.norm_k"):
        t = torch.randn(dim, generator=g)
        w[p + name + ".weight"] = 1.0 + 0.1 * t if affine_jitter else torch.ones(dim)
    t = torch.randn(dim, generator=g)
    w[p + "norm3.weight"] = 1.0 + 0.1 * t if affine_jitter else torch.ones(dim)
    t = torch.randn(dim, generator=g)
    w[p + "norm3.bias"] = 0.1 * t if affine_jitter else torch.zeros(dim)
    w[p + "modulation"] = torch.randn(1, 6, dim, generator=g) / math.sqrt(dim)
    return {k: v.to(dtype) for k, v in w.items()}


IMAGE_TOKENS = 257  # CLIP tokens in front of the text context (transformer_infer.py:217-218)


def make_image_branch_weights(dim: int, seed: int = 0, block_index: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """The extra keys of an i2v block with a CLIP image encoder
    (weights/transformer_weights.py:398-431): `cross_attn.{k_img, v_img}.{weight, bias}`,
    `cross_attn.norm_k_img.weight`.  Own generator: the t2v keys of a seed do not change."""
    g = torch.Generator().manual_seed(seed * 1000003 + block_index + 500009)
    p = f"blocks.{block_index}.cross_attn."
    w = {}
    for n in ("k_img", "v_img"):
        w[p + n + ".weight"] = 0.02 * torch.randn(dim, dim, generator=g)
        w[p + n + ".bias"] = 0.02 * torch.randn(dim, generator=g)
    w[p + "norm_k_img.weight"] = 1.0 + 0.1 * torch.randn(dim, generator=g)
    return {k: v.to(dtype) for k, v in w.items()}


def make_head_weights(dim: int, out_dim: int = 64, seed: int = 0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed * 1000003 + 999983)
    return {
        "head.head.weight": (0.02 * torch.randn(out_dim, dim, generator=g)).to(dtype),
        "head.head.bias": (0.02 * torch.randn(out_dim, generator=g)).to(dtype),
        "head.modulation": (torch.randn(1, 2, dim, generator=g) / math.sqrt(dim)).to(dtype),
    }


def make_inputs(seq_len: int, dim: int, text_len: int = 512, seed: int = 0, dtype=torch.float32):
    """x ~ N(0,1) [S,D]; context ~ N(0,1) [T,D]; embed0 ~ 0.1 N(0,1) [6,D]; embed ~ 0.1 N(0,1) [1,D]."""
    g = torch.Generator().manual_seed(seed * 1000003 + 777767)
    x = torch.randn(seq_len, dim, generator=g)
    context = torch.randn(text_len, dim, generator=g)
    embed0 = 0.1 * torch.randn(6, dim, generator=g)
    embed = 0.1 * torch.randn(1, dim, generator=g)
    return x.to(dtype), context.to(dtype), embed0.to(dtype), embed.to(dtype)


# --------------------------------------------------------------------------- the block
@dataclass
class BlockTrace:
    """Per-stage intermediates (fp32) for stage-by-stage parity checks."""
    stages: Dict[str, torch.Tensor] = field(default_factory=dict)

    def put(self, name, t):
        self.stages[name] = t.detach().clone()


def block_forward(x: torch.Tensor, embed0: torch.Tensor, context: torch.Tensor,
                  w: Dict[str, torch.Tensor], cos_sin: torch.Tensor, num_heads: int,
                  block_index: int = 0, trace: Optional[BlockTrace] = None,
                  attn_fn=sdpa, self_attn_fn=None, image_tokens: int = 0) -> torch.Tensor:
    """`WanTransformerInfer.infer_block`, transformer_infer.py:105-128 (no adapters).  With
    `image_tokens` > 0 (i2v with a CLIP image encoder: 257) the first rows of `context` are image
    tokens with their own k / v projections and a second cross-attention whose output is added to
    the text one before the output projection (:216-219, 247-269).

    x [S,D] (not modified), embed0 [6,D], context [T,D], cos_sin complex [S,1,d/2].
    `self_attn_fn(q,k,v)` lets a sequence-parallel wrapper replace the self-attention call
    (transformer_infer.py:177-199)."""
    p = f"blocks.{block_index}."
    D = x.shape[1]
    H, d = num_heads, D // num_heads
    t = trace.put if trace is not None else (lambda *_: None)

    # pre_process :130-142 -- chunk order shift, scale, gate, c_shift, c_scale, c_gate
    shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = (w[p + "modulation"] + embed0).chunk(6, dim=1)

    # infer_self_attn :144-207
    norm1 = modulate(layer_norm(x), scale_msa, shift_msa)
    t("norm1_out", norm1)
    s = norm1.shape[0]
    q = rms_norm(linear(norm1, w[p + "self_attn.q.weight"], w[p + "self_attn.q.bias"]), w[p + "self_attn.norm_q.weight"]).view(s, H, d)
    k = rms_norm(linear(norm1, w[p + "self_attn.k.weight"], w[p + "self_attn.k.bias"]), w[p + "self_attn.norm_k.weight"]).view(s, H, d)
    v = linear(norm1, w[p + "self_attn.v.weight"], w[p + "self_attn.v.bias"]).view(s, H, d)
    t("q_rms", q.reshape(s, D)); t("k_rms", k.reshape(s, D)); t("v", v.reshape(s, D))
    q, k = rope_apply(q, k, cos_sin)
    t("q_rope", q.reshape(s, D)); t("k_rope", k.reshape(s, D))
    attn = (self_attn_fn or attn_fn)(q, k, v)
    t("self_attn_out", attn)
    y = linear(attn, w[p + "self_attn.o.weight"], w[p + "self_attn.o.bias"])
    t("self_attn_y", y)

    # infer_cross_attn :209-276
    x = x + y * gate_msa.squeeze()
    t("x_after_self", x)
    norm3 = layer_norm(x, w[p + "norm3.weight"], w[p + "norm3.bias"])
    t("norm3_out", norm3)
    context_img, context = (context[:image_tokens], context[image_tokens:]) if image_tokens else (None, context)
    cq = rms_norm(linear(norm3, w[p + "cross_attn.q.weight"], w[p + "cross_attn.q.bias"]), w[p + "cross_attn.norm_q.weight"]).view(-1, H, d)
    ck = rms_norm(linear(context, w[p + "cross_attn.k.weight"], w[p + "cross_attn.k.bias"]), w[p + "cross_attn.norm_k.weight"]).view(-1, H, d)
    cv = linear(context, w[p + "cross_attn.v.weight"], w[p + "cross_attn.v.bias"]).view(-1, H, d)
    t("cross_q", cq.reshape(s, D)); t("cross_k", ck.reshape(-1, D)); t("cross_v", cv.reshape(-1, D))
    cattn = attn_fn(cq, ck, cv)
    if context_img is not None:
        ik = rms_norm(linear(context_img, w[p + "cross_attn.k_img.weight"], w[p + "cross_attn.k_img.bias"]), w[p + "cross_attn.norm_k_img.weight"]).view(-1, H, d)
        iv = linear(context_img, w[p + "cross_attn.v_img.weight"], w[p + "cross_attn.v_img.bias"]).view(-1, H, d)
        cattn = cattn + attn_fn(cq, ik, iv)
    t("cross_attn_out", cattn)
    attn_out = linear(cattn, w[p + "cross_attn.o.weight"], w[p + "cross_attn.o.bias"])
    t("cross_attn_y", attn_out)

    # infer_ffn :278-310
    x = x + attn_out
    t("x_after_cross", x)
    norm2 = modulate(layer_norm(x), c_scale, c_shift)
    t("norm2_out", norm2)
    hmid = F.gelu(linear(norm2, w[p + "ffn.0.weight"], w[p + "ffn.0.bias"]), approximate="tanh")
    t("ffn_mid", hmid)
    y = linear(hmid, w[p + "ffn.2.weight"], w[p + "ffn.2.bias"])
    t("ffn_y", y)

    # post_process :312-321
    x = x + y * c_gate.squeeze()
    t("x_out", x)
    return x


def head_forward(x: torch.Tensor, embed: torch.Tensor, hw: Dict[str, torch.Tensor]) -> torch.Tensor:
    """`infer_non_blocks`, transformer_infer.py:75-97 (2-D embed branch): LN(x)*(1+e1)+e0, Linear D->out."""
    e = (hw["head.modulation"] + embed.unsqueeze(1)).chunk(2, dim=1)
    x = layer_norm(x)
    x = x * (1 + e[1].squeeze()) + e[0].squeeze()
    return linear(x, hw["head.head.weight"], hw["head.head.bias"])


# --------------------------------------------------------------------------- Ulysses reshards
def seq2head_reference(shards: List[torch.Tensor]) -> List[torch.Tensor]:
    """Single-process model of `all2all_seq2head` (lightx2v/common/ops/attn/utils/all2all.py:7-44)
    over N ranks: shards[r] is rank r's [S/N, H, d]; returns per-rank [S, H/N, d] where global
    token order is rank-major chunks and rank r keeps heads r*H/N..(r+1)*H/N."""
    n = len(shards)
    hs = shards[0].shape[1] // n
    full = torch.cat(shards, dim=0)
    return [full[:, r * hs:(r + 1) * hs, :].contiguous() for r in range(n)]


def head2seq_reference(outs: List[torch.Tensor]) -> List[torch.Tensor]:
    """Single-process model of `all2all_head2seq` (all2all.py:48-89): outs[r] is [S, H/N, d];
    returns per-rank [S/N, H, d] with heads ordered by source rank."""
    n = len(outs)
    full = torch.cat(outs, dim=1)
    return list(torch.chunk(full, n, dim=0))


def ulysses_self_attention(q_shards, k_shards, v_shards, attn_fn=sdpa) -> List[torch.Tensor]:
    """`UlyssesAttnWeight.apply` (lightx2v/common/ops/attn/ulysses_attn.py:16-137) for the Wan case
    (no text part): returns per-rank [S/N, H*d]."""
    qs, ks, vs = seq2head_reference(q_shards), seq2head_reference(k_shards), seq2head_reference(v_shards)
    outs = []
    for q, k, v in zip(qs, ks, vs):
        o = attn_fn(q, k, v)
        outs.append(o.reshape(o.shape[0], q.shape[1], q.shape[2]))
    back = head2seq_reference(outs)
    return [b.reshape(b.shape[0], -1) for b in back]


def cfg_combine(cond: torch.Tensor, uncond: torch.Tensor, guide_scale: float) -> torch.Tensor:
    """lightx2v/models/networks/wan/model.py:433: uncond + g*(cond - uncond)."""
    return uncond + guide_scale * (cond - uncond)


def cosine(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


# --------------------------------------------------------------------------- fp8 W8A8 Linear
FP8_MAX = 448.0  # largest finite OCP e4m3 value


def quant_fp8_rowwise(x: torch.Tensor):
    """Dynamic symmetric e4m3 quantisation with one scale per row: scale = max|row| / 448,
    q = e4m3(row * (1/scale)) (round to nearest even).  Per token for activations
    (`sgl_per_token_quant_fp8` / vllm `scaled_fp8_quant(use_per_token_if_dynamic=True)`,
    lightx2v/common/ops/mm/mm_weight.py:492-501), per output channel for weights
    (`FloatQuantizer("e4m3", True, "per_channel")`, mm_weight.py:312-320,
    lightx2v/utils/quant_utils.py:132-161).  Those kernels live in wheels that are absent here
    (sgl-kernel 0.1.8, vllm 0.9.2) and the reference holds no fixture for them: parity unpinned --
    this is their published formula.  Returns (q as float8_e4m3fn, scale fp32 [rows])."""
    x = x.float()
    scale = x.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX
    inv = 1.0 / scale
    return (x * inv[:, None]).to(torch.float8_e4m3fn), scale


def linear_fp8(x: torch.Tensor, w_q: torch.Tensor, w_scale: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`fp8-sgl` apply (mm_weight.py:901-928): quantise x per token, multiply in the quantised
    domain, rescale by both scales, add the bias."""
    x_q, x_scale = quant_fp8_rowwise(x)
    y = (x_q.float() @ w_q.float().t()) * x_scale[:, None] * w_scale[None, :]
    return y if b is None else y + b


def quant_int8_weight(w: torch.Tensor):
    """The reference's per-output-channel int8 weight quantiser, `IntegerQuantizer(8, True, "per_channel")
    .real_quant_tensor` as `load_int8_perchannel_sym` calls it (lightx2v/common/ops/mm/mm_weight.py:322-330;
    lightx2v/utils/quant_utils.py:29-36 range, :39-49 scale = clamp(absmax, 1e-5) / 127, :111-113
    q = clamp(round(w / scale), -128, 127)).  PINNED by that class itself (tests/golden/int8_weight_quant.npz,
    oracle/gen_golden_int8.py).  Returns (q int8, scale fp32 [rows])."""
    w = w.float()
    absmax = torch.max(w.amax(dim=-1, keepdim=True).abs(), w.amin(dim=-1, keepdim=True).abs()).clamp(min=1e-5)
    scale = absmax / 127.0
    q = torch.clamp(torch.round(w / scale), -128, 127)
    return q.to(torch.int8), scale.reshape(-1)


def quant_int8_rowwise(x: torch.Tensor):
    """Dynamic symmetric int8 quantisation with one scale per token, `act_quant_int8_perchannel_sym_vllm`
    (mm_weight.py:503-505) = vllm `ops.scaled_int8_quant(x, scale=None, azp=None, symmetric=True)`.  vllm
    0.9.2 is absent here and the reference holds no fixture for it: parity unpinned -- this is the
    published kernel's arithmetic (csrc/quantization/compressed_tensors/int8_quant_kernels.cu,
    dynamic_scaled_int8_quant_kernel): scale = absmax / 127, q = saturate(nearbyint(x * (127 / absmax))).
    Returns (q int8, scale fp32 [rows])."""
    x = x.float()
    absmax = x.abs().amax(dim=1)
    # a true fp32 division, as the kernel's `127.0f / absmax` (a Python scalar divided by a tensor is reciprocal-then-multiply in torch: two roundings)
    inv = torch.where(absmax > 0, torch.full_like(absmax, 127.0) / absmax, torch.zeros_like(absmax))
    q = torch.clamp(torch.round(x * inv[:, None]), -128, 127)
    return q.to(torch.int8), absmax / 127.0


def linear_int8(x: torch.Tensor, w_q: torch.Tensor, w_scale: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`int8-vllm` apply (mm_weight.py:623-640, cutlass_scaled_mm): quantise x per token, multiply
    exactly in the integer domain, rescale by both scales, add the bias."""
    x_q, x_scale = quant_int8_rowwise(x)
    acc = (x_q.to(torch.int64) @ w_q.to(torch.int64).t()).to(torch.int32)   # exact
    y = acc.float() * x_scale[:, None] * w_scale[None, :]
    return y if b is None else y + b


def dequant_fp8_rowwise(q: torch.Tensor, scale: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """`dequant_fp8_vllm` (lightx2v/utils/quant_utils.py:185-186): q.to(dtype) * scale.to(dtype), one
    scale per row."""
    return q.to(dtype) * scale.to(dtype)[:, None]


def fp8_comm_roundtrip(x: torch.Tensor) -> torch.Tensor:
    """What an activation [tokens, heads, d] looks like after the fp8 Ulysses payload
    (lightx2v/common/ops/attn/ulysses_attn.py:59-73): quantised per (token, head) row of d values,
    moved (a pure permutation), widened back to bf16 (the activation dtype in production)."""
    t, h, d = x.shape
    q, s = quant_fp8_rowwise(x.reshape(-1, d))
    return dequant_fp8_rowwise(q, s, torch.bfloat16).to(x.dtype).view(t, h, d)
