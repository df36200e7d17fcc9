"""Wan DiT transformer-block forward on the HIP kernels: weight containers + block runner.

Containers mirror lightx2v/models/networks/wan/weights/transformer_weights.py:15-490 (same
attribute names, same checkpoint keys, `compute_phases[0..2]` = self-attn / cross-attn / FFN);
the runner mirrors `WanTransformerInfer` (lightx2v/models/networks/wan/infer/transformer_infer.py:16-321)
method for method.  Two execution modes share the same kernels:

  fused_block=True  (default)  AdaLN-LN in one pass; q|k|v as ONE GEMM; RMSNorm+RoPE in one pass
                               in place; O / FFN-2 projections write `x += (acc+b)*gate` from the
                               GEMM epilogue; FFN-0 applies GELU in its epilogue.
  fused_block=False            op-by-op through each plugin's `.apply`, the reference's own call
                               sequence (useful to A/B a single plugin inside the reference).
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

import torch

from . import hip
from . import plugins  # noqa: F401  (registers the HIP plugins)
from . import ulysses  # noqa: F401  (registers hip_ulysses)
from .registry_factory import (
    ATTN_WEIGHT_REGISTER,
    LN_WEIGHT_REGISTER,
    MM_WEIGHT_REGISTER,
    RMS_WEIGHT_REGISTER,
    TENSOR_REGISTER,
)
from .weight_module import WeightModule, WeightModuleList

DEFAULTS = {
    "self_attn_1_type": "hip_flash_attn",
    "cross_attn_1_type": "hip_flash_attn",
    "cross_attn_2_type": "hip_flash_attn",
    "dit_quant_scheme": "hip-bf16",
    "ln_type": "hip",
    "rms_type": "hip",
    "rope_type": "hip",          # "hip": fused into the q/k RMSNorm pass; "hip_attn": rotated inside the attention kernel
    "fused_block": True,
    "seq_parallel": False,
    "task": "t2v",
    "model_cls": "wan2.1",
    # "fp32": the residual stream x stays in fp32 between the Linears (the reference's
    # SENSITIVE_LAYER_DTYPE=FP32, lightx2v/utils/envs.py:39-43); None: read that environment variable
    "sensitive_layer_dtype": None,
}


def sensitive_fp32(config) -> bool:
    v = config.get("sensitive_layer_dtype")
    if v is None:
        v = os.environ.get("SENSITIVE_LAYER_DTYPE", "None")
    v = str(v).lower()
    if v in ("none", "bf16", "bfloat16"):
        return False
    if v in ("fp32", "float32"):
        return True
    raise ValueError(f"sensitive_layer_dtype={v!r}: 'fp32' or None (bf16) are built")


def with_defaults(config: Dict[str, Any]) -> Dict[str, Any]:
    cfg = dict(DEFAULTS)
    cfg.update(config)
    return cfg


# ------------------------------------------------------------------------------- containers
class WanSelfAttention(WeightModule):
    def __init__(self, block_index, block_prefix, task, mm_type, config, is_offload_buffer=False, lazy_load=False, lazy_load_file=None):
        super().__init__()
        self.block_index, self.mm_type, self.task, self.config = block_index, mm_type, task, config
        p = f"{block_prefix}.{block_index}"
        extra = (is_offload_buffer, lazy_load, lazy_load_file)
        mm, ln, rms = MM_WEIGHT_REGISTER[mm_type], LN_WEIGHT_REGISTER[config["ln_type"]], RMS_WEIGHT_REGISTER[config["rms_type"]]
        self.add_module("modulation", TENSOR_REGISTER["Default"](f"{p}.modulation", *extra))
        self.add_module("norm1", ln())
        for n in "qkvo":
            self.add_module(f"self_attn_{n}", mm(f"{p}.self_attn.{n}.weight", f"{p}.self_attn.{n}.bias", *extra))
        self.add_module("self_attn_norm_q", rms(f"{p}.self_attn.norm_q.weight", *extra))
        self.add_module("self_attn_norm_k", rms(f"{p}.self_attn.norm_k.weight", *extra))
        self.add_module("self_attn_1", ATTN_WEIGHT_REGISTER[config["self_attn_1_type"]]())
        if config["seq_parallel"]:
            self.add_module("self_attn_1_parallel", ATTN_WEIGHT_REGISTER[config.get("parallel", {}).get("seq_p_attn_type", "hip_ulysses")]())

    def fused_qkv(self):
        """One Linear with the [3D, D] q|k|v weight (one GEMM instead of three); the q/k/v plugins
        are re-pointed at slices of it so the separate copies are released."""
        return self.fused("qkv", [self.self_attn_q, self.self_attn_k, self.self_attn_v])


class WanCrossAttention(WeightModule):
    def __init__(self, block_index, block_prefix, task, mm_type, config, is_offload_buffer=False, lazy_load=False, lazy_load_file=None):
        super().__init__()
        self.block_index, self.mm_type, self.task, self.config = block_index, mm_type, task, config
        p = f"{block_prefix}.{block_index}"
        extra = (is_offload_buffer, lazy_load, lazy_load_file)
        mm, ln, rms = MM_WEIGHT_REGISTER[mm_type], LN_WEIGHT_REGISTER[config["ln_type"]], RMS_WEIGHT_REGISTER[config["rms_type"]]
        self.add_module("norm3", ln(f"{p}.norm3.weight", f"{p}.norm3.bias", *extra))
        for n in "qkvo":
            self.add_module(f"cross_attn_{n}", mm(f"{p}.cross_attn.{n}.weight", f"{p}.cross_attn.{n}.bias", *extra))
        self.add_module("cross_attn_norm_q", rms(f"{p}.cross_attn.norm_q.weight", *extra))
        self.add_module("cross_attn_norm_k", rms(f"{p}.cross_attn.norm_k.weight", *extra))
        self.add_module("cross_attn_1", ATTN_WEIGHT_REGISTER[config["cross_attn_1_type"]]())
        if task in ("i2v", "flf2v", "animate", "s2v") and config.get("use_image_encoder", True):
            self.add_module("cross_attn_k_img", mm(f"{p}.cross_attn.k_img.weight", f"{p}.cross_attn.k_img.bias", *extra))
            self.add_module("cross_attn_v_img", mm(f"{p}.cross_attn.v_img.weight", f"{p}.cross_attn.v_img.bias", *extra))
            self.add_module("cross_attn_norm_k_img", rms(f"{p}.cross_attn.norm_k_img.weight", *extra))
            self.add_module("cross_attn_2", ATTN_WEIGHT_REGISTER[config["cross_attn_2_type"]]())

    def fused_kv(self):
        return self.fused("kv", [self.cross_attn_k, self.cross_attn_v])


class WanFFN(WeightModule):
    def __init__(self, block_index, block_prefix, task, mm_type, config, is_offload_buffer=False, lazy_load=False, lazy_load_file=None):
        super().__init__()
        self.block_index, self.mm_type, self.task, self.config = block_index, mm_type, task, config
        p = f"{block_prefix}.{block_index}"
        extra = (is_offload_buffer, lazy_load, lazy_load_file)
        mm, ln = MM_WEIGHT_REGISTER[mm_type], LN_WEIGHT_REGISTER[config["ln_type"]]
        self.add_module("norm2", ln())
        self.add_module("ffn_0", mm(f"{p}.ffn.0.weight", f"{p}.ffn.0.bias", *extra))
        self.add_module("ffn_2", mm(f"{p}.ffn.2.weight", f"{p}.ffn.2.bias", *extra))


class WanTransformerAttentionBlock(WeightModule):
    def __init__(self, block_index, task, mm_type, config, is_offload_buffer=False, block_prefix="blocks"):
        super().__init__()
        self.block_index, self.mm_type, self.task, self.config = block_index, mm_type, task, config
        args = (block_index, block_prefix, task, mm_type, config, is_offload_buffer, False, None)
        self.compute_phases = WeightModuleList([WanSelfAttention(*args), WanCrossAttention(*args), WanFFN(*args)])
        self.add_module("compute_phases", self.compute_phases)


class WanTransformerWeights(WeightModule):
    """All blocks + the head (norm, head Linear D->out, head modulation)."""

    def __init__(self, config):
        super().__init__()
        config = with_defaults(config)
        self.config = config
        self.blocks_num = config["num_layers"]
        self.task = config["task"]
        self.mm_type = config["dit_quant_scheme"]
        if self.mm_type not in MM_WEIGHT_REGISTER:
            raise KeyError(f"dit_quant_scheme={self.mm_type!r} is not a registered mm-type; known: {sorted(MM_WEIGHT_REGISTER.keys())}")
        self.blocks = WeightModuleList([WanTransformerAttentionBlock(i, self.task, self.mm_type, config) for i in range(self.blocks_num)])
        self.add_module("blocks", self.blocks)
        self.register_parameter("norm", LN_WEIGHT_REGISTER[config["ln_type"]]())
        # the head stays unquantised whatever the block scheme (weights/transformer_weights.py:32)
        self.add_module("head", MM_WEIGHT_REGISTER["hip-bf16"]("head.head.weight", "head.head.bias"))
        self.register_parameter("head_modulation", TENSOR_REGISTER["Default"]("head.modulation"))


# ------------------------------------------------------------------------------- runner io
@dataclass
class WanPreInferModuleOutput:
    """Same fields as lightx2v/models/networks/wan/infer/module_io.py:13-21."""
    embed: torch.Tensor
    grid_sizes: Any
    x: torch.Tensor
    embed0: torch.Tensor
    context: torch.Tensor
    adapter_args: Dict[str, Any] = field(default_factory=dict)
    conditional_dict: Dict[str, Any] = field(default_factory=dict)


class _Scheduler:
    cos_sin: Optional[torch.Tensor] = None


# ------------------------------------------------------------------------------- runner
class WanTransformerInfer:
    def __init__(self, config):
        config = with_defaults(config)
        self.config = config
        self.task = config["task"]
        self.blocks_num = config["num_layers"]
        self.phases_num = 3
        self.num_heads = config["num_heads"]
        self.head_dim = config["dim"] // config["num_heads"]
        self.fused_block = bool(config["fused_block"])
        self.rope_type = config["rope_type"]
        if self.rope_type not in ("hip", "hip_attn"):
            raise ValueError(f"rope_type={self.rope_type!r}: this runner implements 'hip' and 'hip_attn'")
        if config["seq_parallel"]:
            self.seq_p_group = config.get("seq_p_group")
            self.seq_p_fp8_comm = config.get("parallel", {}).get("seq_p_fp8_comm", False)
            if self.rope_type == "hip_attn":
                raise ValueError("rope_type='hip_attn' rotates inside the attention kernel and needs the un-sharded sequence; use 'hip' with seq_parallel")
        else:
            self.seq_p_group = None
            self.seq_p_fp8_comm = False
        self.fp32_residual = sensitive_fp32(config)
        self.scheduler = _Scheduler()
        self.cos_sin: Optional[torch.Tensor] = None  # fp32 [S_local, head_dim] cos || sin
        self.block_idx = 0

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler
        self.scheduler.transformer_infer = self  # common/transformer_infer/transformer_infer.py:10-12

    def get_scheduler_values(self):
        self.cos_sin = self.scheduler.cos_sin

    # ---- top level (transformer_infer.py:65-103) ---------------------------------------------
    @torch.no_grad()
    def infer(self, weights, pre_infer_out):
        self.get_scheduler_values()
        x = self.infer_main_blocks(weights.blocks, pre_infer_out)
        return self.infer_non_blocks(weights, x, pre_infer_out.embed)

    def infer_main_blocks(self, blocks, pre_infer_out):
        x = pre_infer_out.x
        if self.fp32_residual and x.dtype != torch.float32:
            # the reference upcasts at the first residual sum (transformer_infer.py:210-211); the first
            # norm reads the same values either way
            x = x.float()
        for block_idx in range(len(blocks)):
            self.block_idx = block_idx
            x = self.infer_block(blocks[block_idx], x, pre_infer_out)
        return x

    def infer_non_blocks(self, weights, x, e):
        """Head: LN(x) * (1 + e1) + e0, then Linear D -> out (transformer_infer.py:75-97).  `e` is the
        time embedding [1, D], or [S, D] when the timestep is a per-token vector (the reference's
        2-D branch then broadcasts to a per-token modulation)."""
        if e.dim() == 3 and e.shape[0] == 1:
            # the reference's "diffusion forcing" branch (transformer_infer.py:79-82): e [1, S, D] broadcasts to the
            # same per-token modulation as a 2-D e [S, D] does in its first branch
            e = e[0]
        if e.dim() != 2:
            raise NotImplementedError(f"head modulation from an embedding of shape {tuple(e.shape)}: [1, D], [S, D] and [1, S, D] are built")
        table = weights.head_modulation.tensor.float().reshape(2, -1)
        if e.shape[0] == 1:
            mod = table + e.float()
            shift, scale = mod[0].contiguous(), mod[1].contiguous()
        else:
            if e.shape[0] != x.shape[0]:
                raise ValueError(f"head: {e.shape[0]} embedding rows for {x.shape[0]} tokens")
            rows = e.to(torch.bfloat16).contiguous()
            shift, scale = hip.RowMod(table[0].contiguous(), rows), hip.RowMod(table[1].contiguous(), rows)
        x = weights.norm.apply_modulate(x, scale, shift)
        return weights.head.apply(x)

    # ---- one block (transformer_infer.py:105-128) --------------------------------------------
    @torch.no_grad()
    def infer_block(self, block, x, pre_infer_out):
        sa, ca, ff = block.compute_phases
        shift_msa, scale_msa, gate_msa, c_shift_msa, c_scale_msa, c_gate_msa = self.pre_process(sa.modulation, pre_infer_out.embed0)
        y_out = self.infer_self_attn(sa, x, shift_msa, scale_msa, gate_msa)
        x, attn_out = self.infer_cross_attn(ca, x, pre_infer_out.context, y_out, gate_msa)
        y = self.infer_ffn(ff, x, attn_out, c_shift_msa, c_scale_msa, c_gate_msa)
        return self.post_process(x, y, c_gate_msa, pre_infer_out)

    def pre_process(self, modulation, embed0):
        """(modulation[1,6,D] + embed0) -> shift, scale, gate, c_shift, c_scale, c_gate
        (transformer_infer.py:130-142).  embed0 [6, D]: six fp32 [D] vectors (the sum is taken in fp32
        here).  embed0 [S, 6, D] (per-token timestep, Wan2.2 ti2v: the reference's sum broadcasts):
        six hip.RowMod = the table row plus the token's slice of embed0, added inside the kernels."""
        table = modulation.tensor.float().reshape(6, -1)
        if embed0.dim() == 2:
            m = table + embed0.float()
            return tuple(m[i].contiguous() for i in range(6))
        if embed0.dim() == 3 and embed0.shape[1] == 6:
            rows = embed0 if embed0.dtype == torch.bfloat16 else embed0.to(torch.bfloat16)
            return tuple(hip.RowMod(table[i].contiguous(), rows[:, i]) for i in range(6))
        raise NotImplementedError(f"embed0 of shape {tuple(embed0.shape)}: [6, D] and [S, 6, D] are built (not the diffusion-forcing layout)")

    # ---- self attention (transformer_infer.py:144-207) ---------------------------------------
    def infer_self_attn(self, phase, x, shift_msa, scale_msa, gate_msa=None):
        S, D = x.shape
        H, d = self.num_heads, self.head_dim
        rope_in_attn = self.rope_type == "hip_attn"
        if self.fused_block:
            norm1_out = phase.norm1.apply_modulate(x, scale_msa, shift_msa, quant_for=phase.fused_qkv())  # e4m3 rows for a W8A8 projection
            qkv = phase.fused_qkv().apply(norm1_out)  # [S, 3D]
            q2, k2, v2 = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
            table = None if rope_in_attn else self.cos_sin
            phase.self_attn_norm_q.apply_rope_(q2, table, d)
            phase.self_attn_norm_k.apply_rope_(k2, table, d)
            q, k, v = (t.unflatten(1, (H, d)) for t in (q2, k2, v2))
        else:
            norm1_out = phase.norm1.apply_modulate(x, scale_msa, shift_msa)
            q2 = phase.self_attn_norm_q.apply(phase.self_attn_q.apply(norm1_out))
            k2 = phase.self_attn_norm_k.apply(phase.self_attn_k.apply(norm1_out))
            v = phase.self_attn_v.apply(norm1_out).view(S, H, d)
            if not rope_in_attn:
                q2 = hip.rmsnorm_rope(q2, None, self.cos_sin, head_dim=d, do_norm=False)
                k2 = hip.rmsnorm_rope(k2, None, self.cos_sin, head_dim=d, do_norm=False)
            q, k = q2.view(S, H, d), k2.view(S, H, d)

        if self.config["seq_parallel"]:
            par = phase.self_attn_1_parallel
            cu = torch.tensor([0, S], dtype=torch.int32)
            if self.fused_block and hasattr(par, "apply_segmented") and phase.self_attn_o.supports_kseg:
                # the all-to-all result stays in its [source rank][S/N][H/N*d] receive layout; the
                # output projection reads it as a K-segmented A operand (no gather copy)
                op = par.apply_segmented(q, k, v, attention_module=phase.self_attn_1, seq_p_group=self.seq_p_group, use_fp8_comm=self.seq_p_fp8_comm)
                phase.self_attn_o.apply_residual_(op.first, x, gate_msa, kseg=op.kseg)
                return None
            attn_out = par.apply(q=q, k=k, v=v, img_qkv_len=S, cu_seqlens_qkv=cu, attention_module=phase.self_attn_1,
                                 seq_p_group=self.seq_p_group, use_fp8_comm=self.seq_p_fp8_comm, model_cls=self.config["model_cls"])
        else:
            if rope_in_attn:
                phase.self_attn_1.rope_cos_sin = self.cos_sin
            attn_out = phase.self_attn_1.apply(q=q, k=k, v=v, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=S, max_seqlen_kv=S,
                                               model_cls=self.config["model_cls"])
        if self.fused_block:
            # x += (attn Wo^T + b) * gate_msa from the output projection's epilogue
            phase.self_attn_o.apply_residual_(attn_out, x, gate_msa)
            return None
        return phase.self_attn_o.apply(attn_out)

    # ---- cross attention (transformer_infer.py:209-276) --------------------------------------
    def infer_cross_attn(self, phase, x, context, y_out, gate_msa):
        S, D = x.shape
        H, d = self.num_heads, self.head_dim
        if self.task in ("i2v", "flf2v", "animate", "s2v") and self.config.get("use_image_encoder", True):
            context_img, context = context[:257], context[257:]
        else:
            context_img = None
        if self.fused_block:
            # (the gated self-attention residual was already added by infer_self_attn)
            norm3_out = phase.norm3.apply(x, quant_for=phase.cross_attn_q)
            q2 = phase.cross_attn_q.apply(norm3_out)
            phase.cross_attn_norm_q.apply_(q2)
            kv = phase.fused_kv().apply(context)  # [T, 2D]
            k2, v2 = kv[:, :D], kv[:, D:]
            phase.cross_attn_norm_k.apply_(k2)
            q, k, v = q2.view(S, H, d), k2.unflatten(1, (H, d)), v2.unflatten(1, (H, d))
        else:
            hip.gate_residual_(x, y_out, gate_msa)
            norm3_out = phase.norm3.apply(x)
            q = phase.cross_attn_norm_q.apply(phase.cross_attn_q.apply(norm3_out)).view(-1, H, d)
            k = phase.cross_attn_norm_k.apply(phase.cross_attn_k.apply(context)).view(-1, H, d)
            v = phase.cross_attn_v.apply(context).view(-1, H, d)
        attn_out = phase.cross_attn_1.apply(q=q, k=k, v=v, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=q.size(0),
                                            max_seqlen_kv=k.size(0), model_cls=self.config["model_cls"])
        if context_img is not None:
            k_img = phase.cross_attn_norm_k_img.apply(phase.cross_attn_k_img.apply(context_img)).view(-1, H, d)
            v_img = phase.cross_attn_v_img.apply(context_img).view(-1, H, d)
            img_attn_out = phase.cross_attn_2.apply(q=q, k=k_img, v=v_img, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=q.size(0),
                                                    max_seqlen_kv=k_img.size(0), model_cls=self.config["model_cls"])
            hip.gate_residual_(attn_out, img_attn_out, None)
        if self.fused_block:
            phase.cross_attn_o.apply_residual_(attn_out, x, None)  # x += attn_out Wo^T + b
            return x, None
        return x, phase.cross_attn_o.apply(attn_out)

    # ---- FFN (transformer_infer.py:278-310) ---------------------------------------------------
    def infer_ffn(self, phase, x, attn_out, c_shift_msa, c_scale_msa, c_gate_msa=None):
        if not self.fused_block:
            hip.gate_residual_(x, attn_out, None)
        norm2_out = phase.norm2.apply_modulate(x, c_scale_msa, c_shift_msa, quant_for=phase.ffn_0 if self.fused_block else None)
        if self.fused_block:
            hmid = phase.ffn_0.apply_gelu(norm2_out)
            phase.ffn_2.apply_residual_(hmid, x, c_gate_msa)  # x += (h W2^T + b) * c_gate
            return None
        y = hip.gelu_tanh(phase.ffn_0.apply(norm2_out))
        return phase.ffn_2.apply(y)

    # ---- post (transformer_infer.py:312-321) --------------------------------------------------
    def post_process(self, x, y, c_gate_msa, pre_infer_out=None):
        if y is not None:
            hip.gate_residual_(x, y, c_gate_msa)
        return x
