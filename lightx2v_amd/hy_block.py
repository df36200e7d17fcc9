"""HunyuanVideo-1.5 double-stream (MMDiT) block forward on the HIP kernels.

Containers mirror lightx2v/models/networks/hunyuan_video/weights/transformer_weights.py:10-357
(same attribute names and checkpoint keys: `double_blocks.{i}.{img,txt}_{mod.linear, attn_q, attn_k,
attn_v, attn_q_norm, attn_k_norm, attn_proj, mlp.fc1, mlp.fc2}`, `final_layer.*`); the runner
mirrors `HunyuanVideo15TransformerInfer` (lightx2v/models/networks/hunyuan_video/infer/transformer_infer.py:98-270)
method for method.

What differs from the Wan block and how it maps onto the same kernels:
  * two token streams (image, text) with their own weights, ONE joint attention over
    image tokens followed by text tokens: both streams' q|k|v GEMMs write row ranges of one
    [L_img + L_txt, 3D] buffer that the attention kernel reads in place;
  * q/k RMSNorm is per head (weight [head_dim]); RoPE only on the image rows -- one in-place
    pass per (stream, q/k) with the per-head mode of the RMSNorm+RoPE kernel;
  * gates come from a Linear on the conditioning vector (not a learned table + embedding);
  * residuals (`x + y * gate`) run in the GEMM epilogues of the projection / fc2 Linears.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch

from . import hip
from . import plugins  # noqa: F401  (registers the HIP plugins)
from . import ulysses  # noqa: F401  (registers hip_ulysses)
from .registry_factory import ATTN_WEIGHT_REGISTER, LN_WEIGHT_REGISTER, MM_WEIGHT_REGISTER, RMS_WEIGHT_REGISTER
from .weight_module import WeightModule, WeightModuleList

DEFAULTS = {
    "attn_type": "hip_flash_attn",
    "dit_quant_scheme": "hip-bf16",
    "ln_type": "hip",
    "rms_type": "hip",
    "fused_block": True,
    "seq_parallel": False,
    "cpu_offload": False,
    "task": "t2v",
}


def with_defaults(config: Dict[str, Any]) -> Dict[str, Any]:
    cfg = dict(DEFAULTS)
    cfg.update(config)
    return cfg


# ------------------------------------------------------------------------------- io
@dataclass
class HunyuanVideo15InferModuleOutput:
    """Same fields as lightx2v/models/networks/hunyuan_video/infer/module_io.py:6-11."""
    img: torch.Tensor   # [1, L, D]
    txt: torch.Tensor   # [1, Lt, D]
    vec: torch.Tensor   # [1, D] (silu of the time embedding)
    grid_sizes: tuple


# ------------------------------------------------------------------------------- containers
class _Branch(WeightModule):
    """One stream's weights (transformer_weights.py:77-318); `br` is "img" or "txt"."""

    def __init__(self, br, block_index, task, config, block_prefix="double_blocks", is_offload_buffer=False):
        super().__init__()
        self.br, self.block_index, self.task, self.config = br, block_index, task, config
        self.mm_type, self.ln_type, self.rms_type = config["dit_quant_scheme"], config["ln_type"], config["rms_type"]
        mm, ln, rms = MM_WEIGHT_REGISTER[self.mm_type], LN_WEIGHT_REGISTER[self.ln_type], RMS_WEIGHT_REGISTER[self.rms_type]
        p = f"{block_prefix}.{block_index}.{br}_"
        extra = (is_offload_buffer, False, None)
        self.add_module(f"{br}_mod", mm(p + "mod.linear.weight", p + "mod.linear.bias", *extra))
        self.add_module(f"{br}_norm1", ln())
        for n in ("attn_q", "attn_k", "attn_v"):
            self.add_module(f"{br}_{n}", mm(f"{p}{n}.weight", f"{p}{n}.bias", *extra))
        self.add_module(f"{br}_attn_q_norm", rms(p + "attn_q_norm.weight", *extra))
        self.add_module(f"{br}_attn_k_norm", rms(p + "attn_k_norm.weight", *extra))
        self.add_module(f"{br}_attn_proj", mm(p + "attn_proj.weight", p + "attn_proj.bias", *extra))
        self.add_module(f"{br}_norm2", ln())
        self.add_module(f"{br}_mlp_fc1", mm(p + "mlp.fc1.weight", p + "mlp.fc1.bias", *extra))
        self.add_module(f"{br}_mlp_fc2", mm(p + "mlp.fc2.weight", p + "mlp.fc2.bias", *extra))

    def part(self, name):
        return getattr(self, f"{self.br}_{name}")

    def fused_qkv(self):
        """One Linear with the [3D, D] q|k|v weight; the three plugins then alias it."""
        return self.fused("qkv", [self.part("attn_q"), self.part("attn_k"), self.part("attn_v")])


class MMDoubleStreamBlockImgBranch(_Branch):
    def __init__(self, block_index, task, config, block_prefix="double_blocks", is_offload_buffer=False):
        super().__init__("img", block_index, task, config, block_prefix, is_offload_buffer)


class MMDoubleStreamBlockTxtBranch(_Branch):
    def __init__(self, block_index, task, config, block_prefix="double_blocks", is_offload_buffer=False):
        super().__init__("txt", block_index, task, config, block_prefix, is_offload_buffer)


class MMDoubleStreamBlock(WeightModule):
    def __init__(self, block_index, task, config, block_prefix="double_blocks", is_offload_buffer=False):
        super().__init__()
        self.block_index, self.task, self.config = block_index, task, config
        self.add_module("img_branch", MMDoubleStreamBlockImgBranch(block_index, task, config, block_prefix, is_offload_buffer))
        self.add_module("txt_branch", MMDoubleStreamBlockTxtBranch(block_index, task, config, block_prefix, is_offload_buffer))
        self.add_module("self_attention", ATTN_WEIGHT_REGISTER[config["attn_type"]]())
        if config["seq_parallel"]:
            self.add_module("self_attention_parallel", ATTN_WEIGHT_REGISTER[config.get("parallel", {}).get("seq_p_attn_type", "hip_ulysses")]())


class FinalLayerWeights(WeightModule):
    """transformer_weights.py:321-357: never quantised (mm-type "Default" there, bf16 here)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        mm = MM_WEIGHT_REGISTER["hip-bf16"]
        self.add_module("adaLN_modulation", mm("final_layer.adaLN_modulation.1.weight", "final_layer.adaLN_modulation.1.bias"))
        self.add_module("linear", mm("final_layer.linear.weight", "final_layer.linear.bias"))
        self.add_module("norm_final", LN_WEIGHT_REGISTER[config["ln_type"]]())


class HunyuanVideo15TransformerWeights(WeightModule):
    def __init__(self, config):
        super().__init__()
        config = with_defaults(config)
        self.config = config
        self.task = config["task"]
        self.mm_type = config["dit_quant_scheme"]
        if self.mm_type not in MM_WEIGHT_REGISTER:
            raise KeyError(f"dit_quant_scheme={self.mm_type!r} is not a registered mm-type; known: {sorted(MM_WEIGHT_REGISTER.keys())}")
        if config["cpu_offload"]:
            raise NotImplementedError("block offload buffers are not built: 288 GB of HBM hold the whole model")
        self.double_blocks_num = config["mm_double_blocks_depth"]
        self.add_module("double_blocks", WeightModuleList([MMDoubleStreamBlock(i, self.task, config) for i in range(self.double_blocks_num)]))
        self.add_module("final_layer", FinalLayerWeights(config))

    def non_block_weights_to_cuda(self):
        self.final_layer.to_cuda()

    def non_block_weights_to_cpu(self):
        self.final_layer.to_cpu()


# ------------------------------------------------------------------------------- runner
class _Scheduler:
    cos_sin: Optional[torch.Tensor] = None


def _rows(t: torch.Tensor) -> torch.Tensor:
    """[1, L, D] or [L, D] -> [L, D] view."""
    return t[0] if t.dim() == 3 else t


class HunyuanVideo15TransformerInfer:
    def __init__(self, config):
        config = with_defaults(config)
        self.config = config
        self.double_blocks_num = config["mm_double_blocks_depth"]
        self.heads_num = config["heads_num"]
        self.fused_block = bool(config["fused_block"])
        if config["seq_parallel"]:
            self.seq_p_group = config.get("seq_p_group")
            self.seq_p_fp8_comm = config.get("parallel", {}).get("seq_p_fp8_comm", False)
        else:
            self.seq_p_group = None
            self.seq_p_fp8_comm = False
        self.scheduler = _Scheduler()
        self.infer_func = self.infer_without_offload

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler
        self.scheduler.transformer_infer = self

    # ---- top level (transformer_infer.py:124-142) ---------------------------------------------
    @torch.no_grad()
    def infer(self, weights, infer_module_out):
        self.infer_func(weights, infer_module_out)
        return self.infer_final_layer(weights, infer_module_out)

    @torch.no_grad()
    def infer_without_offload(self, weights, infer_module_out):
        # the blocks update their streams in place: work on private copies so the caller's
        # pre-infer tensors survive (the reference rebinds fresh tensors each block)
        infer_module_out.img = infer_module_out.img.clone()
        infer_module_out.txt = infer_module_out.txt.clone()
        for i in range(self.double_blocks_num):
            infer_module_out.img, infer_module_out.txt = self.infer_double_block(weights.double_blocks[i], infer_module_out, _inplace=True)

    @torch.no_grad()
    def infer_final_layer(self, weights, infer_module_out):
        img = _rows(infer_module_out.img)
        mod = weights.final_layer.adaLN_modulation.apply(infer_module_out.vec).float()
        shift, scale = mod.chunk(2, dim=1)
        img = weights.final_layer.norm_final.apply_modulate(img, scale[0].contiguous(), shift[0].contiguous())
        return weights.final_layer.linear.apply(img).unsqueeze(0)

    # ---- one block (transformer_infer.py:144-151) ---------------------------------------------
    @torch.no_grad()
    def infer_double_block(self, weights, infer_module_out, _inplace=False):
        img, txt = _rows(infer_module_out.img), _rows(infer_module_out.txt)
        if not _inplace:
            img, txt = img.clone(), txt.clone()
        vec = infer_module_out.vec
        L, D = img.shape
        Lt = txt.shape[0]
        H = self.heads_num
        d = D // H
        ib, tb = weights.img_branch, weights.txt_branch
        imod = self._mod(ib, vec, D)
        tmod = self._mod(tb, vec, D)

        qkv = torch.empty((L + Lt, 3 * D), dtype=img.dtype, device=img.device)
        self._branch_before_attn(ib, img, imod, qkv[:L], d, self.scheduler.cos_sin)
        self._branch_before_attn(tb, txt, tmod, qkv[L:], d, None)
        attn = self._infer_attn(weights, qkv, L, H, d)
        if isinstance(attn, tuple):
            # sequence parallel, fused runner: the image rows' result stays in the exchange's receive layout and the
            # output projection (bf16 or W8A8) reads it as a K-segmented operand -- no gather copy
            op, txt_attn = attn
            self._branch_after_attn(ib, op.first, img, imod, kseg=op.kseg)
            self._branch_after_attn(tb, txt_attn, txt, tmod)
        else:
            self._branch_after_attn(ib, attn[:L], img, imod)
            self._branch_after_attn(tb, attn[L:], txt, tmod)
        return img.unsqueeze(0), txt.unsqueeze(0)

    @staticmethod
    def _mod(branch, vec, D):
        """(shift1, scale1, gate1, shift2, scale2, gate2), each fp32 [D] (transformer_infer.py:155-162)."""
        m = branch.part("mod").apply(vec).float().view(6, D)
        return tuple(m[i].contiguous() for i in range(6))

    def _branch_before_attn(self, b, x, mod, qkv_rows, d, cos_sin):
        """LN + modulate, q|k|v projections into `qkv_rows` [rows, 3D], per-head RMSNorm of q and k
        (+ RoPE when a table is given), all in place (transformer_infer.py:153-216)."""
        D = x.shape[1]
        y = b.part("norm1").apply_modulate(x, mod[1], mod[0], quant_for=b.fused_qkv() if self.fused_block else None)
        if self.fused_block:
            b.fused_qkv().apply(y, out=qkv_rows)
        else:
            for i, n in enumerate(("attn_q", "attn_k", "attn_v")):
                b.part(n).apply(y, out=qkv_rows[:, i * D:(i + 1) * D])
        b.part("attn_q_norm").apply_per_head_rope_(qkv_rows[:, :D], cos_sin, d)
        b.part("attn_k_norm").apply_per_head_rope_(qkv_rows[:, D:2 * D], cos_sin, d)

    def _infer_attn(self, weights, qkv, img_len, H, d):
        """Joint attention over image rows followed by text rows (transformer_infer.py:218-250)."""
        D = H * d
        S = qkv.shape[0]
        q, k, v = (qkv[:, i * D:(i + 1) * D].unflatten(1, (H, d)) for i in range(3))
        if self.config["seq_parallel"]:
            par = weights.self_attention_parallel
            if self.fused_block and hasattr(par, "apply_segmented_joint") and getattr(weights.img_branch.part("attn_proj"), "supports_kseg", False) and S > img_len:
                return par.apply_segmented_joint(q, k, v, img_len, attention_module=weights.self_attention, seq_p_group=self.seq_p_group,
                                                 use_fp8_comm=self.seq_p_fp8_comm)
            cu = torch.tensor([0, S], dtype=torch.int32)
            return weights.self_attention_parallel.apply(q=q, k=k, v=v, img_qkv_len=img_len, cu_seqlens_qkv=cu, attention_module=weights.self_attention,
                                                         seq_p_group=self.seq_p_group, use_fp8_comm=self.seq_p_fp8_comm)
        return weights.self_attention.apply(q=q, k=k, v=v, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=S, max_seqlen_kv=S)

    def _branch_after_attn(self, b, attn_rows, x, mod, kseg=None):
        """x += proj(attn) * gate1; x += fc2(gelu(fc1(modulate(LN(x))))) * gate2, in place
        (transformer_infer.py:252-270).  `kseg`: `attn_rows` are the rows of the first segment of a
        K-segmented operand (fused runner only)."""
        if self.fused_block:
            b.part("attn_proj").apply_residual_(attn_rows, x, mod[2], kseg=kseg)
            h = b.part("mlp_fc1").apply_gelu(b.part("norm2").apply_modulate(x, mod[4], mod[3], quant_for=b.part("mlp_fc1")))
            b.part("mlp_fc2").apply_residual_(h, x, mod[5])
        else:
            hip.gate_residual_(x, b.part("attn_proj").apply(attn_rows), mod[2])
            h = hip.gelu_tanh(b.part("mlp_fc1").apply(b.part("norm2").apply_modulate(x, mod[4], mod[3])))
            hip.gate_residual_(x, b.part("mlp_fc2").apply(h), mod[5])
