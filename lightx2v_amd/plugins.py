"""HIP-backed op classes behind the reference's plugin surface.

Every class keeps the constructor arguments, method names and return conventions of the
reference class it stands in for, so a `WanTransformerInfer`-style caller that only uses
`.load()` / `.apply()` / `.to_cuda()` / `.state_dict()` works unchanged:

  ATTN_WEIGHT_REGISTER["hip_flash_attn"]  <-> FlashAttn2Weight / TorchSDPAWeight
        (lightx2v/common/ops/attn/flash_attn.py:29-61, torch_sdpa.py:9-39)
  ATTN_WEIGHT_REGISTER["hip_ulysses"]     <-> UlyssesAttnWeight (ulysses_attn.py:11-137)  [in ulysses.py]
  MM_WEIGHT_REGISTER["hip-bf16"]          <-> MMWeight "Default" (lightx2v/common/ops/mm/mm_weight.py:119-201)
  LN_WEIGHT_REGISTER["hip"]               <-> LNWeight "Default" (lightx2v/common/ops/norm/layer_norm_weight.py:135-169)
  RMS_WEIGHT_REGISTER["hip"]              <-> RMSWeightSgl "sgl-kernel" (lightx2v/common/ops/norm/rms_norm_weight.py:112-146)
  TENSOR_REGISTER["Default"]              <-> DefaultTensor (lightx2v/common/ops/tensor/tensor.py:9-78)

All arithmetic runs in liblx2v_hip.so; there is no torch fallback (a CPU tensor raises).
Extra `apply_*` methods expose the fused forms the block runner uses.
"""

from __future__ import annotations

import re
from typing import Optional

import torch

from . import hip
from .registry_factory import (
    ATTN_WEIGHT_REGISTER,
    LN_WEIGHT_REGISTER,
    MM_WEIGHT_REGISTER,
    RMS_WEIGHT_REGISTER,
    TENSOR_REGISTER,
)

COMPUTE_DTYPE = torch.bfloat16


def _renumber(name: str, index: int) -> str:
    """First `.N` in a checkpoint key -> `.index` (per-block buffer reuse, mm_weight.py:180-199)."""
    return re.sub(r"\.\d+", lambda m: f".{index}", name, count=1)


def _device_ready(t: Optional[torch.Tensor], dtype) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t.to(dtype=dtype).contiguous()


def _cat_bias(mods) -> Optional[torch.Tensor]:
    """Concatenated fp32 bias of Linear plugins that are fused row-wise; a member without a bias
    contributes zeros, and no member having one gives None."""
    if all(m.bias is None for m in mods):
        return None
    like = next(m.bias for m in mods if m.bias is not None)
    return torch.cat([m.bias if m.bias is not None else torch.zeros(m.weight_nk.shape[0], dtype=like.dtype, device=like.device) for m in mods]).contiguous()


def _one_sequence(who: str, name: str, cu, max_seqlen, rows: int):
    """The kernels attend over every row they are given, so a varlen description that says
    anything else (more than one sequence, or fewer rows than the tensor holds: the reference's
    flash_attn_varlen_func would stop at cu_seqlens[-1]) is refused instead of mis-computed."""
    if cu is not None:
        if cu.numel() != 2:
            raise NotImplementedError(f"{who}: {name} must describe one sequence, got {cu.numel() - 1}")
        if not cu.is_cuda and (int(cu[0]) != 0 or int(cu[-1]) != rows):  # a host tensor costs no device sync to read
            raise ValueError(f"{who}: {name}={cu.tolist()} does not cover the {rows} rows of the tensor; slice the tensor to the sequence")
    if max_seqlen is not None and int(max_seqlen) != rows:
        raise ValueError(f"{who}: max_seqlen={int(max_seqlen)} differs from the tensor's {rows} rows; slice the tensor to the sequence")


# ----------------------------------------------------------------------------------- attention
class _AttnBase:
    """No-op weight plumbing every attention plugin inherits (attn/template.py:4-32)."""

    def __init__(self):
        self.config = {}

    def load(self, weight_dict):
        pass

    def set_config(self, config=None):
        if config is not None:
            self.config = config

    def to_cpu(self, non_blocking=False):
        pass

    def to_cuda(self, non_blocking=False):
        pass

    def state_dict(self, destination=None):
        return {} if destination is None else destination

    def load_state_dict(self, destination, block_index, adapter_block_index=None):
        return {}


@ATTN_WEIGHT_REGISTER("hip_flash_attn")
class HipFlashAttnWeight(_AttnBase):
    """softmax(q k^T / sqrt(d)) v on MFMA.  Non-causal, unmasked, one sequence (the only form the
    Wan block uses: cu_seqlens = [0, S]).  `rope_cos_sin` (set by the caller before `apply`, fp32
    [S, d] cos||sin) makes the kernel rotate q and k on load; it is consumed by one call."""

    _workspaces = {}   # per device: partial results of key-split launches, shared by every block (blocks run one after the other on one stream)

    def __init__(self):
        super().__init__()
        self.rope_cos_sin: Optional[torch.Tensor] = None

    @classmethod
    def _workspace_for(cls, q, k):
        """A launch of few heads over a long sequence (a head chunk of a sequence-parallel rank) may split the keys
        of its last round; the partial results live in one buffer per device that grows on demand."""
        need = hip.attn_workspace_bytes(q.shape[0], k.shape[0], q.shape[1])
        if need == 0:
            return None
        ws = cls._workspaces.get(q.device)
        if ws is None or ws.numel() < need:
            ws = cls._workspaces[q.device] = torch.empty(need, dtype=torch.uint8, device=q.device)
        return ws

    def apply(self, q, k, v, drop_rate=0, attn_mask=None, causal=False, cu_seqlens_q=None, cu_seqlens_kv=None,
              max_seqlen_q=None, max_seqlen_kv=None, model_cls=None):
        if drop_rate not in (0, 0.0) or attn_mask is not None or causal:
            raise NotImplementedError("hip_flash_attn: dropout, masks and causal attention are not part of the Wan block path")
        if q.dim() == 4:  # [B,S,H,d] with B == 1 is flattened like flash_attn.py:47-51
            if q.shape[0] != 1:
                raise NotImplementedError("hip_flash_attn: batch > 1 is not part of the Wan block path")
            q, k, v = q[0], k[0], v[0]
        _one_sequence("hip_flash_attn", "cu_seqlens_q", cu_seqlens_q, max_seqlen_q, q.shape[0])
        _one_sequence("hip_flash_attn", "cu_seqlens_kv", cu_seqlens_kv, max_seqlen_kv, k.shape[0])
        rope, self.rope_cos_sin = self.rope_cos_sin, None
        return hip.attn_fwd(q, k, v, rope_cos_sin=rope, workspace=None if rope is not None else self._workspace_for(q, k))


@ATTN_WEIGHT_REGISTER("hip_fp8_attn")
class HipFp8AttnWeight(_AttnBase):
    """The same attention with both contractions on the block-scaled fp8 MFMA (q, k in e4m3 with a
    power-of-two scale per 32 channels, v per 32 keys, P rounded to e4m3; fp32 accumulation).
    Selected where the reference would select a quantised backend such as `sage_attn2`
    (lightx2v/common/ops/attn/sage_attn.py:36-64).  The quantisation workspace is kept between calls
    and grows on demand."""

    def __init__(self):
        super().__init__()
        self._workspace: Optional[torch.Tensor] = None

    def apply(self, q, k, v, drop_rate=0, attn_mask=None, causal=False, cu_seqlens_q=None, cu_seqlens_kv=None,
              max_seqlen_q=None, max_seqlen_kv=None, model_cls=None):
        if drop_rate not in (0, 0.0) or attn_mask is not None or causal:
            raise NotImplementedError("hip_fp8_attn: dropout, masks and causal attention are not part of the block path")
        if q.dim() == 4:
            if q.shape[0] != 1:
                raise NotImplementedError("hip_fp8_attn: batch > 1 is not part of the block path")
            q, k, v = q[0], k[0], v[0]
        _one_sequence("hip_fp8_attn", "cu_seqlens_q", cu_seqlens_q, max_seqlen_q, q.shape[0])
        _one_sequence("hip_fp8_attn", "cu_seqlens_kv", cu_seqlens_kv, max_seqlen_kv, k.shape[0])
        need = hip.attn_fp8_workspace_bytes(q.shape[0], k.shape[0], q.shape[1])
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != q.device:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
        return hip.attn_fwd_fp8(q, k, v, workspace=self._workspace)


@ATTN_WEIGHT_REGISTER("hip_fp8_attn_linear")
class HipFp8AttnLinearWeight(HipFp8AttnWeight):
    """`hip_fp8_attn` with the exp-free P: the e4m3 code round(8 x + 56) instead of exp2 + rounding
    (2^x with a linear mantissa; the row sums run over the same codes, so the common factor of the
    approximation divides out).  ~11 % less kernel time for a cosine against the fp32 softmax of
    0.9983 instead of 0.9986 -- the kind of trade the reference's quantised backends make
    (lightx2v/common/ops/attn/sage_attn.py:36-64).  The choice is a process-wide kernel option, set
    around the launch."""

    def apply(self, *args, **kwargs):
        prev = hip.get_option("attn_fp8_p")
        hip.set_option("attn_fp8_p", 1)
        try:
            return super().apply(*args, **kwargs)
        finally:
            hip.set_option("attn_fp8_p", prev)


# ----------------------------------------------------------------------------------- matmul
@MM_WEIGHT_REGISTER("hip-bf16")
class HipMMWeight:
    """y = x W^T + b with the checkpoint's [out, in] weight used as stored (it is already the
    K-contiguous operand MFMA wants; the reference keeps the same memory as a `.t()` view)."""

    def __init__(self, weight_name, bias_name, create_cuda_buffer=False, lazy_load=False, lazy_load_file=None, is_post_adapter=False):
        self.weight_name = weight_name
        self.bias_name = bias_name
        self.create_cuda_buffer = create_cuda_buffer
        self.lazy_load = lazy_load
        self.lazy_load_file = lazy_load_file
        self.is_post_adapter = is_post_adapter
        self.config = {}
        self.weight_nk: Optional[torch.Tensor] = None  # [N, K] bf16
        self.bias: Optional[torch.Tensor] = None       # [N] fp32

    # -- reference-compatible views
    @property
    def weight(self):
        return None if self.weight_nk is None else self.weight_nk.t()

    def set_config(self, config={}):
        self.config = config

    supports_kseg = True  # apply_residual_ can read a K-segmented A operand
    accepts_quantized = False  # input_tensor may be hip.QuantRows (e4m3 rows + per-row scales)

    def adopt(self, weight_nk: torch.Tensor, bias: Optional[torch.Tensor]):
        """Use existing storage (e.g. a slice of a fused q|k|v weight) instead of copying."""
        self.weight_nk = weight_nk
        self.bias = bias

    @classmethod
    def fused(cls, mods):
        """One plugin whose weight is the row-wise concatenation of `mods`' weights (q|k|v, k|v):
        one GEMM instead of several.  The members are re-pointed at slices of the fused storage so
        their separate copies are released."""
        out = cls(None, None)
        w = torch.cat([m.weight_nk for m in mods], dim=0).contiguous()
        b = _cat_bias(mods)
        out.adopt(w, b)
        o = 0
        for m in mods:
            n = m.weight_nk.shape[0]
            m.adopt(w[o:o + n], None if (b is None or m.bias is None) else b[o:o + n])
            o += n
        return out

    def load(self, weight_dict):
        w = weight_dict[self.weight_name]
        b = weight_dict[self.bias_name] if self.bias_name is not None else None
        self.weight_nk = _device_ready(w, COMPUTE_DTYPE)
        self.bias = _device_ready(b, torch.float32)
        if self.create_cuda_buffer:
            self.weight_cuda_buffer = self.weight_nk.cuda()
            self.bias_cuda_buffer = None if self.bias is None else self.bias.cuda()

    def _calculate_size(self):
        n = self.weight_nk.numel() * self.weight_nk.element_size()
        if self.bias is not None:
            n += self.bias.numel() * self.bias.element_size()
        return n

    def to_cuda(self, non_blocking=False):
        self.weight_nk = self.weight_nk.cuda(non_blocking=non_blocking)
        if self.bias is not None:
            self.bias = self.bias.cuda(non_blocking=non_blocking)

    def to_cpu(self, non_blocking=False):
        self.weight_nk = self.weight_nk.to("cpu", non_blocking=non_blocking)
        if self.bias is not None:
            self.bias = self.bias.to("cpu", non_blocking=non_blocking)

    def state_dict(self, destination=None):
        if destination is None:
            destination = {}
        destination[self.weight_name] = self.weight_nk
        if self.bias_name is not None:
            destination[self.bias_name] = self.bias
        return destination

    def load_state_dict(self, destination, block_index, adapter_block_index=None):
        idx = adapter_block_index if self.is_post_adapter else block_index
        assert idx is not None
        wname = _renumber(self.weight_name, idx)
        if wname not in destination:
            self.weight_nk = None
            return
        self.weight_nk = self.weight_cuda_buffer.copy_(destination[wname], non_blocking=True)
        if self.bias_name is not None:
            self.bias = self.bias_cuda_buffer.copy_(destination[_renumber(self.bias_name, idx)], non_blocking=True)
        else:
            self.bias = None

    # -- compute
    def apply(self, input_tensor, out=None):
        """`out`: optional preallocated [M, N] destination (row stride free), e.g. a row range of a
        larger buffer; the reference signature has no such slot and allocates."""
        return hip.gemm(input_tensor, self.weight_nk, self.bias, out=out, workspace=self._workspace(input_tensor))

    def _workspace(self, input_tensor, kseg=None):
        """The device's shared GEMM workspace when this product's last round of tiles is worth a K split
        (every Linear of a forward runs on one stream, so one workspace per device serves them all)."""
        return None if kseg is not None else hip.gemm_workspace_for(input_tensor, *self.weight_nk.shape)

    def apply_gelu(self, input_tensor):
        return hip.gemm(input_tensor, self.weight_nk, self.bias, epilogue=hip.EPI_GELU_TANH, workspace=self._workspace(input_tensor))

    def apply_residual_(self, input_tensor, resid, gate=None, kseg=None):
        """resid <- resid + (x W^T + b) * gate, in place; returns resid.  `kseg` (hip.KSeg): the
        input's K columns are stored as segments, `input_tensor` being the rows of the first one.
        A per-token gate (hip.RowMod) does not fit the GEMM epilogue's [N] gate vector: the product is
        formed first and the gated add is its own pass."""
        if isinstance(gate, hip.RowMod) or resid.dtype == torch.float32:   # (an fp32 residual stream is not the epilogue's bf16 slot either)
            return hip.gate_residual_(resid, hip.gemm(input_tensor, self.weight_nk, self.bias, kseg=kseg, workspace=self._workspace(input_tensor, kseg)), gate)
        return hip.gemm(input_tensor, self.weight_nk, self.bias, epilogue=hip.EPI_RESIDUAL, resid=resid, gate=gate, out=resid, kseg=kseg,
                        workspace=self._workspace(input_tensor, kseg))


@MM_WEIGHT_REGISTER("hip-fp8")
class HipFp8MMWeight(HipMMWeight):
    """W8A8 in OCP e4m3 on the CDNA4 fp8 MFMA: per-output-channel weight scale, dynamic per-token
    activation scale, bf16 output -- the semantics of the reference's "fp8-sgl" / "fp8-vllm" /
    "fp8-q8f" schemes (lightx2v/common/ops/mm/mm_weight.py:572-605,806-834,901-928).  Loads either
    a pre-quantised checkpoint (`<name>.weight` in float8_e4m3fn + `<name>.weight_scale`, the
    reference's `load_quantized`, :269-290) or, like its `weight_auto_quant`, quantises a bf16/fp32
    weight at load time."""

    def __init__(self, weight_name, bias_name, create_cuda_buffer=False, lazy_load=False, lazy_load_file=None, is_post_adapter=False):
        super().__init__(weight_name, bias_name, create_cuda_buffer, lazy_load, lazy_load_file, is_post_adapter)
        self.weight_scale_name = None if weight_name is None else weight_name.removesuffix(".weight") + ".weight_scale"
        self.weight_scale: Optional[torch.Tensor] = None  # [N] fp32

    def load(self, weight_dict):
        w = weight_dict[self.weight_name]
        b = weight_dict[self.bias_name] if self.bias_name is not None else None
        self.bias = _device_ready(b, torch.float32)
        if w.dtype == torch.float8_e4m3fn:
            self.weight_nk = w.contiguous()
            self.weight_scale = weight_dict[self.weight_scale_name].to(torch.float32).reshape(-1).contiguous()
        else:
            if not w.is_cuda:
                if not torch.cuda.is_available():
                    raise RuntimeError("hip-fp8: quantising a weight at load time runs on the GPU; no GPU is visible (no CPU fallback exists)")
                w = w.cuda()
            # quantised from the checkpoint's own values: fp32 (and fp16, widened) rows go in as fp32, like the
            # reference's `weight.to(torch.float32)` in front of its quantiser (mm_weight.py:312-320)
            self.weight_nk, self.weight_scale = hip.quant_fp8_rowwise((w if w.dtype == COMPUTE_DTYPE else w.to(torch.float32)).contiguous())

    def _calculate_size(self):
        return super()._calculate_size() + self.weight_scale.numel() * 4

    def to_cuda(self, non_blocking=False):
        super().to_cuda(non_blocking)
        self.weight_scale = self.weight_scale.cuda(non_blocking=non_blocking)

    def to_cpu(self, non_blocking=False):
        super().to_cpu(non_blocking)
        self.weight_scale = self.weight_scale.to("cpu", non_blocking=non_blocking)

    def state_dict(self, destination=None):
        destination = super().state_dict(destination)
        destination[self.weight_scale_name] = self.weight_scale
        return destination

    def load_state_dict(self, destination, block_index, adapter_block_index=None):
        raise NotImplementedError("hip-fp8: block-streamed (offload) loading is not built")

    supports_kseg = True  # the row-wise activation quantiser reads the segments and writes the plain e4m3 matrix
    accepts_quantized = True  # a producer that holds whole rows (LayerNorm) may hand over hip.QuantRows

    def adopt(self, weight_nk, bias, weight_scale=None):
        if weight_scale is None:
            raise ValueError("hip-fp8: adopting storage needs the per-channel weight scale with it")
        self.weight_nk, self.bias, self.weight_scale = weight_nk, bias, weight_scale

    @classmethod
    def fused(cls, mods):
        out = cls(None, None)
        w = torch.cat([m.weight_nk for m in mods], dim=0).contiguous()
        b = _cat_bias(mods)
        sc = torch.cat([m.weight_scale for m in mods], dim=0).contiguous()
        out.adopt(w, b, sc)
        o = 0
        for m in mods:
            n = m.weight_nk.shape[0]
            m.adopt(w[o:o + n], None if (b is None or m.bias is None) else b[o:o + n], sc[o:o + n])
            o += n
        return out

    def _run(self, x, epilogue, resid=None, gate=None, out=None, kseg=None):
        if isinstance(x, hip.QuantRows):
            xq, xs = x.q, x.scale
        else:
            xq, xs = hip.quant_fp8_rowwise(x, kseg=kseg, dim=self.weight_nk.shape[1])
        return hip.gemm_fp8(xq, xs, self.weight_nk, self.weight_scale, self.bias, epilogue=epilogue, resid=resid, gate=gate,
                            out=resid if epilogue == hip.EPI_RESIDUAL else out, workspace=hip.gemm_w8a8_workspace_for(xq, *self.weight_nk.shape))

    def apply(self, input_tensor, out=None):
        return self._run(input_tensor, hip.EPI_NONE, out=out)

    def apply_gelu(self, input_tensor):
        return self._run(input_tensor, hip.EPI_GELU_TANH)

    def apply_residual_(self, input_tensor, resid, gate=None, kseg=None):
        """`kseg`: `input_tensor` holds the rows of the first segment of a K-segmented operand (the
        sequence-parallel attention result in its receive layout); the activation quantiser reads the
        segments in place, so the gather copy of the bf16 path's alternative never happens."""
        if isinstance(gate, hip.RowMod) or resid.dtype == torch.float32:  # per-token gate / fp32 residual stream: product first, gated add as its own pass
            return hip.gate_residual_(resid, self._run(input_tensor, hip.EPI_NONE, kseg=kseg), gate)
        return self._run(input_tensor, hip.EPI_RESIDUAL, resid=resid, gate=gate, kseg=kseg)


@MM_WEIGHT_REGISTER("hip-int8")
class HipInt8MMWeight(HipFp8MMWeight):
    """W8A8 in int8 on the int8 MFMA: per-output-channel symmetric weight scale, dynamic per-token
    symmetric activation scale, exact int32 accumulation, bf16 output -- the semantics of the
    reference's "int8-vllm" / "int8-sgl" / "int8-q8f" / "int8-torchao" schemes
    (lightx2v/common/ops/mm/mm_weight.py:607-641,836-866,931-990).  Loads a pre-quantised checkpoint
    (`<name>.weight` int8 + `<name>.weight_scale`, `load_quantized`) or, like `weight_auto_quant`
    (:322-330), quantises a bf16 / fp32 weight at load time with the reference's per-channel formula."""

    accepts_quantized = False  # the LayerNorm kernel's fused output is e4m3

    def load(self, weight_dict):
        w = weight_dict[self.weight_name]
        b = weight_dict[self.bias_name] if self.bias_name is not None else None
        self.bias = _device_ready(b, torch.float32)
        if w.dtype == torch.int8:
            self.weight_nk = w.contiguous()
            self.weight_scale = weight_dict[self.weight_scale_name].to(torch.float32).reshape(-1).contiguous()
        else:
            if not w.is_cuda:
                if not torch.cuda.is_available():
                    raise RuntimeError("hip-int8: quantising a weight at load time runs on the GPU; no GPU is visible (no CPU fallback exists)")
                w = w.cuda()
            # from the checkpoint's own values (mm_weight.py:324-327 quantises `weight.to(torch.float32)`)
            self.weight_nk, self.weight_scale = hip.quant_int8_rowwise((w if w.dtype == COMPUTE_DTYPE else w.to(torch.float32)).contiguous(), weight_mode=True)

    def load_state_dict(self, destination, block_index, adapter_block_index=None):
        raise NotImplementedError("hip-int8: block-streamed (offload) loading is not built")

    def _run(self, x, epilogue, resid=None, gate=None, out=None, kseg=None):
        xq, xs = hip.quant_int8_rowwise(x, kseg=kseg, dim=self.weight_nk.shape[1])
        return hip.gemm_int8(xq, xs, self.weight_nk, self.weight_scale, self.bias, epilogue=epilogue, resid=resid, gate=gate,
                             out=resid if epilogue == hip.EPI_RESIDUAL else out, workspace=hip.gemm_w8a8_workspace_for(xq, *self.weight_nk.shape))


# ----------------------------------------------------------------------------------- norms
def _all_take_quantized(consumers) -> bool:
    if consumers is None:
        return False
    if not isinstance(consumers, (list, tuple)):
        consumers = (consumers,)
    return len(consumers) > 0 and all(getattr(c, "accepts_quantized", False) for c in consumers)


class _NormBase:
    def __init__(self, create_cuda_buffer, lazy_load, lazy_load_file, is_post_adapter, eps):
        self.create_cuda_buffer = create_cuda_buffer
        self.lazy_load = lazy_load
        self.lazy_load_file = lazy_load_file
        self.is_post_adapter = is_post_adapter
        self.eps = eps
        self.config = {}

    def set_config(self, config=None):
        if config is not None:
            self.config = config

    def _names(self):
        raise NotImplementedError

    def _calculate_size(self):
        return sum(t.numel() * t.element_size() for t in (getattr(self, a, None) for a in ("weight", "bias")) if t is not None)

    def to_cuda(self, non_blocking=False):
        for a in ("weight", "bias"):
            t = getattr(self, a, None)
            if t is not None:
                setattr(self, a, t.cuda(non_blocking=non_blocking))

    def to_cpu(self, non_blocking=False):
        for a in ("weight", "bias"):
            t = getattr(self, a, None)
            if t is not None:
                setattr(self, a, t.to("cpu", non_blocking=non_blocking))

    def clear(self):
        self.weight = None
        if hasattr(self, "bias"):
            self.bias = None


@LN_WEIGHT_REGISTER("hip")
class HipLNWeight(_NormBase):
    """LayerNorm over the last dim, eps 1e-6, optional affine (norm3 has one, norm1/norm2 do not:
    lightx2v/models/networks/wan/weights/transformer_weights.py:180-183,322-331,446-449)."""

    def __init__(self, weight_name=None, bias_name=None, create_cuda_buffer=False, lazy_load=False, lazy_load_file=None,
                 is_post_adapter=False, eps=1e-6):
        super().__init__(create_cuda_buffer, lazy_load, lazy_load_file, is_post_adapter, eps)
        self.weight_name = weight_name
        self.bias_name = bias_name
        self.weight = None
        self.bias = None

    def load(self, weight_dict):
        if self.weight_name is not None:
            self.weight = _device_ready(weight_dict[self.weight_name], torch.float32)
            self.bias = _device_ready(weight_dict[self.bias_name], torch.float32) if self.bias_name is not None else torch.zeros_like(self.weight)

    def state_dict(self, destination=None):
        if destination is None:
            destination = {}
        if self.weight_name is not None:
            destination[self.weight_name] = self.weight
        if self.bias_name is not None:
            destination[self.bias_name] = self.bias
        return destination

    def load_state_dict(self, destination, block_index, adapter_block_index=None):
        idx = adapter_block_index if self.is_post_adapter else block_index
        if self.weight_name is not None:
            name = _renumber(self.weight_name, idx)
            self.weight = destination.get(name)
        if self.bias_name is not None and self.weight is not None:
            self.bias = destination.get(_renumber(self.bias_name, idx))

    def apply(self, input_tensor, quant_for=None):
        """`quant_for`: the matmul plugin(s) that alone consume the result; when they all take
        quantised rows (`accepts_quantized`), the result leaves the kernel as hip.QuantRows."""
        return hip.ln_modulate(input_tensor, self.weight, self.bias, eps=self.eps, quant=_all_take_quantized(quant_for) and input_tensor.dtype != torch.float32)

    def apply_modulate(self, input_tensor, scale, shift, quant_for=None):
        """LN fused with `x * (1 + scale) + shift` (transformer_infer.py:12-13,154-157).  fp32 rows (the
        residual stream kept in fp32) give a bf16 result; a W8A8 consumer then quantises it itself."""
        return hip.ln_modulate(input_tensor, self.weight, self.bias, scale, shift, eps=self.eps,
                               quant=_all_take_quantized(quant_for) and input_tensor.dtype != torch.float32)


@RMS_WEIGHT_REGISTER("hip")
class HipRMSWeight(_NormBase):
    """RMSNorm over the WHOLE last dim with fp32 statistics (what sgl_kernel.rmsnorm does for the
    reference, rms_norm_weight.py:131-136); weight [dim]."""

    def __init__(self, weight_name, create_cuda_buffer=False, lazy_load=False, lazy_load_file=None, is_post_adapter=False, eps=1e-6):
        super().__init__(create_cuda_buffer, lazy_load, lazy_load_file, is_post_adapter, eps)
        self.weight_name = weight_name
        self.weight = None

    def load(self, weight_dict):
        self.weight = _device_ready(weight_dict[self.weight_name], torch.float32)

    def state_dict(self, destination=None):
        if destination is None:
            destination = {}
        destination[self.weight_name] = self.weight
        return destination

    def load_state_dict(self, destination, block_index, adapter_block_index=None):
        idx = adapter_block_index if self.is_post_adapter else block_index
        self.weight = destination.get(_renumber(self.weight_name, idx))

    def apply(self, input_tensor):
        shape = input_tensor.shape
        x2 = input_tensor.reshape(-1, shape[-1])
        return hip.rmsnorm_rope(x2, self.weight, None, eps=self.eps).view(shape)

    def apply_rope_(self, x2d, rope_cos_sin, head_dim):
        """In place: RMSNorm then per-head interleaved RoPE (one pass over HBM)."""
        return hip.rmsnorm_rope(x2d, self.weight, rope_cos_sin, head_dim=head_dim, eps=self.eps, out=x2d)

    def apply_(self, x2d):
        return hip.rmsnorm_rope(x2d, self.weight, None, eps=self.eps, out=x2d)

    def apply_per_head_rope_(self, x2d, rope_cos_sin, head_dim):
        """In place on [rows, H*head_dim]: RMSNorm of every head separately with this plugin's
        [head_dim] weight (the result of `apply` on the [rows, H, head_dim] view), then the
        interleaved RoPE when a table is given."""
        return hip.rmsnorm_rope(x2d, self.weight, rope_cos_sin, head_dim=head_dim, eps=self.eps, out=x2d, per_head=True)


# ----------------------------------------------------------------------------------- tensors
@TENSOR_REGISTER("Default")
class DefaultTensor:
    """Plain parameter holder (`modulation`, `head.modulation`)."""

    def __init__(self, tensor_name, create_cuda_buffer=False, lazy_load=False, lazy_load_file=None, is_post_adapter=False):
        self.tensor_name = tensor_name
        self.create_cuda_buffer = create_cuda_buffer
        self.lazy_load = lazy_load
        self.lazy_load_file = lazy_load_file
        self.is_post_adapter = is_post_adapter
        self.tensor = None

    def load(self, weight_dict):
        self.tensor = weight_dict[self.tensor_name]

    def clear(self):
        self.tensor = None

    def _calculate_size(self):
        return self.tensor.numel() * self.tensor.element_size()

    def to_cuda(self, non_blocking=False):
        self.tensor = self.tensor.cuda(non_blocking=non_blocking)

    def to_cpu(self, non_blocking=False):
        self.tensor = self.tensor.to("cpu", non_blocking=non_blocking)

    def state_dict(self, destination=None):
        if destination is None:
            destination = {}
        destination[self.tensor_name] = self.tensor
        return destination

    def load_state_dict(self, destination, block_index, adapter_block_index=None):
        idx = adapter_block_index if self.is_post_adapter else block_index
        self.tensor = destination.get(_renumber(self.tensor_name, idx))
