"""ctypes binding of liblx2v_hip.so (the C ABI declared in include/lx2v_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, the op raises.
Tensors are passed as raw device pointers; kernels are enqueued on torch's CURRENT stream
(the reference's op classes do the same through torch, e.g. its offload path switches streams
with `torch.cuda.stream(...)`, lightx2v/models/networks/wan/infer/offload/transformer_infer.py:65).
"""

from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_LIB_NAME = "liblx2v_hip.so"
ABI_VERSION = 4  # include/lx2v_hip.h: lx2v_abi_version()
_lib: Optional[ctypes.CDLL] = None

EPI_NONE, EPI_GELU_TANH, EPI_RESIDUAL = 0, 1, 2

# name -> argtypes (restype is always c_int unless listed in _SPECIAL)
_P, _I64, _I, _F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
_SIGNATURES = {
    "lx2v_ln_modulate_bf16": [_P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I, _F, _P],
    "lx2v_rmsnorm_rope_bf16": [_P, _I64, _P, _I, _P, _I, _P, _I64, _I64, _I, _F, _P],
    "lx2v_gate_residual_bf16": [_P, _I64, _P, _I64, _P, _I64, _I, _P],
    "lx2v_ln_modulate_rows_bf16": [_P, _I64, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I, _F, _P],
    "lx2v_ln_modulate_quant_fp8": [_P, _I64, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _I64, _I, _F, _P],
    "lx2v_gate_residual_rows_bf16": [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I, _P],
    "lx2v_ln_modulate_f32in": [_P, _I64, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I, _F, _P],
    "lx2v_gate_residual_f32": [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I, _P],
    "lx2v_gelu_tanh_bf16": [_P, _P, _I64, _P],
    "lx2v_gelu_erf_bf16": [_P, _P, _I64, _P],
    "lx2v_gemm_bf16": [_P, _I64, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _I64, _I, _P, _I64, _P, _P],
    "lx2v_gemm_bf16_ws": [_P, _I64, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _I64, _I, _P, _I64, _P, _P, _I64, _P],
    "lx2v_gemm_bf16_workspace_bytes": [_I64, _I64, _I64],
    "lx2v_attn_fwd_bf16": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I, _I, _F, _P, _P],
    "lx2v_attn_fwd_bf16_ws": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I, _I, _F, _P, _P, _I64, _P],
    "lx2v_attn_bf16_workspace_bytes": [_I64, _I64, _I],
    "lx2v_ulysses_pack_qkv_bf16": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _I, _P],
    "lx2v_patchify_bf16": [_P, _P, _I64, _I, _I, _I, _I, _I, _P],
    "lx2v_unpatchify_f32": [_P, _I64, _P, _I, _I, _I, _I, _P],
    "lx2v_channels_to_rows_bf16": [_P, _P, _I64, _I, _I64, _I, _P],
    "lx2v_rows_to_channels_f32": [_P, _I64, _P, _I, _I64, _P],
    "lx2v_weighted_mean_rows_bf16": [_P, _I64, _P, _I, _I, _P, _P],
    "lx2v_add_rowvec_bf16": [_P, _I64, _P, _P, _I64, _I64, _I, _P],
    "lx2v_timestep_embedding_bf16": [_F, _I, _P, _P],
    "lx2v_silu_bf16": [_P, _P, _I64, _P],
    "lx2v_lincomb_f32": [_P, _I64, _I, _P, _P, _P],
    "lx2v_cfg_combine_f32": [_P, _P, _P, _F, _I64, _P],
    "lx2v_attn_fp8_workspace_bytes": [_I64, _I64, _I],
    "lx2v_attn_fwd_fp8": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I, _I, _F, _P, _I64, _P],
    "lx2v_attn_fp8_static_bytes": [],
    "lx2v_attn_fwd_fp8_static": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I, _I, _F, _P],
    "lx2v_set_option": [ctypes.c_char_p, _I],
    "lx2v_get_option": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)],
    "lx2v_quant_fp8_rowwise": [_P, _I64, _P, _I64, _P, _I64, _I, _P],
    "lx2v_quant_fp8_rowwise_kseg": [_P, _I64, _P, _P, _I64, _P, _I64, _I, _P],
    "lx2v_quant_fp8_rowwise_f32": [_P, _I64, _P, _I64, _P, _I64, _I, _P],
    "lx2v_quant_int8_rowwise_f32": [_P, _I64, _P, _I64, _P, _I64, _I, _I, _P],
    "lx2v_dequant_fp8_rowwise": [_P, _I64, _P, _P, _I64, _I64, _I, _P],
    "lx2v_gemm_fp8": [_P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _P, _I64, _P, _P],
    "lx2v_quant_int8_rowwise": [_P, _I64, _P, _I64, _P, _I64, _I, _I, _P],
    "lx2v_quant_int8_rowwise_kseg": [_P, _I64, _P, _P, _I64, _P, _I64, _I, _P],
    "lx2v_gemm_int8": [_P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _P, _I64, _P, _P],
    "lx2v_gemm_fp8_ws": [_P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _P, _I64, _P, _P, _I64, _P],
    "lx2v_gemm_int8_ws": [_P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _P, _I64, _P, _P, _I64, _P],
    "lx2v_gemm_w8a8_workspace_bytes": [_I64, _I64, _I64],
}
EXPORTS = tuple(_SIGNATURES) + ("lx2v_abi_version", "lx2v_last_error")


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load() -> ctypes.CDLL:
    """Load the shared library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `make -C lightx2v_amd/csrc` (or __graft_entry__.build()). "
            "lightx2v_amd has no CPU or PyTorch fallback for its ops."
        )
    lib = ctypes.CDLL(path)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name in ("lx2v_attn_fp8_workspace_bytes", "lx2v_attn_fp8_static_bytes", "lx2v_attn_bf16_workspace_bytes", "lx2v_gemm_bf16_workspace_bytes", "lx2v_gemm_w8a8_workspace_bytes") else ctypes.c_int
    lib.lx2v_abi_version.restype = ctypes.c_int
    lib.lx2v_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


class Lx2vError(RuntimeError):
    pass


class KSeg(ctypes.Structure):
    """`lx2v_kseg`: a GEMM A operand whose K columns are stored as segments.  Region 1: segments of
    `seg` columns, `stride` elements apart; from K column `split` on (0 = one region): segments of
    `seg2` columns, `stride2` apart, the first one `base2` elements from the tensor's start, rows
    `lda2` elements apart."""
    _fields_ = [("seg", _I64), ("stride", _I64), ("split", _I64), ("seg2", _I64), ("stride2", _I64), ("base2", _I64), ("lda2", _I64)]


def _check(status: int, what: str):
    if status != 0:
        msg = load().lx2v_last_error().decode("utf-8", "replace")
        raise Lx2vError(f"{what} failed with status {status}: {msg}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_bf16_cuda(name: str, t: torch.Tensor, inner_contig: bool = True):
    if not t.is_cuda:
        raise Lx2vError(f"{name}: expected a GPU tensor, got device {t.device} (no CPU fallback exists)")
    if t.dtype != torch.bfloat16:
        raise Lx2vError(f"{name}: expected bfloat16, got {t.dtype}")
    if inner_contig and t.stride(-1) != 1:
        raise Lx2vError(f"{name}: innermost dimension must be contiguous")


def _need_f32_rows(name: str, t: torch.Tensor):
    if not t.is_cuda:
        raise Lx2vError(f"{name}: expected a GPU tensor, got device {t.device} (no CPU fallback exists)")
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(-1) != 1 or t.stride(0) % 8 != 0 or t.data_ptr() % 16 != 0:
        raise Lx2vError(f"{name}: expected fp32 [rows, dim] with a contiguous innermost dimension, a row stride multiple of 8 and a 16-byte aligned base, "
                        f"got {t.dtype} {tuple(t.shape)} stride {t.stride()}")


def _need_out(name: str, t: torch.Tensor, shape, like: torch.Tensor, dtype=torch.bfloat16):
    """A caller-supplied output / residual slot: same device as the input, exact shape and dtype,
    contiguous innermost dimension (rows may be strided: the slots are often views)."""
    if not t.is_cuda or t.device != like.device:
        raise Lx2vError(f"{name}: expected a tensor on {like.device}, got {t.device}")
    if t.dtype != dtype:
        raise Lx2vError(f"{name}: expected {dtype}, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise Lx2vError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if t.stride(-1) != 1:
        raise Lx2vError(f"{name}: innermost dimension must be contiguous")


def set_option(name: str, value: int):
    """Kernel-selection switch (see include/lx2v_hip.h); every setting computes the same function."""
    _check(load().lx2v_set_option(name.encode(), int(value)), "lx2v_set_option")


def get_option(name: str) -> int:
    v = ctypes.c_int(0)
    _check(load().lx2v_get_option(name.encode(), ctypes.byref(v)), "lx2v_get_option")
    return v.value


def _vec_f32(name: str, t: Optional[torch.Tensor], n: int, like: torch.Tensor) -> Optional[torch.Tensor]:
    """scale/shift/gate/bias/norm-weight vectors: any float dtype / shape with n elements -> fp32 [n]."""
    if t is None:
        return None
    if t.numel() != n:
        raise Lx2vError(f"{name}: expected {n} elements, got shape {tuple(t.shape)}")
    v = t.reshape(n)
    if v.dtype != torch.float32 or v.device != like.device or not v.is_contiguous():
        v = v.to(device=like.device, dtype=torch.float32).contiguous()
    return v


# ---------------------------------------------------------------------------------------------
class RowMod:
    """One of the six modulation terms when the timestep is a per-token vector: a [dim] vector
    (the block's learned table row, fp32) plus a per-token part `rows` (bf16 [tokens, dim], any row
    stride: usually a column slice of the [tokens, 6*dim] time projection)."""

    def __init__(self, vec: torch.Tensor, rows: torch.Tensor):
        self.vec, self.rows = vec, rows

    def checked(self, name: str, n_rows: int, dim: int, like: torch.Tensor):
        _need_bf16_cuda(name + ".rows", self.rows)
        if tuple(self.rows.shape) != (n_rows, dim) or self.rows.stride(0) % 8 != 0 or self.rows.data_ptr() % 16 != 0:
            raise Lx2vError(f"{name}.rows: expected bf16 [{n_rows}, {dim}] with a 16-byte aligned base and a row stride multiple of 8, "
                            f"got {tuple(self.rows.shape)} stride {self.rows.stride()}")
        return _vec_f32(name + ".vec", self.vec, dim, like), self.rows


class QuantRows:
    """An activation already in the W8A8 operand form: e4m3 rows `q` [rows, dim] and one fp32 scale
    per row (x ~= q * scale), as `quant_fp8_rowwise` returns them."""

    def __init__(self, q: torch.Tensor, scale: torch.Tensor):
        self.q, self.scale = q, scale

    @property
    def shape(self):
        return self.q.shape


def ln_modulate(x: torch.Tensor, ln_w=None, ln_b=None, scale=None, shift=None, eps: float = 1e-6,
                out: Optional[torch.Tensor] = None, quant: bool = False):
    """`scale` / `shift`: fp32 [dim] vectors, or both `RowMod` (per-token modulation).
    `quant`: return the result as `QuantRows` (e4m3 rows + per-row scales, the values
    `quant_fp8_rowwise` gives for the bf16 result) instead of a bf16 tensor."""
    x_f32 = x.dtype == torch.float32   # the residual stream kept in fp32 (the reference's SENSITIVE_LAYER_DTYPE=FP32)
    if x_f32:
        _need_f32_rows("ln_modulate.x", x)
    else:
        _need_bf16_cuda("ln_modulate.x", x)
    assert x.dim() == 2
    rows, dim = x.shape
    w = _vec_f32("ln_w", ln_w, dim, x)
    b = _vec_f32("ln_b", ln_b, dim, x)
    per_token = isinstance(scale, RowMod) or isinstance(shift, RowMod)
    if per_token:
        if not (isinstance(scale, RowMod) and isinstance(shift, RowMod)):
            raise Lx2vError("ln_modulate: scale and shift must both be per-token (RowMod) or both vectors")
        sc, sc_rows = scale.checked("scale", rows, dim, x)
        sh, sh_rows = shift.checked("shift", rows, dim, x)
        if sc_rows.stride(0) != sh_rows.stride(0):
            raise Lx2vError("ln_modulate: per-token scale and shift rows must share one row stride")
    if quant and x_f32:
        raise Lx2vError("ln_modulate: fp32 rows are not combined with the e4m3 output")
    if quant:
        if out is not None:
            raise Lx2vError("ln_modulate: `out` and `quant` exclude each other")
        if not per_token:
            sc, sh, sc_rows, sh_rows = _vec_f32("scale", scale, dim, x), _vec_f32("shift", shift, dim, x), None, None
        q = torch.empty((rows, dim), dtype=torch.float8_e4m3fn, device=x.device)
        qs = torch.empty((rows,), dtype=torch.float32, device=x.device)
        st = load().lx2v_ln_modulate_quant_fp8(x.data_ptr(), x.stride(0), _ptr(w), _ptr(b), _ptr(sc), _ptr(sh), _ptr(sc_rows), _ptr(sh_rows),
                                                0 if sc_rows is None else sc_rows.stride(0), q.data_ptr(), q.stride(0), qs.data_ptr(), rows, dim, eps, _stream())
        _check(st, "lx2v_ln_modulate_quant_fp8")
        return QuantRows(q, qs)
    if out is None:
        out = torch.empty((rows, dim), dtype=torch.bfloat16, device=x.device)
    else:
        _need_out("ln_modulate.out", out, (rows, dim), x)
    if x_f32:
        if not per_token:
            sc, sh, sc_rows, sh_rows = _vec_f32("scale", scale, dim, x), _vec_f32("shift", shift, dim, x), None, None
        st = load().lx2v_ln_modulate_f32in(x.data_ptr(), x.stride(0), _ptr(w), _ptr(b), _ptr(sc), _ptr(sh), _ptr(sc_rows), _ptr(sh_rows),
                                            0 if sc_rows is None else sc_rows.stride(0), out.data_ptr(), out.stride(0), rows, dim, eps, _stream())
        _check(st, "lx2v_ln_modulate_f32in")
        return out
    if per_token:
        st = load().lx2v_ln_modulate_rows_bf16(x.data_ptr(), x.stride(0), _ptr(w), _ptr(b), _ptr(sc), _ptr(sh), sc_rows.data_ptr(), sh_rows.data_ptr(),
                                                sc_rows.stride(0), out.data_ptr(), out.stride(0), rows, dim, eps, _stream())
        _check(st, "lx2v_ln_modulate_rows_bf16")
        return out
    sc = _vec_f32("scale", scale, dim, x)
    sh = _vec_f32("shift", shift, dim, x)
    st = load().lx2v_ln_modulate_bf16(x.data_ptr(), x.stride(0), _ptr(w), _ptr(b), _ptr(sc), _ptr(sh),
                                       out.data_ptr(), out.stride(0), rows, dim, eps, _stream())
    _check(st, "lx2v_ln_modulate_bf16")
    return out


def rmsnorm_rope(x: torch.Tensor, weight=None, rope_cos_sin: Optional[torch.Tensor] = None, head_dim: int = 128,
                 eps: float = 1e-6, out: Optional[torch.Tensor] = None, do_norm: bool = True, per_head: bool = False) -> torch.Tensor:
    """x [rows, dim] (row stride free). rope_cos_sin fp32 [rows, head_dim] (cos || sin) or None.
    `per_head`: statistics over each head's `head_dim` columns with a [head_dim] weight shared by
    all heads, instead of over the whole row with a [dim] weight."""
    _need_bf16_cuda("rmsnorm_rope.x", x)
    assert x.dim() == 2
    rows, dim = x.shape
    if out is None:
        out = torch.empty((rows, dim), dtype=x.dtype, device=x.device)
    else:
        _need_out("rmsnorm_rope.out", out, (rows, dim), x)
    w = _vec_f32("rms weight", weight, head_dim if per_head else dim, x)
    do_norm = 2 if (do_norm and per_head) else int(bool(do_norm))
    if rope_cos_sin is not None:
        if rope_cos_sin.dtype != torch.float32 or not rope_cos_sin.is_contiguous() or tuple(rope_cos_sin.shape) != (rows, head_dim):
            raise Lx2vError(f"rope table must be contiguous fp32 [{rows}, {head_dim}], got {tuple(rope_cos_sin.shape)} {rope_cos_sin.dtype}")
    st = load().lx2v_rmsnorm_rope_bf16(x.data_ptr(), x.stride(0), _ptr(w), int(do_norm), _ptr(rope_cos_sin), head_dim,
                                        out.data_ptr(), out.stride(0), rows, dim, eps, _stream())
    _check(st, "lx2v_rmsnorm_rope_bf16")
    return out


def gate_residual_(x: torch.Tensor, y: torch.Tensor, gate=None) -> torch.Tensor:
    """x += y * gate in place; x is bf16, or fp32 (the residual stream kept in fp32) with y bf16."""
    _need_bf16_cuda("gate_residual.y", y)
    assert x.shape == y.shape and x.dim() == 2
    rows, dim = x.shape
    if x.dtype == torch.float32:
        _need_f32_rows("gate_residual.x", x)
        g, g_rows = gate.checked("gate", rows, dim, x) if isinstance(gate, RowMod) else (_vec_f32("gate", gate, dim, x), None)
        st = load().lx2v_gate_residual_f32(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), _ptr(g), _ptr(g_rows),
                                            0 if g_rows is None else g_rows.stride(0), rows, dim, _stream())
        _check(st, "lx2v_gate_residual_f32")
        return x
    _need_bf16_cuda("gate_residual.x", x)
    if isinstance(gate, RowMod):
        g, g_rows = gate.checked("gate", rows, dim, x)
        st = load().lx2v_gate_residual_rows_bf16(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), _ptr(g), g_rows.data_ptr(), g_rows.stride(0), rows, dim, _stream())
        _check(st, "lx2v_gate_residual_rows_bf16")
        return x
    g = _vec_f32("gate", gate, dim, x)
    st = load().lx2v_gate_residual_bf16(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), _ptr(g), rows, dim, _stream())
    _check(st, "lx2v_gate_residual_bf16")
    return x


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """Exact GELU (bf16 in / out)."""
    _need_bf16_cuda("gelu_erf.x", x)
    if not x.is_contiguous():
        raise Lx2vError("gelu_erf: input must be contiguous")
    out = torch.empty_like(x)
    _check(load().lx2v_gelu_erf_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "lx2v_gelu_erf_bf16")
    return out


def gelu_tanh(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_bf16_cuda("gelu.x", x)
    if not x.is_contiguous():
        raise Lx2vError("gelu_tanh: input must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    else:
        _need_out("gelu.out", out, x.shape, x)
        if not out.is_contiguous():
            raise Lx2vError("gelu_tanh: out must be contiguous")
    st = load().lx2v_gelu_tanh_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    _check(st, "lx2v_gelu_tanh_bf16")
    return out


def gemm_workspace_bytes(M: int, N: int, K: int) -> int:
    """Bytes of partial sums `gemm` can use for a call of this shape (0: its last round of tiles is not split along K)."""
    return int(load().lx2v_gemm_bf16_workspace_bytes(int(M), int(N), int(K)))


_GEMM_WORKSPACES = {}


def gemm_workspace_for(a: torch.Tensor, N: int, K: int) -> Optional[torch.Tensor]:
    """One workspace per device, grown to the largest request, for callers that issue their GEMMs on one stream
    (calls that share a workspace must be ordered); None when the shape would not use one."""
    need = gemm_workspace_bytes(a.shape[0], N, K)
    if need == 0:
        return None
    ws = _GEMM_WORKSPACES.get(a.device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=a.device)
        _GEMM_WORKSPACES[a.device] = ws
    return ws


def gemm(a: torch.Tensor, w_out_in: torch.Tensor, bias=None, epilogue: int = EPI_NONE, resid: Optional[torch.Tensor] = None,
         gate=None, out: Optional[torch.Tensor] = None, kseg: Optional[KSeg] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epilogue(a @ w_out_in.T + bias).  a [M,K] bf16, w_out_in [N,K] bf16 (checkpoint layout).
    With `kseg`, `a` is [M, seg] rows of the FIRST segment and the other segments are found through
    the strides in `kseg` (the caller guarantees they lie inside one allocation).
    `workspace`: an optional uint8 GPU tensor (`gemm_workspace_bytes`) that lets a sparsely filled last round of
    output tiles be split along K; those tiles' outputs then differ from the plain call's by fp32 summation order."""
    _need_bf16_cuda("gemm.a", a)
    _need_bf16_cuda("gemm.w", w_out_in)
    assert a.dim() == 2 and w_out_in.dim() == 2
    M = a.shape[0]
    N, K = w_out_in.shape
    if kseg is None and a.shape[1] != K:
        raise Lx2vError(f"gemm: inner dimensions differ: a {tuple(a.shape)} vs w {tuple(w_out_in.shape)}")
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    else:
        _need_out("gemm.out", out, (M, N), a)
    b = _vec_f32("bias", bias, N, a)
    g = _vec_f32("gate", gate, N, a)
    if resid is not None:
        _need_out("gemm.resid", resid, (M, N), a)
    elif epilogue == EPI_RESIDUAL:
        raise Lx2vError("gemm: the residual epilogue needs `resid`")
    if workspace is not None and (workspace.dtype != torch.uint8 or workspace.device != a.device or not workspace.is_contiguous()):
        raise Lx2vError("gemm: workspace must be a contiguous uint8 tensor on the operands' device")
    st = load().lx2v_gemm_bf16_ws(a.data_ptr(), a.stride(0), None if kseg is None else ctypes.byref(kseg), w_out_in.data_ptr(), w_out_in.stride(0), _ptr(b),
                                  out.data_ptr(), out.stride(0), M, N, K, epilogue, _ptr(resid), 0 if resid is None else resid.stride(0),
                                  _ptr(g), _ptr(workspace), 0 if workspace is None else workspace.numel(), _stream())
    _check(st, "lx2v_gemm_bf16")
    return out


def attn_workspace_bytes(sq: int, skv: int, heads: int) -> int:
    """Bytes of partial results `attn_fwd` can use for a call of this shape (0: its launch plan has no key split)."""
    return int(load().lx2v_attn_bf16_workspace_bytes(int(sq), int(skv), int(heads)))


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None,
             rope_cos_sin: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [Sq,H,d], k/v [Skv,H,d] (token stride free, heads*d contiguous inside a token) -> [Sq, H*d].
    `workspace`: an optional uint8 GPU tensor for the partial results of a key split (launches of few heads over a
    long sequence; `attn_workspace_bytes`); without it such launches run unsplit."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _need_bf16_cuda("attn." + n, t)
        if t.dim() != 3 or t.stride(1) != t.shape[2]:
            raise Lx2vError(f"attn.{n}: expected [S,H,d] with heads contiguous inside a token, got shape {tuple(t.shape)} strides {t.stride()}")
    sq, H, d = q.shape
    skv = k.shape[0]
    if k.shape[1:] != (H, d) or v.shape != k.shape:
        raise Lx2vError(f"attn: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)} do not agree")
    if scale is None:
        scale = d ** -0.5
    if out is None:
        out = torch.empty((sq, H * d), dtype=q.dtype, device=q.device)
    else:
        _need_out("attn.out", out, (sq, H * d), q)
    if rope_cos_sin is not None:
        if rope_cos_sin.dtype != torch.float32 or not rope_cos_sin.is_contiguous() or tuple(rope_cos_sin.shape) != (sq, d):
            raise Lx2vError(f"attn: rope table must be contiguous fp32 [{sq}, {d}]")
    if workspace is not None and (not workspace.is_cuda or workspace.dtype != torch.uint8 or not workspace.is_contiguous()):
        raise Lx2vError("attn: workspace must be a contiguous uint8 GPU tensor")
    st = load().lx2v_attn_fwd_bf16_ws(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                      out.data_ptr(), out.stride(0), sq, skv, H, d, float(scale), _ptr(rope_cos_sin),
                                      _ptr(workspace), 0 if workspace is None else workspace.numel(), _stream())
    _check(st, "lx2v_attn_fwd_bf16_ws")
    return out


def attn_fwd_fp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None, out: Optional[torch.Tensor] = None,
                 workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`attn_fwd` with both contractions in block-scaled e4m3 (bf16 in / bf16 out).  `workspace`: a
    uint8 GPU tensor of at least `attn_fp8_workspace_bytes` bytes (allocated here when absent)."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _need_bf16_cuda("attn_fp8." + n, t)
        if t.dim() != 3 or t.stride(1) != t.shape[2]:
            raise Lx2vError(f"attn_fp8.{n}: expected [S,H,d] with heads contiguous inside a token, got shape {tuple(t.shape)} strides {t.stride()}")
    sq, H, d = q.shape
    skv = k.shape[0]
    if k.shape[1:] != (H, d) or v.shape != k.shape:
        raise Lx2vError(f"attn_fp8: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)} do not agree")
    if scale is None:
        scale = d ** -0.5
    if out is None:
        out = torch.empty((sq, H * d), dtype=q.dtype, device=q.device)
    else:
        _need_out("attn_fp8.out", out, (sq, H * d), q)
    need = attn_fp8_workspace_bytes(sq, skv, H)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
    if not workspace.is_cuda or workspace.dtype != torch.uint8 or not workspace.is_contiguous() or workspace.numel() < need:
        raise Lx2vError(f"attn_fp8: workspace must be a contiguous uint8 GPU tensor of >= {need} bytes")
    st = load().lx2v_attn_fwd_fp8(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), out.data_ptr(), out.stride(0),
                                  sq, skv, H, d, float(scale), workspace.data_ptr(), workspace.numel(), _stream())
    _check(st, "lx2v_attn_fwd_fp8")
    return out


def attn_fwd_fp8_static(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, scale: Optional[float] = None) -> torch.Tensor:
    """`attn_fwd_fp8` through the entry point that allocates nothing: the caller brings `out`, the
    quantised images live in the library's fixed per-GPU array (shapes beyond it raise)."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _need_bf16_cuda("attn_fp8_static." + n, t)
        if t.dim() != 3 or t.stride(1) != t.shape[2]:
            raise Lx2vError(f"attn_fp8_static.{n}: expected [S,H,d] with heads contiguous inside a token, got shape {tuple(t.shape)} strides {t.stride()}")
    sq, H, d = q.shape
    skv = k.shape[0]
    if k.shape[1:] != (H, d) or v.shape != k.shape:
        raise Lx2vError(f"attn_fp8_static: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)} do not agree")
    _need_out("attn_fp8_static.out", out, (sq, H * d), q)
    if scale is None:
        scale = d ** -0.5
    st = load().lx2v_attn_fwd_fp8_static(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), out.data_ptr(), out.stride(0),
                                         sq, skv, H, d, float(scale), _stream())
    _check(st, "lx2v_attn_fwd_fp8_static")
    return out


def attn_fp8_static_bytes() -> int:
    return int(load().lx2v_attn_fp8_static_bytes())


def attn_fp8_workspace_bytes(sq: int, skv: int, heads: int) -> int:
    n = load().lx2v_attn_fp8_workspace_bytes(sq, skv, heads)
    if n < 0:
        raise Lx2vError(f"attn_fp8_workspace_bytes: bad shape sq={sq} skv={skv} heads={heads}")
    return int(n)


def ulysses_pack_qkv(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, world: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q,k,v [rows, dim] -> [world, rows, 3, dim/world] (one all-to-all payload)."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _need_bf16_cuda("pack." + n, t)
    rows, dim = q.shape
    if dim % world != 0:
        raise Lx2vError(f"ulysses_pack_qkv: dim={dim} not divisible by world={world}")
    for n, t in (("k", k), ("v", v)):
        if tuple(t.shape) != (rows, dim):
            raise Lx2vError(f"ulysses_pack_qkv: {n} {tuple(t.shape)} differs from q {(rows, dim)}")
    if out is None:
        out = torch.empty((world, rows, 3, dim // world), dtype=q.dtype, device=q.device)
    else:
        _need_out("ulysses_pack_qkv.out", out, (world, rows, 3, dim // world), q)
        if not out.is_contiguous():
            raise Lx2vError("ulysses_pack_qkv: out must be contiguous")
    st = load().lx2v_ulysses_pack_qkv_bf16(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                           out.data_ptr(), rows, dim, world, _stream())
    _check(st, "lx2v_ulysses_pack_qkv_bf16")
    return out


# ------------------------------------------------------------------------------- step glue
def _need_f32_cuda(name: str, t: torch.Tensor):
    if not t.is_cuda:
        raise Lx2vError(f"{name}: expected a GPU tensor, got device {t.device} (no CPU fallback exists)")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise Lx2vError(f"{name}: expected contiguous float32, got {t.dtype} (contiguous={t.is_contiguous()})")


def patchify(latents: torch.Tensor, round_latents: bool = False, pad_to: int = 0) -> torch.Tensor:
    """fp32 latents [C, F, H, W] -> bf16 [F*(H/2)*(W/2), K], columns (c, ph, pw) then zeros up to
    K = max(4*C, pad_to).  With `round_latents` the latents are rounded through bf16 in place as well."""
    _need_f32_cuda("patchify.latents", latents)
    c, f, h, w = latents.shape
    k = max(4 * c, pad_to)
    rows = f * (h // 2) * (w // 2)
    out = torch.empty((rows, k), dtype=torch.bfloat16, device=latents.device) if k == 4 * c else torch.zeros((rows, k), dtype=torch.bfloat16, device=latents.device)
    _check(load().lx2v_patchify_bf16(latents.data_ptr(), out.data_ptr(), k, c, f, h, w, int(round_latents), _stream()), "lx2v_patchify_bf16")
    return out


def unpatchify(y: torch.Tensor, channels: int, frames: int, height: int, width: int) -> torch.Tensor:
    """bf16 head output [S, 4*C] (columns (ph, pw, c)) -> fp32 [C, F, H, W]."""
    _need_bf16_cuda("unpatchify.y", y)
    if y.shape[0] < frames * (height // 2) * (width // 2) or y.shape[1] != 4 * channels:
        raise Lx2vError(f"unpatchify: y {tuple(y.shape)} does not cover a [{channels},{frames},{height},{width}] latent")
    out = torch.empty((channels, frames, height, width), dtype=torch.float32, device=y.device)
    _check(load().lx2v_unpatchify_f32(y.data_ptr(), y.stride(0), out.data_ptr(), channels, frames, height, width, _stream()), "lx2v_unpatchify_f32")
    return out


def channels_to_rows(planes: torch.Tensor, pad_to: int = 0) -> torch.Tensor:
    """fp32 [C, ...] channel planes -> bf16 [tokens, K] rows (column = channel), zeros in the columns
    C..K-1 with K = max(C rounded up to 8, pad_to)."""
    _need_f32_cuda("channels_to_rows.planes", planes)
    c = planes.shape[0]
    n = planes.numel() // c
    k = max((c + 7) // 8 * 8, pad_to)
    out = torch.empty((n, k), dtype=torch.bfloat16, device=planes.device)
    _check(load().lx2v_channels_to_rows_bf16(planes.data_ptr(), out.data_ptr(), k, c, n, k, _stream()), "lx2v_channels_to_rows_bf16")
    return out


def rows_to_channels(y: torch.Tensor, channels: int, tokens: int) -> torch.Tensor:
    """bf16 rows [>= tokens, >= channels] -> fp32 [channels, tokens]."""
    _need_bf16_cuda("rows_to_channels.y", y)
    if y.dim() != 2 or y.shape[0] < tokens or y.shape[1] < channels:
        raise Lx2vError(f"rows_to_channels: y {tuple(y.shape)} does not cover {tokens} tokens x {channels} channels")
    out = torch.empty((channels, tokens), dtype=torch.float32, device=y.device)
    _check(load().lx2v_rows_to_channels_f32(y.data_ptr(), y.stride(0), out.data_ptr(), channels, tokens, _stream()), "lx2v_rows_to_channels_f32")
    return out


def weighted_mean_rows(x: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """bf16 [rows, dim], fp32 weights [rows] -> bf16 [1, dim] = sum_l w_l x_l / sum_l w_l."""
    _need_bf16_cuda("weighted_mean_rows.x", x)
    _need_f32_cuda("weighted_mean_rows.weights", weights)
    if x.dim() != 2 or weights.numel() != x.shape[0]:
        raise Lx2vError(f"weighted_mean_rows: x {tuple(x.shape)} and weights {tuple(weights.shape)} do not match")
    out = torch.empty((1, x.shape[1]), dtype=torch.bfloat16, device=x.device)
    _check(load().lx2v_weighted_mean_rows_bf16(x.data_ptr(), x.stride(0), weights.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), _stream()),
           "lx2v_weighted_mean_rows_bf16")
    return out


def add_rowvec(x: torch.Tensor, vec: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 [rows, dim] + bf16 [dim] -> bf16 [rows, dim] (out may be x)."""
    _need_bf16_cuda("add_rowvec.x", x)
    _need_bf16_cuda("add_rowvec.vec", vec)
    if x.dim() != 2 or vec.numel() != x.shape[1] or not vec.is_contiguous():
        raise Lx2vError(f"add_rowvec: x {tuple(x.shape)} and vec {tuple(vec.shape)} do not match")
    if out is None:
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
    else:
        _need_out("add_rowvec.out", out, x.shape, x)
    _check(load().lx2v_add_rowvec_bf16(x.data_ptr(), x.stride(0), vec.data_ptr(), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], _stream()),
           "lx2v_add_rowvec_bf16")
    return out


def timestep_embedding(t: float, dim: int, device) -> torch.Tensor:
    out = torch.empty((1, dim), dtype=torch.bfloat16, device=device)
    _check(load().lx2v_timestep_embedding_bf16(float(t), dim, out.data_ptr(), _stream()), "lx2v_timestep_embedding_bf16")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    _need_bf16_cuda("silu.x", x)
    if not x.is_contiguous():
        raise Lx2vError("silu: x must be contiguous")
    out = torch.empty_like(x)
    _check(load().lx2v_silu_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "lx2v_silu_bf16")
    return out


def lincomb(terms, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum of coef * tensor over `terms` = [(coef, fp32 tensor), ...] (1..4 terms, same shape)."""
    if not 1 <= len(terms) <= 4:
        raise Lx2vError(f"lincomb: {len(terms)} terms, 1..4 supported")
    first = terms[0][1]
    for _, t in terms:
        _need_f32_cuda("lincomb.x", t)
        if t.shape != first.shape:
            raise Lx2vError("lincomb: all tensors must have the same shape")
    if out is None:
        out = torch.empty_like(first)
    ptrs = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for _, t in terms])
    coefs = (ctypes.c_float * len(terms))(*[float(c) for c, _ in terms])
    _check(load().lx2v_lincomb_f32(out.data_ptr(), first.numel(), len(terms), ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(coefs, ctypes.c_void_p), _stream()),
           "lx2v_lincomb_f32")
    return out


def cfg_combine(cond: torch.Tensor, uncond: torch.Tensor, guide_scale: float) -> torch.Tensor:
    _need_f32_cuda("cfg_combine.cond", cond)
    _need_f32_cuda("cfg_combine.uncond", uncond)
    if cond.shape != uncond.shape:
        raise Lx2vError("cfg_combine: shapes differ")
    out = torch.empty_like(cond)
    _check(load().lx2v_cfg_combine_f32(out.data_ptr(), cond.data_ptr(), uncond.data_ptr(), float(guide_scale), cond.numel(), _stream()), "lx2v_cfg_combine_f32")
    return out


# ------------------------------------------------------------------------------- fp8 W8A8
def quant_fp8_rowwise(x: torch.Tensor, kseg: Optional["KSeg"] = None, dim: Optional[int] = None):
    """bf16 [rows, dim] -> (e4m3 [rows, dim] as torch.float8_e4m3fn, fp32 scale [rows]); x ~= q * scale.
    `kseg` + `dim`: x holds the rows of the first segment of a K-segmented operand (see KSeg) of `dim`
    logical columns; the result is the plain matrix."""
    f32 = x.dtype == torch.float32 and kseg is None
    if f32:
        if not x.is_cuda or x.dim() != 2 or x.stride(1) != 1:
            raise Lx2vError("quant_fp8.x: fp32 input must be a GPU matrix with a contiguous inner dimension")
    else:
        _need_bf16_cuda("quant_fp8.x", x)
    rows = x.shape[0]
    dim = x.shape[1] if kseg is None else int(dim)
    q = torch.empty((rows, dim), dtype=torch.float8_e4m3fn, device=x.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    if f32:
        _check(load().lx2v_quant_fp8_rowwise_f32(x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(), rows, dim, _stream()), "lx2v_quant_fp8_rowwise_f32")
    elif kseg is None:
        _check(load().lx2v_quant_fp8_rowwise(x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(), rows, dim, _stream()), "lx2v_quant_fp8_rowwise")
    else:
        _check(load().lx2v_quant_fp8_rowwise_kseg(x.data_ptr(), x.stride(0), ctypes.byref(kseg), q.data_ptr(), q.stride(0), scale.data_ptr(), rows, dim, _stream()),
               "lx2v_quant_fp8_rowwise_kseg")
    return q, scale


def quant_int8_rowwise(x: torch.Tensor, weight_mode: bool = False, kseg: Optional["KSeg"] = None, dim: Optional[int] = None):
    """bf16 [rows, dim] -> (int8 [rows, dim], fp32 scale [rows]); x ~= q * scale.  `weight_mode`: the
    reference's per-channel weight quantiser (division by the scale, scale floor 1e-5 / 127) instead of
    the per-token activation form (multiplication by 127 / max).  `kseg` + `dim` as in quant_fp8_rowwise
    (activation form only)."""
    f32 = x.dtype == torch.float32 and kseg is None
    if f32:
        if not x.is_cuda or x.dim() != 2 or x.stride(1) != 1:
            raise Lx2vError("quant_int8.x: fp32 input must be a GPU matrix with a contiguous inner dimension")
    else:
        _need_bf16_cuda("quant_int8.x", x)
    rows = x.shape[0]
    dim = x.shape[1] if kseg is None else int(dim)
    q = torch.empty((rows, dim), dtype=torch.int8, device=x.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    if f32:
        _check(load().lx2v_quant_int8_rowwise_f32(x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(), rows, dim, int(bool(weight_mode)), _stream()),
               "lx2v_quant_int8_rowwise_f32")
    elif kseg is None:
        _check(load().lx2v_quant_int8_rowwise(x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(), rows, dim, int(bool(weight_mode)), _stream()),
               "lx2v_quant_int8_rowwise")
    else:
        if weight_mode:
            raise Lx2vError("quant_int8: a segmented source is an activation; weight_mode does not apply")
        _check(load().lx2v_quant_int8_rowwise_kseg(x.data_ptr(), x.stride(0), ctypes.byref(kseg), q.data_ptr(), q.stride(0), scale.data_ptr(), rows, dim, _stream()),
               "lx2v_quant_int8_rowwise_kseg")
    return q, scale


def dequant_fp8_rowwise(q: torch.Tensor, scale: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """e4m3 [rows, dim] and fp32 scale [rows] -> bf16 [rows, dim] = q.to(bf16) * scale.to(bf16)."""
    if not q.is_cuda or q.dtype != torch.float8_e4m3fn or q.dim() != 2 or q.stride(1) != 1:
        raise Lx2vError("dequant_fp8: q must be a GPU float8_e4m3fn [rows, dim] tensor with a contiguous inner dimension")
    rows, dim = q.shape
    scale = _vec_f32("dequant_fp8.scale", scale, rows, q)
    if out is None:
        out = torch.empty((rows, dim), dtype=torch.bfloat16, device=q.device)
    _need_out("dequant_fp8.out", out, (rows, dim), q)
    _check(load().lx2v_dequant_fp8_rowwise(q.data_ptr(), q.stride(0), scale.data_ptr(), out.data_ptr(), out.stride(0), rows, dim, _stream()), "lx2v_dequant_fp8_rowwise")
    return out


def gemm_w8a8_workspace_bytes(M: int, N: int, K: int) -> int:
    """Bytes of partial sums `gemm_fp8` / `gemm_int8` can use for a call of this shape (0: no K split of the last round)."""
    return int(load().lx2v_gemm_w8a8_workspace_bytes(int(M), int(N), int(K)))


def gemm_w8a8_workspace_for(a8: torch.Tensor, N: int, K: int) -> Optional[torch.Tensor]:
    """The device's shared GEMM workspace (see `gemm_workspace_for`) when this W8A8 product would use one."""
    need = gemm_w8a8_workspace_bytes(a8.shape[0], N, K)
    if need == 0:
        return None
    ws = _GEMM_WORKSPACES.get(a8.device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=a8.device)
        _GEMM_WORKSPACES[a8.device] = ws
    return ws


def gemm_int8(a8: torch.Tensor, a_scale: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor, bias=None, epilogue: int = EPI_NONE,
              resid: Optional[torch.Tensor] = None, gate=None, out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(a8 [M,K] int8) @ (w8 [N,K] int8)^T, exact in int32, scaled per row / per output channel -> bf16 [M,N]."""
    return gemm_fp8(a8, a_scale, w8, w_scale, bias, epilogue, resid, gate, out, workspace, _dtype=torch.int8)


def gemm_fp8(a8: torch.Tensor, a_scale: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor, bias=None, epilogue: int = EPI_NONE,
             resid: Optional[torch.Tensor] = None, gate=None, out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
             _dtype=torch.float8_e4m3fn) -> torch.Tensor:
    """(a8 [M,K] e4m3) @ (w8 [N,K] e4m3)^T scaled per row / per output channel -> bf16 [M,N].
    `workspace`: optional uint8 GPU tensor (`gemm_w8a8_workspace_bytes`) for the K split of a sparsely filled last round
    of output tiles (fp8: those tiles' outputs differ by fp32 summation order; int8: bit-identical)."""
    for n, t in (("a8", a8), ("w8", w8)):
        if not t.is_cuda or t.dtype != _dtype or t.stride(-1) != 1:
            raise Lx2vError(f"gemm_fp8: {n} must be a GPU {_dtype} tensor with a contiguous inner dimension")
    M, K = a8.shape
    N = w8.shape[0]
    if w8.shape[1] != K:
        raise Lx2vError(f"gemm_fp8: K mismatch {K} vs {w8.shape[1]}")
    a_scale = _vec_f32("gemm_fp8.a_scale", a_scale, M, a8)
    w_scale = _vec_f32("gemm_fp8.w_scale", w_scale, N, a8)
    bias = _vec_f32("gemm_fp8.bias", bias, N, a8)
    gate = _vec_f32("gemm_fp8.gate", gate, N, a8)
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a8.device)
    else:
        _need_out("gemm_fp8.out", out, (M, N), a8)
    if resid is not None:
        _need_out("gemm_fp8.resid", resid, (M, N), a8)
    elif epilogue == EPI_RESIDUAL:
        raise Lx2vError("gemm_fp8: the residual epilogue needs `resid`")
    if workspace is not None and (workspace.dtype != torch.uint8 or workspace.device != a8.device or not workspace.is_contiguous()):
        raise Lx2vError("gemm_fp8: workspace must be a contiguous uint8 tensor on the operands' device")
    entry = "lx2v_gemm_int8_ws" if _dtype == torch.int8 else "lx2v_gemm_fp8_ws"
    st = getattr(load(), entry)(a8.data_ptr(), a8.stride(0), a_scale.data_ptr(), w8.data_ptr(), w8.stride(0), w_scale.data_ptr(), _ptr(bias),
                                out.data_ptr(), out.stride(0), M, N, K, epilogue, _ptr(resid), 0 if resid is None else resid.stride(0), _ptr(gate),
                                _ptr(workspace), 0 if workspace is None else workspace.numel(), _stream())
    _check(st, entry)
    return out
