"""Ulysses sequence parallelism over RCCL: head <-> sequence reshard around any attention plugin.

Stands in for `UlyssesAttnWeight` (lightx2v/common/ops/attn/ulysses_attn.py:11-137) and its helpers
`all2all_seq2head` / `all2all_head2seq` (lightx2v/common/ops/attn/utils/all2all.py:7-89) with the
same result -- global token order is rank-major chunks, the output's heads are in global head
order -- and a different data path:

  * Heads are dealt to ranks in C chunks ("chunks", widths as even as H/N allows: 5 heads per rank
    -> 3 + 2), not in one contiguous run per rank: a rank attends its chunks one after the other,
    and the all-to-all of the next chunk and the return of the previous one overlap the attention
    of the current one (async collectives on RCCL's own stream).

  * q, k and v travel in ONE all-to-all (the reference issues three).  The send buffer is
    [dest rank][S/N][q|k|v][H/N*d], written by one pack kernel, so on arrival token (src, s) of
    tensor `which` sits at ((src*S/N + s)*3 + which)*H/N*d: a uniform token stride of 3*H/N*d that
    the attention kernel reads in place (no transpose/contiguous copies on either side).
  * The attention output [S, H/N*d] is already the send layout of the return all-to-all.
  * The receive buffer ([src rank][S/N][w*d] per chunk) can be handed to the output projection as
    a K-segmented operand (`apply_segmented`), which removes the final gather copy too.
  * Send / receive buffers are allocated once per shape and shared by all blocks.

On MI355X every GPU pair has its own xGMI link, so an all-to-all drives all 7 links at once;
`torch.distributed` backend "nccl" is RCCL on ROCm.
"""

from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import comm, hip
from .plugins import _AttnBase
from .registry_factory import ATTN_WEIGHT_REGISTER


class _Both:
    """Two outstanding collectives waited on together."""

    def __init__(self, *works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()
        return True


def default_chunks(world: int) -> int:
    """Head chunks per rank when LX2V_ULYSSES_CHUNKS is not set.  xGMI is point to point: the bytes a
    rank puts on ONE link per exchange are payload / world^2, the block's compute shrinks only with
    1 / world, so the exchange weighs most in a group of two (14B at 720p: 0.58 GB per link and block
    against 15 ms of compute; 0.15 GB / 7.6 ms at four, 36 MB / 3.8 ms at eight).  The exposed part is
    the first chunk's arrival and the last chunk's return, i.e. 1 / chunks of it."""
    return 4 if world == 2 else 2


# Largest q|k|v payload one rank hands ONE peer in one all-to-all.  RCCL 2.26.6 (ROCm 7.0) returns from an
# all_to_all_single whose per-peer payload exceeds 1 GiB with only the first half of it moved and no error
# (tools/probe_rccl_self_a2a.py); the exchange therefore uses more, narrower head chunks when a chunk would
# come near that size.  Multi-GPU groups are far below it (14B at 720p: 145 MB per peer and chunk in a group of
# two); a group of one rank at that size is not (1.16 GB in two chunks).
MAX_PEER_BYTES = int(os.environ.get("LX2V_ULYSSES_MAX_PEER_MB", "512")) << 20


def chunks_for_payload(heads_per_rank: int, chunks: int, rows: int, d: int, elem_bytes: int = 2) -> int:
    """`chunks`, raised until a chunk's q|k|v payload per peer (`rows` tokens x 3 x heads x d) fits MAX_PEER_BYTES."""
    c = max(1, min(chunks, heads_per_rank))
    while c < heads_per_rank and rows * 3 * -(-heads_per_rank // c) * d * elem_bytes > MAX_PEER_BYTES:
        c += 1
    return c


def split_heads(heads_per_rank: int, chunks: int) -> List[int]:
    """Head-chunk widths of one rank: `chunks` parts as even as the count allows, the wider ones
    first (5 heads in 2 chunks -> [3, 2]); fewer parts when there are fewer heads."""
    c = max(1, min(chunks, heads_per_rank))
    lo, extra = divmod(heads_per_rank, c)
    return [lo + 1] * extra + [lo] * (c - extra)


# Exchange buffers are shared by every block of a process (blocks run one after the other on one
# stream, and a buffer's last reader of block i is enqueued before block i+1 writes it).
_POOL: Dict[Tuple, torch.Tensor] = {}


def _pooled(tag: str, shape: Tuple[int, ...], dtype, device) -> torch.Tensor:
    key = (tag, tuple(shape), dtype, str(device))
    t = _POOL.get(key)
    if t is None:
        t = _POOL[key] = torch.empty(shape, dtype=dtype, device=device)
    return t


def release_buffers():
    """Drop the pooled exchange buffers (e.g. before a run at another resolution)."""
    _POOL.clear()


class CommTimer:
    """Optional measurement of EXPOSED communication: GPU time the compute stream spends blocked in
    the waits for the exchanges (events recorded on the current stream around each `wait()`; no
    kernel is launched between the two).  Off unless `enable()` was called; `bench.py` reports it."""

    def __init__(self):
        self.enabled = False
        self.spans: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []

    def enable(self, on: bool = True):
        self.enabled = on
        self.spans.clear()

    def wait(self, work):
        if not self.enabled or not torch.cuda.is_available():
            return work.wait()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        work.wait()
        e1.record(torch.cuda.current_stream())
        self.spans.append((e0, e1))

    def total_ms(self) -> float:
        return sum(a.elapsed_time(b) for a, b in self.spans)


COMM_TIMER = CommTimer()
LAST_WIDTHS: List[int] = []   # head-chunk widths of the most recent exchange (reporting only)


class SegmentedOperand:
    """The attention result of one rank in its receive layout, as a GEMM A operand: logical
    [S/N, H*d], stored per head chunk as [source rank][S/N][width_c*d]; the segments are in global
    head order.  `first` = rows of the first segment (the tensor handed to the GEMM), `kseg` = the
    matching hip.KSeg (None when one segment spans all columns)."""

    def __init__(self, flat: torch.Tensor, sl: int, world: int, widths: List[int], d: int):
        if len(set(widths)) > 2 or sorted(widths, reverse=True) != list(widths):
            raise ValueError(f"segmented operand: chunk widths {widths} must be non-increasing with at most two distinct values")
        self.flat, self.sl, self.world, self.widths, self.d = flat, sl, world, list(widths), d
        w0 = widths[0]
        n0 = sum(1 for w in widths if w == w0)
        seg0 = w0 * d
        self.first = flat[:sl * seg0].view(sl, seg0)
        self.kseg = hip.KSeg(seg=seg0, stride=sl * seg0, split=0, seg2=0, stride2=0, base2=0, lda2=0)
        if n0 < len(widths):
            seg1 = widths[-1] * d
            self.kseg.split = n0 * world * seg0
            self.kseg.seg2, self.kseg.stride2, self.kseg.base2, self.kseg.lda2 = seg1, sl * seg1, n0 * world * sl * seg0, seg1

    def gathered(self) -> torch.Tensor:
        """The plain [S/N, H*d] matrix (a copy)."""
        parts, off = [], 0
        for w in self.widths:
            n = self.world * self.sl * w * self.d
            parts.append(self.flat[off:off + n].view(self.world, self.sl, w * self.d).permute(1, 0, 2).reshape(self.sl, self.world * w * self.d))
            off += n
        return torch.cat(parts, dim=1)


@ATTN_WEIGHT_REGISTER("hip_ulysses")
class HipUlyssesAttnWeight(_AttnBase):
    def __init__(self, pack_fn: Optional[Callable] = None, quant_fn: Optional[Callable] = None, dequant_fn: Optional[Callable] = None):
        """`pack_fn(q2d, k2d, v2d, world, out=) -> [world, rows, 3, dim/world]`; defaults to the HIP pack
        kernel.  `quant_fn(x2d) -> (e4m3, scale)` / `dequant_fn(e4m3, scale, out=)` serve the fp8
        payload and default to the HIP kernels.  (Injectable so the exchange logic can be exercised
        with host tensors in tests.)"""
        super().__init__()
        self._pack = pack_fn or hip.ulysses_pack_qkv
        self._quant = quant_fn or hip.quant_fp8_rowwise
        self._dequant = dequant_fn or hip.dequant_fp8_rowwise
        # Head chunks per rank: the exchange of chunk c+1 (and the return of chunk c-1) overlaps the
        # attention of chunk c.  1 = one exchange, no overlap.
        env = os.environ.get("LX2V_ULYSSES_CHUNKS")
        self.chunks = max(1, int(env)) if env else 0   # 0: default_chunks(group size) at the call

    # ---- exchange core -------------------------------------------------------------------------
    def _exchange_attend(self, q, k, v, attention_module, group, model_cls=None, txt=None, fp8=False) -> Tuple[SegmentedOperand, Optional[torch.Tensor]]:
        """Returns (operand, txt_out): `operand` holds this rank's [S/N, H*d] attention output in the
        receive layout (segments in global head order).

        Heads are dealt in chunks: with widths w_0 >= w_1 >= ... (heads per rank per chunk) chunk c
        covers the next N*w_c heads, of which rank r attends heads r*w_c .. (r+1)*w_c - 1, so every
        chunk is one contiguous column range and one all-to-all, and the returned segments
        (chunk-major, then source rank) are in global head order.

        `txt` = (tq, tk, tv), each [Lt, H, d] and identical on every rank (HunyuanVideo layout,
        ulysses_attn.py:80-88,105-116): the text rows are appended after the gathered image rows of
        every head group this rank attends, and their outputs are gathered along the head axis into
        `txt_out` [Lt, H*d] (None without a text part).

        `fp8` (`seq_p_fp8_comm`, ulysses_attn.py:59-73,125-131): what crosses the links is e4m3 with one
        fp32 scale per (token, head) -- q, k, v on the way out, the attention output on the way back --
        and is widened to bf16 on arrival; 0.53x the bytes.  Text rows never leave the rank and stay bf16.

        Issue order of the collectives (identical on every rank of the group): arrivals of all
        chunks; then per chunk its return (and its text gather).  Each is waited for on the compute
        stream right before its first consumer."""
        tp = comm.transport()
        if q.dim() == 4:
            q, k, v = (t.reshape(-1, t.shape[-2], t.shape[-1]) for t in (q, k, v))
        world = comm.group_size(group)
        rank = comm.group_rank(group)
        sl, heads, d = q.shape
        if heads % world != 0:
            raise ValueError(f"hip_ulysses: num_heads={heads} is not divisible by the sequence-parallel size {world}")
        widths = split_heads(heads // world, chunks_for_payload(heads // world, self.chunks or default_chunks(world), sl, d, q.element_size()))
        global LAST_WIDTHS
        LAST_WIDTHS = list(widths)
        dim = heads * d
        lt = 0 if txt is None else txt[0].shape[0]
        dev, dt = q.device, q.dtype
        q2, k2, v2 = (t.reshape(sl, dim) for t in (q, k, v))
        rows = world * sl + lt

        # ---- pack + issue every chunk's arrival ----
        head0 = [0]
        for w in widths:
            head0.append(head0[-1] + world * w)   # first head of each chunk
        recv, arrivals, staged = [], [], []
        for c, w in enumerate(widths):
            cols = slice(head0[c] * d, head0[c + 1] * d)
            send = self._pack(q2[:, cols], k2[:, cols], v2[:, cols], world, out=_pooled(f"send{c}", (world, sl, 3, w * d), dt, dev))
            r = _pooled(f"recv{c}", (rows, 3, w * d), dt, dev)   # image rows of all ranks, then text rows
            recv.append(r)
            if fp8:
                s8, ssc = self._quant(send.view(-1, d))
                s8, ssc = s8.view(torch.uint8).view(world, sl, 3, w * d), ssc.view(world, sl, 3 * w)
                r8, rsc = _pooled(f"recv8_{c}", tuple(s8.shape), torch.uint8, dev), _pooled(f"recvsc{c}", tuple(ssc.shape), torch.float32, dev)
                staged.append((s8, ssc, r8, rsc))
                arrivals.append(_Both(tp.all_to_all_single(r8, s8, group=group, async_op=True), tp.all_to_all_single(rsc, ssc, group=group, async_op=True)))
            else:
                staged.append((send,))
                arrivals.append(tp.all_to_all_single(r[:world * sl].view(world, sl, 3, w * d), send, group=group, async_op=True))
            if lt:
                h0 = head0[c] + rank * w
                for i, t in enumerate(txt):
                    r[world * sl:, i] = t[:, h0:h0 + w].reshape(lt, w * d)

        # ---- attend chunk by chunk; each output leaves while the next chunk is attended ----
        total = sl * dim
        back = _pooled("back", (total,), dt, dev)   # per chunk [source rank][sl][w*d], chunk after chunk
        txt_parts: List[torch.Tensor] = []
        returns, keep, off = [], [], 0
        for c, w in enumerate(widths):
            COMM_TIMER.wait(arrivals[c])
            if fp8:
                _, _, r8, rsc = staged[c]
                self._dequant(r8.view(-1, d).view(torch.float8_e4m3fn), rsc.view(-1), out=recv[c][:world * sl].view(-1, d))
            r = recv[c].view(rows, 3, w, d)
            attn = attention_module.apply(q=r[:, 0], k=r[:, 1], v=r[:, 2], cu_seqlens_q=None, cu_seqlens_kv=None,
                                          max_seqlen_q=rows, max_seqlen_kv=rows, model_cls=model_cls)  # [S (+Lt), w*d]
            attn = attn.contiguous()
            keep.append(attn)
            n = world * sl * w * d
            dst = back[off:off + n].view(world, sl, w * d)
            off += n
            if fp8:
                a8, asc = self._quant(attn[:world * sl].view(-1, d))
                a8, asc = a8.view(torch.uint8).view(world, sl, w * d), asc.view(world, sl, w)
                b8, bsc = _pooled(f"back8_{c}", tuple(a8.shape), torch.uint8, dev), _pooled(f"backsc{c}", tuple(asc.shape), torch.float32, dev)
                keep += [a8, asc]
                returns.append((_Both(tp.all_to_all_single(b8, a8, group=group, async_op=True), tp.all_to_all_single(bsc, asc, group=group, async_op=True)), (b8, bsc, dst)))
            else:
                returns.append((tp.all_to_all_single(dst, attn[:world * sl].view(world, sl, w * d), group=group, async_op=True), None))
            if lt:
                tg = torch.empty((world, lt, w * d), dtype=dt, device=dev)
                txt_parts.append(tg)
                returns.append((tp.all_gather(list(tg.unbind(0)), attn[world * sl:].contiguous(), group=group, async_op=True), None))
        for work, fin in returns:
            COMM_TIMER.wait(work)
            if fin is not None:
                b8, bsc, dst = fin
                self._dequant(b8.view(-1, d).view(torch.float8_e4m3fn), bsc.view(-1), out=dst.view(-1, d))
        # text outputs: chunk c, source rank r holds heads head0[c] + r*w .. -> concatenated they are in head order
        txt_out = torch.cat([tg.permute(1, 0, 2).reshape(lt, -1) for tg in txt_parts], dim=1) if lt else None
        return SegmentedOperand(back, sl, world, widths, d), txt_out

    # ---- reference-compatible entry ------------------------------------------------------------
    def apply(self, q, k, v, img_qkv_len, cu_seqlens_qkv, attention_module=None, seq_p_group=None, model_cls=None, use_fp8_comm=False):
        """q,k,v [S/N (+ Lt), H, d] -> [S/N (+ Lt), H*d] (ulysses_attn.py:16): the first `img_qkv_len`
        rows are this rank's shard of the image tokens, rows after them (up to cu_seqlens_qkv[1]) are
        text tokens replicated on every rank."""
        if len(cu_seqlens_qkv) != 2:
            raise NotImplementedError("hip_ulysses: a masked text tail (three-entry cu_seqlens_qkv) is not built")
        if q.dim() == 4:
            q, k, v = (t.reshape(-1, t.shape[-2], t.shape[-1]) for t in (q, k, v))
        img_qkv_len = int(img_qkv_len)
        total = int(cu_seqlens_qkv[-1])
        if q.shape[0] != total or total < img_qkv_len:
            raise ValueError(f"hip_ulysses: q has {q.shape[0]} rows, cu_seqlens_qkv says {total}, image part {img_qkv_len}")
        txt = None if total == img_qkv_len else tuple(t[img_qkv_len:] for t in (q, k, v))
        operand, txt_out = self._exchange_attend(q[:img_qkv_len], k[:img_qkv_len], v[:img_qkv_len], attention_module, seq_p_group, model_cls, txt, fp8=bool(use_fp8_comm))
        img_out = operand.gathered()
        return img_out if txt_out is None else torch.cat([img_out, txt_out])

    # ---- zero-copy form for the fused block runner ---------------------------------------------
    def apply_segmented(self, q, k, v, attention_module=None, seq_p_group=None, model_cls=None, use_fp8_comm=False) -> SegmentedOperand:
        """The [S/N, H*d] result left in its receive layout (valid until this process's next
        exchange): hand `operand.first` and `operand.kseg` to the output projection."""
        operand, _ = self._exchange_attend(q, k, v, attention_module, seq_p_group, model_cls, fp8=bool(use_fp8_comm))
        return operand


    def apply_segmented_joint(self, q, k, v, img_qkv_len, attention_module=None, seq_p_group=None, model_cls=None, use_fp8_comm=False):
        """`apply_segmented` for the joint image + text layout (HunyuanVideo): rows [0, img_qkv_len) are this
        rank's image shard, the rest text rows replicated on every rank.  Returns (operand, txt_out): the
        image rows' result in its receive layout, the text rows' result [Lt, H*d] (None without text)."""
        img_qkv_len = int(img_qkv_len)
        if q.dim() == 4:
            q, k, v = (t.reshape(-1, t.shape[-2], t.shape[-1]) for t in (q, k, v))
        txt = None if q.shape[0] == img_qkv_len else tuple(t[img_qkv_len:] for t in (q, k, v))
        return self._exchange_attend(q[:img_qkv_len], k[:img_qkv_len], v[:img_qkv_len], attention_module, seq_p_group, model_cls, txt, fp8=bool(use_fp8_comm))


def make_groups(cfg_p_size: int, seq_p_size: int):
    """2-D layout (cfg_p, seq_p) with seq_p the fast (contiguous-rank) dimension, as
    lightx2v/utils/set_config.py:90-105 builds it.  Returns (cfg_index, seq_index, seq_group, cfg_group);
    every rank creates every group (collective requirement of `new_group`)."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if cfg_p_size * seq_p_size != world:
        raise ValueError(f"cfg_p_size*seq_p_size = {cfg_p_size * seq_p_size} != world size {world}")
    cfg_idx, seq_idx = divmod(rank, seq_p_size)
    seq_group = cfg_group = None
    for c in range(cfg_p_size):
        g = dist.new_group(ranks=[c * seq_p_size + s for s in range(seq_p_size)])
        if c == cfg_idx:
            seq_group = g
    for s in range(seq_p_size):
        g = dist.new_group(ranks=[c * seq_p_size + s for c in range(cfg_p_size)])
        if s == seq_idx:
            cfg_group = g
    return cfg_idx, seq_idx, seq_group, cfg_group


def cfg_parallel_combine(noise_pred_local: torch.Tensor, cfg_group, guide_scale: float) -> torch.Tensor:
    """CFG split over 2 ranks (lightx2v/models/networks/wan/model.py:412-433): rank 0 of the group ran
    the conditional pass, rank 1 the unconditional one; one all_gather, then u + g (c - u)."""
    parts = [torch.empty_like(noise_pred_local) for _ in range(2)]
    comm.transport().all_gather(parts, noise_pred_local.contiguous(), group=cfg_group)
    cond, uncond = parts
    return uncond + guide_scale * (cond - uncond)
