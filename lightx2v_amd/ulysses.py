"""Ulysses sequence parallelism over RCCL: head <-> sequence reshard around any attention plugin.

Stands in for `UlyssesAttnWeight` (lightx2v/common/ops/attn/ulysses_attn.py:11-137) and its helpers
`all2all_seq2head` / `all2all_head2seq` (lightx2v/common/ops/attn/utils/all2all.py:7-89) with the
same result -- global token order is rank-major chunks, the output's heads are in global head
order -- and a different data path:

  * Heads are dealt to ranks in groups of H/(N*C), round-robin (group g -> rank g % N), not in
    one contiguous run per rank: with C > 1 ("chunks") a rank attends its C groups one after the
    other, and the all-to-all of the next group and the return of the previous one overlap the
    attention of the current one (async collectives on RCCL's own stream).

  * q, k and v travel in ONE all-to-all (the reference issues three).  The send buffer is
    [dest rank][S/N][q|k|v][H/N*d], written by one pack kernel, so on arrival token (src, s) of
    tensor `which` sits at ((src*S/N + s)*3 + which)*H/N*d: a uniform token stride of 3*H/N*d that
    the attention kernel reads in place (no transpose/contiguous copies on either side).
  * The attention output [S, H/N*d] is already the send layout of the return all-to-all.
  * The receive buffer [src rank][S/N][H/N*d] can be handed to the output projection as a
    K-segmented operand (`apply_segmented`), which removes the final gather copy too.

On MI355X every GPU pair has its own xGMI link, so an all-to-all drives all 7 links at once;
`torch.distributed` backend "nccl" is RCCL on ROCm.
"""

from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import hip
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


@ATTN_WEIGHT_REGISTER("hip_ulysses")
class HipUlyssesAttnWeight(_AttnBase):
    def __init__(self, pack_fn: Optional[Callable] = None, quant_fn: Optional[Callable] = None, dequant_fn: Optional[Callable] = None):
        """`pack_fn(q2d, k2d, v2d, world) -> [world, rows, 3, dim/world]`; defaults to the HIP pack
        kernel.  `quant_fn(x2d) -> (e4m3, scale)` / `dequant_fn(e4m3, scale, out=)` serve the fp8
        payload and default to the HIP kernels.  (Injectable so the exchange logic can be exercised
        with host tensors in tests.)"""
        super().__init__()
        self._pack = pack_fn or hip.ulysses_pack_qkv
        self._quant = quant_fn or hip.quant_fp8_rowwise
        self._dequant = dequant_fn or hip.dequant_fp8_rowwise
        # Head chunks per rank: the exchange of chunk c+1 (and the return of chunk c-1) overlaps the
        # attention of chunk c.  1 = one exchange, no overlap.
        self.chunks = max(1, int(os.environ.get("LX2V_ULYSSES_CHUNKS", "2")))

    # ---- exchange core -------------------------------------------------------------------------
    def _exchange_attend(self, q, k, v, attention_module, group, model_cls=None, txt=None, fp8=False) -> Tuple[torch.Tensor, int, int, int, Optional[torch.Tensor]]:
        """Returns (back, segments, sl, seg, txt_out): `back` is [segments][sl][seg] with the segments in
        global head order, i.e. the logical [sl, H*d] output has columns [g*seg, (g+1)*seg) at back[g].

        `txt` = (tq, tk, tv), each [Lt, H, d] and identical on every rank (HunyuanVideo layout,
        ulysses_attn.py:80-88,105-116): the text rows are appended after the gathered image rows of
        every head group this rank attends, and their outputs are gathered along the head axis into
        `txt_out` [Lt, H*d] (None without a text part).

        `fp8` (`seq_p_fp8_comm`, ulysses_attn.py:59-73,125-131): what crosses the links is e4m3 with one
        fp32 scale per (token, head) -- q, k, v on the way out, the attention output on the way back --
        and is widened to bf16 on arrival; 0.53x the bytes.  Text rows never leave the rank and stay bf16."""
        if q.dim() == 4:
            q, k, v = (t.reshape(-1, t.shape[-2], t.shape[-1]) for t in (q, k, v))
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        sl, heads, d = q.shape
        if heads % world != 0:
            raise ValueError(f"hip_ulysses: num_heads={heads} is not divisible by the sequence-parallel size {world}")
        hs = heads // world
        chunks = max(c for c in range(1, self.chunks + 1) if hs % c == 0)
        hc = hs // chunks
        dim = heads * d
        lt = 0 if txt is None else txt[0].shape[0]
        # Head group g = heads [g*hc, (g+1)*hc) goes to rank g % world as its chunk g // world, so the
        # pack kernel's natural order [g][sl][q|k|v][hc*d] is already [chunk][dest rank][...]: every
        # chunk is one contiguous all-to-all, and the returned segments come back in head order.
        send = self._pack(q.reshape(sl, dim), k.reshape(sl, dim), v.reshape(sl, dim), world * chunks).view(chunks, world, sl, 3, hc * d)
        recv = torch.empty((chunks, world * sl + lt, 3, hc * d), dtype=send.dtype, device=send.device)  # image rows of all ranks, then text rows
        if fp8:
            s8, ssc = self._quant(send.view(-1, d))
            s8, ssc = s8.view(torch.uint8).view(chunks, world, sl, 3, hc * d), ssc.view(chunks, world, sl, 3 * hc)
            r8, rsc = torch.empty_like(s8), torch.empty_like(ssc)
            arrivals = [_Both(dist.all_to_all_single(r8[c], s8[c], group=group, async_op=True), dist.all_to_all_single(rsc[c], ssc[c], group=group, async_op=True))
                        for c in range(chunks)]
        else:
            arrivals = [dist.all_to_all_single(recv[c, :world * sl].view(world, sl, 3, hc * d), send[c], group=group, async_op=True) for c in range(chunks)]
        if lt:
            for c in range(chunks):
                g = c * world + rank
                for i, t in enumerate(txt):
                    recv[c, world * sl:, i] = t[:, g * hc:(g + 1) * hc].reshape(lt, hc * d)
        back = torch.empty((chunks, world, sl, hc * d), dtype=send.dtype, device=send.device)  # [chunk][src rank][sl][hc*d]
        txt_g = torch.empty((chunks, world, lt, hc * d), dtype=send.dtype, device=send.device) if lt else None
        if fp8:
            b8 = torch.empty((chunks, world, sl, hc * d), dtype=torch.uint8, device=send.device)
            bsc = torch.empty((chunks, world, sl, hc), dtype=torch.float32, device=send.device)
        returns, keep = [], []
        for c in range(chunks):
            # chunk c+1.. are still in flight while chunk c is attended; its output leaves while chunk c+1 is attended
            arrivals[c].wait()
            if fp8:
                self._dequant(r8[c].view(-1, d).view(torch.float8_e4m3fn), rsc[c].view(-1), out=recv[c, :world * sl].view(-1, d))
            r = recv[c].view(world * sl + lt, 3, hc, d)
            attn = attention_module.apply(q=r[:, 0], k=r[:, 1], v=r[:, 2], cu_seqlens_q=None, cu_seqlens_kv=None,
                                          max_seqlen_q=world * sl + lt, max_seqlen_kv=world * sl + lt, model_cls=model_cls)  # [S (+Lt), hc*d]
            attn = attn.contiguous()
            keep.append(attn)
            if fp8:
                a8, asc = self._quant(attn[:world * sl].view(-1, d))
                a8, asc = a8.view(torch.uint8).view(world, sl, hc * d), asc.view(world, sl, hc)
                keep += [a8, asc]
                returns.append(dist.all_to_all_single(b8[c], a8, group=group, async_op=True))
                returns.append(dist.all_to_all_single(bsc[c], asc, group=group, async_op=True))
            else:
                returns.append(dist.all_to_all_single(back[c], attn[:world * sl].view(world, sl, hc * d), group=group, async_op=True))
            if lt:
                returns.append(dist.all_gather(list(txt_g[c].unbind(0)), attn[world * sl:], group=group, async_op=True))
        for w in returns:
            w.wait()
        if fp8:
            self._dequant(b8.view(-1, d).view(torch.float8_e4m3fn), bsc.view(-1), out=back.view(-1, d))
        txt_out = txt_g.permute(2, 0, 1, 3).reshape(lt, dim) if lt else None  # group g = c*world + r -> columns [g*hc*d, ...)
        return back, world * chunks, sl, hc * d, txt_out

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
        back, nseg, sl, seg, txt_out = self._exchange_attend(q[:img_qkv_len], k[:img_qkv_len], v[:img_qkv_len], attention_module, seq_p_group, model_cls, txt, fp8=bool(use_fp8_comm))
        img_out = back.view(nseg, sl, seg).permute(1, 0, 2).reshape(sl, nseg * seg)
        return img_out if txt_out is None else torch.cat([img_out, txt_out])

    # ---- zero-copy form for the fused block runner ---------------------------------------------
    def apply_segmented(self, q, k, v, attention_module=None, seq_p_group=None, model_cls=None, use_fp8_comm=False):
        """Returns (buffer, a_kseg, a_kseg_stride): logical [S/N, H*d] matrix whose K columns
        [g*seg, (g+1)*seg) live at buffer + g*a_kseg_stride + row*seg."""
        back, nseg, sl, seg, _ = self._exchange_attend(q, k, v, attention_module, seq_p_group, model_cls, fp8=bool(use_fp8_comm))
        return back.view(nseg * sl, seg)[:sl], seg, sl * seg


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
    dist.all_gather(parts, noise_pred_local.contiguous(), group=cfg_group)
    cond, uncond = parts
    return uncond + guide_scale * (cond - uncond)
