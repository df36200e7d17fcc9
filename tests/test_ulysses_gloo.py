"""The N>1 exchange logic on CPU: world_size 2 (and 4) gloo processes drive HipUlyssesAttnWeight with
host tensors.  The pack step and the attention module are test doubles (a torch permutation and the
oracle's SDPA) injected through the plugin's constructor/arguments, so what is checked here is the
product's all-to-all layout, group construction and CFG gather -- not a compute fallback."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wan_block_ref as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _torch_pack(q, k, v, world, out=None):
    rows, dim = q.shape
    seg = dim // world
    st = torch.stack([q, k, v], dim=1).view(rows, 3, world, seg).permute(2, 0, 1, 3)
    return st.contiguous() if out is None else out.copy_(st)


def _logical(op):
    """Walk a SegmentedOperand the way the GEMM's A cursor does: K column -> (region, segment, column)."""
    ks, sl = op.kseg, op.sl
    K = sum(op.widths) * op.world * op.d
    flat = op.flat
    cols = []
    for kcol in range(0, K, 64):
        if ks.split == 0 or kcol < ks.split:
            s_, c_ = divmod(kcol, ks.seg)
            base, lda = s_ * ks.stride + c_, ks.seg
        else:
            s_, c_ = divmod(kcol - ks.split, ks.seg2)
            base, lda = ks.base2 + s_ * ks.stride2 + c_, ks.lda2
        cols.append(torch.stack([flat[base + m * lda: base + m * lda + 64] for m in range(sl)]))
    return torch.cat(cols, dim=1)


class _OracleAttn:
    def apply(self, q, k, v, **kw):
        return R.sdpa(q.float(), k.float(), v.float()).to(q.dtype)


def _worker(rank, world, port, cfg_p, seq_p, out_dir, heads_per_rank=2, chunks=2):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lightx2v_amd import ulysses

    cfg_idx, seq_idx, seq_group, cfg_group = ulysses.make_groups(cfg_p, seq_p)
    assert rank == cfg_idx * seq_p + seq_idx
    if chunks:
        os.environ["LX2V_ULYSSES_CHUNKS"] = str(chunks)
    else:  # unset: the count follows the group size (4 chunks in a group of two, 2 otherwise)
        os.environ.pop("LX2V_ULYSSES_CHUNKS", None)
        chunks = ulysses.default_chunks(seq_p)
        assert chunks == (4 if seq_p == 2 else 2)
    S, H, d = 8 * seq_p, heads_per_rank * seq_p, 128
    g = torch.Generator().manual_seed(100 + cfg_idx)  # cond / uncond branches see different data
    q, k, v = (torch.randn(S, H, d, generator=g) for _ in range(3))
    ql, kl, vl = (t.chunk(seq_p)[seq_idx].contiguous() for t in (q, k, v))
    Sl = S // seq_p
    uly = ulysses.HipUlyssesAttnWeight(pack_fn=_torch_pack)
    out = uly.apply(ql, kl, vl, img_qkv_len=Sl, cu_seqlens_qkv=torch.tensor([0, Sl], dtype=torch.int32),
                    attention_module=_OracleAttn(), seq_p_group=seq_group)
    ref = R.sdpa(q, k, v).chunk(seq_p)[seq_idx]
    err = (out - ref).abs().max().item()
    # zero-copy form: logical matrix [Sl, H*d] described by K segments
    op = uly.apply_segmented(ql, kl, vl, attention_module=_OracleAttn(), seq_p_group=seq_group)
    assert op.widths == ulysses.split_heads(heads_per_rank, chunks) and op.first.shape == (Sl, op.widths[0] * d)
    err2 = max((_logical(op) - ref).abs().max().item(), (op.gathered() - ref).abs().max().item())
    # H not divisible by N must raise, not floor silently
    raised = False
    try:
        uly.apply(ql[:, :H - 1], kl[:, :H - 1], vl[:, :H - 1], img_qkv_len=Sl, cu_seqlens_qkv=torch.tensor([0, Sl]), attention_module=_OracleAttn(), seq_p_group=seq_group)
    except ValueError:
        raised = True
    # fp8 payload: q, k, v and the attention output are quantised per (token, head) around the exchanges
    rt = R.fp8_comm_roundtrip
    uly8 = ulysses.HipUlyssesAttnWeight(pack_fn=_torch_pack, quant_fn=R.quant_fp8_rowwise,
                                        dequant_fn=lambda q8, sc, out: out.copy_(R.dequant_fp8_rowwise(q8, sc, torch.bfloat16)))
    out8 = uly8.apply(ql, kl, vl, img_qkv_len=Sl, cu_seqlens_qkv=torch.tensor([0, Sl], dtype=torch.int32),
                      attention_module=_OracleAttn(), seq_p_group=seq_group, use_fp8_comm=True)
    ref8 = rt(R.sdpa(rt(q), rt(k), rt(v)).view(S, H, d)).view(S, H * d).chunk(seq_p)[seq_idx]
    err8 = ((out8 - ref8).abs().max() / ref8.abs().max()).item()  # a value on a rounding boundary may land one e4m3 step away
    cos8 = R.cosine(out8, ref8)
    ok_cfg = True
    if cfg_p == 2:
        local = torch.full((4, 3), float(cfg_idx + 1))  # cond rank holds 1s, uncond rank 2s
        comb = ulysses.cfg_parallel_combine(local, cfg_group, 5.0)
        ok_cfg = torch.allclose(comb, torch.full((4, 3), 2.0 + 5.0 * (1.0 - 2.0)))
    torch.save({"err": err, "err2": err2, "raised": raised or seq_p == 1, "ok_cfg": ok_cfg, "err8": err8, "cos8": cos8, "fp8_changes": (out8 - out).abs().max().item()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cfg_p,seq_p,heads_per_rank,chunks", [(1, 2, 2, 2), (2, 2, 2, 2), (1, 4, 2, 2),
                                                              (1, 2, 5, 2),    # 5 heads per rank -> chunks of 3 + 2 (14B at seq_p = 8)
                                                              (1, 2, 5, 3),    # 2 + 2 + 1
                                                              (1, 2, 3, 1),    # one exchange, no overlap
                                                              (1, 2, 8, 0),    # default in a group of two: 4 chunks of 2 heads
                                                              (1, 4, 3, 0)])   # default in a group of four: 2 chunks, 2 + 1
def test_ulysses_and_cfg_groups_gloo(tmp_path, cfg_p, seq_p, heads_per_rank, chunks):
    world = cfg_p * seq_p
    mp.spawn(_worker, args=(world, _free_port(), cfg_p, seq_p, str(tmp_path), heads_per_rank, chunks), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert res["err"] < 1e-5 and res["err2"] < 1e-5, res
        assert res["raised"] and res["ok_cfg"], res
        assert res["err8"] <= 0.07 and res["cos8"] >= 0.9999 and res["fp8_changes"] > 0, res


def _txt_worker(rank, world, port, golden_path, out_dir):
    import numpy as np

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lightx2v_amd import ulysses

    rec = np.load(golden_path)
    li, lt, H, d, seed = (int(rec[k]) for k in ("li", "lt", "H", "d", "seed"))
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(li + lt, H, d, generator=g) for _ in range(3))
    sl = li // world
    ql, kl, vl = (torch.cat([t[rank * sl:(rank + 1) * sl], t[li:]]) for t in (q, k, v))
    errs = {}
    for chunks in (1, 2):
        os.environ["LX2V_ULYSSES_CHUNKS"] = str(chunks)
        uly = ulysses.HipUlyssesAttnWeight(pack_fn=_torch_pack)
        out = uly.apply(ql, kl, vl, img_qkv_len=sl, cu_seqlens_qkv=torch.tensor([0, sl + lt], dtype=torch.int32), attention_module=_OracleAttn(), seq_p_group=None)
        errs[chunks] = (out - torch.from_numpy(rec["out"][rank])).abs().max().item()
    torch.save(errs, os.path.join(out_dir, f"t{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ulysses_with_replicated_text_rows_matches_reference_run(tmp_path, world):
    """Image rows sharded, text rows replicated (HunyuanVideo layout): every rank's output equals
    what the reference's UlyssesAttnWeight produced under gloo (fixture from oracle/gen_golden_hy.py)."""
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"hy_ulysses_txt_n{world}.npz")
    mp.spawn(_txt_worker, args=(world, _free_port(), golden, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        errs = torch.load(os.path.join(str(tmp_path), f"t{r}.pt"))
        assert all(e < 1e-5 for e in errs.values()), errs
