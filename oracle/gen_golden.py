"""Generate tests/golden/*.npz by running the REFERENCE's own hot-path modules (CPU, fp32).

Run only where the reference checkout exists (default /root/reference); the GPU box never runs
this.  The reference package is imported as a bare namespace (its `__init__` pulls every runner),
with empty stand-in modules for optional third-party imports that are absent here and unused on
this path (loguru, gguf, imageio, imageio_ffmpeg, torchvision).  The reference's dtype switch is
pointed at fp32 before its first use so `torch_sdpa` + `Default` matmul + torch norms run in fp32.

For every case it (1) builds one `WanTransformerAttentionBlock` with the synthetic checkpoint
from `wan_block_ref.make_block_weights`, (2) runs the reference `WanTransformerInfer.infer_block`
and `infer_non_blocks`, (3) asserts this repo's restatement agrees stage by stage, and (4) stores
inputs recipe + expected outputs (full for tiny cases, strided slices for large ones).

Usage: python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden] [--big]
"""

from __future__ import annotations

import argparse
import hashlib
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import wan_block_ref as R  # noqa: E402


def _install_reference(ref_root: str):
    class _AnyCall:
        def __getattr__(self, name):
            return lambda *a, **k: None

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("loguru", logger=_AnyCall())

    class _GG:
        def __getattr__(self, name):
            return name

    stub("gguf", GGMLQuantizationType=_GG())
    stub("imageio")
    stub("imageio_ffmpeg")
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    tv.transforms.functional = stub("torchvision.transforms.functional")
    tv.io = stub("torchvision.io")

    pkg = types.ModuleType("lightx2v")
    pkg.__path__ = [os.path.join(ref_root, "lightx2v")]
    sys.modules["lightx2v"] = pkg
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)

    import lightx2v.utils.envs as envs

    envs.DTYPE_MAP["BF16"] = torch.float32
    envs.GET_DTYPE.cache_clear()
    envs.GET_SENSITIVE_DTYPE.cache_clear()
    assert envs.GET_DTYPE() == torch.float32


def _ref_block_objects(cfg, w, block_index=0):
    import lightx2v.common.ops  # noqa: F401  (registers plugins)
    from lightx2v.models.networks.wan.infer.transformer_infer import WanTransformerInfer
    from lightx2v.models.networks.wan.weights.transformer_weights import WanTransformerAttentionBlock

    blk = WanTransformerAttentionBlock(block_index, cfg.get("task", "t2v"), "Default", cfg)
    sa, ca, ff = blk.compute_phases
    p = f"blocks.{block_index}."

    def set_mm(mm, name):
        mm.weight = w[p + name + ".weight"].t()
        mm.bias = w[p + name + ".bias"]

    sa.modulation.tensor = w[p + "modulation"]
    sa.norm1.weight = sa.norm1.bias = None
    for n in "qkvo":
        set_mm(getattr(sa, f"self_attn_{n}"), f"self_attn.{n}")
        set_mm(getattr(ca, f"cross_attn_{n}"), f"cross_attn.{n}")
    sa.self_attn_norm_q.weight = w[p + "self_attn.norm_q.weight"]
    sa.self_attn_norm_k.weight = w[p + "self_attn.norm_k.weight"]
    ca.norm3.weight = w[p + "norm3.weight"]
    ca.norm3.bias = w[p + "norm3.bias"]
    ca.cross_attn_norm_q.weight = w[p + "cross_attn.norm_q.weight"]
    ca.cross_attn_norm_k.weight = w[p + "cross_attn.norm_k.weight"]
    if hasattr(ca, "cross_attn_k_img"):  # i2v with a CLIP image encoder
        set_mm(ca.cross_attn_k_img, "cross_attn.k_img")
        set_mm(ca.cross_attn_v_img, "cross_attn.v_img")
        ca.cross_attn_norm_k_img.weight = w[p + "cross_attn.norm_k_img.weight"]
    ff.norm2.weight = ff.norm2.bias = None
    set_mm(ff.ffn_0, "ffn.0")
    set_mm(ff.ffn_2, "ffn.2")
    infer = WanTransformerInfer(cfg)
    return blk, infer


def _ref_cos_sin(cfg, grid):
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler

    sch = WanScheduler(cfg)
    return sch.prepare_cos_sin(grid)


def _sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def run_case(name, grid, dim, heads, ffn, text_len, seed, out_dir, full, stride_rows=None, stages_keep=None):
    from lightx2v.models.networks.wan.infer.module_io import GridOutput, WanPreInferModuleOutput

    f, h, w_ = grid
    S = f * h * w_
    cfg = {
        "task": "t2v", "model_cls": "wan2.1", "num_layers": 1, "num_heads": heads, "dim": dim,
        "self_attn_1_type": "torch_sdpa", "cross_attn_1_type": "torch_sdpa", "cross_attn_2_type": "torch_sdpa",
        "modulate_type": "torch", "rope_type": "torch", "seq_parallel": False, "cpu_offload": False,
        "run_device": "cpu", "infer_steps": 4, "target_video_length": 81, "sample_shift": 5,
        "sample_guide_scale": 5, "feature_caching": "NoCaching",
    }
    w = R.make_block_weights(dim, ffn, seed=seed)
    hw = R.make_head_weights(dim, 64, seed=seed)
    x, context, embed0, embed = R.make_inputs(S, dim, text_len, seed=seed)

    blk, infer = _ref_block_objects(cfg, w)
    cos_sin_ref = _ref_cos_sin(cfg, grid)
    infer.cos_sin = cos_sin_ref
    pre = WanPreInferModuleOutput(embed=embed, grid_sizes=GridOutput(tensor=torch.tensor([list(grid)], dtype=torch.int32), tuple=grid),
                                  x=x.clone(), embed0=embed0, context=context)
    t0 = time.time()
    with torch.no_grad():
        x_ref = infer.infer_block(blk, x.clone(), pre)
    t_ref = time.time() - t0

    # head (infer_non_blocks) through the reference's own classes
    from lightx2v.utils.registry_factory import LN_WEIGHT_REGISTER, MM_WEIGHT_REGISTER, TENSOR_REGISTER

    class _HW:
        pass

    hwobj = _HW()
    hwobj.norm = LN_WEIGHT_REGISTER["Default"]()
    hwobj.norm.weight = hwobj.norm.bias = None
    hwobj.head = MM_WEIGHT_REGISTER["Default"]("head.head.weight", "head.head.bias")
    hwobj.head.weight = hw["head.head.weight"].t()
    hwobj.head.bias = hw["head.head.bias"]
    hwobj.head_modulation = TENSOR_REGISTER["Default"]("head.modulation")
    hwobj.head_modulation.tensor = hw["head.modulation"]
    with torch.no_grad():
        head_ref = infer.infer_non_blocks(hwobj, x_ref.clone(), embed)

    # restatement
    cos_sin = R.build_cos_sin(grid, dim // heads)
    assert torch.equal(torch.view_as_real(cos_sin), torch.view_as_real(cos_sin_ref)), "RoPE table differs from reference"
    trace = R.BlockTrace()
    t0 = time.time()
    with torch.no_grad():
        x_mine = R.block_forward(x, embed0, context, w, cos_sin, heads, trace=trace)
        head_mine = R.head_forward(x_mine, embed, hw)
    t_mine = time.time() - t0
    err = (x_mine - x_ref).abs().max().item()
    herr = (head_mine - head_ref).abs().max().item()
    print(f"[{name}] S={S} D={dim} H={heads} F={ffn}: ref {t_ref:.2f}s mine {t_mine:.2f}s  max|x_mine-x_ref|={err:.3e} head={herr:.3e}")
    assert err <= 2e-5 * max(1.0, x_ref.abs().max().item()), err
    assert herr <= 2e-5 * max(1.0, head_ref.abs().max().item()), herr

    # stage checks against the reference's own ops, stage by stage
    from lightx2v.models.networks.wan.infer.utils import apply_wan_rope_with_torch

    sa, ca, ff = blk.compute_phases
    with torch.no_grad():
        n1 = sa.norm1.apply(x)
        q_ref = sa.self_attn_norm_q.apply(sa.self_attn_q.apply(trace.stages["norm1_out"]))
        assert (q_ref - trace.stages["q_rms"]).abs().max().item() < 1e-5
        qr, kr = apply_wan_rope_with_torch(trace.stages["q_rms"].view(S, heads, -1), trace.stages["k_rms"].view(S, heads, -1), cos_sin_ref)
        assert (qr.reshape(S, -1) - trace.stages["q_rope"]).abs().max().item() < 1e-5
        a_ref = sa.self_attn_1.apply(q=qr, k=kr, v=trace.stages["v"].view(S, heads, -1))
        assert (a_ref - trace.stages["self_attn_out"]).abs().max().item() < 1e-5
        assert (n1 - R.layer_norm(x)).abs().max().item() < 1e-5

    rec = {
        "grid": np.array(grid, dtype=np.int64), "dim": np.int64(dim), "heads": np.int64(heads), "ffn": np.int64(ffn),
        "text_len": np.int64(text_len), "seed": np.int64(seed),
        "sha_x": np.array(_sha(x)), "sha_wq": np.array(_sha(w["blocks.0.self_attn.q.weight"])),
        "ref_seconds": np.float64(t_ref),
    }
    if full:
        rec["x_out"] = x_ref.numpy()
        rec["head_out"] = head_ref.numpy()
        for k, v in trace.stages.items():
            rec["stage_" + k] = v.numpy()
    else:
        rows = torch.arange(0, S, stride_rows)
        rec["rows"] = rows.numpy()
        rec["x_out"] = x_ref[rows].numpy()
        rec["head_out"] = head_ref[rows].numpy()
        for k, v in trace.stages.items():
            if v.shape[0] == S and (stages_keep is None or k in stages_keep):
                rec["stage_" + k] = v[rows].numpy()
    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **rec)


def run_ulysses_case(out_dir, world=4, S=64, H=8, d=128, seed=3):
    """Reference `all2all_seq2head/head2seq` + `UlyssesAttnWeight` under gloo vs the single-process
    model in wan_block_ref (token/head permutation golden)."""
    import torch.multiprocessing as mp

    g = torch.Generator().manual_seed(seed)
    q = torch.randn(S, H, d, generator=g)
    k = torch.randn(S, H, d, generator=g)
    v = torch.randn(S, H, d, generator=g)
    tmp = os.path.join(out_dir, "_uly_tmp")
    os.makedirs(tmp, exist_ok=True)
    ref_root = [p for p in sys.path if os.path.isdir(os.path.join(p, "lightx2v"))][0]
    mp.spawn(_uly_worker, args=(world, q, k, v, tmp, ref_root), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp, f"o{r}.pt")) for r in range(world)]
    s2h = [torch.load(os.path.join(tmp, f"s2h{r}.pt")) for r in range(world)]
    mine = R.ulysses_self_attention(list(q.chunk(world)), list(k.chunk(world)), list(v.chunk(world)))
    mine_s2h = R.seq2head_reference(list(q.chunk(world)))
    for r in range(world):
        assert torch.equal(s2h[r], mine_s2h[r])
        assert (outs[r] - mine[r]).abs().max().item() < 1e-5
    full = R.sdpa(q, k, v)
    assert (torch.cat(outs, 0) - full).abs().max().item() < 1e-5
    np.savez_compressed(os.path.join(out_dir, f"ulysses_n{world}.npz"), world=np.int64(world), S=np.int64(S), H=np.int64(H), d=np.int64(d),
                        seed=np.int64(seed), out=torch.cat(outs, 0).numpy(), s2h_rank1=s2h[1].numpy())
    for f_ in os.listdir(tmp):
        os.remove(os.path.join(tmp, f_))
    os.rmdir(tmp)
    print(f"[ulysses_n{world}] reference gloo run == single-process model")


def _uly_worker(rank, world, q, k, v, tmp, ref_root):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29617"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_reference(ref_root)
    import lightx2v.common.ops  # noqa: F401
    from lightx2v.common.ops.attn.utils.all2all import all2all_seq2head
    from lightx2v.utils.registry_factory import ATTN_WEIGHT_REGISTER

    uly = ATTN_WEIGHT_REGISTER["ulysses"]()
    uly.set_config({"run_device": "cpu"})
    sd = ATTN_WEIGHT_REGISTER["torch_sdpa"]()
    ql, kl, vl = (t.chunk(world)[rank].contiguous() for t in (q, k, v))
    Sl = ql.shape[0]
    o = uly.apply(ql, kl, vl, img_qkv_len=Sl, cu_seqlens_qkv=torch.tensor([0, Sl], dtype=torch.int32), attention_module=sd, seq_p_group=None)
    torch.save(o, os.path.join(tmp, f"o{rank}.pt"))
    torch.save(all2all_seq2head(ql), os.path.join(tmp, f"s2h{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--big", action="store_true", help="also run the full 32760-token config-1 block (20+ s, 201 MB tensors)")
    ap.add_argument("--dims14b", action="store_true", help="only the block at Wan2.1-14B dims (D=5120, H=40, F=13824; 1.4 GB of fp32 weights)")
    args = ap.parse_args()
    _install_reference(args.ref)
    torch.manual_seed(0)
    if args.dims14b:
        keep = {"self_attn_out", "self_attn_y", "x_after_self", "cross_attn_y", "x_after_cross", "ffn_y", "x_out"}
        run_case("wan14b_g3x30x52", (3, 30, 52), 5120, 40, 13824, 512, seed=6, out_dir=args.out, full=False, stride_rows=199, stages_keep=keep)
        return
    # tiny full-tensor goldens (reduced D) and 1.3B-dim strided goldens
    run_case("tiny_d256_g2x4x4", (2, 4, 4), 256, 2, 512, 16, seed=1, out_dir=args.out, full=True)
    run_case("tiny_d256_g3x6x8", (3, 6, 8), 256, 2, 512, 40, seed=2, out_dir=args.out, full=True)
    run_case("small_d512_g3x5x7", (3, 5, 7), 512, 4, 1024, 77, seed=3, out_dir=args.out, full=True)
    run_case("wan1p3b_g3x6x8", (3, 6, 8), 1536, 12, 8960, 512, seed=4, out_dir=args.out, full=False, stride_rows=9)
    run_case("wan1p3b_g3x30x52", (3, 30, 52), 1536, 12, 8960, 512, seed=5, out_dir=args.out, full=False, stride_rows=293)
    if args.big:
        run_case("wan1p3b_g21x30x52", (21, 30, 52), 1536, 12, 8960, 512, seed=0, out_dir=args.out, full=False, stride_rows=2047)
    for n in (2, 4):
        run_ulysses_case(args.out, world=n)


if __name__ == "__main__":
    main()
