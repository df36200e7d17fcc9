"""Generate tests/golden/hy_*.npz by running the REFERENCE's HunyuanVideo-1.5 transformer classes
(CPU, fp32): `HunyuanVideo15TransformerWeights`, `HunyuanVideo15TransformerInfer.infer_double_block`
/ `infer_final_layer`, `HunyuanVideo15Scheduler.prepare_cos_sin`, and -- under gloo -- its
`UlyssesAttnWeight` with text tokens after the image tokens.

Run only where the reference checkout exists; the GPU box never runs this.  The reference is
imported the same way as in gen_golden.py (bare namespace package + empty stand-ins for absent
optional third-party modules).  For every case the script asserts that the restatement in
hy_block_ref.py agrees with the reference and then stores recipe + expected outputs.

Usage: python oracle/gen_golden_hy.py [--ref /root/reference] [--out tests/golden]
"""

from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import hy_block_ref as HY  # noqa: E402
from oracle import wan_block_ref as R  # noqa: E402
from oracle.gen_golden import _install_reference, _sha  # noqa: E402


def _cfg(dim, heads, depth, rope_dim_list):
    return {
        "task": "t2v", "mm_double_blocks_depth": depth, "heads_num": heads, "hidden_size": dim, "run_device": "cpu",
        "seq_parallel": False, "cpu_offload": False, "modulate_type": "torch", "rope_type": "torch", "attn_type": "torch_sdpa",
        "ln_type": "Default", "rms_type": "sgl-kernel", "rope_dim_list": list(rope_dim_list), "rope_theta": HY.ROPE_THETA,
        "sample_shift": 5.0, "sample_guide_scale": -1.0, "infer_steps": 4,
    }


def _ref_objects(cfg, blocks, fw):
    import lightx2v.common.ops  # noqa: F401  (registers plugins)
    from lightx2v.models.networks.hunyuan_video.infer.transformer_infer import HunyuanVideo15TransformerInfer
    from lightx2v.models.networks.hunyuan_video.weights.transformer_weights import HunyuanVideo15TransformerWeights

    W = HunyuanVideo15TransformerWeights(cfg)

    def set_mm(mm, w, name):
        mm.weight = w[name + ".weight"].t()
        mm.bias = w[name + ".bias"]

    for i, w in enumerate(blocks):
        p = f"double_blocks.{i}."
        blk = W.double_blocks[i]
        for br, mod in (("img", blk.img_branch), ("txt", blk.txt_branch)):
            set_mm(getattr(mod, f"{br}_mod"), w, f"{p}{br}_mod.linear")
            for n in ("attn_q", "attn_k", "attn_v", "attn_proj"):
                set_mm(getattr(mod, f"{br}_{n}"), w, f"{p}{br}_{n}")
            set_mm(getattr(mod, f"{br}_mlp_fc1"), w, f"{p}{br}_mlp.fc1")
            set_mm(getattr(mod, f"{br}_mlp_fc2"), w, f"{p}{br}_mlp.fc2")
            getattr(mod, f"{br}_attn_q_norm").weight = w[f"{p}{br}_attn_q_norm.weight"]
            getattr(mod, f"{br}_attn_k_norm").weight = w[f"{p}{br}_attn_k_norm.weight"]
            for ln in ("norm1", "norm2"):
                getattr(mod, f"{br}_{ln}").weight = getattr(mod, f"{br}_{ln}").bias = None
    set_mm(W.final_layer.adaLN_modulation, fw, "final_layer.adaLN_modulation.1")
    set_mm(W.final_layer.linear, fw, "final_layer.linear")
    W.final_layer.norm_final.weight = W.final_layer.norm_final.bias = None
    return W, HunyuanVideo15TransformerInfer(cfg)


def _ref_cos_sin(cfg, grid):
    from lightx2v.models.schedulers.hunyuan_video import scheduler as S

    sch = S.HunyuanVideo15Scheduler.__new__(S.HunyuanVideo15Scheduler)  # prepare_cos_sin needs only these three fields
    sch.config, sch.run_device, sch.seq_p_group = cfg, torch.device("cpu"), None
    import lightx2v.models.schedulers.hunyuan_video.posemb_layers as P

    orig = P.get_1d_rotary_pos_embed
    S.get_nd_rotary_pos_embed.__globals__["get_1d_rotary_pos_embed"] = lambda *a, **k: orig(*a, **{**k, "run_device": "cpu"})
    try:
        return sch.prepare_cos_sin(grid)
    finally:
        S.get_nd_rotary_pos_embed.__globals__["get_1d_rotary_pos_embed"] = orig


def run_case(name, grid, dim, heads, mlp, txt_len, depth, seed, out_dir, rope_dim_list, full):
    from lightx2v.models.networks.hunyuan_video.infer.module_io import HunyuanVideo15InferModuleOutput

    L = grid[0] * grid[1] * grid[2]
    cfg = _cfg(dim, heads, depth, rope_dim_list)
    blocks = [HY.make_double_block_weights(dim, heads, mlp, seed=seed + i, block_index=i) for i in range(depth)]
    fw = HY.make_final_weights(dim, 32, seed=seed)
    img, txt, vec = HY.make_inputs(L, txt_len, dim, seed=seed)
    W, infer = _ref_objects(cfg, blocks, fw)

    class _Sch:
        pass

    sch = _Sch()
    sch.cos_sin = _ref_cos_sin(cfg, grid)
    infer.set_scheduler(sch)
    cos_sin = HY.build_cos_sin(grid, dim // heads, rope_dim_list)
    assert torch.equal(cos_sin, sch.cos_sin), "RoPE table differs from the reference"

    io = HunyuanVideo15InferModuleOutput(img=img.clone(), txt=txt.clone(), vec=vec, grid_sizes=grid)
    with torch.no_grad():
        i1, t1 = infer.infer_double_block(W.double_blocks[0], io)
        io2 = HunyuanVideo15InferModuleOutput(img=img.clone(), txt=txt.clone(), vec=vec, grid_sizes=grid)
        out_ref = infer.infer(W, io2)  # all blocks + final layer, [1, L, 32]
    trace = HY.DoubleBlockTrace()
    with torch.no_grad():
        i1m, t1m = HY.double_block_forward(img[0], txt[0], vec, blocks[0], cos_sin, heads, trace=trace)
        out_m = HY.forward(img[0], txt[0], vec, blocks, fw, cos_sin, heads)
    e_img = (i1m - i1[0]).abs().max().item()
    e_txt = (t1m - t1[0]).abs().max().item()
    e_out = (out_m - out_ref[0]).abs().max().item()
    print(f"[{name}] L={L}+{txt_len} D={dim} H={heads}: block img {e_img:.3e} txt {e_txt:.3e}; {depth} blocks + final {e_out:.3e}")
    assert max(e_img, e_txt) <= 2e-5 * max(1.0, i1.abs().max().item())
    assert e_out <= 2e-5 * max(1.0, out_ref.abs().max().item())

    rec = {"grid": np.array(grid, dtype=np.int64), "dim": np.int64(dim), "heads": np.int64(heads), "mlp": np.int64(mlp), "txt_len": np.int64(txt_len),
           "depth": np.int64(depth), "seed": np.int64(seed), "rope_dim_list": np.array(rope_dim_list, dtype=np.int64),
           "sha_img": np.array(_sha(img)), "sha_wq": np.array(_sha(blocks[0]["double_blocks.0.img_attn_q.weight"])),
           "img_out": i1[0].numpy(), "txt_out": t1[0].numpy(), "final_out": out_ref[0].numpy()}
    if full:
        for k, v in trace.stages.items():
            rec["stage_" + k] = v.numpy()
        rec["cos_sin"] = sch.cos_sin.numpy()
    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **rec)


def _uly_worker(rank, world, q, k, v, li, tmp, ref_root):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29619"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_reference(ref_root)
    import lightx2v.common.ops  # noqa: F401
    from lightx2v.utils.registry_factory import ATTN_WEIGHT_REGISTER

    uly = ATTN_WEIGHT_REGISTER["ulysses"]()
    uly.set_config({"run_device": "cpu"})
    sd = ATTN_WEIGHT_REGISTER["torch_sdpa"]()
    sl = li // world
    S = q.shape[0]
    ql, kl, vl = (torch.cat([t[rank * sl:(rank + 1) * sl], t[li:]]) for t in (q, k, v))
    o = uly.apply(ql, kl, vl, img_qkv_len=sl, cu_seqlens_qkv=torch.tensor([0, sl + (S - li)], dtype=torch.int32), attention_module=sd, seq_p_group=None)
    torch.save(o, os.path.join(tmp, f"o{rank}.pt"))
    dist.destroy_process_group()


def run_ulysses_txt_case(out_dir, world=2, li=48, lt=11, H=4, d=128, seed=5):
    """Reference Ulysses with a replicated text part vs the single-process model and vs plain joint
    attention."""
    import torch.multiprocessing as mp

    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(li + lt, H, d, generator=g) for _ in range(3))
    tmp = os.path.join(out_dir, "_uly_tmp")
    os.makedirs(tmp, exist_ok=True)
    ref_root = [p for p in sys.path if os.path.isdir(os.path.join(p, "lightx2v"))][0]
    mp.spawn(_uly_worker, args=(world, q, k, v, li, tmp, ref_root), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp, f"o{r}.pt")) for r in range(world)]
    sl = li // world
    shards = [[torch.cat([t[r * sl:(r + 1) * sl], t[li:]]) for r in range(world)] for t in (q, k, v)]
    mine = HY.ulysses_joint_attention(*shards, img_len_local=sl)
    full = R.sdpa(q, k, v)
    for r in range(world):
        assert (outs[r] - mine[r]).abs().max().item() < 1e-5
        assert (outs[r][:sl] - full[r * sl:(r + 1) * sl]).abs().max().item() < 1e-5
        assert (outs[r][sl:] - full[li:]).abs().max().item() < 1e-5
    np.savez_compressed(os.path.join(out_dir, f"hy_ulysses_txt_n{world}.npz"), world=np.int64(world), li=np.int64(li), lt=np.int64(lt), H=np.int64(H),
                        d=np.int64(d), seed=np.int64(seed), out=torch.stack(outs).numpy())
    for f_ in os.listdir(tmp):
        os.remove(os.path.join(tmp, f_))
    os.rmdir(tmp)
    print(f"[hy_ulysses_txt_n{world}] reference gloo run == single-process model == joint attention")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    a = ap.parse_args()
    _install_reference(a.ref)
    torch.manual_seed(0)
    run_case("hy_tiny_d256", (2, 4, 6), 256, 2, 1024, 19, 2, 11, a.out, HY.ROPE_DIM_LIST, full=True)
    run_case("hy_d2048_g3x6x8", (3, 6, 8), 2048, 16, 8192, 40, 1, 12, a.out, HY.ROPE_DIM_LIST, full=False)
    run_ulysses_txt_case(a.out, world=2)
    run_ulysses_txt_case(a.out, world=4, li=64, lt=7, H=8, seed=6)
