"""Generate tests/golden/step_teacache_d256.npz by running the REFERENCE's TeaCache runner
(`WanTransformerInferTeaCaching`) through a CFG denoise loop on a tiny model (CPU, fp32).

Same import method as gen_golden.py.  Two environment stand-ins that do not touch arithmetic: the
reference moves its cached tensors with `.cuda()`, which has no device here, so `Tensor.cuda` is the
identity for the duration of the run; the polynomial coefficients are the ones the reference ships
in configs/caching/teacache/wan_t2v_1_3b_tea_480p.json (read as data).  A random-init time MLP
changes its output by ~80 % from step to step (a trained one by a few percent), which the shipped
polynomial maps to several hundred, so the threshold is set to 1500 (every decision then has a margin of >= 18 %) to make the tiny model skip
some steps and compute others.

Usage: python oracle/gen_golden_teacache.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import gen_golden as G  # noqa: E402
from oracle import gen_golden_step as GS  # noqa: E402
from oracle import wan_step_ref as T  # noqa: E402


def run(name, ref_root, out_dir, latent_shape=(16, 3, 8, 12), dim=256, heads=2, ffn=512, layers=2, text_len=16, text_dim=128, steps=12, shift=5.0,
        guide=5.0, thresh=1500.0, use_ret_steps=False, seed=9):
    coefficients = json.load(open(os.path.join(ref_root, "configs/caching/teacache/wan_t2v_1_3b_tea_480p.json")))["coefficients"]
    cfg = {
        "dim": dim, "num_heads": heads, "num_layers": layers, "ffn_dim": ffn, "in_dim": 16, "out_dim": 16, "freq_dim": 256, "text_len": text_len,
        "task": "t2v", "model_cls": "wan2.1", "run_device": "cpu", "cpu_offload": False, "seq_parallel": False, "cfg_parallel": False,
        "self_attn_1_type": "torch_sdpa", "cross_attn_1_type": "torch_sdpa", "cross_attn_2_type": "torch_sdpa", "modulate_type": "torch",
        "rope_type": "torch", "infer_steps": steps, "target_video_length": 4 * (latent_shape[1] - 1) + 1, "sample_shift": shift,
        "sample_guide_scale": guide, "enable_cfg": True, "feature_caching": "Tea", "coefficients": coefficients, "use_ret_steps": use_ret_steps,
        "teacache_thresh": thresh,
    }
    model = GS.build_model(dim, ffn, layers, text_dim, seed)
    pre_w, blocks, head, pre, _, post, sch = GS.reference_objects(cfg, model, layers)
    from lightx2v.models.networks.wan.infer.feature_caching.transformer_infer import WanTransformerInferTeaCaching

    infer = WanTransformerInferTeaCaching(cfg)
    infer.set_scheduler(sch)
    g = torch.Generator().manual_seed(seed * 7919 + 17)
    context = torch.randn(text_len, text_dim, generator=g)
    context_null = torch.randn(text_len, text_dim, generator=g)
    inputs = {"text_encoder_output": {"context": context.unsqueeze(0), "context_null": context_null.unsqueeze(0)}}
    sch.prepare(seed, latent_shape)
    latents0 = sch.latents.clone()
    infer.cos_sin = sch.cos_sin

    def ref_forward(cond):
        sch.infer_condition = cond
        out = pre.infer(pre_w, inputs)
        x = infer.infer_main_blocks(blocks, out)
        y = infer.infer_non_blocks(head, x, out.embed)
        return post.infer(y, out)[0]

    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        store = {"latents0": latents0.numpy(), "context": context.numpy(), "context_null": context_null.numpy()}
        for i in range(steps):
            sch.step_pre(step_index=i)
            cond = ref_forward(True)
            uncond = ref_forward(False)
            sch.noise_pred = uncond + sch.sample_guide_scale * (cond - uncond)
            sch.step_post()
            store[f"latents_{i + 1}"] = sch.latents.numpy()
    finally:
        torch.Tensor.cuda = orig_cuda
    rec, rec2 = list(sch.caching_records), list(sch.caching_records_2)
    final, traj, my_rec, my_rec2 = T.sample_loop_teacache(model, latents0, context, context_null, heads, layers, steps, shift, guide, thresh, coefficients, use_ret_steps)
    assert my_rec == rec and my_rec2 == rec2, (rec, my_rec, rec2, my_rec2)
    assert not all(rec) and any(rec[1:]), f"threshold gives a degenerate pattern: {rec}"
    worst = max((a - torch.from_numpy(store[f"latents_{i + 1}"])).abs().max().item() for i, a in enumerate(traj))
    assert worst < 2e-4, worst
    meta = dict(latent_shape=latent_shape, dim=dim, heads=heads, ffn=ffn, layers=layers, text_len=text_len, text_dim=text_dim, steps=steps, shift=shift,
                guide=guide, seed=seed, thresh=thresh, use_ret_steps=use_ret_steps, coefficients=coefficients)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=np.array(repr(meta)), records=np.array(rec), records_2=np.array(rec2), **store)
    print(f"{name}: records {''.join('C' if r else '-' for r in rec)} / {''.join('C' if r else '-' for r in rec2)}; worst |restatement - reference| = {worst:.3e}")


def run_mag(name, ref_root, out_dir, latent_shape=(16, 3, 8, 12), dim=256, heads=2, ffn=512, layers=2, text_len=16, text_dim=128, steps=12, shift=5.0,
            guide=5.0, thresh=0.12, K=2, retention_ratio=0.2, seed=10):
    """MagCache (`WanTransformerInferMagCaching`, feature_caching "Mag").  The magnitude ratios are the first
    `steps` entries of the two lists the reference ships in configs/caching/magcache/ (read as data);
    threshold and K are set so that the 12-step loop skips some forwards and computes others."""
    import glob

    shipped = json.load(open(sorted(glob.glob(os.path.join(ref_root, "configs/caching/magcache/*.json")))[0]))["magcache_ratios"]
    ratios = [list(shipped[0][:steps]), list(shipped[1][:steps])]
    cfg = {
        "dim": dim, "num_heads": heads, "num_layers": layers, "ffn_dim": ffn, "in_dim": 16, "out_dim": 16, "freq_dim": 256, "text_len": text_len,
        "task": "t2v", "model_cls": "wan2.1", "run_device": "cpu", "cpu_offload": False, "seq_parallel": False, "cfg_parallel": False,
        "self_attn_1_type": "torch_sdpa", "cross_attn_1_type": "torch_sdpa", "cross_attn_2_type": "torch_sdpa", "modulate_type": "torch",
        "rope_type": "torch", "infer_steps": steps, "target_video_length": 4 * (latent_shape[1] - 1) + 1, "sample_shift": shift,
        "sample_guide_scale": guide, "enable_cfg": True, "feature_caching": "Mag", "magcache_calibration": False, "magcache_K": K,
        "magcache_thresh": thresh, "magcache_retention_ratio": retention_ratio, "magcache_ratios": ratios,
    }
    model = GS.build_model(dim, ffn, layers, text_dim, seed)
    pre_w, blocks, head, pre, _, post, sch = GS.reference_objects(cfg, model, layers)
    from lightx2v.models.networks.wan.infer.feature_caching.transformer_infer import WanTransformerInferMagCaching

    infer = WanTransformerInferMagCaching(cfg)
    infer.set_scheduler(sch)
    g = torch.Generator().manual_seed(seed * 7919 + 17)
    context = torch.randn(text_len, text_dim, generator=g)
    context_null = torch.randn(text_len, text_dim, generator=g)
    inputs = {"text_encoder_output": {"context": context.unsqueeze(0), "context_null": context_null.unsqueeze(0)}}
    sch.prepare(seed, latent_shape)
    latents0 = sch.latents.clone()
    infer.cos_sin = sch.cos_sin
    ran = {True: [], False: []}
    real_calc = infer.infer_calculating

    def traced(*a, **k):
        ran[sch.infer_condition][-1] = True
        return real_calc(*a, **k)

    infer.infer_calculating = traced

    def ref_forward(cond):
        sch.infer_condition = cond
        ran[cond].append(False)
        out = pre.infer(pre_w, inputs)
        x = infer.infer_main_blocks(blocks, out)
        return post.infer(infer.infer_non_blocks(head, x, out.embed), out)[0]

    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        store = {"latents0": latents0.numpy(), "context": context.numpy(), "context_null": context_null.numpy()}
        for i in range(steps):
            sch.step_pre(step_index=i)
            cond = ref_forward(True)
            uncond = ref_forward(False)
            sch.noise_pred = uncond + sch.sample_guide_scale * (cond - uncond)
            sch.step_post()
            store[f"latents_{i + 1}"] = sch.latents.numpy()
    finally:
        torch.Tensor.cuda = orig_cuda
    rec, rec2 = ran[True], ran[False]
    final, traj, my_rec, my_rec2 = T.sample_loop_magcache(model, latents0, context, context_null, heads, layers, steps, shift, guide, thresh, K, retention_ratio, ratios)
    assert my_rec == rec and my_rec2 == rec2, (rec, my_rec, rec2, my_rec2)
    assert not all(rec) and any(rec[3:]), f"threshold gives a degenerate pattern: {rec}"
    worst = max((a - torch.from_numpy(store[f"latents_{i + 1}"])).abs().max().item() for i, a in enumerate(traj))
    assert worst < 2e-4, worst
    meta = dict(latent_shape=latent_shape, dim=dim, heads=heads, ffn=ffn, layers=layers, text_len=text_len, text_dim=text_dim, steps=steps, shift=shift,
                guide=guide, seed=seed, thresh=thresh, K=K, retention_ratio=retention_ratio, ratios=ratios)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=np.array(repr(meta)), records=np.array(rec), records_2=np.array(rec2), **store)
    print(f"{name}: ran {''.join('C' if r else '-' for r in rec)} / {''.join('C' if r else '-' for r in rec2)}; worst |restatement - reference| = {worst:.3e}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--thresh", type=float, default=1500.0)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    G._install_reference(a.ref)
    with torch.no_grad():
        if a.only in (None, "tea"):
            run("step_teacache_d256", a.ref, a.out, thresh=a.thresh)
        if a.only in (None, "mag"):
            run_mag("step_magcache_d256", a.ref, a.out)
