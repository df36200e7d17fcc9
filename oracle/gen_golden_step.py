"""Generate tests/golden/step_*.npz by running the REFERENCE's own step glue (CPU, fp32).

Same scaffolding as gen_golden.py (bare-namespace import of the reference, fp32 dtype switch).
For a tiny Wan-shaped model it builds the reference's `WanPreWeights` / `WanPreInfer`,
`WanTransformerWeights`-style blocks / `WanTransformerInfer`, `WanPostInfer` and `WanScheduler`,
injects the synthetic checkpoint, runs the reference denoise loop (step_pre -> cond and uncond
forwards -> CFG combine -> step_post, the order of default_runner.py:137-159), asserts that
`oracle/wan_step_ref.py` reproduces every intermediate, and stores the expected tensors.

Usage: python oracle/gen_golden_step.py [--ref /root/reference] [--out tests/golden]
"""

from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import gen_golden as G  # noqa: E402
from oracle import wan_block_ref as R  # noqa: E402
from oracle import wan_step_ref as T  # noqa: E402


def build_model(dim, ffn, layers, text_dim, seed, in_dim=16):
    model = {}
    for i in range(layers):
        model.update(R.make_block_weights(dim, ffn, seed=seed, block_index=i))
    model.update(R.make_head_weights(dim, 64, seed=seed))
    model.update(T.make_pre_weights(dim, in_dim, 256, text_dim, seed=seed))
    return model


def reference_objects(cfg, model, layers, scheduler_cls=None):
    import lightx2v.common.ops  # noqa: F401
    from lightx2v.models.networks.wan.infer.post_infer import WanPostInfer
    from lightx2v.models.networks.wan.infer.pre_infer import WanPreInfer
    from lightx2v.models.networks.wan.weights.pre_weights import WanPreWeights
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler

    pre_w = WanPreWeights(cfg)
    pre_w.patch_embedding.weight = model["patch_embedding.weight"]
    pre_w.patch_embedding.bias = model["patch_embedding.bias"]
    for attr, key in [("text_embedding_0", "text_embedding.0"), ("text_embedding_2", "text_embedding.2"), ("time_embedding_0", "time_embedding.0"),
                      ("time_embedding_2", "time_embedding.2"), ("time_projection_1", "time_projection.1")]:
        mm = getattr(pre_w, attr)
        mm.weight = model[key + ".weight"].t()
        mm.bias = model[key + ".bias"]
    if hasattr(pre_w, "cfg_cond_proj_1"):  # wan2.1_distill with enable_dynamic_cfg
        for attr, key in (("cfg_cond_proj_1", "guidance_embedding.linear_1"), ("cfg_cond_proj_2", "guidance_embedding.linear_2")):
            getattr(pre_w, attr).weight, getattr(pre_w, attr).bias = model[key + ".weight"].t(), model[key + ".bias"]
    if hasattr(pre_w, "proj_0"):  # i2v with a CLIP image encoder: img_emb.proj.{0,1,3,4}
        for attr, key in (("proj_0", "img_emb.proj.0"), ("proj_4", "img_emb.proj.4")):
            getattr(pre_w, attr).weight, getattr(pre_w, attr).bias = model[key + ".weight"], model[key + ".bias"]
        for attr, key in (("proj_1", "img_emb.proj.1"), ("proj_3", "img_emb.proj.3")):
            getattr(pre_w, attr).weight, getattr(pre_w, attr).bias = model[key + ".weight"].t(), model[key + ".bias"]
    blocks = []
    infer = None
    for i in range(layers):
        blk, infer = G._ref_block_objects(cfg, model, block_index=i)
        blocks.append(blk)

    class Head:  # the three attributes infer_non_blocks reads (transformer_infer.py:75-97)
        pass

    from lightx2v.utils.registry_factory import LN_WEIGHT_REGISTER, MM_WEIGHT_REGISTER, TENSOR_REGISTER

    head = Head()
    head.norm = LN_WEIGHT_REGISTER["Default"]()
    head.norm.weight = head.norm.bias = None
    head.head = MM_WEIGHT_REGISTER["Default"]("head.head.weight", "head.head.bias")
    head.head.weight = model["head.head.weight"].t()
    head.head.bias = model["head.head.bias"]
    head.head_modulation = TENSOR_REGISTER["Default"]("head.modulation")
    head.head_modulation.tensor = model["head.modulation"]
    sch = (scheduler_cls or WanScheduler)(cfg)
    pre = WanPreInfer(cfg)
    post = WanPostInfer(cfg)
    for o in (pre, post, infer):
        o.set_scheduler(sch)
    return pre_w, blocks, head, pre, infer, post, sch


def run_case(name, latent_shape, dim, heads, ffn, layers, text_len, text_dim, steps, shift, guide, seed, out_dir):
    cfg = {
        "dim": dim, "num_heads": heads, "num_layers": layers, "ffn_dim": ffn, "in_dim": 16, "out_dim": 16, "freq_dim": 256, "text_len": text_len,
        "task": "t2v", "model_cls": "wan2.1", "run_device": "cpu", "cpu_offload": False, "seq_parallel": False, "cfg_parallel": False,
        "self_attn_1_type": "torch_sdpa", "cross_attn_1_type": "torch_sdpa", "cross_attn_2_type": "torch_sdpa", "modulate_type": "torch",
        "rope_type": "torch", "infer_steps": steps, "target_video_length": 4 * (latent_shape[1] - 1) + 1, "sample_shift": shift,
        "sample_guide_scale": guide, "enable_cfg": True, "feature_caching": "NoCaching",
    }
    model = build_model(dim, ffn, layers, text_dim, seed)
    pre_w, blocks, head, pre, infer, post, sch = reference_objects(cfg, model, layers)
    g = torch.Generator().manual_seed(seed * 7919 + 17)
    context = torch.randn(text_len, text_dim, generator=g)
    context_null = torch.randn(text_len, text_dim, generator=g)
    inputs = {"text_encoder_output": {"context": context.unsqueeze(0), "context_null": context_null.unsqueeze(0)}}

    sch.prepare(seed, latent_shape)
    latents0 = sch.latents.clone()
    infer.cos_sin = sch.cos_sin

    def ref_forward(cond: bool):
        sch.infer_condition = cond
        out = pre.infer(pre_w, inputs)
        x = out.x
        for blk in blocks:
            x = infer.infer_block(blk, x, out)
        y = infer.infer_non_blocks(head, x, out.embed)
        return post.infer(y, out)[0], out

    mine = T.UniPC(steps, shift)
    assert torch.equal(mine.sigmas, sch.sigmas) and torch.equal(mine.timesteps, sch.timesteps)
    store = {"latents0": latents0.numpy(), "context": context.numpy(), "context_null": context_null.numpy(),
             "sigmas": sch.sigmas.numpy(), "timesteps": sch.timesteps.numpy()}
    lat = latents0
    worst = 0.0
    for i in range(steps):
        sch.step_pre(step_index=i)
        cond, out = ref_forward(True)
        uncond, _ = ref_forward(False)
        sch.noise_pred = uncond + sch.sample_guide_scale * (cond - uncond)
        # --- restatement, same inputs ---
        t = mine.timestep_input(i)
        assert torch.equal(t, sch.timestep_input)
        if i == 0:
            x, embed, embed0, ctx, grid = T.pre_infer(model, lat, t, context)
            sch.infer_condition = True
            pre_out = pre.infer(pre_w, inputs)  # fresh: the blocks update pre_infer_out.x in place
            d = max((x - pre_out.x).abs().max().item(), (embed - pre_out.embed).abs().max().item(), (embed0 - pre_out.embed0).abs().max().item(),
                    (ctx - pre_out.context).abs().max().item())
            assert d < 2e-5, f"pre_infer mismatch {d}"
            assert grid == pre_out.grid_sizes.tuple
            store.update(pre_x=x.numpy(), pre_embed=embed.numpy(), pre_embed0=embed0.numpy(), pre_context=ctx.numpy())
        my_cond = T.forward(model, lat, t, context, heads, layers)
        my_uncond = T.forward(model, lat, t, context_null, heads, layers)
        d = max((my_cond - cond).abs().max().item(), (my_uncond - uncond).abs().max().item())
        worst = max(worst, d)
        assert d < 5e-5, f"forward mismatch at step {i}: {d}"
        my_noise = T.cfg_combine(my_cond, my_uncond, guide)
        sch.step_post()
        lat = mine.step(i, my_noise, lat)
        d = (lat - sch.latents).abs().max().item()
        worst = max(worst, d)
        assert d < 2e-4, f"scheduler mismatch at step {i}: {d}"
        store[f"noise_pred_{i}"] = sch.noise_pred.numpy()
        store[f"latents_{i + 1}"] = sch.latents.numpy()
    # the scheduler alone, fed the reference's own noise predictions: must agree to rounding
    solo = T.UniPC(steps, shift)
    lat = latents0
    for i in range(steps):
        lat = solo.step(i, torch.from_numpy(store[f"noise_pred_{i}"]), lat)
        d = (lat - torch.from_numpy(store[f"latents_{i + 1}"])).abs().max().item()
        assert d < 2e-6 * max(1.0, lat.abs().max().item()), f"UniPC restatement differs at step {i}: {d}"
    meta = dict(latent_shape=latent_shape, dim=dim, heads=heads, ffn=ffn, layers=layers, text_len=text_len, text_dim=text_dim, steps=steps,
                shift=shift, guide=guide, seed=seed)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=np.array(repr(meta)), **store)
    print(f"{name}: {steps} steps, worst |restatement - reference| = {worst:.3e}")


def run_two_expert_case(name, latent_shape, dim, heads, ffn, layers, text_len, text_dim, steps, shift, guides, boundary, seed, out_dir):
    """Wan2.2-A14B i2v shape of the loop: in_dim = 16 + 20 (latents + mask/VAE conditioning, no CLIP
    branch), two experts sharing one scheduler.  The reference's WanPreInfer (task i2v,
    use_image_encoder false), blocks, head, WanPostInfer and WanScheduler run per expert; the
    expert switch is the restated MultiModelStruct rule (see wan_step_ref.sample_loop_two_experts)."""
    cfg = {
        "dim": dim, "num_heads": heads, "num_layers": layers, "ffn_dim": ffn, "in_dim": 36, "out_dim": 16, "freq_dim": 256, "text_len": text_len,
        "task": "i2v", "model_cls": "wan2.2_moe", "use_image_encoder": False, "run_device": "cpu", "cpu_offload": False, "seq_parallel": False,
        "cfg_parallel": False, "self_attn_1_type": "torch_sdpa", "cross_attn_1_type": "torch_sdpa", "cross_attn_2_type": "torch_sdpa",
        "modulate_type": "torch", "rope_type": "torch", "infer_steps": steps, "target_video_length": 4 * (latent_shape[1] - 1) + 1,
        "sample_shift": shift, "sample_guide_scale": list(guides), "enable_cfg": True, "feature_caching": "NoCaching",
    }
    models = [build_model(dim, ffn, layers, text_dim, seed + e, in_dim=36) for e in range(2)]
    experts = [reference_objects(cfg, m, layers) for m in models]
    sch = experts[0][6]
    for pre_w, blocks, head, pre, infer, post, _ in experts:
        for o in (pre, post, infer):
            o.set_scheduler(sch)
    g = torch.Generator().manual_seed(seed * 7919 + 29)
    context = torch.randn(text_len, text_dim, generator=g)
    context_null = torch.randn(text_len, text_dim, generator=g)
    y = torch.randn(20, *latent_shape[1:], generator=g)
    inputs = {"text_encoder_output": {"context": context.unsqueeze(0), "context_null": context_null.unsqueeze(0)},
              "image_encoder_output": {"vae_encoder_out": y}}
    sch.prepare(seed, latent_shape)
    latents0 = sch.latents.clone()
    for e in experts:
        e[4].cos_sin = sch.cos_sin
    store = {"latents0": latents0.numpy(), "context": context.numpy(), "context_null": context_null.numpy(), "y": y.numpy()}
    picks = []
    for i in range(steps):
        sch.step_pre(step_index=i)
        which = 0 if float(sch.timesteps[i]) >= boundary * 1000 else 1
        pre_w, blocks, head, pre, infer, post, _ = experts[which]
        preds = []
        for cond in (True, False):
            sch.infer_condition = cond
            out = pre.infer(pre_w, inputs)
            x = out.x
            for blk in blocks:
                x = infer.infer_block(blk, x, out)
            preds.append(post.infer(infer.infer_non_blocks(head, x, out.embed), out)[0])
        sch.noise_pred = preds[1] + guides[which] * (preds[0] - preds[1])
        sch.step_post()
        store[f"noise_pred_{i}"] = sch.noise_pred.numpy()
        store[f"latents_{i + 1}"] = sch.latents.numpy()
        picks.append(which)
    assert 0 in picks and 1 in picks, picks
    mine, traj, my_picks = T.sample_loop_two_experts(models, boundary, guides, latents0, context, context_null, heads, layers, steps, shift, y=y)
    assert my_picks == picks
    worst = max((a - torch.from_numpy(store[f"latents_{i + 1}"])).abs().max().item() for i, a in enumerate(traj))
    assert worst < 2e-4, worst
    meta = dict(latent_shape=latent_shape, dim=dim, heads=heads, ffn=ffn, layers=layers, text_len=text_len, text_dim=text_dim, steps=steps,
                shift=shift, guides=list(guides), boundary=boundary, seed=seed, picks=picks)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=np.array(repr(meta)), **store)
    print(f"{name}: {steps} steps, experts {picks}, worst |restatement - reference| = {worst:.3e}")


def run_clip_i2v_case(name, latent_shape, dim, heads, ffn, layers, text_len, text_dim, clip_dim, steps, shift, guide, seed, out_dir):
    """Wan2.1 i2v with a CLIP image encoder: in_dim = 16 + 20 (latents + mask / VAE conditioning), 257 image
    tokens through img_emb in front of the text context, k_img / v_img / norm_k_img and a second
    cross-attention in every block.  The reference's WanPreInfer (task i2v), blocks, head,
    WanPostInfer and WanScheduler run a CFG loop; wan_step_ref must reproduce every step."""
    cfg = {
        "dim": dim, "num_heads": heads, "num_layers": layers, "ffn_dim": ffn, "in_dim": 36, "out_dim": 16, "freq_dim": 256, "text_len": text_len,
        "task": "i2v", "model_cls": "wan2.1", "use_image_encoder": True, "run_device": "cpu", "cpu_offload": False, "seq_parallel": False,
        "cfg_parallel": False, "self_attn_1_type": "torch_sdpa", "cross_attn_1_type": "torch_sdpa", "cross_attn_2_type": "torch_sdpa",
        "modulate_type": "torch", "rope_type": "torch", "infer_steps": steps, "target_video_length": 4 * (latent_shape[1] - 1) + 1,
        "sample_shift": shift, "sample_guide_scale": guide, "enable_cfg": True, "feature_caching": "NoCaching",
    }
    model = build_model(dim, ffn, layers, text_dim, seed, in_dim=36)
    for i in range(layers):
        model.update(R.make_image_branch_weights(dim, seed=seed, block_index=i))
    model.update(T.make_img_emb_weights(dim, clip_dim, seed=seed))
    pre_w, blocks, head, pre, infer, post, sch = reference_objects(cfg, model, layers)
    g = torch.Generator().manual_seed(seed * 7919 + 41)
    context = torch.randn(text_len, text_dim, generator=g)
    context_null = torch.randn(text_len, text_dim, generator=g)
    y = torch.randn(20, *latent_shape[1:], generator=g)
    clip_fea = torch.randn(R.IMAGE_TOKENS, clip_dim, generator=g)
    inputs = {"text_encoder_output": {"context": context.unsqueeze(0), "context_null": context_null.unsqueeze(0)},
              "image_encoder_output": {"vae_encoder_out": y, "clip_encoder_out": clip_fea}}
    sch.prepare(seed, latent_shape)
    latents0 = sch.latents.clone()
    infer.cos_sin = sch.cos_sin
    store = {"latents0": latents0.numpy(), "context": context.numpy(), "context_null": context_null.numpy(), "y": y.numpy(), "clip_fea": clip_fea.numpy()}
    mine = T.UniPC(steps, shift)
    lat = latents0
    worst = 0.0
    for i in range(steps):
        sch.step_pre(step_index=i)
        preds = []
        for cond in (True, False):
            sch.infer_condition = cond
            out = pre.infer(pre_w, inputs)
            if i == 0 and cond:
                store["pre_context"] = out.context.numpy().copy()
                ctx_mine = T.pre_infer(model, lat, sch.timestep_input, context, y=y, clip_fea=clip_fea)[3]
                assert (ctx_mine - out.context).abs().max().item() < 2e-5
            x = out.x
            for blk in blocks:
                x = infer.infer_block(blk, x, out)
            preds.append(post.infer(infer.infer_non_blocks(head, x, out.embed), out)[0])
        sch.noise_pred = preds[1] + guide * (preds[0] - preds[1])
        t = mine.timestep_input(i)
        my = [T.forward(model, lat, t, c, heads, layers, y=y, clip_fea=clip_fea) for c in (context, context_null)]
        d = max((my[0] - preds[0]).abs().max().item(), (my[1] - preds[1]).abs().max().item())
        assert d < 5e-5, f"forward mismatch at step {i}: {d}"
        sch.step_post()
        lat = mine.step(i, T.cfg_combine(my[0], my[1], guide), lat)
        d2 = (lat - sch.latents).abs().max().item()
        assert d2 < 2e-4, f"scheduler mismatch at step {i}: {d2}"
        worst = max(worst, d, d2)
        store[f"noise_pred_{i}"] = sch.noise_pred.numpy()
        store[f"latents_{i + 1}"] = sch.latents.numpy()
    meta = dict(latent_shape=latent_shape, dim=dim, heads=heads, ffn=ffn, layers=layers, text_len=text_len, text_dim=text_dim, clip_dim=clip_dim,
                steps=steps, shift=shift, guide=guide, seed=seed)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=np.array(repr(meta)), **store)
    print(f"{name}: {steps} steps, worst |restatement - reference| = {worst:.3e}")


def run_per_token_case(name, latent_shape, dim, heads, ffn, layers, text_len, text_dim, seed, out_dir):
    """One forward with a PER-TOKEN timestep vector (Wan2.2 ti2v: 0 on the tokens of the conditioning
    frame, t elsewhere; scheduler.py:388-392): the reference's pre-infer then hands the blocks
    embed0 [S, 6, D] and the head embed [S, D], and its modulation sums broadcast per token."""
    cfg = {
        "dim": dim, "num_heads": heads, "num_layers": layers, "ffn_dim": ffn, "in_dim": 16, "out_dim": 16, "freq_dim": 256, "text_len": text_len,
        "task": "t2v", "model_cls": "wan2.1", "run_device": "cpu", "cpu_offload": False, "seq_parallel": False, "cfg_parallel": False,
        "self_attn_1_type": "torch_sdpa", "cross_attn_1_type": "torch_sdpa", "cross_attn_2_type": "torch_sdpa", "modulate_type": "torch",
        "rope_type": "torch", "infer_steps": 4, "target_video_length": 4 * (latent_shape[1] - 1) + 1, "sample_shift": 5.0,
        "sample_guide_scale": 5.0, "enable_cfg": True, "feature_caching": "NoCaching",
    }
    model = build_model(dim, ffn, layers, text_dim, seed)
    pre_w, blocks, head, pre, infer, post, sch = reference_objects(cfg, model, layers)
    g = torch.Generator().manual_seed(seed * 7919 + 53)
    context = torch.randn(text_len, text_dim, generator=g)
    inputs = {"text_encoder_output": {"context": context.unsqueeze(0), "context_null": context.unsqueeze(0)}}
    sch.prepare(seed, latent_shape)
    latents0 = sch.latents.clone()
    infer.cos_sin = sch.cos_sin
    sch.step_pre(step_index=1)
    f, h, w = latent_shape[1], latent_shape[2] // 2, latent_shape[3] // 2
    mask = torch.ones(f, h, w)
    mask[0] = 0.0  # the first latent frame is the conditioning image: its tokens see t = 0
    t_vec = (mask * sch.timestep_input.float()).flatten()
    sch.timestep_input = t_vec
    sch.infer_condition = True
    out = pre.infer(pre_w, inputs)
    assert out.embed0.shape == (f * h * w, 6, dim) and out.embed.shape == (f * h * w, dim)
    x = out.x
    for blk in blocks:
        x = infer.infer_block(blk, x, out)
    pred = post.infer(infer.infer_non_blocks(head, x, out.embed), out)[0]
    mine = T.forward(model, latents0, t_vec, context, heads, layers)
    d = (mine - pred).abs().max().item()
    assert d < 5e-5, f"per-token forward mismatch: {d}"
    meta = dict(latent_shape=latent_shape, dim=dim, heads=heads, ffn=ffn, layers=layers, text_len=text_len, text_dim=text_dim, seed=seed)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=np.array(repr(meta)), latents0=latents0.numpy(), context=context.numpy(),
                        t_vec=t_vec.numpy(), noise_pred=pred.numpy(), block_out=x.numpy())
    print(f"{name}: per-token forward, |restatement - reference| = {d:.3e}")


def _ref_forward(pre_w, blocks, head, pre, infer, post, sch, inputs, cond):
    sch.infer_condition = cond
    out = pre.infer(pre_w, inputs)
    x = out.x
    for blk in blocks:
        x = infer.infer_block(blk, x, out)
    return post.infer(infer.infer_non_blocks(head, x, out.embed), out)[0]


def run_ti2v_case(name, latent_shape, dim, heads, ffn, layers, text_len, text_dim, steps, shift, guide, seed, out_dir):
    """model_cls "wan2.2" (the dense ti2v model) with task i2v: the reference's WanScheduler keeps the VAE
    latents of the conditioning image on the first latent frame (prepare_latents / step_post) and hands
    the model a per-token timestep vector (step_pre); the runner clears `vae_encoder_out` from the
    inputs after `prepare` (default_runner.py:282-285), so the pre-infer concatenates nothing."""
    cfg = {
        "dim": dim, "num_heads": heads, "num_layers": layers, "ffn_dim": ffn, "in_dim": 16, "out_dim": 16, "freq_dim": 256, "text_len": text_len,
        "task": "i2v", "model_cls": "wan2.2", "use_image_encoder": False, "run_device": "cpu", "cpu_offload": False, "seq_parallel": False,
        "cfg_parallel": False, "self_attn_1_type": "torch_sdpa", "cross_attn_1_type": "torch_sdpa", "cross_attn_2_type": "torch_sdpa",
        "modulate_type": "torch", "rope_type": "torch", "infer_steps": steps, "target_video_length": 4 * (latent_shape[1] - 1) + 1,
        "sample_shift": shift, "sample_guide_scale": guide, "enable_cfg": True, "feature_caching": "NoCaching",
    }
    model = build_model(dim, ffn, layers, text_dim, seed)
    objs = reference_objects(cfg, model, layers)
    pre_w, blocks, head, pre, infer, post, sch = objs
    g = torch.Generator().manual_seed(seed * 7919 + 61)
    context = torch.randn(text_len, text_dim, generator=g)
    context_null = torch.randn(text_len, text_dim, generator=g)
    z = torch.randn(16, 1, *latent_shape[2:], generator=g)   # VAE latents of the one conditioning image
    inputs = {"text_encoder_output": {"context": context.unsqueeze(0), "context_null": context_null.unsqueeze(0)},
              "image_encoder_output": {"vae_encoder_out": z}}
    sch.prepare(seed, latent_shape, image_encoder_output=inputs["image_encoder_output"])
    inputs["image_encoder_output"]["vae_encoder_out"] = None
    noise0 = torch.randn(*latent_shape, generator=torch.Generator().manual_seed(seed))   # what prepare_latents drew before the blend
    latents0 = sch.latents.clone()
    assert torch.equal(latents0[:, 1:], noise0[:, 1:]) and torch.equal(latents0[:, :1], z)
    infer.cos_sin = sch.cos_sin
    store = {"noise0": noise0.numpy(), "latents0": latents0.numpy(), "context": context.numpy(), "context_null": context_null.numpy(), "z": z.numpy()}
    for i in range(steps):
        sch.step_pre(step_index=i)
        store[f"timestep_input_{i}"] = sch.timestep_input.numpy().copy()
        cond = _ref_forward(*objs, inputs, True)
        uncond = _ref_forward(*objs, inputs, False)
        sch.noise_pred = uncond + sch.sample_guide_scale * (cond - uncond)
        sch.step_post()
        store[f"noise_pred_{i}"] = sch.noise_pred.numpy()
        store[f"latents_{i + 1}"] = sch.latents.numpy()
    my0, my_preds, my_traj = T.sample_loop_ti2v(model, noise0, z, context, context_null, heads, layers, steps, shift, guide)
    assert torch.equal(my0, latents0)
    worst = 0.0
    for i in range(steps):
        assert torch.equal(T.ti2v_timestep(T.ti2v_mask(noise0), T.UniPC(steps, shift).timestep_input(i)), torch.from_numpy(store[f"timestep_input_{i}"]))
        worst = max(worst, (my_preds[i] - torch.from_numpy(store[f"noise_pred_{i}"])).abs().max().item(),
                    (my_traj[i] - torch.from_numpy(store[f"latents_{i + 1}"])).abs().max().item())
    assert worst < 2e-4, worst
    meta = dict(latent_shape=latent_shape, dim=dim, heads=heads, ffn=ffn, layers=layers, text_len=text_len, text_dim=text_dim, steps=steps,
                shift=shift, guide=guide, seed=seed)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=np.array(repr(meta)), **store)
    print(f"{name}: {steps} steps, worst |restatement - reference| = {worst:.3e}")


def run_distill_case(name, latent_shape, dim, heads, ffn, layers, text_len, text_dim, step_list, shift, seed, out_dir, two_experts=False, boundary_step_index=2,
                     cfg_scale=None):
    """The step-distilled loops: the reference's WanStepDistillScheduler (Wan2.1 distill, t2v, one model) or
    Wan22StepDistillScheduler (Wan2.2 two-expert distill, i2v with 20 conditioning channels), no CFG, one
    forward per step; for two experts the switch is MultiDistillModelStruct's rule (restated: its module's
    import chain needs absent packages)."""
    from lightx2v.models.schedulers.wan.step_distill.scheduler import Wan22StepDistillScheduler, WanStepDistillScheduler

    in_dim = 36 if two_experts else 16
    cfg = {
        "dim": dim, "num_heads": heads, "num_layers": layers, "ffn_dim": ffn, "in_dim": in_dim, "out_dim": 16, "freq_dim": 256, "text_len": text_len,
        "task": "i2v" if two_experts else "t2v", "model_cls": "wan2.2_moe_distill" if two_experts else "wan2.1_distill", "use_image_encoder": False,
        "run_device": "cpu", "cpu_offload": False, "seq_parallel": False, "cfg_parallel": False,
        "self_attn_1_type": "torch_sdpa", "cross_attn_1_type": "torch_sdpa", "cross_attn_2_type": "torch_sdpa", "modulate_type": "torch", "rope_type": "torch",
        "infer_steps": len(step_list), "denoising_step_list": list(step_list), "boundary_step_index": boundary_step_index,
        "target_video_length": 4 * (latent_shape[1] - 1) + 1, "sample_shift": shift, "sample_guide_scale": 5.0, "enable_cfg": False, "feature_caching": "NoCaching",
    }
    n_models = 2 if two_experts else 1
    models = [build_model(dim, ffn, layers, text_dim, seed + e, in_dim=in_dim) for e in range(n_models)]
    if cfg_scale is not None:  # the distilled guidance embedding (configs/distill/wan_t2v_distill_4step_cfg_dynamic.json)
        cfg.update({"enable_dynamic_cfg": True, "cfg_scale": cfg_scale})
        for e, m in enumerate(models):
            m.update(T.make_guidance_weights(dim, 256, seed=seed + e))
    experts = [reference_objects(cfg, m, layers, scheduler_cls=Wan22StepDistillScheduler if two_experts else WanStepDistillScheduler) for m in models]
    sch = experts[0][6]
    for e in experts:
        for o in (e[3], e[5], e[4]):
            o.set_scheduler(sch)
    g = torch.Generator().manual_seed(seed * 7919 + 71)
    context = torch.randn(text_len, text_dim, generator=g)
    inputs = {"text_encoder_output": {"context": context.unsqueeze(0), "context_null": context.unsqueeze(0)}}
    y = None
    if two_experts:
        y = torch.randn(20, *latent_shape[1:], generator=g)
        inputs["image_encoder_output"] = {"vae_encoder_out": y}
    sch.prepare(seed, latent_shape)
    latents0 = sch.latents.clone()
    for e in experts:
        e[4].cos_sin = sch.cos_sin
    store = {"latents0": latents0.numpy(), "context": context.numpy(), "timesteps": sch.timesteps.numpy(), "sigmas": sch.sigmas.numpy()}
    if y is not None:
        store["y"] = y.numpy()
    picks = []
    for i in range(len(step_list)):
        sch.step_pre(step_index=i)
        which = 0 if (not two_experts or sch.step_index < boundary_step_index) else 1
        pre_w, blocks, head, pre, infer, post, _ = experts[which]
        sch.noise_pred = _ref_forward(pre_w, blocks, head, pre, infer, post, sch, inputs, True)
        sch.step_post()
        store[f"noise_pred_{i}"] = sch.noise_pred.numpy()
        store[f"latents_{i + 1}"] = sch.latents.numpy()
        picks.append(which)
    mine = T.StepDistill(step_list, shift)
    assert torch.equal(mine.timesteps, sch.timesteps) and torch.equal(mine.sigmas, sch.sigmas)
    my_preds, my_traj, my_picks = T.sample_loop_distill(models, boundary_step_index, latents0, context, heads, layers, step_list, shift, y=y, cfg_scale=cfg_scale)
    assert my_picks == picks
    worst = max(max((a - torch.from_numpy(store[f"noise_pred_{i}"])).abs().max().item() for i, a in enumerate(my_preds)),
                max((a - torch.from_numpy(store[f"latents_{i + 1}"])).abs().max().item() for i, a in enumerate(my_traj)))
    assert worst < 2e-4, worst
    # the scheduler alone on the reference's own predictions: bit for bit
    lat = latents0
    for i in range(len(step_list)):
        lat = mine.step(i, torch.from_numpy(store[f"noise_pred_{i}"]), lat)
        assert torch.equal(lat, torch.from_numpy(store[f"latents_{i + 1}"])), i
    meta = dict(latent_shape=latent_shape, dim=dim, heads=heads, ffn=ffn, layers=layers, text_len=text_len, text_dim=text_dim, step_list=list(step_list),
                shift=shift, seed=seed, two_experts=two_experts, boundary_step_index=boundary_step_index, picks=picks, cfg_scale=cfg_scale)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=np.array(repr(meta)), **store)
    print(f"{name}: {len(step_list)} distilled steps, experts {picks}, worst |restatement - reference| = {worst:.3e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--only", default=None, help="run only the case of this name")
    a = ap.parse_args()
    G._install_reference(a.ref)
    torch.manual_seed(0)
    if a.only == "step_per_token_d256":
        with torch.no_grad():
            run_per_token_case("step_per_token_d256", (16, 3, 8, 12), 256, 2, 512, 2, 16, 128, 13, a.out)
        return
    extra = {
        "step_ti2v_d256": lambda: run_ti2v_case("step_ti2v_d256", (16, 3, 8, 12), 256, 2, 512, 2, 16, 128, 4, 5.0, 5.0, 21, a.out),
        "step_distill_d256": lambda: run_distill_case("step_distill_d256", (16, 3, 8, 12), 256, 2, 512, 2, 16, 128, [1000, 750, 500, 250], 5.0, 23, a.out),
        "step_distill_dyncfg_d256": lambda: run_distill_case("step_distill_dyncfg_d256", (16, 3, 8, 12), 256, 2, 512, 2, 16, 128, [1000, 750, 500, 250], 5.0, 31, a.out,
                                                             cfg_scale=3.4),
        "step_distill_moe_i2v_d256": lambda: run_distill_case("step_distill_moe_i2v_d256", (16, 3, 8, 12), 256, 2, 512, 1, 16, 128, [1000, 750, 500, 250], 5.0, 27, a.out,
                                                              two_experts=True, boundary_step_index=2),
    }
    if a.only in extra:
        with torch.no_grad():
            extra[a.only]()
        return
    if a.only == "step_clip_i2v_d256":
        with torch.no_grad():
            run_clip_i2v_case("step_clip_i2v_d256", (16, 3, 8, 12), 256, 2, 512, 2, 16, 128, 128, 3, 5.0, 5.0, 9, a.out)
        return
    with torch.no_grad():
        run_clip_i2v_case("step_clip_i2v_d256", (16, 3, 8, 12), 256, 2, 512, 2, 16, 128, 128, 3, 5.0, 5.0, 9, a.out)
        run_per_token_case("step_per_token_d256", (16, 3, 8, 12), 256, 2, 512, 2, 16, 128, 13, a.out)
        run_case("step_tiny_d256", (16, 3, 8, 12), 256, 2, 512, 2, 16, 128, 6, 5.0, 5.0, 3, a.out)
        run_case("step_small_d512", (16, 2, 12, 16), 512, 4, 1024, 2, 32, 256, 4, 3.0, 4.0, 5, a.out)
        run_two_expert_case("step_moe_i2v_d256", (16, 3, 8, 12), 256, 2, 512, 1, 16, 128, 6, 5.0, (3.5, 3.0), 0.6, 7, a.out)
        for fn in extra.values():
            fn()


if __name__ == "__main__":
    main()
