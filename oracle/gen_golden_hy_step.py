"""Generate tests/golden/hy_step_*.npz by running the REFERENCE's HunyuanVideo-1.5 step glue (CPU, fp32):
`HunyuanVideo15PreWeights` + `HunyuanVideo15PreInfer.infer`, `HunyuanVideo15TransformerInfer.infer`,
`HunyuanVideo15PostInfer.infer`, `HunyuanVideo15Scheduler` and `HunyuanVideo15StepDistillScheduler`,
looped the way `HunyuanVideo15Model.infer` and the runner loop them.

Run only where the reference checkout exists; the GPU box never runs this.  Import harness as in
gen_golden.py, plus:
  * the pre-infer module casts to `torch.bfloat16` by name; its module-level `torch` is replaced by a
    proxy whose `bfloat16` is fp32, so the reference's pre-infer runs in fp32 like the rest;
  * `infer/attn_no_pad.py` imports names from flash-attn at import time; an empty stand-in module
    supplies the names (all None), and the pre-infer's `attention` -- the one call that would reach
    them -- is replaced by `hy_step_ref.refiner_attention` (drop padded tokens, attend, pad with
    zeros: the documented behaviour of unpad_input -> flash_attn_varlen -> pad_input).  That one
    operation is therefore NOT pinned by the reference; everything around it is.
For every case the script asserts that hy_step_ref agrees with the reference at the pre-infer
outputs, at every step's prediction and at every step's latents, then stores inputs + outputs.

Usage: python oracle/gen_golden_hy_step.py [--ref /root/reference] [--out tests/golden]
"""

from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import hy_block_ref as HY  # noqa: E402
from oracle import hy_step_ref as S  # noqa: E402
from oracle.gen_golden import _install_reference  # noqa: E402
from oracle.gen_golden_hy import _ref_objects  # noqa: E402


class _TorchFp32:
    """`torch` as the reference's pre-infer module sees it: bf16 spelled fp32."""

    bfloat16 = torch.float32

    def __getattr__(self, name):
        return getattr(torch, name)


def _ref_pre_objects(cfg, w):
    import lightx2v.common.ops  # noqa: F401

    for name, attrs in (("flash_attn", ("flash_attn_varlen_qkvpacked_func",)), ("flash_attn.bert_padding", ("pad_input", "unpad_input"))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m
    import lightx2v.models.networks.hunyuan_video.infer.pre_infer as PI
    from lightx2v.models.networks.hunyuan_video.infer.post_infer import HunyuanVideo15PostInfer
    from lightx2v.models.networks.hunyuan_video.weights.pre_weights import HunyuanVideo15PreWeights

    PI.torch = _TorchFp32()

    def attention(q, k, v, drop_rate=0.0, attn_mask=None, causal=False, attn_type="flash_attn2"):
        assert q.shape[0] == 1 and not causal and drop_rate == 0.0
        return S.refiner_attention(q[0], k[0], v[0], attn_mask[0].bool()).unsqueeze(0)

    PI.attention = attention
    W = HunyuanVideo15PreWeights(cfg)

    def set_mm(mm):
        mm.weight = w[mm.weight_name].t()
        mm.bias = w[mm.bias_name]

    W.img_in.weight, W.img_in.bias = w["img_in.proj.weight"], w["img_in.proj.bias"]
    for n in ("time_in_0", "time_in_2", "txt_in_t_embedder_0", "txt_in_t_embedder_2", "txt_in_c_embedder_0", "txt_in_c_embedder_2", "txt_in_input_embedder"):
        set_mm(getattr(W, n))
    for blk in W.individual_token_refiner:
        for n in ("self_attn_qkv", "self_attn_proj", "mlp_fc1", "mlp_fc2", "adaLN_modulation"):
            set_mm(getattr(blk, n))
        for n in ("norm1", "norm2"):
            ln = getattr(blk, n)
            ln.weight, ln.bias = w[ln.weight_name], w[ln.bias_name]
    W.cond_type_embedding.weight = w["cond_type_embedding.weight"]
    return W, PI.HunyuanVideo15PreInfer(cfg), HunyuanVideo15PostInfer(cfg)


def _ref_scheduler(cfg, distill):
    import lightx2v.models.schedulers.hunyuan_video.posemb_layers as P
    from lightx2v.models.schedulers.hunyuan_video import scheduler as SCH
    from lightx2v.models.schedulers.hunyuan_video.step_distill.scheduler import HunyuanVideo15StepDistillScheduler

    sch = (HunyuanVideo15StepDistillScheduler if distill else SCH.HunyuanVideo15Scheduler)(cfg)
    orig = P.get_1d_rotary_pos_embed
    real_prepare = sch.prepare

    def prepare(*a, **k):  # the table builder defaults to a CUDA device (posemb_layers.py:235-283)
        SCH.get_nd_rotary_pos_embed.__globals__["get_1d_rotary_pos_embed"] = lambda *aa, **kk: orig(*aa, **{**kk, "run_device": "cpu"})
        try:
            return real_prepare(*a, **k)
        finally:
            SCH.get_nd_rotary_pos_embed.__globals__["get_1d_rotary_pos_embed"] = orig

    sch.prepare = prepare
    return sch


def run_case(name, latent_shape, dim, heads, mlp, depth, text_dim, lens, task, distill, steps_or_list, shift, guide, seed, out_dir):
    C, T, H, Wd = latent_shape
    cfg = {
        "task": task, "mm_double_blocks_depth": depth, "heads_num": heads, "hidden_size": dim, "run_device": "cpu", "seq_parallel": False,
        "cpu_offload": False, "modulate_type": "torch", "rope_type": "torch", "attn_type": "torch_sdpa", "ln_type": "Default",
        "rms_type": "sgl-kernel", "rope_dim_list": list(HY.ROPE_DIM_LIST), "rope_theta": HY.ROPE_THETA, "sample_shift": shift,
        "sample_guide_scale": -1.0 if guide is None else guide, "enable_cfg": guide is not None, "cfg_parallel": False,
        "infer_steps": len(steps_or_list) if distill else steps_or_list, "patch_size": (1, 1, 1), "out_channels": C, "is_sr_running": False,
    }
    if distill:
        cfg["denoising_step_list"] = list(steps_or_list)
    pre_w = S.make_pre_weights(dim, heads, C, text_dim, seed=seed)
    blocks = [HY.make_double_block_weights(dim, heads, mlp, seed=seed + i, block_index=i) for i in range(depth)]
    fw = HY.make_final_weights(dim, C, seed=seed)
    inputs = S.make_step_inputs(dim, text_dim, *lens, seed=seed)
    if task == "i2v":
        g = torch.Generator().manual_seed(seed + 31)
        inputs["image_encoder_output"]["cond_latents"] = torch.randn(1, C, 1, H, Wd, generator=g)

    PW, pre, post = _ref_pre_objects(cfg, pre_w)
    TW, infer = _ref_objects(cfg, blocks, fw)
    sch = _ref_scheduler(cfg, distill)
    for o in (pre, post, infer):
        o.set_scheduler(sch)
    sch.prepare(seed, latent_shape, image_encoder_output=inputs["image_encoder_output"])
    latents0 = sch.latents.clone()
    assert latents0.dtype == torch.bfloat16

    mine = S.StepDistillScheduler(steps_or_list, shift, task) if distill else S.EulerScheduler(steps_or_list, shift, task)
    mine.prepare(latents0.clone(), inputs["image_encoder_output"]["cond_latents"])
    assert torch.equal(mine.sigmas, sch.sigmas) and torch.equal(mine.timesteps, sch.timesteps)
    assert torch.equal(mine.cond_latents_concat, sch.cond_latents_concat.float()) and torch.equal(mine.mask_concat, sch.mask_concat)

    def ref_forward(cond):
        sch.infer_condition = cond
        out = pre.infer(PW, inputs)
        x = infer.infer(TW, out)
        return post.infer(x, out)[0], out

    store = {"latents0": latents0.float().numpy(), "sigmas": sch.sigmas.numpy(), "timesteps": sch.timesteps.numpy()}
    worst = 0.0
    for i in range(sch.infer_steps):
        sch.step_pre(step_index=i)
        mine.step_index = i
        cond, out = ref_forward(True)
        if i == 0:
            t = mine.timesteps[0]
            for flag, tag in ((True, "cond"), (False, "uncond")):
                sch.infer_condition = flag
                o = pre.infer(PW, inputs)
                img, txt, vec, grid = S.pre_infer(pre_w, mine.latents, mine.cond_latents_concat, mine.mask_concat, t, inputs, flag, heads)
                assert grid == tuple(o.grid_sizes) and txt.shape == o.txt.shape, (txt.shape, o.txt.shape)
                d = max((img - o.img).abs().max().item(), (txt - o.txt).abs().max().item(), (vec - o.vec).abs().max().item())
                assert d < 2e-5, f"pre_infer mismatch ({tag}): {d}"
                store.update({f"pre_img_{tag}": o.img[0].numpy(), f"pre_txt_{tag}": o.txt[0].numpy(), f"pre_vec_{tag}": o.vec.numpy()})
        my_cond = S.model_forward(pre_w, blocks, fw, mine, inputs, True, heads)
        d = (my_cond - cond).abs().max().item()
        if guide is not None:
            uncond, _ = ref_forward(False)
            my_uncond = S.model_forward(pre_w, blocks, fw, mine, inputs, False, heads)
            d = max(d, (my_uncond - uncond).abs().max().item())
            sch.noise_pred = uncond + sch.sample_guide_scale * (cond - uncond)
            mine.noise_pred = my_uncond + guide * (my_cond - my_uncond)
        else:
            sch.noise_pred, mine.noise_pred = cond, my_cond
        assert d < 5e-5 * max(1.0, cond.abs().max().item()), f"forward mismatch at step {i}: {d}"
        worst = max(worst, d)
        sch.step_post()
        mine.step_post()
        assert mine.latents.dtype == sch.latents.dtype
        d = (mine.latents.float() - sch.latents.float()).abs().max().item()
        tol = 2e-4 if sch.latents.dtype == torch.float32 else 2 ** -6  # bf16 latents: one rounding step apart at most
        assert d <= tol * max(1.0, sch.latents.float().abs().max().item()), f"scheduler mismatch at step {i}: {d}"
        worst = max(worst, d)
        mine.latents = sch.latents.clone()  # both continue from the reference's latents
        store[f"noise_pred_{i}"] = sch.noise_pred.float().numpy()
        store[f"latents_{i + 1}"] = sch.latents.float().numpy()
    # the scheduler alone, on the reference's own predictions, must reproduce its latents exactly
    solo = S.StepDistillScheduler(steps_or_list, shift, task) if distill else S.EulerScheduler(steps_or_list, shift, task)
    solo.prepare(latents0.clone(), inputs["image_encoder_output"]["cond_latents"])
    for i in range(solo.infer_steps):
        solo.step_index = i
        solo.noise_pred = torch.from_numpy(store[f"noise_pred_{i}"])
        solo.step_post()
        assert torch.equal(solo.latents.float(), torch.from_numpy(store[f"latents_{i + 1}"])), f"scheduler restatement differs at step {i}"
    meta = dict(latent_shape=list(latent_shape), dim=dim, heads=heads, mlp=mlp, depth=depth, text_dim=text_dim, lens=list(lens), task=task,
                distill=bool(distill), steps=list(steps_or_list) if distill else steps_or_list, shift=shift, guide=guide, seed=seed,
                latents_dtype=str(sch.latents.dtype))
    if task == "i2v":
        store["cond_latents"] = inputs["image_encoder_output"]["cond_latents"].numpy()
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=np.array(repr(meta)), **store)
    print(f"{name}: {sch.infer_steps} steps, worst |restatement - reference| = {worst:.3e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    a = ap.parse_args()
    _install_reference(a.ref)
    torch.manual_seed(0)
    with torch.no_grad():
        # lens = (text tokens, valid, byT5 tokens, valid, image tokens, valid)
        run_case("hy_step_distill_d256", (8, 2, 4, 6), 256, 2, 1024, 2, 128, (24, 17, 16, 5, 8, 0), "t2v", True, (1000, 750, 500, 250), 5.0, None, 21, a.out)
        run_case("hy_step_cfg_i2v_d256", (8, 3, 4, 4), 256, 2, 1024, 1, 128, (20, 9, 12, 0, 8, 8), "i2v", False, 3, 7.0, 6.0, 22, a.out)


if __name__ == "__main__":
    main()
