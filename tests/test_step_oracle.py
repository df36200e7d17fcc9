"""CPU: the step-glue restatement (oracle/wan_step_ref.py) against the fixtures written from the
reference's own WanPreInfer / WanPostInfer / WanScheduler / denoise loop, and the product's
host-side scheduler arithmetic against the restatement."""
import ast

import numpy as np
import pytest
import torch

from oracle import wan_block_ref as R
from oracle import wan_step_ref as T
from tests.util import load_golden

STEP_CASES = ["step_tiny_d256", "step_small_d512"]
MOE_CASE = "step_moe_i2v_d256"


def step_case(name):
    rec = load_golden(name)
    meta = ast.literal_eval(str(rec["meta"]))
    model = {}
    for i in range(meta["layers"]):
        model.update(R.make_block_weights(meta["dim"], meta["ffn"], seed=meta["seed"], block_index=i))
    model.update(R.make_head_weights(meta["dim"], 64, seed=meta["seed"]))
    model.update(T.make_pre_weights(meta["dim"], 16, 256, meta["text_dim"], seed=meta["seed"]))
    return rec, meta, model


def moe_case(name=MOE_CASE):
    rec = load_golden(name)
    meta = ast.literal_eval(str(rec["meta"]))
    models = []
    for e in range(2):
        m = {}
        for i in range(meta["layers"]):
            m.update(R.make_block_weights(meta["dim"], meta["ffn"], seed=meta["seed"] + e, block_index=i))
        m.update(R.make_head_weights(meta["dim"], 64, seed=meta["seed"] + e))
        m.update(T.make_pre_weights(meta["dim"], 36, 256, meta["text_dim"], seed=meta["seed"] + e))
        models.append(m)
    return rec, meta, models


CLIP_CASE = "step_clip_i2v_d256"


def clip_case(name=CLIP_CASE):
    rec = load_golden(name)
    meta = ast.literal_eval(str(rec["meta"]))
    model = {}
    for i in range(meta["layers"]):
        model.update(R.make_block_weights(meta["dim"], meta["ffn"], seed=meta["seed"], block_index=i))
        model.update(R.make_image_branch_weights(meta["dim"], seed=meta["seed"], block_index=i))
    model.update(R.make_head_weights(meta["dim"], 64, seed=meta["seed"]))
    model.update(T.make_pre_weights(meta["dim"], 36, 256, meta["text_dim"], seed=meta["seed"]))
    model.update(T.make_img_emb_weights(meta["dim"], meta["clip_dim"], seed=meta["seed"]))
    return rec, meta, model


def test_clip_i2v_loop_matches_reference_trajectory():
    """i2v with a CLIP image encoder: 257 image tokens through img_emb in front of the text, k_img /
    v_img / norm_k_img and the second cross-attention in every block, 36 input channels."""
    rec, meta, model = clip_case()
    y, clip_fea = torch.from_numpy(rec["y"]), torch.from_numpy(rec["clip_fea"])
    lat = torch.from_numpy(rec["latents0"])
    sch = T.UniPC(meta["steps"], meta["shift"])
    ctx = T.pre_infer(model, lat, sch.timestep_input(0), torch.from_numpy(rec["context"]), y=y, clip_fea=clip_fea)[3]
    assert ctx.shape[0] == R.IMAGE_TOKENS + meta["text_len"]
    assert (ctx - torch.from_numpy(rec["pre_context"])).abs().max().item() <= 2e-5
    for i in range(meta["steps"]):
        t = sch.timestep_input(i)
        c, u = (T.forward(model, lat, t, torch.from_numpy(rec[k]), meta["heads"], meta["layers"], y=y, clip_fea=clip_fea) for k in ("context", "context_null"))
        noise = T.cfg_combine(c, u, meta["guide"])
        assert (noise - torch.from_numpy(rec[f"noise_pred_{i}"])).abs().max().item() <= 2e-4, i
        lat = sch.step(i, noise, lat)
        assert (lat - torch.from_numpy(rec[f"latents_{i + 1}"])).abs().max().item() <= 5e-4, i


def per_token_case(name="step_per_token_d256"):
    rec = load_golden(name)
    meta = ast.literal_eval(str(rec["meta"]))
    model = {}
    for i in range(meta["layers"]):
        model.update(R.make_block_weights(meta["dim"], meta["ffn"], seed=meta["seed"], block_index=i))
    model.update(R.make_head_weights(meta["dim"], 64, seed=meta["seed"]))
    model.update(T.make_pre_weights(meta["dim"], 16, 256, meta["text_dim"], seed=meta["seed"]))
    return rec, meta, model


def test_per_token_timestep_forward_matches_reference():
    """Timestep as a per-token vector (0 on the conditioning frame's tokens): every modulation term
    becomes a per-token row through the reference's broadcasting sums."""
    rec, meta, model = per_token_case()
    t_vec = torch.from_numpy(rec["t_vec"])
    assert t_vec.unique().numel() == 2 and float(t_vec[0]) == 0.0
    pred = T.forward(model, torch.from_numpy(rec["latents0"]), t_vec, torch.from_numpy(rec["context"]), meta["heads"], meta["layers"])
    assert (pred - torch.from_numpy(rec["noise_pred"])).abs().max().item() <= 5e-5
    # and it differs from the scalar-timestep forward (the conditioning tokens do see another modulation)
    scalar = T.forward(model, torch.from_numpy(rec["latents0"]), t_vec.max().reshape(1), torch.from_numpy(rec["context"]), meta["heads"], meta["layers"])
    assert (pred - scalar).abs().max().item() > 1e-3


def test_two_expert_loop_matches_reference_trajectory():
    rec, meta, models = moe_case()
    lat, traj, picks = T.sample_loop_two_experts(models, meta["boundary"], meta["guides"], torch.from_numpy(rec["latents0"]),
                                                 torch.from_numpy(rec["context"]), torch.from_numpy(rec["context_null"]), meta["heads"],
                                                 meta["layers"], meta["steps"], meta["shift"], y=torch.from_numpy(rec["y"]))
    assert picks == meta["picks"] and 0 in picks and 1 in picks
    for i, got in enumerate(traj):
        assert (got - torch.from_numpy(rec[f"latents_{i + 1}"])).abs().max().item() <= 5e-4


def test_product_expert_switch_rule():
    import lightx2v_amd as L

    class FakeModel:
        def __init__(self):
            self.calls = 0

        def set_scheduler(self, s):
            self.s = s

        def infer(self, inputs):
            self.calls += 1

    cfg = {"infer_steps": 6, "sample_shift": 5.0, "sample_guide_scale": [3.5, 3.0], "dim": 256, "num_heads": 2, "run_device": "cpu"}
    sch = L.WanScheduler(cfg)
    sch.prepare(0, (16, 2, 4, 4), latents=torch.zeros(16, 2, 4, 4))
    hi, lo = FakeModel(), FakeModel()
    mm = L.MultiModelStruct([hi, lo], cfg, boundary=0.6)
    mm.set_scheduler(sch)
    picks, guides = [], []
    for i in range(6):
        sch.step_index = i
        mm.infer(None)
        picks.append(mm.cur_model_index)
        guides.append(sch.sample_guide_scale)
    ref = T.UniPC(6, 5.0)
    assert picks == [0 if float(ref.timesteps[i]) >= 600 else 1 for i in range(6)]
    assert guides == [[3.5, 3.0][p] for p in picks] and hi.calls + lo.calls == 6


@pytest.mark.parametrize("name", STEP_CASES)
def test_scheduler_restatement_matches_reference_trajectory(name):
    rec, meta, _ = step_case(name)
    sch = T.UniPC(meta["steps"], meta["shift"])
    assert np.array_equal(sch.sigmas.numpy(), rec["sigmas"]) and np.array_equal(sch.timesteps.numpy(), rec["timesteps"])
    lat = torch.from_numpy(rec["latents0"])
    for i in range(meta["steps"]):
        lat = sch.step(i, torch.from_numpy(rec[f"noise_pred_{i}"]), lat)
        ref = torch.from_numpy(rec[f"latents_{i + 1}"])
        assert (lat - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


def test_pre_infer_and_first_forward_match_reference():
    rec, meta, model = step_case("step_tiny_d256")
    lat, ctx = torch.from_numpy(rec["latents0"]), torch.from_numpy(rec["context"])
    t = torch.from_numpy(rec["timesteps"][:1])
    x, embed, embed0, c, grid = T.pre_infer(model, lat, t, ctx)
    for got, key in [(x, "pre_x"), (embed, "pre_embed"), (embed0, "pre_embed0"), (c, "pre_context")]:
        assert (got - torch.from_numpy(rec[key])).abs().max().item() <= 1e-4
    cond = T.forward(model, lat, t, ctx, meta["heads"], meta["layers"])
    uncond = T.forward(model, lat, t, torch.from_numpy(rec["context_null"]), meta["heads"], meta["layers"])
    noise = T.cfg_combine(cond, uncond, meta["guide"])
    assert (noise - torch.from_numpy(rec["noise_pred_0"])).abs().max().item() <= 2e-4


def test_unpatchify_inverts_patchify_up_to_column_order():
    lat = torch.randn(16, 2, 6, 8, generator=torch.Generator().manual_seed(1))
    rows = T.patchify(lat)  # columns (c, ph, pw)
    as_head = rows.view(-1, 16, 2, 2).permute(0, 2, 3, 1).reshape(rows.shape[0], 64)  # columns (ph, pw, c)
    assert torch.equal(T.unpatchify(as_head, (2, 3, 4)), lat)


@pytest.mark.parametrize("steps,shift", [(6, 5.0), (4, 3.0), (50, 5.0)])
def test_product_scheduler_scalars_equal_restatement(steps, shift):
    import lightx2v_amd as L

    cfg = {"infer_steps": steps, "sample_shift": shift, "sample_guide_scale": 5.0, "dim": 256, "num_heads": 2, "run_device": "cpu"}
    sch = L.WanScheduler(cfg)
    sch.prepare(0, (16, 2, 4, 4), latents=torch.zeros(16, 2, 4, 4))
    ref = T.UniPC(steps, shift)
    assert torch.equal(sch.sigmas, ref.sigmas) and torch.equal(sch.timesteps, ref.timesteps)
    for i in range(steps):
        sch.step_index = i
        for order in (1, 2):
            for corrector in (False, True):
                if (corrector and i < order) or (not corrector and i < order - 1) or (not corrector and i + 1 >= len(ref.sigmas)):
                    continue
                if not corrector and i == steps - 1 and order == 2:
                    continue
                a, b, c, rk, rho = sch._coefficients(order, corrector)
                ra, rb, rc, rrk, rrho = ref.coefficients(i, order, corrector)
                assert (a, b, c) == (float(ra), float(rb), float(rc))
                assert rk is None and rrk is None or rk == float(rrk)
                assert rho is None and rrho is None or torch.equal(rho, rrho)


def test_teacache_loop_matches_reference_run():
    """The TeaCache restatement against the reference's `WanTransformerInferTeaCaching` run: same
    skip pattern on both CFG branches and the same latents after every step."""
    rec, meta, model = step_case("step_teacache_d256")
    final, traj, r1, r2 = T.sample_loop_teacache(model, torch.from_numpy(rec["latents0"]), torch.from_numpy(rec["context"]), torch.from_numpy(rec["context_null"]),
                                                 meta["heads"], meta["layers"], meta["steps"], meta["shift"], meta["guide"], meta["thresh"], meta["coefficients"],
                                                 meta["use_ret_steps"])
    assert r1 == rec["records"].tolist() and r2 == rec["records_2"].tolist()
    assert not all(r1) and sum(r1) > 2
    for i, lat in enumerate(traj):
        assert (lat - torch.from_numpy(rec[f"latents_{i + 1}"])).abs().max().item() < 2e-4


def test_magcache_loop_matches_reference_run():
    """The MagCache restatement against the reference's `WanTransformerInferMagCaching` run (ratios:
    the first 12 entries of the lists the reference ships): same forwards skipped on both CFG
    branches, same latents after every step."""
    rec, meta, model = step_case("step_magcache_d256")
    final, traj, r1, r2 = T.sample_loop_magcache(model, torch.from_numpy(rec["latents0"]), torch.from_numpy(rec["context"]), torch.from_numpy(rec["context_null"]),
                                                 meta["heads"], meta["layers"], meta["steps"], meta["shift"], meta["guide"], meta["thresh"], meta["K"],
                                                 meta["retention_ratio"], meta["ratios"])
    assert r1 == rec["records"].tolist() and r2 == rec["records_2"].tolist()
    assert not all(r1) and sum(r1) > 2 and all(r1[: int(meta["steps"] * meta["retention_ratio"])])
    for i, lat in enumerate(traj):
        assert (lat - torch.from_numpy(rec[f"latents_{i + 1}"])).abs().max().item() < 2e-4


def test_magcache_decision_is_pure_host_logic():
    """The product's skip decision (no GPU involved) against the restatement's, over the shipped-ratio
    prefix and a few thresholds, both branches interleaved as the CFG loop calls them."""
    from lightx2v_amd.wan_caching import WanTransformerInferMagCaching

    rec, meta, _ = step_case("step_magcache_d256")
    for thresh, K in ((0.12, 2), (0.04, 4), (1.0, 1), (0.0, 3)):
        mine = WanTransformerInferMagCaching.__new__(WanTransformerInferMagCaching)
        mine.config = {"infer_steps": meta["steps"]}
        mine.magcache_thresh, mine.K, mine.retention_ratio, mine.calibration = thresh, K, meta["retention_ratio"], False
        mine.mag_ratios = np.array(meta["ratios"])
        mine.clear()
        ref = T.MagCache(meta["steps"], thresh, K, meta["retention_ratio"], meta["ratios"])
        for i in range(meta["steps"]):
            for cond in (True, False):
                assert mine.should_skip(i, cond) == (not ref.should_calc(i, cond)), (thresh, K, i, cond)


# ---- Wan2.2 ti2v (image on the first latent frame) and the step-distilled schedules ----
def plain_case(name, in_dim=16, experts=1):
    rec = load_golden(name)
    meta = ast.literal_eval(str(rec["meta"]))
    models = []
    for e in range(experts):
        m = {}
        for i in range(meta["layers"]):
            m.update(R.make_block_weights(meta["dim"], meta["ffn"], seed=meta["seed"] + e, block_index=i))
        m.update(R.make_head_weights(meta["dim"], 64, seed=meta["seed"] + e))
        m.update(T.make_pre_weights(meta["dim"], in_dim, 256, meta["text_dim"], seed=meta["seed"] + e))
        if meta.get("cfg_scale") is not None:
            m.update(T.make_guidance_weights(meta["dim"], 256, seed=meta["seed"] + e))
        models.append(m)
    return rec, meta, models


def test_guidance_scale_embedding_equals_restatement():
    from lightx2v_amd.wan_model import guidance_scale_embedding

    for g in (0.0, 1.0, 3.4, 4.0, 5.5, 9.0):
        assert torch.equal(guidance_scale_embedding(g), T.guidance_scale_embedding(torch.tensor([g])))


def test_ti2v_loop_matches_reference_trajectory():
    """model_cls "wan2.2", task i2v: the reference's own scheduler keeps the image's VAE latents on the
    first latent frame and hands the model a per-token timestep vector; the restatement reproduces the
    vectors bit for bit and the trajectory to rounding."""
    rec, meta, (model,) = plain_case("step_ti2v_d256")
    noise0, z = torch.from_numpy(rec["noise0"]), torch.from_numpy(rec["z"])
    lat0, preds, traj = T.sample_loop_ti2v(model, noise0, z, torch.from_numpy(rec["context"]), torch.from_numpy(rec["context_null"]), meta["heads"],
                                           meta["layers"], meta["steps"], meta["shift"], meta["guide"])
    assert torch.equal(lat0, torch.from_numpy(rec["latents0"]))
    mask = T.ti2v_mask(noise0)
    sch = T.UniPC(meta["steps"], meta["shift"])
    for i in range(meta["steps"]):
        assert torch.equal(T.ti2v_timestep(mask, sch.timestep_input(i)), torch.from_numpy(rec[f"timestep_input_{i}"]))
        assert (preds[i] - torch.from_numpy(rec[f"noise_pred_{i}"])).abs().max().item() <= 5e-4
        assert (traj[i] - torch.from_numpy(rec[f"latents_{i + 1}"])).abs().max().item() <= 5e-4
        assert torch.equal(traj[i][:, :1], z)   # the image frame never moves


def test_product_ti2v_scheduler_host_logic():
    """The product scheduler's image bookkeeping on host tensors: blended initial latents and the
    per-token timestep vectors equal the reference's."""
    import lightx2v_amd as L

    rec, meta, _ = plain_case("step_ti2v_d256")
    cfg = {"infer_steps": meta["steps"], "sample_shift": meta["shift"], "sample_guide_scale": meta["guide"], "dim": meta["dim"], "num_heads": meta["heads"],
           "run_device": "cpu", "model_cls": "wan2.2", "task": "i2v", "keep_latents_dtype_in_scheduler": True}
    sch = L.WanScheduler(cfg)
    sch.prepare(meta["seed"], meta["latent_shape"], image_encoder_output={"vae_encoder_out": torch.from_numpy(rec["z"])}, latents=torch.from_numpy(rec["noise0"]))
    assert torch.equal(sch.latents, torch.from_numpy(rec["latents0"]))
    for i in range(meta["steps"]):
        sch.step_pre(i)
        assert torch.equal(sch.timestep_input, torch.from_numpy(rec[f"timestep_input_{i}"])), i
    # a scheduler without an image keeps the scalar timestep
    plain = L.WanScheduler(dict(cfg, model_cls="wan2.1", task="t2v"))
    plain.prepare(0, meta["latent_shape"], latents=torch.from_numpy(rec["noise0"]))
    plain.step_pre(0)
    assert plain.timestep_input.numel() == 1


@pytest.mark.parametrize("name,in_dim,experts", [("step_distill_d256", 16, 1), ("step_distill_moe_i2v_d256", 36, 2), ("step_distill_dyncfg_d256", 16, 1)])
def test_distill_loops_match_reference_trajectory(name, in_dim, experts):
    rec, meta, models = plain_case(name, in_dim, experts)
    y = torch.from_numpy(rec["y"]) if "y" in rec else None
    preds, traj, picks = T.sample_loop_distill(models, meta["boundary_step_index"], torch.from_numpy(rec["latents0"]), torch.from_numpy(rec["context"]),
                                               meta["heads"], meta["layers"], meta["step_list"], meta["shift"], y=y, cfg_scale=meta.get("cfg_scale"))
    assert picks == meta["picks"]
    for i in range(len(meta["step_list"])):
        assert (preds[i] - torch.from_numpy(rec[f"noise_pred_{i}"])).abs().max().item() <= 5e-4
        assert (traj[i] - torch.from_numpy(rec[f"latents_{i + 1}"])).abs().max().item() <= 5e-4
    # the scheduler alone on the reference's predictions: bit for bit
    sch = T.StepDistill(meta["step_list"], meta["shift"])
    assert torch.equal(sch.timesteps, torch.from_numpy(rec["timesteps"])) and torch.equal(sch.sigmas, torch.from_numpy(rec["sigmas"]))
    lat = torch.from_numpy(rec["latents0"])
    for i in range(len(meta["step_list"])):
        lat = sch.step(i, torch.from_numpy(rec[f"noise_pred_{i}"]), lat)
        assert torch.equal(lat, torch.from_numpy(rec[f"latents_{i + 1}"])), i


def test_product_distill_scheduler_tables_and_switch():
    """The product's step-distilled scheduler holds the reference's timesteps and sigmas bit for bit;
    the two-expert struct switches on the step index."""
    import lightx2v_amd as L

    rec, meta, _ = plain_case("step_distill_moe_i2v_d256", 36, 2)
    cfg = {"denoising_step_list": meta["step_list"], "boundary_step_index": meta["boundary_step_index"], "infer_steps": 50, "sample_shift": meta["shift"],
           "sample_guide_scale": 5.0, "dim": meta["dim"], "num_heads": meta["heads"], "run_device": "cpu"}
    sch = L.Wan22StepDistillScheduler(cfg)
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    assert sch.infer_steps == 4
    assert torch.equal(sch.timesteps, torch.from_numpy(rec["timesteps"])) and torch.equal(sch.sigmas, torch.from_numpy(rec["sigmas"]))
    assert sch.sigma_bound == float(rec["sigmas"][meta["boundary_step_index"]])

    class FakeModel:
        def set_scheduler(self, s):
            pass

        def infer(self, inputs):
            pass

    mm = L.MultiDistillModelStruct([FakeModel(), FakeModel()], cfg, boundary_step_index=meta["boundary_step_index"])
    mm.set_scheduler(sch)
    picks = []
    for i in range(4):
        sch.step_pre(i)
        mm.infer(None)
        picks.append(mm.cur_model_index)
    assert picks == meta["picks"] and sch.sample_guide_scale == 5.0
