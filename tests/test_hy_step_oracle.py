"""CPU: the HunyuanVideo-1.5 step-glue restatement (oracle/hy_step_ref.py) against the fixtures written
from the reference's own HunyuanVideo15PreInfer / PostInfer / schedulers / denoise loop
(tests/golden/hy_step_*.npz), and the product's host-side scheduler tables against the restatement.

The masked attention inside the token refiner is parity unpinned (flash-attn is absent here; see the
header of oracle/hy_step_ref.py); every other operation of these fixtures ran in the reference."""
import ast

import pytest
import torch

from oracle import hy_block_ref as HY
from oracle import hy_step_ref as S
from tests.util import load_golden

HY_STEP_CASES = ["hy_step_distill_d256", "hy_step_cfg_i2v_d256"]


def hy_step_case(name):
    rec = load_golden(name)
    meta = ast.literal_eval(str(rec["meta"]))
    C = meta["latent_shape"][0]
    pre_w = S.make_pre_weights(meta["dim"], meta["heads"], C, meta["text_dim"], seed=meta["seed"])
    blocks = [HY.make_double_block_weights(meta["dim"], meta["heads"], meta["mlp"], seed=meta["seed"] + i, block_index=i) for i in range(meta["depth"])]
    fw = HY.make_final_weights(meta["dim"], C, seed=meta["seed"])
    inputs = S.make_step_inputs(meta["dim"], meta["text_dim"], *meta["lens"], seed=meta["seed"])
    if meta["task"] == "i2v":
        inputs["image_encoder_output"]["cond_latents"] = torch.from_numpy(rec["cond_latents"])
    return rec, meta, pre_w, blocks, fw, inputs


def scheduler_of(meta):
    if meta["distill"]:
        return S.StepDistillScheduler(meta["steps"], meta["shift"], meta["task"])
    return S.EulerScheduler(meta["steps"], meta["shift"], meta["task"])


@pytest.mark.parametrize("name", HY_STEP_CASES)
def test_pre_infer_matches_reference(name):
    rec, meta, pre_w, blocks, fw, inputs = hy_step_case(name)
    sch = scheduler_of(meta)
    sch.prepare(torch.from_numpy(rec["latents0"]).to(torch.bfloat16), inputs["image_encoder_output"]["cond_latents"])
    assert torch.equal(sch.sigmas, torch.from_numpy(rec["sigmas"])) and torch.equal(sch.timesteps, torch.from_numpy(rec["timesteps"]))
    for flag, tag in ((True, "cond"), (False, "uncond")):
        img, txt, vec, grid = S.pre_infer(pre_w, sch.latents, sch.cond_latents_concat, sch.mask_concat, sch.timesteps[0], inputs, flag, meta["heads"])
        assert grid == tuple(meta["latent_shape"][1:])
        for got, key in ((img[0], f"pre_img_{tag}"), (txt[0], f"pre_txt_{tag}"), (vec, f"pre_vec_{tag}")):
            want = torch.from_numpy(rec[key])
            assert got.shape == want.shape
            assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), key
    lens = meta["lens"]
    valid_cond = lens[1] + lens[3] + lens[5]
    assert rec["pre_txt_cond"].shape[0] == valid_cond  # [image-encoder valid | byT5 valid | text valid], padding dropped


@pytest.mark.parametrize("name", HY_STEP_CASES)
def test_denoise_loop_matches_reference_trajectory(name):
    rec, meta, pre_w, blocks, fw, inputs = hy_step_case(name)
    sch = scheduler_of(meta)
    sch.prepare(torch.from_numpy(rec["latents0"]).to(torch.bfloat16), inputs["image_encoder_output"]["cond_latents"])
    preds, lats = S.denoise_loop(pre_w, blocks, fw, sch, inputs, meta["heads"], guide_scale=meta["guide"])
    bf16 = "bfloat16" in meta["latents_dtype"]
    for i, (p, x) in enumerate(zip(preds, lats)):
        want_p, want_x = torch.from_numpy(rec[f"noise_pred_{i}"]), torch.from_numpy(rec[f"latents_{i + 1}"])
        # bf16 latents (distilled schedule): an fp32 difference of 1e-6 can flip a rounding, and the flip feeds the next step
        tol = (2.0 ** -6 if bf16 and i > 0 else 5e-5) * max(1.0, want_p.abs().max().item())
        assert (p - want_p).abs().max().item() <= tol, f"prediction {i}"
        assert (x.float() - want_x).abs().max().item() <= (2.0 ** -6 if bf16 else 2e-4) * max(1.0, want_x.abs().max().item()), f"latents {i + 1}"
    assert lats[-1].dtype == (torch.bfloat16 if bf16 else torch.float32)


@pytest.mark.parametrize("name", HY_STEP_CASES)
def test_scheduler_alone_is_exact_on_reference_predictions(name):
    rec, meta, *_ , inputs = hy_step_case(name)
    sch = scheduler_of(meta)
    sch.prepare(torch.from_numpy(rec["latents0"]).to(torch.bfloat16), inputs["image_encoder_output"]["cond_latents"])
    for i in range(sch.infer_steps):
        sch.step_index = i
        sch.noise_pred = torch.from_numpy(rec[f"noise_pred_{i}"])
        sch.step_post()
        assert torch.equal(sch.latents.float(), torch.from_numpy(rec[f"latents_{i + 1}"]))


def test_refiner_attention_drops_padding_and_pads_with_zeros():
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(9, 2, 16, generator=g) for _ in range(3))
    mask = torch.tensor([1, 1, 0, 1, 0, 0, 1, 0, 0]).bool()
    out = S.refiner_attention(q, k, v, mask)
    assert torch.equal(out[~mask], torch.zeros(5, 32))
    from oracle import wan_block_ref as R
    assert torch.allclose(out[mask], R.sdpa_naive(q[mask], k[mask], v[mask]).float(), atol=1e-5)


@pytest.mark.parametrize("name", HY_STEP_CASES)
def test_product_scheduler_tables_match(name):
    import lightx2v_amd as L

    rec, meta, *_ = hy_step_case(name)
    cfg = {"infer_steps": meta["steps"] if not meta["distill"] else len(meta["steps"]), "sample_shift": meta["shift"], "sample_guide_scale": meta["guide"] or -1.0,
           "task": meta["task"], "hidden_size": meta["dim"], "heads_num": meta["heads"], "run_device": "cpu"}
    if meta["distill"]:
        cfg["denoising_step_list"] = meta["steps"]
        sch = L.HunyuanVideo15StepDistillScheduler(cfg)
    else:
        sch = L.HunyuanVideo15Scheduler(cfg)
    sch.set_timesteps(sch.infer_steps, device="cpu", shift=sch.sample_shift)
    assert torch.equal(sch.sigmas, torch.from_numpy(rec["sigmas"])) and torch.equal(sch.timesteps, torch.from_numpy(rec["timesteps"]))
    lat = torch.zeros(1, *meta["latent_shape"])
    cond = torch.ones(1, meta["latent_shape"][0], 1, *meta["latent_shape"][2:]) if meta["task"] == "i2v" else None
    cc, mc = sch._prepare_cond_latents_and_mask(meta["task"], cond, lat, sch.get_task_mask(meta["task"], meta["latent_shape"][1]))
    ref = S.EulerScheduler(1, 1.0, meta["task"])
    ref.prepare(lat, cond)
    assert torch.equal(cc, ref.cond_latents_concat) and torch.equal(mc, ref.mask_concat)
