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


def step_case(name):
    rec = load_golden(name)
    meta = ast.literal_eval(str(rec["meta"]))
    model = {}
    for i in range(meta["layers"]):
        model.update(R.make_block_weights(meta["dim"], meta["ffn"], seed=meta["seed"], block_index=i))
    model.update(R.make_head_weights(meta["dim"], 64, seed=meta["seed"]))
    model.update(T.make_pre_weights(meta["dim"], 16, 256, meta["text_dim"], seed=meta["seed"]))
    return rec, meta, model


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
