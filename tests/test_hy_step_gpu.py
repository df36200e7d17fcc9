"""GPU: the HunyuanVideo-1.5 step glue (pre-infer with the token refiner, unpatchify, Euler and
distilled schedulers, model-level infer with and without CFG) through the C ABI, against the
fixtures from the reference's own run (tests/golden/hy_step_*.npz) and the CPU restatement.

Tolerances: row moves are exact; single bf16 kernels within one bf16 rounding of the fp32 value;
pre-infer outputs cosine >= 0.9995 against the reference's fp32 run (bf16 GEMM chains); per-step
predictions cosine >= 0.999; final latents reported as max relative error against the reference's
trajectory and held to 2e-2 (bf16 model, 3-4 steps)."""
import pytest
import torch

import lightx2v_amd as L
from lightx2v_amd import hip
from oracle import hy_step_ref as S
from tests.test_hy_step_oracle import HY_STEP_CASES, hy_step_case
from tests.util import cos, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c,n,pad", [(65, 1000, 128), (9, 48, 64), (32, 37, 0), (3, 7, 8)])
def test_channels_rows_transposes_exact(cuda, c, n, pad):
    g = torch.Generator().manual_seed(c * 1000 + n)
    planes = torch.randn(c, n, generator=g)
    rows = hip.channels_to_rows(planes.to(cuda), pad_to=pad)
    k = max((c + 7) // 8 * 8, pad)
    assert rows.shape == (n, k)
    assert torch.equal(rows[:, :c].cpu(), planes.t().to(torch.bfloat16))
    assert torch.equal(rows[:, c:].cpu(), torch.zeros(n, k - c, dtype=torch.bfloat16))
    y = torch.randn(n + 3, c, generator=g).to(torch.bfloat16)  # rows past `tokens` (sequence-parallel padding) are ignored
    back = hip.rows_to_channels(y.to(cuda), c, n)
    assert torch.equal(back.cpu(), y[:n].float().t())
    with pytest.raises(hip.Lx2vError):
        hip.rows_to_channels(y.to(cuda), c, n + 10)


def test_weighted_mean_rows_and_add_rowvec(cuda):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(77, 136, generator=g).to(torch.bfloat16)
    mask = (torch.rand(77, generator=g) > 0.4).float()
    got = hip.weighted_mean_rows(x.to(cuda), mask.to(cuda))
    ref = (x.float() * mask[:, None]).sum(0) / mask.sum()
    assert (got.float().cpu()[0] - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6
    v = torch.randn(136, generator=g).to(torch.bfloat16)
    out = hip.add_rowvec(x.to(cuda), v.to(cuda))
    assert torch.equal(out.cpu(), (x + v))
    view = torch.zeros(77, 272, dtype=torch.bfloat16, device=cuda)
    hip.add_rowvec(x.to(cuda), v.to(cuda), out=view[:, 136:])
    assert torch.equal(view[:, 136:].cpu(), x + v) and not view[:, :136].any()


def _model(meta, pre_w, blocks, fw, cuda, **kw):
    C = meta["latent_shape"][0]
    cfg = {"mm_double_blocks_depth": meta["depth"], "heads_num": meta["heads"], "hidden_size": meta["dim"], "patch_size": (1, 1, 1), "out_channels": C,
           "task": meta["task"], "sample_shift": meta["shift"], "sample_guide_scale": -1.0 if meta["guide"] is None else meta["guide"],
           "enable_cfg": meta["guide"] is not None, "infer_steps": len(meta["steps"]) if meta["distill"] else meta["steps"], "run_device": "cuda",
           "rope_dim_list": [16, 56, 56], "rope_theta": 256.0}
    if meta["distill"]:
        cfg["denoising_step_list"] = meta["steps"]
    cfg.update(kw)
    wd = dict(pre_w)
    for b in blocks:
        wd.update(b)
    wd.update(fw)
    m = L.HunyuanVideo15Model(cfg, weight_dict={k: v.to(cuda) for k, v in wd.items()})
    m.to_cuda()
    sch = (L.HunyuanVideo15StepDistillScheduler if meta["distill"] else L.HunyuanVideo15Scheduler)(cfg)
    m.set_scheduler(sch)
    return m, sch


def _to_cuda(inputs, cuda):
    te, ie = inputs["text_encoder_output"], inputs["image_encoder_output"]
    mv = lambda t: None if t is None else t.to(cuda)
    return {"text_encoder_output": {"context": [mv(te["context"][0]), mv(te["context"][1])], "context_null": [mv(te["context_null"][0]), mv(te["context_null"][1])],
                                    "byt5_features": mv(te["byt5_features"]), "byt5_masks": mv(te["byt5_masks"])},
            "image_encoder_output": {"siglip_output": mv(ie["siglip_output"]), "siglip_mask": mv(ie["siglip_mask"]), "cond_latents": mv(ie["cond_latents"])}}


@pytest.mark.parametrize("name", HY_STEP_CASES)
def test_pre_infer_matches_reference(cuda, name):
    rec, meta, pre_w, blocks, fw, inputs = hy_step_case(name)
    m, sch = _model(meta, pre_w, blocks, fw, cuda)
    dev_inputs = _to_cuda(inputs, cuda)
    sch.prepare(meta["seed"], meta["latent_shape"], image_encoder_output=dev_inputs["image_encoder_output"], latents=torch.from_numpy(rec["latents0"]))
    sch.step_pre(0)
    assert torch.equal(sch.sigmas, torch.from_numpy(rec["sigmas"])) and torch.equal(sch.timesteps.cpu(), torch.from_numpy(rec["timesteps"]))
    for flag, tag in ((True, "cond"), (False, "uncond")):
        sch.infer_condition = flag
        out = m.pre_infer.infer(m.pre_weight, dev_inputs)
        assert out.grid_sizes == tuple(meta["latent_shape"][1:])
        for got, key in ((out.img[0], f"pre_img_{tag}"), (out.txt[0], f"pre_txt_{tag}"), (out.vec, f"pre_vec_{tag}")):
            want = torch.from_numpy(rec[key])
            assert tuple(got.shape) == tuple(want.shape), key
            assert cos(got, want) >= 0.9995, (key, cos(got, want))
            assert rel_err(got, want) <= 3e-2, (key, rel_err(got, want))


def test_refiner_with_scattered_padding(cuda):
    """Masks that are not a prefix (left padding, holes): rows are gathered, attended, scattered back."""
    rec, meta, pre_w, blocks, fw, inputs = hy_step_case("hy_step_distill_d256")
    te = inputs["text_encoder_output"]
    L_txt = te["context"][1].shape[1]
    mask = torch.zeros(1, L_txt, dtype=torch.int64)
    mask[0, [2, 3, 5, 8, 13, 21]] = 1  # token 0 padded: the refiner still counts it
    te["context"] = [te["context"][0], mask]
    m, sch = _model(meta, pre_w, blocks, fw, cuda)
    dev_inputs = _to_cuda(inputs, cuda)
    sch.prepare(meta["seed"], meta["latent_shape"], image_encoder_output=dev_inputs["image_encoder_output"], latents=torch.from_numpy(rec["latents0"]))
    sch.step_pre(0)
    out = m.pre_infer.infer(m.pre_weight, dev_inputs)
    ref = S.EulerScheduler(1, 1.0)
    ref.prepare(torch.from_numpy(rec["latents0"]).to(torch.bfloat16))
    img, txt, vec, _ = S.pre_infer(pre_w, ref.latents, ref.cond_latents_concat, ref.mask_concat, sch.timesteps[0].cpu(), inputs, True, meta["heads"])
    assert tuple(out.txt.shape) == tuple(txt.shape)
    assert cos(out.txt, txt) >= 0.9995 and cos(out.img, img) >= 0.9995


@pytest.mark.parametrize("name", HY_STEP_CASES)
def test_denoise_loop_matches_reference_trajectory(cuda, name):
    rec, meta, pre_w, blocks, fw, inputs = hy_step_case(name)
    m, sch = _model(meta, pre_w, blocks, fw, cuda)
    dev_inputs = _to_cuda(inputs, cuda)
    sch.prepare(meta["seed"], meta["latent_shape"], image_encoder_output=dev_inputs["image_encoder_output"], latents=torch.from_numpy(rec["latents0"]))
    worst_cos = 1.0
    for i in range(sch.infer_steps):
        sch.step_pre(i)
        m.infer(dev_inputs)
        c = cos(sch.noise_pred, torch.from_numpy(rec[f"noise_pred_{i}"]))
        worst_cos = min(worst_cos, c)
        assert c >= 0.999, f"step {i}: prediction cosine {c}"
        sch.step_post()
    want = torch.from_numpy(rec[f"latents_{sch.infer_steps}"])
    assert sch.latents.dtype == (torch.bfloat16 if "bfloat16" in meta["latents_dtype"] else torch.float32)
    err = rel_err(sch.latents[0], want[0] if want.dim() == 5 else want)
    print(f"{name}: worst prediction cosine {worst_cos:.6f}, final-latent max-rel-err {err:.3e}")
    assert err <= 2e-2


def test_scheduler_updates_on_reference_predictions(cuda):
    for name in HY_STEP_CASES:
        rec, meta, pre_w, blocks, fw, inputs = hy_step_case(name)
        m, sch = _model(meta, pre_w, blocks, fw, cuda)
        sch.prepare(meta["seed"], meta["latent_shape"], image_encoder_output=_to_cuda(inputs, cuda)["image_encoder_output"], latents=torch.from_numpy(rec["latents0"]))
        for i in range(sch.infer_steps):
            sch.step_pre(i)
            sch.noise_pred = torch.from_numpy(rec[f"noise_pred_{i}"]).to(cuda)
            sch.step_post()
            want = torch.from_numpy(rec[f"latents_{i + 1}"])
            assert torch.equal(sch.latents.float().cpu().reshape(want.shape), want), f"{name} step {i}"
