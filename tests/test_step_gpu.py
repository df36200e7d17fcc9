"""GPU: step-glue kernels and the whole denoise loop through the C ABI, against the CPU
restatement and the reference's golden trajectory.

Tolerances: the gathers are exact (bf16 rounding of the same values); the MLPs are bf16 GEMMs
against fp32 (cosine); the denoise loop reports the final-latent max relative error against the
reference's fp32 trajectory, the north star's second numerics figure."""
import pytest
import torch

import lightx2v_amd as L
from lightx2v_amd import hip
from oracle import wan_step_ref as T
from tests.test_step_oracle import clip_case, moe_case, per_token_case, step_case
from tests.util import cos, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(16, 1, 2, 2), (16, 3, 8, 12), (4, 2, 6, 10), (16, 5, 30, 52)])
def test_patchify_unpatchify_exact(cuda, shape):
    lat = torch.randn(*shape, generator=torch.Generator().manual_seed(5))
    d = lat.to(cuda)
    rows = hip.patchify(d)
    assert torch.equal(rows.cpu(), T.patchify(lat).to(torch.bfloat16))
    assert torch.equal(d.cpu(), lat)  # untouched without round_latents
    rows2 = hip.patchify(d, round_latents=True)
    assert torch.equal(rows2, rows) and torch.equal(d.cpu(), lat.to(torch.bfloat16).float())
    c, f, h, w = shape
    y = torch.randn(f * (h // 2) * (w // 2), 4 * c, generator=torch.Generator().manual_seed(6)).to(torch.bfloat16)
    out = hip.unpatchify(y.to(cuda), c, f, h, w)
    assert torch.equal(out.cpu(), T.unpatchify(y.float(), (f, h // 2, w // 2), c))


@pytest.mark.parametrize("t", [999.0, 500.0, 1.0, 0.0])
def test_timestep_embedding_and_silu(cuda, t):
    ref = T.sinusoidal_embedding_1d(256, torch.tensor([t]))
    got = hip.timestep_embedding(t, 256, cuda)
    assert (got.float().cpu() - ref).abs().max().item() <= 4e-3  # bf16 rounding of values in [-1, 1]
    x = torch.randn(1, 1000, generator=torch.Generator().manual_seed(int(t))).to(torch.bfloat16)
    s = hip.silu(x.to(cuda))
    assert (s.float().cpu() - torch.nn.functional.silu(x.float())).abs().max().item() <= 2e-2


def test_lincomb_and_cfg_combine(cuda):
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(16, 3, 8, 12, generator=g) for _ in range(4)]
    cs = [0.75, -1.25, 0.3, 2.0]
    for k in range(1, 5):
        got = hip.lincomb([(c, x.to(cuda)) for c, x in zip(cs[:k], xs[:k])])
        ref = sum(c * x.double() for c, x in zip(cs[:k], xs[:k]))
        assert (got.double().cpu() - ref).abs().max().item() <= 1e-5
    a = xs[0].to(cuda)
    hip.lincomb([(2.0, a), (1.0, xs[1].to(cuda))], out=a)  # aliasing an input
    assert (a.cpu() - (2.0 * xs[0] + xs[1])).abs().max().item() <= 1e-5
    got = hip.cfg_combine(xs[0].to(cuda), xs[1].to(cuda), 5.0)
    assert torch.equal(got.cpu(), T.cfg_combine(xs[0], xs[1], 5.0))
    with pytest.raises(hip.Lx2vError):
        hip.lincomb([(1.0, xs[0].to(cuda))] * 5)


def _model(meta, model, cuda, **kw):
    cfg = {"dim": meta["dim"], "num_heads": meta["heads"], "num_layers": meta["layers"], "ffn_dim": meta["ffn"], "in_dim": 16, "out_dim": 16,
           "freq_dim": 256, "task": "t2v", "infer_steps": meta["steps"], "sample_shift": meta["shift"], "sample_guide_scale": meta["guide"],
           "enable_cfg": True, "run_device": "cuda"}
    cfg.update(kw)
    m = L.WanModel(cfg, weight_dict=dict(model))
    m.to_cuda()
    sch = L.WanScheduler(cfg)
    m.set_scheduler(sch)
    return m, sch


def test_pre_infer_matches_reference(cuda):
    rec, meta, model = step_case("step_tiny_d256")
    m, sch = _model(meta, model, cuda)
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    sch.step_pre(0)
    inputs = {"text_encoder_output": {"context": torch.from_numpy(rec["context"]).to(cuda), "context_null": torch.from_numpy(rec["context_null"]).to(cuda)}}
    out = m.pre_infer.infer(m.pre_weight, inputs)
    for got, key in [(out.x, "pre_x"), (out.embed, "pre_embed"), (out.embed0, "pre_embed0"), (out.context, "pre_context")]:
        assert cos(got, torch.from_numpy(rec[key])) >= 0.9995, key
    assert out.grid_sizes.tuple == (3, 4, 6)


@pytest.mark.parametrize("name", ["step_tiny_d256", "step_small_d512"])
def test_denoise_loop_against_reference_trajectory(cuda, name):
    rec, meta, model = step_case(name)
    inputs_host = {"context": torch.from_numpy(rec["context"]), "context_null": torch.from_numpy(rec["context_null"])}
    inputs = {"text_encoder_output": {k: v.to(cuda) for k, v in inputs_host.items()}}
    report = {}
    for keep in (True, False):  # fp32 latents between steps / the reference's production bf16 rounding
        m, sch = _model(meta, model, cuda, keep_latents_dtype_in_scheduler=keep)
        sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
        for i in range(meta["steps"]):
            sch.step_pre(i)
            m.infer(inputs)
            assert cos(sch.noise_pred, torch.from_numpy(rec[f"noise_pred_{i}"])) >= 0.999, (keep, i)
            sch.step_post()
        final = torch.from_numpy(rec[f"latents_{meta['steps']}"])
        report[keep] = (rel_err(sch.latents, final), cos(sch.latents, final))
    print(f"\n{name}: final-latent max-rel-err / cosine vs the reference fp32 trajectory: "
          f"fp32 latents {report[True][0]:.3e} / {report[True][1]:.6f}; bf16-rounded latents {report[False][0]:.3e} / {report[False][1]:.6f}")
    assert report[True][0] <= 3e-2 and report[True][1] >= 0.9995
    assert report[False][0] <= 5e-2 and report[False][1] >= 0.999


def test_denoise_loop_with_fp32_residual_stream(cuda):
    """The whole loop with `sensitive_layer_dtype: "fp32"` (x in fp32 between the Linears, through all
    blocks and into the head): closer to the reference's fp32 trajectory than the bf16 stream."""
    name = "step_small_d512"
    rec, meta, model = step_case(name)
    inputs = {"text_encoder_output": {"context": torch.from_numpy(rec["context"]).to(cuda), "context_null": torch.from_numpy(rec["context_null"]).to(cuda)}}
    final = torch.from_numpy(rec[f"latents_{meta['steps']}"])
    errs = {}
    for mode in ("fp32", None):
        m, sch = _model(meta, model, cuda, keep_latents_dtype_in_scheduler=True, sensitive_layer_dtype=mode)
        assert m.transformer_infer.fp32_residual == (mode == "fp32")
        sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
        for i in range(meta["steps"]):
            sch.step_pre(i)
            m.infer(inputs)
            assert cos(sch.noise_pred, torch.from_numpy(rec[f"noise_pred_{i}"])) >= 0.999, (mode, i)
            sch.step_post()
        errs[mode] = (rel_err(sch.latents, final), cos(sch.latents, final))
    print(f"\n{name}: final-latent max-rel-err / cosine, fp32 residual stream {errs['fp32'][0]:.3e} / {errs['fp32'][1]:.6f}; bf16 stream {errs[None][0]:.3e} / {errs[None][1]:.6f}")
    assert errs["fp32"][0] <= 3e-2 and errs["fp32"][1] >= 0.9995
    assert errs["fp32"][0] <= errs[None][0] * 1.05


def test_scheduler_on_reference_noise_is_tight(cuda):
    """UniPC alone, fed the reference's own noise predictions: only fp32 rounding separates them."""
    rec, meta, model = step_case("step_tiny_d256")
    m, sch = _model(meta, model, cuda, keep_latents_dtype_in_scheduler=True)
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    for i in range(meta["steps"]):
        sch.step_pre(i)
        sch.noise_pred = torch.from_numpy(rec[f"noise_pred_{i}"]).to(cuda)
        sch.step_post()
        ref = torch.from_numpy(rec[f"latents_{i + 1}"])
        assert (sch.latents.cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), i


def test_two_expert_i2v_loop_against_reference_trajectory(cuda):
    """Wan2.2-A14B shape of the loop: 36 input channels (latents + conditioning), high-/low-noise
    experts switched at the boundary timestep, a guide scale per expert."""
    rec, meta, models = moe_case()
    cfg = {"dim": meta["dim"], "num_heads": meta["heads"], "num_layers": meta["layers"], "ffn_dim": meta["ffn"], "in_dim": 36, "out_dim": 16,
           "freq_dim": 256, "task": "i2v", "use_image_encoder": False, "model_cls": "wan2.2_moe", "infer_steps": meta["steps"],
           "sample_shift": meta["shift"], "sample_guide_scale": list(meta["guides"]), "enable_cfg": True, "run_device": "cuda",
           "keep_latents_dtype_in_scheduler": True}
    experts = [L.WanModel(cfg, weight_dict=dict(m)) for m in models]
    mm = L.MultiModelStruct(experts, cfg, boundary=meta["boundary"])
    mm.to_cuda()
    sch = L.WanScheduler(cfg)
    mm.set_scheduler(sch)
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    inputs = {"text_encoder_output": {"context": torch.from_numpy(rec["context"]).to(cuda), "context_null": torch.from_numpy(rec["context_null"]).to(cuda)},
              "image_encoder_output": {"vae_encoder_out": torch.from_numpy(rec["y"]).to(cuda)}}
    picks = []
    for i in range(meta["steps"]):
        sch.step_pre(i)
        mm.infer(inputs)
        picks.append(mm.cur_model_index)
        assert cos(sch.noise_pred, torch.from_numpy(rec[f"noise_pred_{i}"])) >= 0.999, i
        sch.step_post()
    assert picks == meta["picks"]
    final = torch.from_numpy(rec[f"latents_{meta['steps']}"])
    err, c = rel_err(sch.latents, final), cos(sch.latents, final)
    print(f"\ntwo-expert i2v loop: final-latent max-rel-err {err:.3e}, cosine {c:.6f}")
    assert err <= 3e-2 and c >= 0.9995


def test_clip_i2v_loop_against_reference_trajectory(cuda):
    """Wan2.1 i2v with a CLIP image encoder: img_emb (LN, Linear, exact GELU, Linear, LN) on the 257 image
    tokens, k_img / v_img / norm_k_img and the second cross-attention in every block."""
    rec, meta, model = clip_case()
    cfg = {"dim": meta["dim"], "num_heads": meta["heads"], "num_layers": meta["layers"], "ffn_dim": meta["ffn"], "in_dim": 36, "out_dim": 16,
           "freq_dim": 256, "task": "i2v", "use_image_encoder": True, "model_cls": "wan2.1", "infer_steps": meta["steps"],
           "sample_shift": meta["shift"], "sample_guide_scale": meta["guide"], "enable_cfg": True, "run_device": "cuda",
           "keep_latents_dtype_in_scheduler": True}
    inputs = {"text_encoder_output": {"context": torch.from_numpy(rec["context"]).to(cuda), "context_null": torch.from_numpy(rec["context_null"]).to(cuda)},
              "image_encoder_output": {"vae_encoder_out": torch.from_numpy(rec["y"]).to(cuda), "clip_encoder_out": torch.from_numpy(rec["clip_fea"]).to(cuda)}}
    for fused in (True, False):
        m = L.WanModel(dict(cfg, fused_block=fused), weight_dict=dict(model))
        m.to_cuda()
        sch = L.WanScheduler(cfg)
        m.set_scheduler(sch)
        sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
        sch.step_pre(0)
        ctx = m.pre_infer.infer(m.pre_weight, inputs).context
        want = torch.from_numpy(rec["pre_context"])
        assert tuple(ctx.shape) == tuple(want.shape) and cos(ctx[:257], want[:257]) >= 0.9995 and cos(ctx[257:], want[257:]) >= 0.9995
        for i in range(meta["steps"]):
            sch.step_pre(i)
            m.infer(inputs)
            assert cos(sch.noise_pred, torch.from_numpy(rec[f"noise_pred_{i}"])) >= 0.999, (fused, i)
            sch.step_post()
        final = torch.from_numpy(rec[f"latents_{meta['steps']}"])
        err, c = rel_err(sch.latents, final), cos(sch.latents, final)
        print(f"\nCLIP i2v loop (fused_block={fused}): final-latent max-rel-err {err:.3e}, cosine {c:.6f}")
        assert err <= 3e-2 and c >= 0.9995


def test_per_token_timestep_forward_against_reference(cuda):
    """Per-token timestep vector: embeddings per distinct value dealt to the tokens, per-token LN
    modulation and gates in every block (fused and op-by-op) and in the head."""
    rec, meta, model = per_token_case()
    cfg = {"dim": meta["dim"], "num_heads": meta["heads"], "num_layers": meta["layers"], "ffn_dim": meta["ffn"], "in_dim": 16, "out_dim": 16,
           "freq_dim": 256, "task": "t2v", "infer_steps": 4, "sample_shift": 5.0, "sample_guide_scale": 5.0, "enable_cfg": False, "run_device": "cuda",
           "keep_latents_dtype_in_scheduler": True}
    inputs = {"text_encoder_output": {"context": torch.from_numpy(rec["context"]).to(cuda), "context_null": torch.from_numpy(rec["context"]).to(cuda)}}
    want = torch.from_numpy(rec["noise_pred"])
    for fused in (True, False):
        m = L.WanModel(dict(cfg, fused_block=fused), weight_dict=dict(model))
        m.to_cuda()
        sch = L.WanScheduler(cfg)
        m.set_scheduler(sch)
        sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
        sch.step_pre(1)
        sch.timestep_input = torch.from_numpy(rec["t_vec"])
        out = m.pre_infer.infer(m.pre_weight, inputs)
        S = out.x.shape[0]
        assert tuple(out.embed0.shape) == (S, 6, meta["dim"]) and tuple(out.embed.shape) == (S, meta["dim"])
        m.infer(inputs)
        c, err = cos(sch.noise_pred, want), rel_err(sch.noise_pred, want)
        print(f"\nper-token timestep forward (fused_block={fused}): cosine {c:.6f}, max-rel-err {err:.3e}")
        assert c >= 0.9995 and err <= 3e-2


def test_head_accepts_the_three_dimensional_per_token_embedding(cuda):
    """`infer_non_blocks` with e [1, S, D] (the reference's diffusion-forcing branch, transformer_infer.py:79-82)
    is the per-token modulation its 2-D branch forms from e [S, D]."""
    rec, meta, model = per_token_case()
    m, sch = _model(dict(meta, steps=4, shift=5.0, guide=5.0), model, cuda)
    g = torch.Generator().manual_seed(3)
    S, D = 3 * 4 * 6, meta["dim"]
    x = torch.randn(S, D, generator=g).to(torch.bfloat16).to(cuda)
    e = (0.1 * torch.randn(S, D, generator=g)).to(torch.bfloat16).to(cuda)
    two = m.transformer_infer.infer_non_blocks(m.transformer_weights, x, e)
    three = m.transformer_infer.infer_non_blocks(m.transformer_weights, x, e.unsqueeze(0))
    assert torch.equal(two, three)
    with pytest.raises(NotImplementedError):
        m.transformer_infer.infer_non_blocks(m.transformer_weights, x, e.reshape(2, S // 2, D))


def test_row_modulation_kernels(cuda):
    g = torch.Generator().manual_seed(8)
    rows, dim = 37, 256
    x = torch.randn(rows, dim, generator=g).to(torch.bfloat16)
    vec_s, vec_h, vec_g = (0.1 * torch.randn(dim, generator=g) for _ in range(3))
    e = (0.2 * torch.randn(rows, 6, dim, generator=g)).to(torch.bfloat16)
    d = e.to(cuda)
    got = hip.ln_modulate(x.to(cuda), None, None, hip.RowMod(vec_s.to(cuda), d[:, 1]), hip.RowMod(vec_h.to(cuda), d[:, 0]))
    ref = torch.nn.functional.layer_norm(x.float(), (dim,), eps=1e-6) * (1 + vec_s + e[:, 1].float()) + vec_h + e[:, 0].float()
    assert (got.float().cpu() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    y = torch.randn(rows, dim, generator=g).to(torch.bfloat16)
    xx = x.to(cuda).clone()
    hip.gate_residual_(xx, y.to(cuda), hip.RowMod(vec_g.to(cuda), d[:, 2]))
    ref = x.float() + y.float() * (vec_g + e[:, 2].float())
    assert (xx.float().cpu() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    with pytest.raises(hip.Lx2vError):
        hip.ln_modulate(x.to(cuda), None, None, hip.RowMod(vec_s.to(cuda), d[:, 1]), vec_h.to(cuda))
    with pytest.raises(hip.Lx2vError):
        hip.gate_residual_(xx, y.to(cuda), hip.RowMod(vec_g.to(cuda), d[:5, 2]))


def test_gelu_erf_kernel(cuda):
    x = (3.0 * torch.randn(4096, generator=torch.Generator().manual_seed(4))).to(torch.bfloat16)
    got = hip.gelu_erf(x.to(cuda)).float().cpu()
    ref = torch.nn.functional.gelu(x.float(), approximate="none")
    assert (got - ref).abs().max().item() <= 2.0 ** -8 * max(1.0, ref.abs().max().item())


def test_teacache_loop_against_reference_run(cuda):
    """`feature_caching: "Tea"`: the same forwards skip their blocks as in the reference's run (both
    CFG branches), and the trajectory stays with the reference's."""
    rec, meta, model = step_case("step_teacache_d256")
    inputs = {"text_encoder_output": {k: torch.from_numpy(rec[k]).to(cuda) for k in ("context", "context_null")}}
    m, sch = _model(meta, model, cuda, keep_latents_dtype_in_scheduler=True, feature_caching="Tea", teacache_thresh=meta["thresh"],
                    coefficients=meta["coefficients"], use_ret_steps=meta["use_ret_steps"])
    assert type(m.transformer_infer).__name__ == "WanTransformerInferTeaCaching" and sch.transformer_infer is m.transformer_infer
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    for i in range(meta["steps"]):
        sch.step_pre(i)
        m.infer(inputs)
        sch.step_post()
        assert cos(sch.latents, torch.from_numpy(rec[f"latents_{i + 1}"])) >= 0.999, i
    assert sch.caching_records == rec["records"].tolist() and sch.caching_records_2 == rec["records_2"].tolist()
    final = torch.from_numpy(rec[f"latents_{meta['steps']}"])
    print(f"\nteacache loop: records {''.join('C' if r else '-' for r in sch.caching_records)}, final-latent max-rel-err {rel_err(sch.latents, final):.3e}")
    assert rel_err(sch.latents, final) <= 5e-2
    sch.clear()
    assert m.transformer_infer.previous_residual_even is None and all(sch.caching_records)
    with pytest.raises(NotImplementedError):
        _model(meta, model, cuda, feature_caching="TaylorSeer")


def test_magcache_loop_against_reference_run(cuda):
    """`feature_caching: "Mag"`: the forwards the reference's `WanTransformerInferMagCaching` run
    skipped are skipped here (both CFG branches), and the trajectory stays with the reference's."""
    rec, meta, model = step_case("step_magcache_d256")
    inputs = {"text_encoder_output": {k: torch.from_numpy(rec[k]).to(cuda) for k in ("context", "context_null")}}
    m, sch = _model(meta, model, cuda, keep_latents_dtype_in_scheduler=True, feature_caching="Mag", magcache_thresh=meta["thresh"], magcache_K=meta["K"],
                    magcache_retention_ratio=meta["retention_ratio"], magcache_ratios=meta["ratios"], magcache_calibration=False)
    infer = m.transformer_infer
    assert type(infer).__name__ == "WanTransformerInferMagCaching" and sch.transformer_infer is infer
    ran = {True: [], False: []}
    real = infer.infer_calculating

    def traced(*a, **k):
        ran[sch.infer_condition][-1] = True
        return real(*a, **k)

    real_main = infer.infer_main_blocks

    def main(*a, **k):
        ran[sch.infer_condition].append(False)
        return real_main(*a, **k)

    infer.infer_calculating, infer.infer_main_blocks = traced, main
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    for i in range(meta["steps"]):
        sch.step_pre(i)
        m.infer(inputs)
        sch.step_post()
        assert cos(sch.latents, torch.from_numpy(rec[f"latents_{i + 1}"])) >= 0.999, i
    assert ran[True] == rec["records"].tolist() and ran[False] == rec["records_2"].tolist()
    final = torch.from_numpy(rec[f"latents_{meta['steps']}"])
    print(f"\nmagcache loop: ran {''.join('C' if r else '-' for r in ran[True])}, final-latent max-rel-err {rel_err(sch.latents, final):.3e}")
    assert rel_err(sch.latents, final) <= 5e-2
    sch.clear()
    assert infer.residual_cache == {True: None, False: None} and infer.accumulated_steps == {True: 0, False: 0}
    with pytest.raises(ValueError):
        _model(meta, model, cuda, feature_caching="Mag", magcache_thresh=0.1, magcache_K=2, magcache_retention_ratio=0.2, magcache_ratios=[[1.0] * 3, [1.0] * 3])


def test_magcache_calibration_statistics(cuda):
    """Calibration mode never skips and records, per branch and step, mean and std of the per-token
    norm ratio of successive residuals and their mean cosine distance (fp32 torch on the same
    residuals as the check; transformer_infer.py:1057-1064)."""
    rec, meta, model = step_case("step_magcache_d256")
    inputs = {"text_encoder_output": {k: torch.from_numpy(rec[k]).to(cuda) for k in ("context", "context_null")}}
    m, sch = _model(meta, model, cuda, keep_latents_dtype_in_scheduler=True, feature_caching="Mag", magcache_thresh=meta["thresh"], magcache_K=meta["K"],
                    magcache_retention_ratio=meta["retention_ratio"], magcache_ratios=[], magcache_calibration=True)
    infer = m.transformer_infer
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    kept = []
    for i in range(3):
        sch.step_pre(i)
        m.infer(inputs)
        kept.append(infer.residual_cache[True].float().cpu())
        sch.step_post()
        # without skipping, the loop is the plain loop: compare with the cache-free start of the trajectory
        if i < 2:
            assert cos(sch.latents, torch.from_numpy(rec[f"latents_{i + 1}"])) >= 0.999
    assert len(infer.norm_ratio[0]) == 3 and len(infer.norm_ratio[1]) == 3 and len(infer.cos_dis[0]) == 3
    ratio = kept[2].norm(dim=-1) / kept[1].norm(dim=-1)
    assert abs(infer.norm_ratio[0][2] - ratio.mean().item()) < 2e-2 and abs(infer.norm_std[0][2] - ratio.std().item()) < 2e-2
    want = (1 - torch.nn.functional.cosine_similarity(kept[2], kept[1], dim=-1)).mean().item()
    assert abs(infer.cos_dis[0][2] - want) < 2e-2


def test_ti2v_loop_against_reference_trajectory(cuda):
    """model_cls "wan2.2" with task i2v (the dense ti2v model): the image's VAE latents stay on the first
    latent frame through prepare / every step_post, the model sees a per-token timestep vector (row
    modulation kernels), nothing is concatenated in the pre-infer."""
    from tests.test_step_oracle import plain_case

    rec, meta, (model,) = plain_case("step_ti2v_d256")
    m, sch = _model(meta, model, cuda, task="i2v", model_cls="wan2.2", use_image_encoder=False, keep_latents_dtype_in_scheduler=True)
    z = torch.from_numpy(rec["z"])
    inputs = {"text_encoder_output": {"context": torch.from_numpy(rec["context"]).to(cuda), "context_null": torch.from_numpy(rec["context_null"]).to(cuda)},
              "image_encoder_output": {"vae_encoder_out": z.to(cuda)}}
    sch.prepare(meta["seed"], meta["latent_shape"], image_encoder_output=inputs["image_encoder_output"], latents=torch.from_numpy(rec["noise0"]))
    inputs["image_encoder_output"]["vae_encoder_out"] = None   # as the reference's runner does after prepare
    assert torch.equal(sch.latents.cpu(), torch.from_numpy(rec["latents0"]))
    for i in range(meta["steps"]):
        sch.step_pre(i)
        assert torch.equal(sch.timestep_input.cpu(), torch.from_numpy(rec[f"timestep_input_{i}"]))
        m.infer(inputs)
        assert cos(sch.noise_pred, torch.from_numpy(rec[f"noise_pred_{i}"])) >= 0.999, i
        sch.step_post()
        assert torch.equal(sch.latents[:, :1].cpu(), z)
    final = torch.from_numpy(rec[f"latents_{meta['steps']}"])
    err, c = rel_err(sch.latents, final), cos(sch.latents, final)
    print(f"\nti2v loop: final-latent max-rel-err {err:.3e}, cosine {c:.6f}")
    assert err <= 3e-2 and c >= 0.9995


@pytest.mark.parametrize("name,in_dim,experts", [("step_distill_d256", 16, 1), ("step_distill_moe_i2v_d256", 36, 2), ("step_distill_dyncfg_d256", 16, 1)])
def test_distill_loops_against_reference_trajectory(cuda, name, in_dim, experts):
    """The 4-step distilled loops without CFG: Wan2.1 distill (one model, t2v) and the Wan2.2 two-expert
    distill (i2v, 36 input channels, experts switched on the step index)."""
    from tests.test_step_oracle import plain_case

    rec, meta, models = plain_case(name, in_dim, experts)
    cfg = {"dim": meta["dim"], "num_heads": meta["heads"], "num_layers": meta["layers"], "ffn_dim": meta["ffn"], "in_dim": in_dim, "out_dim": 16,
           "freq_dim": 256, "task": "i2v" if experts == 2 else "t2v", "use_image_encoder": False, "infer_steps": 4, "denoising_step_list": meta["step_list"],
           "boundary_step_index": meta["boundary_step_index"], "sample_shift": meta["shift"], "sample_guide_scale": 5.0, "enable_cfg": False,
           "run_device": "cuda", "keep_latents_dtype_in_scheduler": True}
    if meta.get("cfg_scale") is not None:   # the distilled guidance embedding
        cfg.update({"model_cls": "wan2.1_distill", "enable_dynamic_cfg": True, "cfg_scale": meta["cfg_scale"]})
    nets = [L.WanModel(cfg, weight_dict=dict(w)) for w in models]
    if experts == 2:
        model = L.MultiDistillModelStruct(nets, cfg, boundary_step_index=meta["boundary_step_index"])
        sch = L.Wan22StepDistillScheduler(cfg)
    else:
        model, sch = nets[0], L.WanStepDistillScheduler(cfg)
    model.to_cuda()
    model.set_scheduler(sch)
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    inputs = {"text_encoder_output": {"context": torch.from_numpy(rec["context"]).to(cuda), "context_null": torch.from_numpy(rec["context"]).to(cuda)}}
    if "y" in rec:
        inputs["image_encoder_output"] = {"vae_encoder_out": torch.from_numpy(rec["y"]).to(cuda)}
    picks = []
    for i in range(4):
        sch.step_pre(i)
        model.infer(inputs)
        picks.append(getattr(model, "cur_model_index", 0))
        assert cos(sch.noise_pred, torch.from_numpy(rec[f"noise_pred_{i}"])) >= 0.999, i
        sch.step_post()
    assert picks == meta["picks"]
    if meta.get("cfg_scale") is not None:
        # the guidance embedding is live: the same model without it is measurably further from the reference's first prediction
        plain = L.WanModel(dict(cfg, enable_dynamic_cfg=False), weight_dict=dict(models[0]))
        plain.to_cuda()
        sch0 = L.WanStepDistillScheduler(cfg)
        for mdl, s_ in ((plain, sch0), (model, sch)):
            mdl.set_scheduler(s_)
            s_.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
            s_.step_pre(0)
            mdl.infer(inputs)
        gold0 = torch.from_numpy(rec["noise_pred_0"])
        c_with, c_without = cos(sch.noise_pred, gold0), cos(sch0.noise_pred, gold0)
        print(f"\ndynamic cfg: cosine of the first prediction with / without the guidance embedding {c_with:.6f} / {c_without:.6f}")
        assert c_with >= 0.99995 and c_without <= 0.9999
        # (restore the trajectory state for the checks below)
        sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
        for i in range(4):
            sch.step_pre(i)
            model.infer(inputs)
            sch.step_post()
    final = torch.from_numpy(rec["latents_4"])
    err, c = rel_err(sch.latents, final), cos(sch.latents, final)
    print(f"\n{name}: final-latent max-rel-err {err:.3e}, cosine {c:.6f}")
    assert err <= 3e-2 and c >= 0.9995
    # the scheduler alone on the reference's predictions
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    for i in range(4):
        sch.step_pre(i)
        sch.noise_pred = torch.from_numpy(rec[f"noise_pred_{i}"]).to(cuda)
        sch.step_post()
        ref = torch.from_numpy(rec[f"latents_{i + 1}"])
        assert (sch.latents.cpu() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item()), i
