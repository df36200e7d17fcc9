"""CPU restatement (torch fp32) of the reference's denoise-step glue around the Wan DiT blocks.

TEST INFRASTRUCTURE ONLY: imported by tests/, `__graft_entry__.smoke()` and the golden generator;
the product path never touches this file.  Parity is pinned: `oracle/gen_golden_step.py` runs the
reference's own `WanPreInfer`, `WanTransformerInfer`, `WanPostInfer` and `WanScheduler` and
asserts this restatement agrees before writing `tests/golden/step_*.npz`.

Restated pieces (reference file:line):
  * sinusoidal timestep embedding      lightx2v/models/networks/wan/infer/utils.py:205-215
  * pre-infer (patch embed, time/text MLPs)   .../wan/infer/pre_infer.py:27-126
  * unpatchify                         .../wan/infer/post_infer.py:26-31
  * one forward = pre -> blocks -> head -> unpatchify   .../wan/model.py:447-466
  * CFG combine                        .../wan/model.py:433
  * UniPC multistep scheduler (flow sigmas, order 2, bh2 corrector/predictor)
                                       lightx2v/models/schedulers/wan/scheduler.py:49-76,126-434
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from oracle import wan_block_ref as R

PATCH = (1, 2, 2)


# --------------------------------------------------------------------------- synthetic checkpoint
def make_pre_weights(dim: int, in_dim: int = 16, freq_dim: int = 256, text_dim: int = 4096, seed: int = 0, dtype=torch.float32):
    """Checkpoint keys of `WanPreWeights` (weights/pre_weights.py:17-40), synthetic values."""
    g = torch.Generator().manual_seed(seed * 1000003 + 424243)

    def lin(o, i):
        return (0.02 * torch.randn(o, i, generator=g)).to(dtype), (0.02 * torch.randn(o, generator=g)).to(dtype)

    w = {}
    w["patch_embedding.weight"] = (0.05 * torch.randn(dim, in_dim, *PATCH, generator=g)).to(dtype)
    w["patch_embedding.bias"] = (0.02 * torch.randn(dim, generator=g)).to(dtype)
    for name, (o, i) in {
        "text_embedding.0": (dim, text_dim),
        "text_embedding.2": (dim, dim),
        "time_embedding.0": (dim, freq_dim),
        "time_embedding.2": (dim, dim),
        "time_projection.1": (6 * dim, dim),
    }.items():
        w[name + ".weight"], w[name + ".bias"] = lin(o, i)
    return w


def make_img_emb_weights(dim: int, clip_dim: int, seed: int = 0, dtype=torch.float32):
    """`img_emb.proj.{0,1,3,4}` of an i2v checkpoint with a CLIP image encoder (weights/pre_weights.py:43-59):
    LayerNorm(clip_dim), Linear(clip_dim, clip_dim), [GELU], Linear(clip_dim, dim), LayerNorm(dim)."""
    g = torch.Generator().manual_seed(seed * 1000003 + 616161)
    w = {"img_emb.proj.0.weight": 1 + 0.1 * torch.randn(clip_dim, generator=g), "img_emb.proj.0.bias": 0.05 * torch.randn(clip_dim, generator=g),
         "img_emb.proj.1.weight": 0.05 * torch.randn(clip_dim, clip_dim, generator=g), "img_emb.proj.1.bias": 0.02 * torch.randn(clip_dim, generator=g),
         "img_emb.proj.3.weight": 0.05 * torch.randn(dim, clip_dim, generator=g), "img_emb.proj.3.bias": 0.02 * torch.randn(dim, generator=g),
         "img_emb.proj.4.weight": 1 + 0.1 * torch.randn(dim, generator=g), "img_emb.proj.4.bias": 0.05 * torch.randn(dim, generator=g)}
    return {k: v.to(dtype) for k, v in w.items()}


def img_emb(pw: Dict[str, torch.Tensor], clip_fea: torch.Tensor) -> torch.Tensor:
    """pre_infer.py:99-113 (i2v): clip_fea [257, clip_dim] -> image context tokens [257, D]; the
    GELU between the two Linears is the exact (erf) one."""
    h = R.layer_norm(clip_fea, pw["img_emb.proj.0.weight"], pw["img_emb.proj.0.bias"])  # eps 1e-6: the plugin default, not nn.LayerNorm's
    h = F.gelu(R.linear(h, pw["img_emb.proj.1.weight"], pw["img_emb.proj.1.bias"]), approximate="none")
    h = R.linear(h, pw["img_emb.proj.3.weight"], pw["img_emb.proj.3.bias"])
    return R.layer_norm(h, pw["img_emb.proj.4.weight"], pw["img_emb.proj.4.bias"])


# --------------------------------------------------------------------------- pre / post infer
def make_guidance_weights(dim: int, freq_dim: int = 256, seed: int = 0, dtype=torch.float32):
    """Synthetic `guidance_embedding.linear_{1,2}` (pre_weights.py:60-69; model_cls "wan2.1_distill" with
    `enable_dynamic_cfg`): 256 -> dim -> freq_dim, so that the result adds to the sinusoidal timestep embedding."""
    g = torch.Generator().manual_seed(seed * 104729 + 977)
    w = {}
    for name, (o, i) in {"guidance_embedding.linear_1": (dim, 256), "guidance_embedding.linear_2": (freq_dim, dim)}.items():
        w[name + ".weight"] = (torch.randn(o, i, generator=g) * 0.05).to(dtype)
        w[name + ".bias"] = (torch.randn(o, generator=g) * 0.02).to(dtype)
    return w


def guidance_scale_embedding(w: torch.Tensor, embedding_dim: int = 256, cfg_range=(1.0, 6.0), target_range: float = 1000.0) -> torch.Tensor:
    """lightx2v/models/networks/wan/infer/utils.py:218-242: the guide scale rounded, clamped to `cfg_range`,
    mapped to [0, target_range] and embedded as sin || cos over 10000^(-i/(half-1))."""
    lo, hi = cfg_range
    w = torch.clamp(torch.round(w), min=lo, max=hi)
    w = (w - lo) / (hi - lo) * target_range
    half = embedding_dim // 2
    step = torch.log(torch.tensor(10000.0)) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -step)
    ang = w.to(torch.float32)[:, None] * freqs[None, :]
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)


def sinusoidal_embedding_1d(dim: int, position: torch.Tensor) -> torch.Tensor:
    """utils.py:205-215: [cos(p * 10000^(-i/half)) || sin(...)], i < half."""
    half = dim // 2
    position = position.type(torch.float32)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def patchify(latents: torch.Tensor) -> torch.Tensor:
    """[C, F, H, W] -> [F*(H/2)*(W/2), C*1*2*2]: row = token (f, h/2, w/2), column = (c, pf, ph, pw),
    the flattening order of the conv3d weight [D, C, 1, 2, 2] (pre_infer.py:56,63-64)."""
    c, f, h, w = latents.shape
    x = latents.view(c, f, h // 2, 2, w // 2, 2)
    return x.permute(1, 2, 4, 0, 3, 5).reshape(f * (h // 2) * (w // 2), c * 4)


def pre_infer(pw: Dict[str, torch.Tensor], latents: torch.Tensor, t: torch.Tensor, context: torch.Tensor, freq_dim: int = 256,
              y: Optional[torch.Tensor] = None, clip_fea: Optional[torch.Tensor] = None, cfg_scale: Optional[float] = None):
    """pre_infer.py:27-126 for task t2v, and for i2v without the CLIP branch (`use_image_encoder`
    false, the Wan2.2 two-expert configuration): `y` = mask + VAE latents [C_y, F, H, W] is
    concatenated to the latents along channels before the patch embedding (:37-53).
    Returns (x [S,D], embed [1,D], embed0 [6,D], context [T,D], grid)."""
    dim = pw["patch_embedding.bias"].shape[0]
    _, f, h, w = latents.shape
    if y is not None:
        latents = torch.cat([latents, y], dim=0)
    x = R.linear(patchify(latents), pw["patch_embedding.weight"].reshape(dim, -1), pw["patch_embedding.bias"])
    embed = sinusoidal_embedding_1d(freq_dim, t.flatten())
    if cfg_scale is not None:  # `enable_dynamic_cfg` (pre_infer.py:67-73): the guide scale enters through the timestep embedding
        ce = guidance_scale_embedding(torch.tensor([cfg_scale], dtype=torch.float32))
        ce = R.linear(ce, pw["guidance_embedding.linear_1.weight"], pw["guidance_embedding.linear_1.bias"])
        ce = R.linear(F.silu(ce), pw["guidance_embedding.linear_2.weight"], pw["guidance_embedding.linear_2.bias"])
        embed = embed + ce
    embed = R.linear(embed, pw["time_embedding.0.weight"], pw["time_embedding.0.bias"])
    embed = F.silu(embed)
    embed = R.linear(embed, pw["time_embedding.2.weight"], pw["time_embedding.2.bias"])
    embed0 = F.silu(embed)
    embed0 = R.linear(embed0, pw["time_projection.1.weight"], pw["time_projection.1.bias"]).unflatten(1, (6, dim)).squeeze(0)
    out = R.linear(context, pw["text_embedding.0.weight"], pw["text_embedding.0.bias"])
    out = F.gelu(out, approximate="tanh")
    ctx = R.linear(out, pw["text_embedding.2.weight"], pw["text_embedding.2.bias"])
    if clip_fea is not None:  # i2v with a CLIP image encoder: image tokens go in front of the text (:99-113)
        ctx = torch.cat([img_emb(pw, clip_fea), ctx], dim=0)
    return x, embed, embed0, ctx, (f // PATCH[0], h // PATCH[1], w // PATCH[2])


def unpatchify(x: torch.Tensor, grid: Tuple[int, int, int], out_dim: int = 16) -> torch.Tensor:
    """post_infer.py:26-31: [S, 1*2*2*c] -> [c, F, 2H', 2W'] (head columns ordered (pf, ph, pw, c))."""
    x = x[: math.prod(grid)].view(*grid, *PATCH, out_dim)
    x = torch.einsum("fhwpqrc->cfphqwr", x)
    return x.reshape(out_dim, *[i * j for i, j in zip(grid, PATCH)])


def forward(model: Dict[str, torch.Tensor], latents: torch.Tensor, t: torch.Tensor, context: torch.Tensor, num_heads: int, num_layers: int,
            cos_sin=None, y: Optional[torch.Tensor] = None, clip_fea: Optional[torch.Tensor] = None, cfg_scale: Optional[float] = None) -> torch.Tensor:
    """`_infer_cond_uncond` (model.py:447-466): one DiT forward -> noise prediction [16, F, H, W] fp32."""
    x, embed, embed0, ctx, grid = pre_infer(model, latents, t, context, y=y, clip_fea=clip_fea, cfg_scale=cfg_scale)
    if cos_sin is None:
        cos_sin = R.build_cos_sin(grid, model["patch_embedding.bias"].shape[0] // num_heads)
    for i in range(num_layers):
        x = R.block_forward(x, embed0, ctx, model, cos_sin, num_heads, block_index=i, image_tokens=R.IMAGE_TOKENS if clip_fea is not None else 0)
    y = R.head_forward(x, embed, model)
    return unpatchify(y, grid, model["head.head.bias"].shape[0] // 4).float()


def cfg_combine(cond: torch.Tensor, uncond: torch.Tensor, guide: float) -> torch.Tensor:
    """model.py:433."""
    return uncond + guide * (cond - uncond)


# --------------------------------------------------------------------------- UniPC scheduler
class UniPC:
    """`WanScheduler` restated (scheduler.py): flow-matching sigmas with `sample_shift`, solver
    order 2, bh2 ("B(h) = expm1(h)"), corrector on every step after the first.  Scalars are
    computed exactly as the reference does (fp32 torch scalars from the fp32 sigma table)."""

    def __init__(self, infer_steps: int, sample_shift: float, num_train_timesteps: int = 1000, solver_order: int = 2):
        self.infer_steps = infer_steps
        self.solver_order = solver_order
        self.num_train_timesteps = num_train_timesteps
        # prepare(): :55-72
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigma_min, sigma_max = sigmas[-1].item(), sigmas[0].item()
        # set_timesteps(): :139-167
        s = np.linspace(sigma_max, sigma_min, infer_steps + 1).copy()[:-1]
        s = sample_shift * s / (1 + (sample_shift - 1) * s)
        self.timesteps = torch.from_numpy(s * num_train_timesteps).to(dtype=torch.int64)
        self.sigmas = torch.from_numpy(np.concatenate([s, [0]]).astype(np.float32))
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * solver_order
        self.last_sample = None
        self.this_order = None
        self.lower_order_nums = 0
        self.step_index = 0

    def timestep_input(self, step_index: int) -> torch.Tensor:
        return torch.stack([self.timesteps[step_index]])  # step_pre(): :389-391

    def _lambda(self, sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def coefficients(self, step_index: int, order: int, corrector: bool):
        """Scalar part of multistep_uni_{p,c}_bh_update (:200-279 / :281-386): returns
        (a, b, c, rk, rho) with  x_t = a*x - b*m0 - c*res,  res built from rk / rho."""
        if corrector:
            sigma_t, sigma_s0 = self.sigmas[step_index], self.sigmas[step_index - 1]
        else:
            sigma_t, sigma_s0 = self.sigmas[step_index + 1], self.sigmas[step_index]
        alpha_t = 1 - sigma_t
        h = self._lambda(sigma_t) - self._lambda(sigma_s0)
        rk = None
        if order == 2:
            si = step_index - 2 if corrector else step_index - 1
            rk = (self._lambda(self.sigmas[si]) - self._lambda(sigma_s0)) / h
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        rho = None
        if corrector:
            if order == 1:
                rho = torch.tensor([0.5])
            else:
                rks = torch.tensor([rk, 1.0])
                Rm, b = [], []
                h_phi_k = h_phi_1 / hh - 1
                fact = 1
                for i in range(1, order + 1):
                    Rm.append(torch.pow(rks, i - 1))
                    b.append(h_phi_k * fact / B_h)
                    fact *= i + 1
                    h_phi_k = h_phi_k / hh - 1 / fact
                rho = torch.linalg.solve(torch.stack(Rm), torch.tensor(b))
        return sigma_t / sigma_s0, alpha_t * h_phi_1, alpha_t * B_h, rk, rho

    def step(self, step_index: int, noise_pred: torch.Tensor, latents: torch.Tensor) -> torch.Tensor:
        """step_post(): :394-434.  Returns the next latents."""
        model_output = noise_pred.to(torch.float32)
        sample = latents.to(torch.float32)
        use_corrector = step_index > 0 and self.last_sample is not None
        x0 = sample - self.sigmas[step_index] * model_output  # convert_model_output(): :172-188
        if use_corrector:
            a, b, c, rk, rho = self.coefficients(step_index, self.this_order, corrector=True)
            m0 = self.model_outputs[-1]
            x_t_ = a * self.last_sample - b * m0
            if self.this_order == 2:
                corr = rho[0] * ((self.model_outputs[-2] - m0) / rk)
            else:
                corr = 0
            sample = x_t_ - c * (corr + rho[-1] * (x0 - m0))
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = x0
        this_order = min(self.solver_order, self.infer_steps - step_index)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        a, b, c, rk, _ = self.coefficients(step_index, self.this_order, corrector=False)
        m0 = self.model_outputs[-1]
        x_t_ = a * sample - b * m0
        if self.this_order == 2:
            pred = 0.5 * ((self.model_outputs[-2] - m0) / rk)
        else:
            pred = 0
        prev = x_t_ - c * pred
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return prev


def sample_loop(model, latents0: torch.Tensor, context: torch.Tensor, context_null: torch.Tensor, num_heads: int, num_layers: int,
                infer_steps: int, sample_shift: float, guide: float, round_latents=None):
    """The reference's denoise loop (default_runner `run_main` + model.infer + scheduler): per step
    step_pre -> cond / uncond forwards -> CFG -> step_post.  `round_latents`: the dtype the
    reference casts the stored latents to in `step_pre` (BaseScheduler.step_pre, scheduler.py:16-19;
    bf16 in production, None in the fp32 oracle run).  Returns (final latents, per-step latents)."""
    sch = UniPC(infer_steps, sample_shift)
    latents = latents0
    traj = []
    for i in range(infer_steps):
        if round_latents is not None:
            latents = latents.to(round_latents)
        t = sch.timestep_input(i)
        cond = forward(model, latents.float(), t, context, num_heads, num_layers)
        uncond = forward(model, latents.float(), t, context_null, num_heads, num_layers)
        latents = sch.step(i, cfg_combine(cond, uncond, guide), latents)
        traj.append(latents)
    return latents, traj


def sample_loop_two_experts(models, boundary: float, guides, latents0: torch.Tensor, context: torch.Tensor, context_null: torch.Tensor,
                            num_heads: int, num_layers: int, infer_steps: int, sample_shift: float, y: Optional[torch.Tensor] = None):
    """Wan2.2-A14B style loop: `MultiModelStruct` (lightx2v/models/runners/wan/wan_runner.py:394-444)
    picks the high-noise expert (models[0], guide scale guides[0]) while
    timesteps[step] >= boundary * 1000 and the low-noise expert (models[1], guides[1]) after; both
    share one scheduler.  (That class is restated from its source text: its module cannot be
    imported in the golden generator's environment, its import chain needs packages that are absent;
    everything it drives -- per-expert forwards, CFG, scheduler -- is pinned by the reference run.)
    Returns (final latents, per-step latents, per-step expert index)."""
    sch = UniPC(infer_steps, sample_shift)
    latents, traj, picks = latents0, [], []
    for i in range(infer_steps):
        which = 0 if float(sch.timesteps[i]) >= boundary * sch.num_train_timesteps else 1
        t = sch.timestep_input(i)
        cond = forward(models[which], latents, t, context, num_heads, num_layers, y=y)
        uncond = forward(models[which], latents, t, context_null, num_heads, num_layers, y=y)
        latents = sch.step(i, cfg_combine(cond, uncond, guides[which]), latents)
        traj.append(latents)
        picks.append(which)
    return latents, traj, picks


# --------------------------------------------------------------------------- Wan2.2 ti2v (dense, i2v / s2v)
def ti2v_mask(latents: torch.Tensor) -> torch.Tensor:
    """`masks_like(latents, zero=True)` without a generator (lightx2v/utils/utils.py:451-461): ones,
    zeros on the first latent frame."""
    m = torch.ones_like(latents)
    m[:, :1] = 0
    return m


def ti2v_blend(latents: torch.Tensor, z: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """scheduler.py:127-129 and :433-434: the VAE latents of the conditioning image `z` ([C, 1, H, W]
    or [C, F, H, W]) replace the masked frame of the latents."""
    return (1.0 - mask) * z + mask * latents


def ti2v_timestep(mask: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """scheduler.py:391-392: one timestep per token of the (1, 2, 2) patch grid -- 0 on the tokens of
    the conditioning frame, t elsewhere."""
    return (mask[0][:, ::2, ::2] * t).flatten()


def sample_loop_ti2v(model, noise0: torch.Tensor, z: torch.Tensor, context: torch.Tensor, context_null: torch.Tensor, num_heads: int,
                     num_layers: int, infer_steps: int, sample_shift: float, guide: float):
    """The denoise loop of model_cls "wan2.2" with task i2v (default_runner.py:275-285 +
    scheduler.py:49-51,116-129,388-392,433-434): the image enters through the latents, not through
    extra channels -- `prepare_latents` and every `step_post` put `z` back on the first frame, and
    every forward sees a per-token timestep vector.  Returns (latents0, per-step predictions, per-step latents)."""
    sch = UniPC(infer_steps, sample_shift)
    mask = ti2v_mask(noise0)
    latents = ti2v_blend(noise0, z, mask)
    latents0, preds, traj = latents, [], []
    for i in range(infer_steps):
        t = ti2v_timestep(mask, sch.timestep_input(i))
        cond = forward(model, latents, t, context, num_heads, num_layers)
        uncond = forward(model, latents, t, context_null, num_heads, num_layers)
        pred = cfg_combine(cond, uncond, guide)
        latents = ti2v_blend(sch.step(i, pred, latents), z, mask)
        preds.append(pred)
        traj.append(latents)
    return latents0, preds, traj


# --------------------------------------------------------------------------- step-distilled schedule
class StepDistill:
    """`WanStepDistillScheduler` restated (lightx2v/models/schedulers/wan/step_distill/scheduler.py:9-50):
    the shifted sigma table of 1000 training steps sampled at `denoising_step_list`, and an Euler-type
    update x <- (x - sigma_i f) + sigma_{i+1} f (no second term on the last step)."""

    def __init__(self, denoising_step_list, sample_shift: float, num_train_timesteps: int = 1000):
        self.infer_steps = len(denoising_step_list)
        sigmas = torch.linspace(1.0, 0.0, num_train_timesteps + 1)[:-1]                   # :27-28 (sigma_max 1, sigma_min 0)
        sigmas = sample_shift * sigmas / (1 + (sample_shift - 1) * sigmas)               # :29
        timesteps = sigmas * num_train_timesteps                                         # :30
        index = [num_train_timesteps - x for x in denoising_step_list]                    # :32
        self.timesteps = timesteps[index]
        self.sigmas = sigmas[index]

    def timestep_input(self, step_index: int) -> torch.Tensor:
        return torch.stack([self.timesteps[step_index]])

    def step(self, step_index: int, flow_pred: torch.Tensor, latents: torch.Tensor) -> torch.Tensor:
        """step_post(): :43-50."""
        flow_pred = flow_pred.to(torch.float32)
        sigma = self.sigmas[step_index].item()
        out = latents.to(torch.float32) - sigma * flow_pred
        if step_index < self.infer_steps - 1:
            out = out + flow_pred * self.sigmas[step_index + 1].item()
        return out.to(latents.dtype)


def sample_loop_distill(models, boundary_step_index: int, latents0: torch.Tensor, context: torch.Tensor, num_heads: int, num_layers: int,
                        denoising_step_list, sample_shift: float, y: Optional[torch.Tensor] = None, cfg_scale: Optional[float] = None):
    """The step-distilled loop, no CFG (`enable_cfg` false in configs/distill/*.json): one forward per
    step.  `models` = [model] for Wan2.1 distill, or [high, low] for the Wan2.2 two-expert distill,
    where `MultiDistillModelStruct` (lightx2v/models/runners/wan/wan_distill_runner.py:44-75) runs the
    high-noise expert while step_index < boundary_step_index.  `Wan22StepDistillScheduler.step_post`
    (step_distill/scheduler.py:66-76) is the same update as its base class.
    Returns (per-step predictions, per-step latents, per-step expert index)."""
    sch = StepDistill(denoising_step_list, sample_shift)
    latents, preds, traj, picks = latents0, [], [], []
    for i in range(sch.infer_steps):
        which = 0 if (len(models) == 1 or i < boundary_step_index) else 1
        pred = forward(models[which], latents, sch.timestep_input(i), context, num_heads, num_layers, y=y, cfg_scale=cfg_scale)
        latents = sch.step(i, pred, latents)
        preds.append(pred)
        traj.append(latents)
        picks.append(which)
    return preds, traj, picks


# --------------------------------------------------------------------------- TeaCache
class TeaCache:
    """Skip logic of `WanTransformerInferTeaCaching`
    (lightx2v/models/networks/wan/infer/feature_caching/transformer_infer.py:25-157): per CFG branch,
    the relative L1 change of the timestep modulation input (embed0 with `use_ret_steps`, else
    embed) is rescaled by a degree-4 polynomial and accumulated; while the sum stays under the
    threshold the blocks are skipped and the previous pass's residual (blocks' output - input) is
    added to the patch-embedded tokens instead.  The first `ret_steps` and the steps from
    `cutoff_steps` on are always computed."""

    def __init__(self, infer_steps: int, thresh: float, coefficients, use_ret_steps: bool):
        self.thresh = thresh
        self.use_ret_steps = use_ret_steps
        if use_ret_steps:
            self.coefficients, self.ret_steps, self.cutoff_steps = coefficients[0], 5, infer_steps
        else:
            self.coefficients, self.ret_steps, self.cutoff_steps = coefficients[1], 1, infer_steps - 1
        self.acc = {True: 0.0, False: 0.0}
        self.prev_inp = {True: None, False: None}
        self.prev_residual = {True: None, False: None}

    def should_calc(self, step_index: int, cond: bool, embed: torch.Tensor, embed0: torch.Tensor) -> bool:
        inp = embed0 if self.use_ret_steps else embed
        if step_index < self.ret_steps or step_index >= self.cutoff_steps:
            calc = True
            self.acc[cond] = 0.0
        else:
            prev = self.prev_inp[cond]
            rel = ((inp - prev).abs().mean() / prev.abs().mean()).item()
            self.acc[cond] += float(np.poly1d(self.coefficients)(rel))
            calc = not (self.acc[cond] < self.thresh)
            if calc:
                self.acc[cond] = 0.0
        self.prev_inp[cond] = inp.clone()
        return calc


def forward_teacache(model, latents, t, context, num_heads, num_layers, tea: TeaCache, step_index: int, cond: bool):
    """`forward` with the blocks replaced by the cached residual when TeaCache says so.  Returns
    (noise prediction, whether the blocks ran)."""
    x, embed, embed0, ctx, grid = pre_infer(model, latents, t, context)
    calc = tea.should_calc(step_index, cond, embed, embed0)
    if calc:
        cos_sin = R.build_cos_sin(grid, model["patch_embedding.bias"].shape[0] // num_heads)
        x0 = x.clone()
        for i in range(num_layers):
            x = R.block_forward(x, embed0, ctx, model, cos_sin, num_heads, block_index=i)
        tea.prev_residual[cond] = x - x0
    else:
        x = x + tea.prev_residual[cond]
    y = R.head_forward(x, embed, model)
    return unpatchify(y, grid, model["head.head.bias"].shape[0] // 4).float(), calc


def sample_loop_teacache(model, latents0, context, context_null, num_heads, num_layers, infer_steps, sample_shift, guide, thresh, coefficients, use_ret_steps):
    """Denoise loop with TeaCache.  Returns (final latents, per-step latents, records of the
    conditional branch, records of the unconditional branch)."""
    sch = UniPC(infer_steps, sample_shift)
    tea = TeaCache(infer_steps, thresh, coefficients, use_ret_steps)
    latents, traj, rec, rec2 = latents0, [], [], []
    for i in range(infer_steps):
        t = sch.timestep_input(i)
        cond, c1 = forward_teacache(model, latents, t, context, num_heads, num_layers, tea, i, True)
        uncond, c2 = forward_teacache(model, latents, t, context_null, num_heads, num_layers, tea, i, False)
        latents = sch.step(i, cfg_combine(cond, uncond, guide), latents)
        traj.append(latents)
        rec.append(c1)
        rec2.append(c2)
    return latents, traj, rec, rec2


# --------------------------------------------------------------------------- MagCache
class MagCache:
    """Skip logic of `WanTransformerInferMagCaching`
    (lightx2v/models/networks/wan/infer/feature_caching/transformer_infer.py:992-1091): per CFG branch
    and from step `infer_steps * retention_ratio` on, the calibrated magnitude ratio of the step
    (`magcache_ratios[branch][step]`) is multiplied into a running ratio, |1 - ratio| is accumulated,
    and while that sum stays under the threshold and at most K steps were skipped in a row the
    blocks are skipped and the previous pass's residual is re-used."""

    def __init__(self, infer_steps: int, thresh: float, K: int, retention_ratio: float, ratios):
        self.first = int(infer_steps * retention_ratio)
        self.thresh, self.K, self.ratios = thresh, K, np.array(ratios)
        self.err = {True: 0.0, False: 0.0}
        self.steps = {True: 0, False: 0}
        self.ratio = {True: 1.0, False: 1.0}
        self.prev_residual = {True: None, False: None}

    def should_calc(self, step_index: int, cond: bool) -> bool:
        if step_index < self.first:
            return True
        self.ratio[cond] = self.ratio[cond] * self.ratios[0 if cond else 1][step_index]
        self.steps[cond] += 1
        self.err[cond] += np.abs(1 - self.ratio[cond])
        if self.err[cond] < self.thresh and self.steps[cond] <= self.K:
            return False
        self.err[cond], self.steps[cond], self.ratio[cond] = 0, 0, 1.0
        return True


def sample_loop_magcache(model, latents0, context, context_null, num_heads, num_layers, infer_steps, sample_shift, guide, thresh, K, retention_ratio, ratios):
    """Denoise loop with MagCache.  Returns (final latents, per-step latents, which forwards ran
    their blocks: conditional branch, unconditional branch)."""
    sch = UniPC(infer_steps, sample_shift)
    mag = MagCache(infer_steps, thresh, K, retention_ratio, ratios)
    latents, traj, rec, rec2 = latents0, [], [], []

    def fwd(ctx, cond, i, t):
        x, embed, embed0, c, grid = pre_infer(model, latents, t, ctx)
        calc = mag.should_calc(i, cond)
        if calc:
            cos_sin = R.build_cos_sin(grid, model["patch_embedding.bias"].shape[0] // num_heads)
            x0 = x.clone()
            for b in range(num_layers):
                x = R.block_forward(x, embed0, c, model, cos_sin, num_heads, block_index=b)
            mag.prev_residual[cond] = x - x0
        else:
            x = x + mag.prev_residual[cond]
        return unpatchify(R.head_forward(x, embed, model), grid, model["head.head.bias"].shape[0] // 4).float(), calc

    for i in range(infer_steps):
        t = sch.timestep_input(i)
        cond, c1 = fwd(context, True, i, t)
        uncond, c2 = fwd(context_null, False, i, t)
        latents = sch.step(i, cfg_combine(cond, uncond, guide), latents)
        traj.append(latents)
        rec.append(c1)
        rec2.append(c2)
    return latents, traj, rec, rec2
