"""CPU restatement (torch fp32) of the reference's HunyuanVideo-1.5 denoise-step glue around the
double-stream blocks: pre-infer (patch embedding with patch (1,1,1), timestep MLP, the text branch
with its two-block token refiner, condition-type embedding, token reorder), unpatchify, the flow
schedulers (Euler and the 4-step distilled one) and the model-level forward / denoise loop.

TEST INFRASTRUCTURE ONLY: imported by tests/ and the golden generator; the product path never
touches this file.  Parity: `oracle/gen_golden_hy_step.py` runs the reference's own
`HunyuanVideo15PreInfer`, `HunyuanVideo15TransformerInfer`, `HunyuanVideo15PostInfer`,
`HunyuanVideo15Scheduler` and `HunyuanVideo15StepDistillScheduler` (fp32) and asserts this
restatement agrees before writing `tests/golden/hy_step_*.npz`.  ONE piece is parity unpinned: the
masked self-attention inside the token refiner calls flash-attn's `flash_attn_varlen_qkvpacked_func`
between `unpad_input` / `pad_input` (infer/attn_no_pad.py:33-56), a wheel that is absent here; the
generator substitutes `refiner_attention` below (attention among the valid tokens, zeros for the
padded ones = what unpad -> varlen attention -> pad computes) for it.

Restated pieces (reference file:line, under lightx2v/):
  * timestep embedding                 models/networks/hunyuan_video/infer/pre_infer.py:190-210
  * pre-infer                          .../infer/pre_infer.py:74-167 (non-SR branch)
  * token refiner                      .../infer/pre_infer.py:169-188
  * token reorder                      .../infer/pre_infer.py:212-240
  * unpatchify                         .../infer/post_infer.py:21-39
  * one forward, CFG combine           models/networks/hunyuan_video/model.py:199-258
  * Euler flow scheduler               models/schedulers/hunyuan_video/scheduler.py:26-121
  * 4-step distilled scheduler         models/schedulers/hunyuan_video/step_distill/scheduler.py:6-33
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from oracle import hy_block_ref as HY
from oracle import wan_block_ref as R

FREQ_DIM = 256
MAX_PERIOD = 10000
REFINER_BLOCKS = 2
RB = "txt_in.individual_token_refiner.blocks"


# --------------------------------------------------------------------------- synthetic checkpoint / inputs
def make_pre_weights(dim: int, heads: int, latent_channels: int, text_dim: int, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Checkpoint keys of `HunyuanVideo15PreWeights` (weights/pre_weights.py:17-147), synthetic values.
    `img_in` sees latents, conditioning latents and one mask channel: 2 C + 1 input channels."""
    g = torch.Generator().manual_seed(seed * 1000003 + 515171)

    def lin(o, i, s=0.02):
        return (s * torch.randn(o, i, generator=g)).to(dtype), (0.02 * torch.randn(o, generator=g)).to(dtype)

    w = {}
    w["img_in.proj.weight"] = (0.05 * torch.randn(dim, 2 * latent_channels + 1, 1, 1, 1, generator=g)).to(dtype)
    w["img_in.proj.bias"] = (0.02 * torch.randn(dim, generator=g)).to(dtype)
    shapes = {
        "time_in.mlp.0": (dim, FREQ_DIM), "time_in.mlp.2": (dim, dim),
        "txt_in.t_embedder.mlp.0": (dim, FREQ_DIM), "txt_in.t_embedder.mlp.2": (dim, dim),
        "txt_in.c_embedder.linear_1": (dim, text_dim), "txt_in.c_embedder.linear_2": (dim, dim),
        "txt_in.input_embedder": (dim, text_dim),
    }
    for i in range(REFINER_BLOCKS):
        shapes.update({f"{RB}.{i}.self_attn_qkv": (3 * dim, dim), f"{RB}.{i}.self_attn_proj": (dim, dim), f"{RB}.{i}.mlp.fc1": (4 * dim, dim),
                       f"{RB}.{i}.mlp.fc2": (dim, 4 * dim), f"{RB}.{i}.adaLN_modulation.1": (2 * dim, dim)})
    for name, (o, i) in shapes.items():
        w[name + ".weight"], w[name + ".bias"] = lin(o, i, 0.05 if "qkv" in name else 0.02)
    for i in range(REFINER_BLOCKS):
        for n in ("norm1", "norm2"):
            w[f"{RB}.{i}.{n}.weight"] = (1 + 0.1 * torch.randn(dim, generator=g)).to(dtype)
            w[f"{RB}.{i}.{n}.bias"] = (0.05 * torch.randn(dim, generator=g)).to(dtype)
    w["cond_type_embedding.weight"] = (0.1 * torch.randn(3, dim, generator=g)).to(dtype)
    return w


def make_step_inputs(dim: int, text_dim: int, txt_len: int, txt_valid: int, byt5_len: int, byt5_valid: int, vis_len: int, vis_valid: int,
                     seed: int = 0, dtype=torch.float32):
    """The dictionaries the runner hands to `infer` (hunyuan_video_15_runner.py:223-224, 372-383): text
    features [1, L, text_dim] + mask [1, L] for the prompt and the empty prompt, byT5 glyph features
    [1, Lb, D] + mask, image-encoder tokens [1, Lv, D] + mask (all zero for t2v).  Valid tokens come
    first, as tokenizers pad on the right."""
    g = torch.Generator().manual_seed(seed + 7919)

    def masked(n, valid):
        m = torch.zeros(1, n, dtype=torch.int64)
        m[:, :valid] = 1
        return m

    ctx = torch.randn(1, txt_len, text_dim, generator=g).to(dtype)
    ctx_null = torch.randn(1, txt_len, text_dim, generator=g).to(dtype)
    byt5 = torch.randn(1, byt5_len, dim, generator=g).to(dtype)
    vis = (torch.randn(1, vis_len, dim, generator=g) if vis_valid else torch.zeros(1, vis_len, dim)).to(dtype)
    return {
        "text_encoder_output": {"context": [ctx, masked(txt_len, txt_valid)], "context_null": [ctx_null, masked(txt_len, max(1, txt_valid // 3))],
                                "byt5_features": byt5, "byt5_masks": masked(byt5_len, byt5_valid)},
        "image_encoder_output": {"siglip_output": vis, "siglip_mask": masked(vis_len, vis_valid).to(dtype), "cond_latents": None},
    }


# --------------------------------------------------------------------------- pre-infer
def timestep_embedding(t: torch.Tensor, dim: int = FREQ_DIM, max_period: int = MAX_PERIOD) -> torch.Tensor:
    """pre_infer.py:190-210: [cos(t f_i) || sin(t f_i)], f_i = exp(-ln(max_period) i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def refiner_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """`attention(..., attn_mask=mask)` of pre_infer.py:33-61 for one sequence: q, k, v [L, H, d],
    mask [L] bool -> [L, H*d].  flash-attn's no-pad wrapper drops the masked tokens, attends among
    the rest and pads the result back with zeros (attn_no_pad.py:33-56)."""
    out = torch.zeros(q.shape[0], q.shape[1] * q.shape[2], dtype=q.dtype)
    idx = mask.nonzero().squeeze(1)
    out[idx] = R.sdpa(q[idx], k[idx], v[idx])
    return out


def token_refiner(w: Dict[str, torch.Tensor], x: torch.Tensor, mask: torch.Tensor, c: torch.Tensor, heads: int) -> torch.Tensor:
    """`run_individual_token_refiner` (pre_infer.py:169-188): x [L, D], mask [1, L], c [1, D]."""
    mask = mask[0].clone().bool()
    mask[0] = True
    for i in range(REFINER_BLOCKS):
        p = f"{RB}.{i}."
        gate_msa, gate_mlp = R.linear(F.silu(c), w[p + "adaLN_modulation.1.weight"], w[p + "adaLN_modulation.1.bias"]).chunk(2, dim=1)
        y = R.layer_norm(x, w[p + "norm1.weight"], w[p + "norm1.bias"])
        qkv = R.linear(y, w[p + "self_attn_qkv.weight"], w[p + "self_attn_qkv.bias"])
        q, k, v = qkv.view(x.shape[0], 3, heads, -1).unbind(1)  # "L (K H D) -> K L H D"
        a = refiner_attention(q, k, v, mask)
        x = x + R.linear(a, w[p + "self_attn_proj.weight"], w[p + "self_attn_proj.bias"]) * gate_msa
        h = R.linear(R.layer_norm(x, w[p + "norm2.weight"], w[p + "norm2.bias"]), w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])
        h = R.linear(F.silu(h), w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
        x = x + h * gate_mlp
    return x


def reorder_txt_token(first: torch.Tensor, txt: torch.Tensor, first_mask: torch.Tensor, txt_mask: torch.Tensor, zero_feat: bool = False):
    """pre_infer.py:212-240 for batch 1: [first valid, txt valid, first padded, txt padded] and the
    matching mask; with `zero_feat` the padded rows are zeroed."""
    fm, tm = first_mask[0].bool(), txt_mask[0].bool()
    a, b = first[0], txt[0]
    pad_a, pad_b = (torch.zeros_like(a[~fm]), torch.zeros_like(b[~tm])) if zero_feat else (a[~fm], b[~tm])
    out = torch.cat([a[fm], b[tm], pad_a, pad_b], dim=0)
    mask = torch.cat([fm[fm], tm[tm], fm[~fm], tm[~tm]], dim=0).to(torch.int64)
    return out.unsqueeze(0), mask.unsqueeze(0)


def pre_infer(w: Dict[str, torch.Tensor], latents: torch.Tensor, cond_latents_concat: torch.Tensor, mask_concat: torch.Tensor, t: torch.Tensor,
              inputs: Dict, infer_condition: bool, heads: int):
    """`HunyuanVideo15PreInfer.infer` (pre_infer.py:74-167, non-SR): latents [1, C, T, H, W], t a 0-d or
    [1] timestep -> (img [1, S, D], txt [1, Lt', D], vec [1, D] = silu(time MLP), grid)."""
    te = inputs["text_encoder_output"]
    txt, text_mask = te["context"] if infer_condition else te["context_null"]
    byt5_txt, byt5_mask = te["byt5_features"], te["byt5_masks"]
    vis, vis_mask = inputs["image_encoder_output"]["siglip_output"], inputs["image_encoder_output"]["siglip_mask"]
    grid = tuple(latents.shape[2:])
    x = torch.cat([latents.float(), cond_latents_concat.float(), mask_concat.float()], dim=1)
    t_expand = t.reshape(-1)[:1].repeat(x.shape[0])
    img = F.conv3d(x, w["img_in.proj.weight"], w["img_in.proj.bias"], stride=(1, 1, 1)).flatten(2).transpose(1, 2)
    t_freq = timestep_embedding(t_expand)
    vec = R.linear(F.silu(R.linear(t_freq, w["time_in.mlp.0.weight"], w["time_in.mlp.0.bias"])), w["time_in.mlp.2.weight"], w["time_in.mlp.2.bias"])
    t_aware = R.linear(F.silu(R.linear(t_freq, w["txt_in.t_embedder.mlp.0.weight"], w["txt_in.t_embedder.mlp.0.bias"])),
                       w["txt_in.t_embedder.mlp.2.weight"], w["txt_in.t_embedder.mlp.2.bias"])
    mask_float = text_mask.float().unsqueeze(-1)
    c_aware = (txt * mask_float).sum(dim=1) / mask_float.sum(dim=1)
    c_aware = R.linear(F.silu(R.linear(c_aware, w["txt_in.c_embedder.linear_1.weight"], w["txt_in.c_embedder.linear_1.bias"])),
                       w["txt_in.c_embedder.linear_2.weight"], w["txt_in.c_embedder.linear_2.bias"])
    c = t_aware + c_aware
    out = R.linear(txt[0], w["txt_in.input_embedder.weight"], w["txt_in.input_embedder.bias"])
    txt = token_refiner(w, out, text_mask, c, heads).unsqueeze(0)
    emb = w["cond_type_embedding.weight"]
    txt = txt + emb[0]
    byt5_txt = byt5_txt + emb[1]
    txt, text_mask = reorder_txt_token(byt5_txt, txt, byt5_mask, text_mask, zero_feat=True)
    vis = vis + emb[2]
    txt, text_mask = reorder_txt_token(vis, txt, vis_mask, text_mask)
    txt = txt[:, : int(text_mask.sum())]
    return img.contiguous(), txt.contiguous(), F.silu(vec), grid


def unpatchify(x: torch.Tensor, grid: Tuple[int, int, int], channels: int) -> torch.Tensor:
    """post_infer.py:21-39 with patch (1,1,1): [1, S(+padding), C] -> [1, C, T, H, W]."""
    t, h, w = grid
    x = x[:, : t * h * w]
    return x.reshape(x.shape[0], t, h, w, channels).permute(0, 4, 1, 2, 3).contiguous()


# --------------------------------------------------------------------------- schedulers
def sd3_time_shift(t: torch.Tensor, shift: float) -> torch.Tensor:
    return (shift * t) / (1 + (shift - 1) * t)


class EulerScheduler:
    """`HunyuanVideo15Scheduler` (scheduler.py:11-121): sigmas = shift(linspace(1, 0, n + 1)),
    timesteps = 1000 sigma_i, x <- x + v (sigma_{i+1} - sigma_i) in fp32; the latents start as
    bf16 noise and become fp32 with the first update."""

    def __init__(self, infer_steps: int, sample_shift: float, task: str = "t2v"):
        self.infer_steps, self.sample_shift, self.task = infer_steps, sample_shift, task
        self.step_index = 0
        self.latents = self.noise_pred = None

    def set_timesteps(self):
        sigmas = torch.linspace(1, 0, self.infer_steps + 1)
        if self.sample_shift != 1.0:
            sigmas = sd3_time_shift(sigmas, self.sample_shift)
        self.sigmas = sigmas
        self.timesteps = (sigmas[:-1] * 1000).to(torch.float32)

    def prepare(self, latents: torch.Tensor, cond_latents: Optional[torch.Tensor] = None):
        """scheduler.py:26-31, 62-102 with the noise given (bf16 [1, C, T, H, W])."""
        self.latents = latents
        self.set_timesteps()
        mask = torch.zeros(latents.shape[2])
        if self.task == "i2v":
            mask[0] = 1.0
        if cond_latents is not None and self.task == "i2v":
            self.cond_latents_concat = cond_latents.repeat(1, 1, latents.shape[2], 1, 1)
            self.cond_latents_concat[:, :, 1:] = 0.0
        else:
            self.cond_latents_concat = torch.zeros(latents.shape)
        self.mask_concat = torch.zeros(latents.shape[0], 1, *latents.shape[2:])
        self.mask_concat[:, :, mask.nonzero().squeeze(1)] = 1.0

    def step_post(self):
        dt = self.sigmas[self.step_index + 1] - self.sigmas[self.step_index]
        self.latents = self.latents.to(torch.float32) + self.noise_pred.to(torch.float32) * dt


class StepDistillScheduler(EulerScheduler):
    """`HunyuanVideo15StepDistillScheduler` (step_distill/scheduler.py:6-33): the 1000-entry shifted
    sigma table sampled at 1000 - denoising_step_list; x0 = x - sigma v, re-noised along the same
    v to the next sigma; the latents keep their dtype."""

    def __init__(self, denoising_step_list: Sequence[int], sample_shift: float, task: str = "t2v"):
        super().__init__(len(denoising_step_list), sample_shift, task)
        self.denoising_step_list = list(denoising_step_list)

    def set_timesteps(self):
        sigmas = torch.linspace(1.0, 0.0, 1001)[:-1]
        sigmas = self.sample_shift * sigmas / (1 + (self.sample_shift - 1) * sigmas)
        idx = [1000 - x for x in self.denoising_step_list]
        self.timesteps = (sigmas * 1000)[idx]
        self.sigmas = sigmas[idx]

    def step_post(self):
        v = self.noise_pred.to(torch.float32)
        sigma = self.sigmas[self.step_index].item()
        x = self.latents.to(torch.float32) - sigma * v
        if self.step_index < self.infer_steps - 1:
            x = x + v * self.sigmas[self.step_index + 1].item()
        self.latents = x.to(self.latents.dtype)


# --------------------------------------------------------------------------- model
def model_forward(pre_w, blocks: List[Dict[str, torch.Tensor]], final_w, sch: EulerScheduler, inputs: Dict, infer_condition: bool, heads: int,
                  rope_dim_list=HY.ROPE_DIM_LIST) -> torch.Tensor:
    """`_infer_cond_uncond` (model.py:232-249) without sequence parallelism -> [C, T, H, W]."""
    t = sch.timesteps[sch.step_index]
    img, txt, vec, grid = pre_infer(pre_w, sch.latents, sch.cond_latents_concat, sch.mask_concat, t, inputs, infer_condition, heads)
    cos_sin = HY.build_cos_sin(grid, img.shape[-1] // heads, rope_dim_list)
    x = HY.forward(img[0], txt[0], vec, blocks, final_w, cos_sin, heads)
    return unpatchify(x.unsqueeze(0), grid, x.shape[-1])[0]


def denoise_loop(pre_w, blocks, final_w, sch: EulerScheduler, inputs: Dict, heads: int, guide_scale: Optional[float] = None):
    """The runner's loop (default_runner.py:137-169) around `HunyuanVideo15Model.infer`
    (model.py:199-230): per step one forward, or two combined as u + g (c - u).  Returns the
    per-step noise predictions and latents."""
    preds, lats = [], []
    for i in range(sch.infer_steps):
        sch.step_index = i
        cond = model_forward(pre_w, blocks, final_w, sch, inputs, True, heads)
        if guide_scale is not None:
            unc = model_forward(pre_w, blocks, final_w, sch, inputs, False, heads)
            sch.noise_pred = unc + guide_scale * (cond - unc)
        else:
            sch.noise_pred = cond
        sch.step_post()
        preds.append(sch.noise_pred.clone())
        lats.append(sch.latents.clone())
    return preds, lats
