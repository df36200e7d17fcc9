"""Denoise-step glue around the Wan DiT blocks, on the HIP kernels: pre-infer (patch embedding as a
GEMM, timestep / text MLPs), post-infer (unpatchify), the UniPC flow scheduler and the model-level
`infer` with classifier-free guidance (serial, or split over a CFG group) and Ulysses sharding.

Mirrors, class for class and method for method (t2v task):
  WanPreWeights    lightx2v/models/networks/wan/weights/pre_weights.py:10-80
  WanPreInfer      lightx2v/models/networks/wan/infer/pre_infer.py:9-126
  WanPostInfer     lightx2v/models/networks/wan/infer/post_infer.py:8-31
  WanScheduler     lightx2v/models/schedulers/wan/scheduler.py:12-434 (+ BaseScheduler, scheduler.py:4-23)
  WanModel.infer   lightx2v/models/networks/wan/model.py:395-499
The per-step scalar coefficients of UniPC are host arithmetic on the fp32 sigma table exactly as
the reference computes them; every tensor operation is a kernel of liblx2v_hip.so.
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import comm, hip
from .registry_factory import CONV3D_WEIGHT_REGISTER, LN_WEIGHT_REGISTER, MM_WEIGHT_REGISTER
from .rope_table import build_rope_table, shard_rope_table
from .wan_block import WanPreInferModuleOutput, WanTransformerInfer, WanTransformerWeights, with_defaults
from .weight_module import WeightModule

PATCH_SIZE = (1, 2, 2)


# ------------------------------------------------------------------------------- patch embedding
@CONV3D_WEIGHT_REGISTER("hip-patch")
class HipPatchEmbedWeight:
    """conv3d whose kernel equals its stride -- (1,2,2) for Wan, (1,1,1) for HunyuanVideo-1.5 -- = a
    GEMM over gathered patches.  Same constructor as the reference's Conv3dWeight
    (lightx2v/common/ops/conv/conv3d.py:5-14); `apply` takes the latents [C,F,H,W] (fp32) and returns
    tokens [S, D] (bf16) directly, i.e. the reference's `apply` + `flatten(2).transpose(1,2)`
    (wan/infer/pre_infer.py:56,63-64; hunyuan_video/infer/pre_infer.py:108-109)."""

    STRIDES = ((1, 2, 2), (1, 1, 1))

    def __init__(self, weight_name, bias_name, stride=1, padding=0, dilation=1, groups=1):
        stride = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        if stride not in self.STRIDES or padding != 0 or dilation != 1 or groups != 1:
            raise ValueError("hip-patch implements kernel = stride = (1,2,2) or (1,1,1), no padding / dilation / groups")
        self.weight_name, self.bias_name = weight_name, bias_name
        self.stride = stride
        self.config = {}
        self.weight_nk: Optional[torch.Tensor] = None
        self.bias: Optional[torch.Tensor] = None

    def set_config(self, config=None):
        if config is not None:
            self.config = config

    def load(self, weight_dict):
        w = weight_dict[self.weight_name]
        if tuple(w.shape[2:]) != self.stride:
            raise ValueError(f"{self.weight_name}: kernel {tuple(w.shape[2:])}, expected {self.stride}")
        w2 = w.reshape(w.shape[0], -1).to(torch.bfloat16)
        k = w2.shape[1]
        kp = (k + 63) // 64 * 64  # the GEMM's K granule: zero columns beyond the patch (Wan 36 channels -> 144 -> 192; HunyuanVideo 65 -> 128)
        self.weight_nk = torch.nn.functional.pad(w2, (0, kp - k)).contiguous()
        self.bias = None if self.bias_name is None else weight_dict[self.bias_name].to(torch.float32).contiguous()

    def _calculate_size(self):
        return self.weight_nk.numel() * 2 + (0 if self.bias is None else self.bias.numel() * 4)

    def to_cuda(self, non_blocking=False):
        self.weight_nk = self.weight_nk.cuda(non_blocking=non_blocking)
        if self.bias is not None:
            self.bias = self.bias.cuda(non_blocking=non_blocking)

    def to_cpu(self, non_blocking=False):
        self.weight_nk = self.weight_nk.to("cpu", non_blocking=non_blocking)
        if self.bias is not None:
            self.bias = self.bias.to("cpu", non_blocking=non_blocking)

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        destination[self.weight_name] = self.weight_nk
        if self.bias_name is not None:
            destination[self.bias_name] = self.bias
        return destination

    def apply(self, latents, round_latents=False):
        if self.stride == (1, 1, 1):
            if round_latents:
                raise ValueError("round_latents belongs to the (1,2,2) patch gather")
            return hip.gemm(hip.channels_to_rows(latents, pad_to=self.weight_nk.shape[1]), self.weight_nk, self.bias)
        return hip.gemm(hip.patchify(latents, round_latents=round_latents, pad_to=self.weight_nk.shape[1]), self.weight_nk, self.bias)


class WanPreWeights(WeightModule):
    def __init__(self, config):
        super().__init__()
        config = with_defaults(config)
        if config["task"] not in ("t2v", "i2v"):
            raise NotImplementedError(f"task={config['task']!r}: t2v and i2v (with or without a CLIP image encoder) are built; flf2v / animate / s2v are not")
        self.in_dim, self.dim, self.patch_size, self.config = config["in_dim"], config["dim"], PATCH_SIZE, config
        mm = MM_WEIGHT_REGISTER[config["dit_quant_scheme"]]
        self.add_module("patch_embedding", CONV3D_WEIGHT_REGISTER["hip-patch"]("patch_embedding.weight", "patch_embedding.bias", stride=PATCH_SIZE))
        self.add_module("text_embedding_0", mm("text_embedding.0.weight", "text_embedding.0.bias"))
        self.add_module("text_embedding_2", mm("text_embedding.2.weight", "text_embedding.2.bias"))
        self.add_module("time_embedding_0", mm("time_embedding.0.weight", "time_embedding.0.bias"))
        self.add_module("time_embedding_2", mm("time_embedding.2.weight", "time_embedding.2.bias"))
        self.add_module("time_projection_1", mm("time_projection.1.weight", "time_projection.1.bias"))
        if config.get("model_cls") == "wan2.1_distill" and config.get("enable_dynamic_cfg", False):
            # the distilled guidance embedding (pre_weights.py:60-69)
            bf = MM_WEIGHT_REGISTER["hip-bf16"]
            self.add_module("cfg_cond_proj_1", bf("guidance_embedding.linear_1.weight", "guidance_embedding.linear_1.bias"))
            self.add_module("cfg_cond_proj_2", bf("guidance_embedding.linear_2.weight", "guidance_embedding.linear_2.bias"))
        if config["task"] == "i2v" and config.get("use_image_encoder", True):
            # CLIP image tokens -> LN, Linear, GELU, Linear, LN (pre_weights.py:43-59); never quantised there
            ln, bf = LN_WEIGHT_REGISTER["hip"], MM_WEIGHT_REGISTER["hip-bf16"]
            self.add_module("proj_0", ln("img_emb.proj.0.weight", "img_emb.proj.0.bias"))
            self.add_module("proj_1", bf("img_emb.proj.1.weight", "img_emb.proj.1.bias"))
            self.add_module("proj_3", bf("img_emb.proj.3.weight", "img_emb.proj.3.bias"))
            self.add_module("proj_4", ln("img_emb.proj.4.weight", "img_emb.proj.4.bias"))


def guidance_scale_embedding(cfg_scale: float, embedding_dim: int = 256, cfg_range=(1.0, 6.0), target_range: float = 1000.0) -> torch.Tensor:
    """The constant [1, 256] input of the distilled guidance embedding (lightx2v/models/networks/wan/infer/utils.py:218-242):
    the guide scale rounded, clamped to `cfg_range`, mapped to [0, target_range], embedded as sin || cos.
    One vector per model configuration, formed on the host like the RoPE table."""
    lo, hi = cfg_range
    w = torch.clamp(torch.round(torch.tensor([float(cfg_scale)], dtype=torch.float32)), min=lo, max=hi)
    w = (w - lo) / (hi - lo) * target_range
    half = embedding_dim // 2
    step = torch.log(torch.tensor(10000.0)) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -step)
    ang = w[:, None] * freqs[None, :]
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)


class GridOutput:
    """module_io.py:7-10."""

    def __init__(self, tensor, tuple):  # noqa: A002 (the reference's field name)
        self.tensor, self.tuple = tensor, tuple


class WanPreInfer:
    def __init__(self, config):
        assert (config["dim"] % config["num_heads"]) == 0 and (config["dim"] // config["num_heads"]) % 2 == 0
        self.config = config
        self.task = config.get("task", "t2v")
        self.freq_dim = config["freq_dim"]
        self.dim = config["dim"]
        self.enable_dynamic_cfg = config.get("enable_dynamic_cfg", False)
        self.cfg_scale = config.get("cfg_scale", 4.0)
        self._cfg_embed_in: Optional[torch.Tensor] = None
        self.scheduler = None

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    @torch.no_grad()
    def infer(self, weights, inputs, kv_start=0, kv_end=0):
        sch = self.scheduler
        latents = sch.latents
        key = "context" if sch.infer_condition else "context_null"
        context = inputs["text_encoder_output"][key]
        c, f, h, w = latents.shape
        # the first forward of a step also rounds the stored latents through bf16 (see step_pre)
        round_now, sch.round_pending = getattr(sch, "round_pending", False), False
        if self.task == "i2v":
            # mask + VAE latents of the conditioning image ride along as extra input channels
            # (pre_infer.py:37-53; in_dim = 16 + 20 for Wan2.2-A14B i2v)
            y = inputs["image_encoder_output"]["vae_encoder_out"]
        else:
            y = None
        if y is not None:
            if round_now:
                hip.patchify(latents, round_latents=True)
            x = weights.patch_embedding.apply(torch.cat([latents, y.to(latents.dtype)], dim=0))
        else:
            # (also model_cls "wan2.2" i2v: the runner clears `vae_encoder_out` after scheduler.prepare,
            # default_runner.py:282-285 -- the image lives on the first latent frame instead)
            x = weights.patch_embedding.apply(latents, round_latents=round_now)
        grid = (f // PATCH_SIZE[0], h // PATCH_SIZE[1], w // PATCH_SIZE[2])
        # One embedding per DISTINCT timestep.  A per-token timestep vector (Wan2.2 ti2v: 0 on the
        # conditioning frame's tokens, t elsewhere; scheduler.py:388-392) has few distinct values: the
        # MLPs run on those, the rows are then dealt to the tokens.
        t_all = sch.timestep_input.flatten().to("cpu")
        t_unique, inverse = (t_all, None) if t_all.numel() == 1 else torch.unique(t_all, return_inverse=True)
        embed = torch.cat([hip.timestep_embedding(float(t), self.freq_dim, latents.device) for t in t_unique], dim=0)
        if self.enable_dynamic_cfg:
            # pre_infer.py:67-73: the guide scale, embedded and projected, is added to the sinusoidal embedding
            if self._cfg_embed_in is None or self._cfg_embed_in.device != latents.device:
                self._cfg_embed_in = guidance_scale_embedding(self.cfg_scale).to(torch.bfloat16).to(latents.device)
            ce = weights.cfg_cond_proj_2.apply(hip.silu(weights.cfg_cond_proj_1.apply(self._cfg_embed_in)))
            embed = hip.add_rowvec(embed, ce.reshape(-1).contiguous())
        embed = weights.time_embedding_0.apply(embed)
        embed = hip.silu(embed)
        embed = weights.time_embedding_2.apply(embed)
        embed0 = hip.silu(embed)
        embed0 = weights.time_projection_1.apply(embed0).unflatten(1, (6, self.dim))
        if inverse is not None:
            if inverse.numel() != x.shape[0]:
                raise ValueError(f"per-token timestep: {inverse.numel()} values for {x.shape[0]} tokens")
            inverse = inverse.to(latents.device)
            embed, embed0 = embed.index_select(0, inverse), embed0.index_select(0, inverse)  # [S, D], [S, 6, D]
        ctx = context.squeeze(0) if context.dim() == 3 else context
        if ctx.dtype != torch.bfloat16:
            ctx = ctx.to(torch.bfloat16)
        out = weights.text_embedding_0.apply_gelu(ctx.contiguous())
        ctx = weights.text_embedding_2.apply(out)
        if self.task == "i2v" and self.config.get("use_image_encoder", True):
            # 257 CLIP tokens through img_emb, in front of the text tokens (pre_infer.py:99-113)
            clip_fea = inputs["image_encoder_output"]["clip_encoder_out"]
            clip_fea = (clip_fea[0] if clip_fea.dim() == 3 else clip_fea).to(torch.bfloat16).contiguous()
            h = hip.gelu_erf(weights.proj_1.apply(weights.proj_0.apply(clip_fea)))
            ctx = torch.cat([weights.proj_4.apply(weights.proj_3.apply(h)), ctx], dim=0)
        grid_sizes = GridOutput(tensor=torch.tensor([list(grid)], dtype=torch.int32, device=x.device), tuple=grid)
        return WanPreInferModuleOutput(embed=embed, grid_sizes=grid_sizes, x=x, embed0=embed0.squeeze(0) if inverse is None else embed0, context=ctx,
                                       adapter_args={"motion_vec": None})


class WanPostInfer:
    def __init__(self, config):
        self.out_dim = config["out_dim"]
        self.patch_size = PATCH_SIZE
        self.scheduler = None

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    @torch.no_grad()
    def infer(self, x, pre_infer_out):
        return [self.unpatchify(x, pre_infer_out.grid_sizes.tuple)]

    def unpatchify(self, x, grid_sizes):
        f, h, w = grid_sizes
        return hip.unpatchify(x, self.out_dim, f * PATCH_SIZE[0], h * PATCH_SIZE[1], w * PATCH_SIZE[2])


# ------------------------------------------------------------------------------- scheduler
class WanScheduler:
    """UniPC multistep (order 2, bh2) on flow-matching sigmas with `sample_shift`."""

    def __init__(self, config):
        self.config = config
        self.device = torch.device(config.get("run_device", "cuda"))
        self.infer_steps = config["infer_steps"]
        self.sample_shift = config["sample_shift"]
        self.sample_guide_scale = config["sample_guide_scale"]
        self.patch_size = PATCH_SIZE
        self.shift = 1
        self.num_train_timesteps = 1000
        self.disable_corrector: List[int] = []
        self.solver_order = 2
        self.head_size = config["dim"] // config["num_heads"]
        # The reference casts its stored latents to the compute dtype at every step_pre unless
        # keep_latents_dtype_in_scheduler (BaseScheduler, scheduler.py:13-19); same switch here.
        self.keep_latents_dtype_in_scheduler = bool(config.get("keep_latents_dtype_in_scheduler", False))
        self.latents: Optional[torch.Tensor] = None
        self.noise_pred: Optional[torch.Tensor] = None
        self.infer_condition = True
        self.step_index = 0
        # which forwards ran their blocks (feature caching; BaseScheduler scheduler.py:10, WanScheduler :30)
        self.caching_records = [True] * self.infer_steps
        self.caching_records_2 = [True] * self.infer_steps
        self.transformer_infer = None
        self.cos_sin = None
        self.seq_p_group = config.get("seq_p_group") if config.get("seq_parallel") else None
        # model_cls "wan2.2" (the dense ti2v model) with an image: the image's VAE latents sit on the
        # first latent frame of the sample itself (scheduler.py:50-51,127-129,391-392,433-434)
        self.image_in_latents = config.get("model_cls") == "wan2.2" and config.get("task") in ("i2v", "s2v")
        self.vae_encoder_out: Optional[torch.Tensor] = None

    def clear(self):
        """Drop cached features between clips (schedulers/wan/feature_caching/scheduler.py:8-9)."""
        if self.transformer_infer is not None and hasattr(self.transformer_infer, "clear"):
            self.transformer_infer.clear()
        self.caching_records = [True] * self.infer_steps
        self.caching_records_2 = [True] * self.infer_steps

    # scheduler.py:49-76
    def prepare(self, seed, latent_shape, image_encoder_output=None, latents: Optional[torch.Tensor] = None):
        if self.image_in_latents:
            self.vae_encoder_out = image_encoder_output["vae_encoder_out"].to(device=self.device, dtype=torch.float32)
        if latents is not None:
            self.latents = latents.to(device=self.device, dtype=torch.float32).contiguous().clone()
            self._keep_image_frame()
        else:
            self.prepare_latents(seed, latent_shape)
        alphas = np.linspace(1, 1 / self.num_train_timesteps, self.num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigmas = self.shift * sigmas / (1 + (self.shift - 1) * sigmas)
        self.sigma_min, self.sigma_max = sigmas[-1].item(), sigmas[0].item()
        self.set_timesteps(self.infer_steps, shift=self.sample_shift)
        grid = (latent_shape[1] // PATCH_SIZE[0], latent_shape[2] // PATCH_SIZE[1], latent_shape[3] // PATCH_SIZE[2])
        self.cos_sin = self.prepare_cos_sin(grid)

    def prepare_latents(self, seed, latent_shape, dtype=torch.float32):
        self.generator = torch.Generator(device=self.device).manual_seed(seed)
        self.latents = torch.randn(*latent_shape, dtype=dtype, device=self.device, generator=self.generator)
        self._keep_image_frame()

    def _keep_image_frame(self):
        """`(1 - mask) * vae_encoder_out + mask * latents` with the reference's mask (ones, zeros on the
        first latent frame; lightx2v/utils/utils.py:451-461): a copy of that one frame, the same values."""
        if self.image_in_latents:
            self.latents[:, :1].copy_(self.vae_encoder_out[:, :1])

    def prepare_cos_sin(self, grid_sizes):
        """fp32 [S_local, head_dim] cos || sin table (scheduler.py:77-114), sharded under Ulysses."""
        table = build_rope_table(grid_sizes, self.head_size, device=self.device)
        if self.seq_p_group is not None:
            table = shard_rope_table(table, comm.group_size(self.seq_p_group), comm.group_rank(self.seq_p_group))
        return table

    # scheduler.py:131-167
    def set_timesteps(self, infer_steps, shift=None):
        sigmas = np.linspace(self.sigma_max, self.sigma_min, infer_steps + 1).copy()[:-1]
        shift = self.shift if shift is None else shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.num_train_timesteps
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.int64)
        assert len(self.timesteps) == self.infer_steps
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * self.solver_order
        self.timestep_list: List[Any] = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = None

    def step_pre(self, step_index):
        self.step_index = step_index
        self.timestep_input = torch.stack([self.timesteps[self.step_index]])
        if self.image_in_latents:
            # one timestep per token of the (1, 2, 2) patch grid: 0 on the image frame's tokens (scheduler.py:391-392)
            _, f, h, w = self.latents.shape
            per_frame = ((h + 1) // 2) * ((w + 1) // 2)
            t = self.timestep_input.to(torch.float32).cpu()
            self.timestep_input = torch.cat([torch.zeros(per_frame), t.expand((f - 1) * per_frame)])
        if not self.keep_latents_dtype_in_scheduler:
            # latents <- bf16(latents): done by the patch gather of the first forward of this step
            self.round_pending = True

    @staticmethod
    def _lam(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _coefficients(self, order, corrector):
        """Scalars of multistep_uni_{c,p}_bh_update (scheduler.py:281-386 / 200-279):
        x_t = a*x - b*m0 - c*res.  Returns a, b, c, rk (order 2) and rho (corrector)."""
        i = self.step_index
        sigma_t, sigma_s0 = (self.sigmas[i], self.sigmas[i - 1]) if corrector else (self.sigmas[i + 1], self.sigmas[i])
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        rk = None
        if order == 2:
            si = i - 2 if corrector else i - 1
            rk = (self._lam(self.sigmas[si]) - self._lam(sigma_s0)) / h
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        rho = None
        if corrector:
            if order == 1:
                rho = torch.tensor([0.5])
            else:
                rks = torch.tensor([rk, 1.0])
                rows, b = [], []
                h_phi_k = h_phi_1 / hh - 1
                fact = 1
                for j in range(1, order + 1):
                    rows.append(torch.pow(rks, j - 1))
                    b.append(h_phi_k * fact / B_h)
                    fact *= j + 1
                    h_phi_k = h_phi_k / hh - 1 / fact
                rho = torch.linalg.solve(torch.stack(rows), torch.tensor(b))
        return float(sigma_t / sigma_s0), float(alpha_t * h_phi_1), float(alpha_t * B_h), (None if rk is None else float(rk)), rho

    # scheduler.py:394-434
    def step_post(self):
        model_output = self.noise_pred
        sample = self.latents
        i = self.step_index
        use_corrector = i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None
        x0 = hip.lincomb([(1.0, sample), (-float(self.sigmas[i]), model_output)])  # convert_model_output
        if use_corrector:
            a, b, c, rk, rho = self._coefficients(self.this_order, corrector=True)
            m0 = self.model_outputs[-1]
            r_last = float(rho[-1])
            terms = [(a, self.last_sample), (-b + c * r_last, m0), (-c * r_last, x0)]
            if self.this_order == 2:
                r0 = float(rho[0]) / rk
                terms[1] = (terms[1][0] + c * r0, m0)
                terms.append((-c * r0, self.model_outputs[-2]))
            sample = hip.lincomb(terms)
        for j in range(self.solver_order - 1):
            self.model_outputs[j] = self.model_outputs[j + 1]
            self.timestep_list[j] = self.timestep_list[j + 1]
        self.model_outputs[-1] = x0
        self.timestep_list[-1] = self.timesteps[i]
        this_order = min(self.solver_order, len(self.timesteps) - i)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        a, b, c, rk, _ = self._coefficients(self.this_order, corrector=False)
        terms = [(a, sample), (-b, x0)]
        if self.this_order == 2:
            terms[1] = (-b + c * 0.5 / rk, x0)
            terms.append((-c * 0.5 / rk, self.model_outputs[-2]))
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.latents = hip.lincomb(terms)
        self._keep_image_frame()


class WanStepDistillScheduler(WanScheduler):
    """The step-distilled schedule (lightx2v/models/schedulers/wan/step_distill/scheduler.py:9-50): the
    shifted sigma table of 1000 training steps sampled at `denoising_step_list`, Euler-type update
    x <- x - sigma_i f + sigma_{i+1} f.  Used without CFG (configs/distill/*.json)."""

    def __init__(self, config):
        config = dict(config, infer_steps=len(config["denoising_step_list"]))
        super().__init__(config)
        self.denoising_step_list = list(config["denoising_step_list"])
        self.sigma_max, self.sigma_min = 1.0, 0.0

    def prepare(self, seed, latent_shape, image_encoder_output=None, latents: Optional[torch.Tensor] = None):
        if latents is not None:
            self.latents = latents.to(device=self.device, dtype=torch.float32).contiguous().clone()
        else:
            self.prepare_latents(seed, latent_shape)
        self.set_denoising_timesteps()
        self.cos_sin = self.prepare_cos_sin((latent_shape[1] // PATCH_SIZE[0], latent_shape[2] // PATCH_SIZE[1], latent_shape[3] // PATCH_SIZE[2]))

    def set_denoising_timesteps(self):
        sigmas = torch.linspace(self.sigma_max, self.sigma_min, self.num_train_timesteps + 1)[:-1]
        sigmas = self.sample_shift * sigmas / (1 + (self.sample_shift - 1) * sigmas)
        timesteps = sigmas * self.num_train_timesteps
        index = [self.num_train_timesteps - x for x in self.denoising_step_list]
        self.timesteps = timesteps[index]   # fp32, host side (the pre-infer reads it there)
        self.sigmas = sigmas[index]

    def reset(self, seed, latent_shape, step_index=None):
        self.prepare_latents(seed, latent_shape)

    def add_noise(self, original_samples, noise, sigma):
        return hip.lincomb([(1.0 - float(sigma), original_samples), (float(sigma), noise)])

    def step_post(self):
        i = self.step_index
        terms = [(1.0, self.latents), (-self.sigmas[i].item(), self.noise_pred)]
        if i < self.infer_steps - 1:
            terms.append((self.sigmas[i + 1].item(), self.noise_pred))
        self.latents = hip.lincomb(terms)


class Wan22StepDistillScheduler(WanStepDistillScheduler):
    """step_distill/scheduler.py:52-76: the same update; keeps the boundary step and its sigma."""

    def __init__(self, config):
        super().__init__(config)
        self.boundary_step_index = config["boundary_step_index"]

    def set_denoising_timesteps(self):
        super().set_denoising_timesteps()
        self.sigma_bound = self.sigmas[self.boundary_step_index].item()


# ------------------------------------------------------------------------------- model
class WanModel:
    """pre -> blocks -> head -> unpatchify, with CFG and Ulysses handling (model.py:395-499)."""

    def __init__(self, config, weight_dict: Optional[Dict[str, torch.Tensor]] = None):
        self.config = with_defaults(config)
        cfg = self.config
        self.pre_weight = WanPreWeights(cfg)
        self.transformer_weights = WanTransformerWeights(cfg)
        self.pre_infer = WanPreInfer(cfg)
        caching = cfg.get("feature_caching", "NoCaching")
        if caching == "NoCaching":
            self.transformer_infer = WanTransformerInfer(cfg)
        elif caching == "Tea":
            from .wan_caching import WanTransformerInferTeaCaching

            self.transformer_infer = WanTransformerInferTeaCaching(cfg)
        elif caching == "Mag":
            from .wan_caching import WanTransformerInferMagCaching

            self.transformer_infer = WanTransformerInferMagCaching(cfg)
        else:
            raise NotImplementedError(f"feature_caching={caching!r}: built are 'NoCaching', 'Tea' and 'Mag'")
        self.post_infer = WanPostInfer(cfg)
        self.seq_p_group = cfg.get("seq_p_group") if cfg.get("seq_parallel") else None
        self.cfg_p_group = cfg.get("cfg_p_group") if cfg.get("cfg_parallel") else None
        self.scheduler = None
        if weight_dict is not None:
            self.pre_weight.load(weight_dict)
            self.transformer_weights.load(weight_dict)

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler
        for o in (self.pre_infer, self.transformer_infer, self.post_infer):
            o.set_scheduler(scheduler)

    def to_cuda(self):
        self.pre_weight.to_cuda()
        self.transformer_weights.to_cuda()

    @torch.no_grad()
    def infer(self, inputs):
        sch = self.scheduler
        if self.config.get("enable_cfg", True):
            if self.cfg_p_group is not None:
                assert comm.group_size(self.cfg_p_group) == 2, "cfg_p_world_size must be equal to 2"
                mine = self._infer_cond_uncond(inputs, infer_condition=comm.group_rank(self.cfg_p_group) == 0)
                parts = [torch.empty_like(mine) for _ in range(2)]
                comm.transport().all_gather(parts, mine.contiguous(), group=self.cfg_p_group)
                cond, uncond = parts
            else:
                cond = self._infer_cond_uncond(inputs, infer_condition=True)
                uncond = self._infer_cond_uncond(inputs, infer_condition=False)
            sch.noise_pred = hip.cfg_combine(cond, uncond, sch.sample_guide_scale)
        else:
            sch.noise_pred = self._infer_cond_uncond(inputs, infer_condition=True)

    @torch.no_grad()
    def _infer_cond_uncond(self, inputs, infer_condition=True):
        sch = self.scheduler
        sch.infer_condition = infer_condition
        pre_infer_out = self.pre_infer.infer(self.pre_weight, inputs)
        if self.seq_p_group is not None:
            pre_infer_out = self._seq_parallel_pre_process(pre_infer_out)
        x = self.transformer_infer.infer(self.transformer_weights, pre_infer_out)
        if self.seq_p_group is not None:
            x = self._seq_parallel_post_process(x)
        return self.post_infer.infer(x, pre_infer_out)[0]

    def _seq_parallel_pre_process(self, pre_infer_out):
        x = pre_infer_out.x
        world, rank = comm.group_size(self.seq_p_group), comm.group_rank(self.seq_p_group)
        pad = (world - (x.shape[0] % world)) % world
        if pad > 0:
            x = torch.nn.functional.pad(x, (0, 0, 0, pad))
        pre_infer_out.x = torch.chunk(x, world, dim=0)[rank].contiguous()
        # A per-token timestep (Wan2.2 dense ti2v with an image) makes the modulation inputs per-token
        # too: embed [S, D] and embed0 [S, 6, D] are padded and chunked like the tokens they belong to
        # (model.py:480-489 does it for model_cls wan2.2 / wan2.2_audio with task i2v / s2v, the cases
        # that produce them; keyed on the shape here so that every producer of per-token rows is covered).
        embed, embed0 = pre_infer_out.embed, pre_infer_out.embed0
        if embed0 is not None and embed0.dim() == 3 and embed0.shape[1] == 6:
            pad = (world - (embed0.shape[0] % world)) % world
            if pad > 0:
                embed0 = torch.nn.functional.pad(embed0, (0, 0, 0, 0, 0, pad))
            pre_infer_out.embed0 = torch.chunk(embed0, world, dim=0)[rank].contiguous()
            if embed is not None and embed.dim() == 2 and embed.shape[0] > 1:
                if pad > 0:
                    embed = torch.nn.functional.pad(embed, (0, 0, 0, pad))
                pre_infer_out.embed = torch.chunk(embed, world, dim=0)[rank].contiguous()
        return pre_infer_out

    def _seq_parallel_post_process(self, x):
        world = comm.group_size(self.seq_p_group)
        gathered = [torch.empty_like(x) for _ in range(world)]
        comm.transport().all_gather(gathered, x.contiguous(), group=self.seq_p_group)
        return torch.cat(gathered, dim=0)


class MultiModelStruct:
    """Two experts sharing one scheduler (Wan2.2-A14B): the high-noise model runs while
    timesteps[step_index] >= boundary * num_train_timesteps, the low-noise model after; each has its
    own guide scale (`sample_guide_scale` is a pair).  Mirrors lightx2v/models/runners/wan/wan_runner.py:394-444
    without the offload branches: both weight sets (2 x 28 GB in bf16) stay resident in HBM."""

    def __init__(self, model_list, config, boundary=0.875, num_train_timesteps=1000):
        self.model = model_list  # [high_noise_model, low_noise_model]
        assert len(self.model) == 2, "MultiModelStruct only supports 2 models now."
        self.config = config
        self.boundary = boundary
        self.boundary_timestep = self.boundary * num_train_timesteps
        self.cur_model_index = -1
        self.scheduler = None

    def set_scheduler(self, shared_scheduler):
        self.scheduler = shared_scheduler
        for model in self.model:
            model.set_scheduler(shared_scheduler)

    def to_cuda(self):
        for model in self.model:
            model.to_cuda()

    def infer(self, inputs):
        self.get_current_model_index()
        self.model[self.cur_model_index].infer(inputs)

    def get_current_model_index(self):
        high = float(self.scheduler.timesteps[self.scheduler.step_index]) >= self.boundary_timestep
        self.cur_model_index = 0 if high else 1
        self.scheduler.sample_guide_scale = self.config["sample_guide_scale"][self.cur_model_index]


class MultiDistillModelStruct(MultiModelStruct):
    """The two-expert switch of the Wan2.2 step-distilled model (lightx2v/models/runners/wan/wan_distill_runner.py:44-75):
    the high-noise expert runs while step_index < boundary_step_index; the guide scale is left alone."""

    def __init__(self, model_list, config, boundary_step_index=2):
        super().__init__(model_list, config)
        self.boundary_step_index = boundary_step_index

    def get_current_model_index(self):
        self.cur_model_index = 0 if self.scheduler.step_index < self.boundary_step_index else 1


def run_denoise_loop(model, inputs, infer_steps: Optional[int] = None):
    """default_runner.py:137-169: step_pre -> model.infer -> step_post for every step."""
    sch = model.scheduler
    for step_index in range(sch.infer_steps if infer_steps is None else infer_steps):
        sch.step_pre(step_index=step_index)
        model.infer(inputs)
        sch.step_post()
    return sch.latents
