"""Denoise-step glue around the HunyuanVideo-1.5 double-stream blocks, on the HIP kernels:
pre-infer (patch embedding with patch (1,1,1) as a GEMM, timestep MLP, the text branch with its
two-block token refiner, condition-type embedding, token reorder), post-infer (unpatchify), the
Euler flow scheduler and its 4-step distilled variant, and the model-level `infer` with
classifier-free guidance and Ulysses sharding.

Mirrors, class for class and method for method (t2v / i2v, no super-resolution branch):
  HunyuanVideo15PreWeights             lightx2v/models/networks/hunyuan_video/weights/pre_weights.py:10-147
  HunyuanVideo15PreInfer               lightx2v/models/networks/hunyuan_video/infer/pre_infer.py:64-240
  HunyuanVideo15PostInfer              lightx2v/models/networks/hunyuan_video/infer/post_infer.py:6-39
  HunyuanVideo15Scheduler              lightx2v/models/schedulers/hunyuan_video/scheduler.py:11-142
  HunyuanVideo15StepDistillScheduler   lightx2v/models/schedulers/hunyuan_video/step_distill/scheduler.py:6-33
  HunyuanVideo15Model.infer            lightx2v/models/networks/hunyuan_video/model.py:199-278
Every arithmetic operation is a kernel of liblx2v_hip.so; torch is used for memory, row selection
and concatenation (the token reorder moves rows, it computes nothing), streams and collectives.
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist

from . import comm, hip
from .hy_block import HunyuanVideo15InferModuleOutput, HunyuanVideo15TransformerInfer, HunyuanVideo15TransformerWeights, with_defaults
from .registry_factory import CONV3D_WEIGHT_REGISTER, EMBEDDING_WEIGHT_REGISTER, LN_WEIGHT_REGISTER, MM_WEIGHT_REGISTER
from .rope_table import build_hunyuan_rope_table
from .weight_module import WeightModule, WeightModuleList
from . import wan_model  # noqa: F401  (registers the conv3d plugin "hip-patch")

PATCH_SIZE = (1, 1, 1)
REFINER_PREFIX = "txt_in.individual_token_refiner.blocks"


# ------------------------------------------------------------------------------- embedding plugin
@EMBEDDING_WEIGHT_REGISTER("hip")
class HipEmbeddingWeight:
    """Lookup table [entries, D] in bf16.  Constructor and `apply(indices)` as the reference's
    EmbeddingWeight (lightx2v/common/ops/embedding/embedding_weight.py); `row(i)` hands one entry
    to the row-broadcast add that the pre-infer needs."""

    def __init__(self, weight_name, create_cuda_buffer=False, lazy_load=False, lazy_load_file=None, is_post_adapter=False):
        self.weight_name = weight_name
        self.config: Dict[str, Any] = {}
        self.weight: Optional[torch.Tensor] = None

    def set_config(self, config=None):
        if config is not None:
            self.config = config

    def load(self, weight_dict):
        self.weight = weight_dict[self.weight_name].to(torch.bfloat16).contiguous()

    def _calculate_size(self):
        return self.weight.numel() * 2

    def to_cuda(self, non_blocking=False):
        self.weight = self.weight.cuda(non_blocking=non_blocking)

    def to_cpu(self, non_blocking=False):
        self.weight = self.weight.to("cpu", non_blocking=non_blocking)

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        destination[self.weight_name] = self.weight
        return destination

    def row(self, i: int) -> torch.Tensor:
        return self.weight[i]

    def apply(self, input_indices):
        return self.weight.index_select(0, input_indices.reshape(-1)).view(*input_indices.shape, -1)


# ------------------------------------------------------------------------------- weights
class IndividualTokenRefinerBlock(WeightModule):
    """pre_weights.py:115-147.  The refiner is never quantised there (mm-type "Default"); bf16 here."""

    def __init__(self, block_idx, mm_type, config, block_prefix):
        super().__init__()
        self.config, self.mm_type = config, mm_type
        mm, ln = MM_WEIGHT_REGISTER["hip-bf16"], LN_WEIGHT_REGISTER[config["ln_type"]]
        p = f"{block_prefix}.{block_idx}."
        self.add_module("norm1", ln(p + "norm1.weight", p + "norm1.bias"))
        self.add_module("self_attn_qkv", mm(p + "self_attn_qkv.weight", p + "self_attn_qkv.bias"))
        self.add_module("self_attn_proj", mm(p + "self_attn_proj.weight", p + "self_attn_proj.bias"))
        self.add_module("norm2", ln(p + "norm2.weight", p + "norm2.bias"))
        self.add_module("mlp_fc1", mm(p + "mlp.fc1.weight", p + "mlp.fc1.bias"))
        self.add_module("mlp_fc2", mm(p + "mlp.fc2.weight", p + "mlp.fc2.bias"))
        self.add_module("adaLN_modulation", mm(p + "adaLN_modulation.1.weight", p + "adaLN_modulation.1.bias"))


class HunyuanVideo15PreWeights(WeightModule):
    def __init__(self, config):
        super().__init__()
        config = with_defaults(config)
        self.config = config
        self.mm_type = config["dit_quant_scheme"]
        self.patch_size = tuple(config.get("patch_size", PATCH_SIZE))
        if config.get("is_sr_running", False):
            raise NotImplementedError("is_sr_running (super-resolution pass: time_r_in, condition channels) is not built")
        mm = MM_WEIGHT_REGISTER["hip-bf16"]
        self.add_module("img_in", CONV3D_WEIGHT_REGISTER["hip-patch"]("img_in.proj.weight", "img_in.proj.bias", stride=self.patch_size))
        for attr, key in (("time_in_0", "time_in.mlp.0"), ("time_in_2", "time_in.mlp.2"), ("txt_in_t_embedder_0", "txt_in.t_embedder.mlp.0"),
                          ("txt_in_t_embedder_2", "txt_in.t_embedder.mlp.2"), ("txt_in_c_embedder_0", "txt_in.c_embedder.linear_1"),
                          ("txt_in_c_embedder_2", "txt_in.c_embedder.linear_2"), ("txt_in_input_embedder", "txt_in.input_embedder")):
            self.add_module(attr, mm(key + ".weight", key + ".bias"))
        self.add_module("individual_token_refiner", WeightModuleList([IndividualTokenRefinerBlock(i, self.mm_type, config, REFINER_PREFIX) for i in range(2)]))
        self.add_module("cond_type_embedding", EMBEDDING_WEIGHT_REGISTER["hip"]("cond_type_embedding.weight"))


# ------------------------------------------------------------------------------- pre-infer
def _bf16_rows(t: torch.Tensor) -> torch.Tensor:
    """[1, L, D] or [L, D], any float dtype -> contiguous bf16 [L, D]."""
    t = t[0] if t.dim() == 3 else t
    return t.to(torch.bfloat16).contiguous()


class _MaskCache:
    """Host copies of the (per-clip constant) token masks: index lists for row selection."""

    def __init__(self):
        self._items: Dict[Any, Any] = {}

    def valid_rows(self, mask: torch.Tensor, force_first: bool = False):
        """(device index list of the set entries, their number, whether they are rows 0..n-1).
        `force_first`: row 0 counts as set whatever the mask says."""
        key = (mask.data_ptr(), tuple(mask.shape), mask._version, force_first)
        hit = self._items.get(key)
        if hit is None:
            m = mask.reshape(-1).to("cpu").bool()
            if force_first:
                m[0] = True
            idx = torch.nonzero(m).squeeze(1)
            prefix = bool(idx.numel() == 0 or int(idx[-1]) + 1 == idx.numel())
            hit = (idx.to(mask.device), int(idx.numel()), prefix, mask)  # holds `mask`: the key is its address
            if len(self._items) > 64:
                self._items.clear()
            self._items[key] = hit
        return hit[:3]


class HunyuanVideo15PreInfer:
    def __init__(self, config):
        self.config = config
        self.patch_size = tuple(config.get("patch_size", PATCH_SIZE))
        self.heads_num = config["heads_num"]
        self.frequency_embedding_size = 256
        self.max_period = 10000
        if self.patch_size != PATCH_SIZE:
            raise NotImplementedError(f"patch_size={self.patch_size}: HunyuanVideo-1.5 uses (1, 1, 1)")
        if config.get("is_sr_running", False):
            raise NotImplementedError("is_sr_running is not built")
        self.scheduler = None
        self._masks = _MaskCache()

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    @torch.no_grad()
    def infer(self, weights, inputs):
        sch = self.scheduler
        latents = sch.latents
        grid = tuple(latents.shape[2:])
        t = float(sch.timesteps[sch.step_index])
        te = inputs["text_encoder_output"]
        txt, text_mask = te["context"] if sch.infer_condition else te["context_null"]
        byt5_txt, byt5_mask = te["byt5_features"], te["byt5_masks"]
        vis, vis_mask = inputs["image_encoder_output"]["siglip_output"], inputs["image_encoder_output"]["siglip_mask"]
        dev = latents.device

        # image tokens: [latents | conditioning latents | mask] channel planes -> rows -> Linear
        planes = torch.cat([latents[0].float(), sch.cond_latents_concat[0].float(), sch.mask_concat[0].float()], dim=0).contiguous()
        img = weights.img_in.apply(planes)

        # conditioning vector and the text branch's modulation input
        t_freq = hip.timestep_embedding(t, self.frequency_embedding_size, dev)
        vec = weights.time_in_2.apply(hip.silu(weights.time_in_0.apply(t_freq)))
        t_aware = weights.txt_in_t_embedder_2.apply(hip.silu(weights.txt_in_t_embedder_0.apply(t_freq)))
        txt_rows = _bf16_rows(txt)
        c_aware = hip.weighted_mean_rows(txt_rows, text_mask.reshape(-1).to(device=dev, dtype=torch.float32).contiguous())
        c_aware = weights.txt_in_c_embedder_2.apply(hip.silu(weights.txt_in_c_embedder_0.apply(c_aware)))
        c = hip.add_rowvec(t_aware, c_aware[0])
        out = weights.txt_in_input_embedder.apply(txt_rows)
        out = self.run_individual_token_refiner(weights, out, text_mask, c)

        # condition-type embedding, then [image-encoder valid | byT5 valid | text valid] rows
        emb = weights.cond_type_embedding
        parts = []
        for rows, mask, kind in ((_bf16_rows(vis), vis_mask, 2), (_bf16_rows(byt5_txt), byt5_mask, 1), (out, text_mask, 0)):
            idx, n, prefix = self._masks.valid_rows(mask)
            if n == 0:
                continue
            sel = rows[:n] if prefix else rows.index_select(0, idx)
            parts.append(hip.add_rowvec(sel, emb.row(kind)))
        txt_out = torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]
        return HunyuanVideo15InferModuleOutput(img=img.unsqueeze(0), txt=txt_out.unsqueeze(0).contiguous(), vec=hip.silu(vec), grid_sizes=grid)

    def run_individual_token_refiner(self, weights, out, mask, c):
        """pre_infer.py:169-188.  Padded tokens take no part in the attention and receive a zero
        attention output (the reference's unpad -> varlen attention -> pad); token 0 always counts."""
        L, D = out.shape
        H = self.heads_num
        idx, n, prefix = self._masks.valid_rows(mask, force_first=True)  # token 0 always counts (pre_infer.py:170-171)
        silu_c = hip.silu(c)
        for block in weights.individual_token_refiner:
            gates = block.adaLN_modulation.apply(silu_c).float().view(2, D)
            gate_msa, gate_mlp = gates[0].contiguous(), gates[1].contiguous()
            qkv = block.self_attn_qkv.apply(block.norm1.apply(out))
            sel = qkv[:n] if prefix else qkv.index_select(0, idx)
            q, k, v = (sel[:, i * D:(i + 1) * D].unflatten(1, (H, D // H)) for i in range(3))
            a = hip.attn_fwd(q, k, v)
            if n == L and prefix:
                attn = a
            else:
                attn = torch.zeros((L, D), dtype=out.dtype, device=out.device)
                if prefix:
                    attn[:n] = a
                else:
                    attn.index_copy_(0, idx, a)
            block.self_attn_proj.apply_residual_(attn, out, gate_msa)
            h = hip.silu(block.mlp_fc1.apply(block.norm2.apply(out)))
            block.mlp_fc2.apply_residual_(h, out, gate_mlp)
        return out


# ------------------------------------------------------------------------------- post-infer
class HunyuanVideo15PostInfer:
    def __init__(self, config):
        self.config = config
        self.unpatchify_channels = config["out_channels"]
        self.patch_size = tuple(config.get("patch_size", PATCH_SIZE))
        if self.patch_size != PATCH_SIZE:
            raise NotImplementedError(f"patch_size={self.patch_size}: HunyuanVideo-1.5 uses (1, 1, 1)")
        self.scheduler = None

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    @torch.no_grad()
    def infer(self, x, pre_infer_out):
        return self.unpatchify(x, *pre_infer_out.grid_sizes)

    def unpatchify(self, x, t, h, w):
        """[1, S (+ padding), C] bf16 -> fp32 [1, C, t, h, w]."""
        rows = x[0] if x.dim() == 3 else x
        return hip.rows_to_channels(rows, self.unpatchify_channels, t * h * w).view(1, self.unpatchify_channels, t, h, w)


# ------------------------------------------------------------------------------- schedulers
class HunyuanVideo15Scheduler:
    """Euler steps on shifted flow sigmas (scheduler.py:11-142)."""

    def __init__(self, config):
        self.config = config
        self.device = torch.device(config.get("run_device", "cuda"))
        self.infer_steps = config["infer_steps"]
        self.reverse = True
        self.num_train_timesteps = 1000
        self.sample_shift = config["sample_shift"]
        self.sample_guide_scale = config["sample_guide_scale"]
        self.keep_latents_dtype_in_scheduler = True
        self.latents: Optional[torch.Tensor] = None
        self.noise_pred: Optional[torch.Tensor] = None
        self.infer_condition = True
        self.step_index = 0
        self.caching_records = [True] * self.infer_steps
        self.transformer_infer = None
        self.cos_sin = None
        self.seq_p_group = config.get("seq_p_group") if config.get("seq_parallel") else None

    def step_pre(self, step_index):
        self.step_index = step_index  # the latents keep their dtype (BaseScheduler.step_pre with keep_latents_dtype_in_scheduler)

    def clear(self):
        pass

    def prepare(self, seed, latent_shape, image_encoder_output=None, latents: Optional[torch.Tensor] = None):
        """`latent_shape` = (C, T, H, W); `latents` (bf16 [1, C, T, H, W]) replaces the seeded noise."""
        if latents is not None:
            self.latents = latents.to(device=self.device, dtype=torch.bfloat16).contiguous().clone()
        else:
            self.prepare_latents(seed, latent_shape, dtype=torch.bfloat16)
        self.set_timesteps(self.infer_steps, device=self.device, shift=self.sample_shift)
        self.multitask_mask = self.get_task_mask(self.config["task"], latent_shape[-3])
        cond = None if image_encoder_output is None else image_encoder_output.get("cond_latents")
        self.cond_latents_concat, self.mask_concat = self._prepare_cond_latents_and_mask(self.config["task"], cond, self.latents, self.multitask_mask)
        self.cos_sin = self.prepare_cos_sin((latent_shape[1], latent_shape[2], latent_shape[3]))

    def prepare_latents(self, seed, latent_shape, dtype=torch.bfloat16):
        self.generator = torch.Generator(device=self.device).manual_seed(seed)
        self.latents = torch.randn(1, *latent_shape, dtype=dtype, device=self.device, generator=self.generator)

    def set_timesteps(self, num_inference_steps, device, shift):
        sigmas = torch.linspace(1, 0, num_inference_steps + 1)
        if shift != 1.0:
            sigmas = (shift * sigmas) / (1 + (shift - 1) * sigmas)
        if not self.reverse:
            sigmas = 1 - sigmas
        self.sigmas = sigmas  # host: the per-step scalars of the update
        self.timesteps = (sigmas[:-1] * self.num_train_timesteps).to(dtype=torch.float32)

    @staticmethod
    def get_task_mask(task_type, latent_target_length):
        if task_type not in ("t2v", "i2v"):
            raise ValueError(f"{task_type} is not supported !")
        mask = torch.zeros(latent_target_length)
        if task_type == "i2v":
            mask[0] = 1.0
        return mask

    @staticmethod
    def _prepare_cond_latents_and_mask(task_type, cond_latents, latents, multitask_mask):
        """scheduler.py:72-102: first-frame conditioning latents (i2v) or zeros, and a one-channel
        mask that is 1 on the conditioned frames."""
        if cond_latents is not None and task_type == "i2v":
            latents_concat = cond_latents.to(latents.device).repeat(1, 1, latents.shape[2], 1, 1)
            latents_concat[:, :, 1:] = 0.0
        else:
            latents_concat = torch.zeros(latents.shape, device=latents.device)
        mask_concat = torch.zeros(latents.shape[0], 1, *latents.shape[2:], device=latents.device)
        mask_concat[:, :, torch.nonzero(multitask_mask).squeeze(1).to(latents.device)] = 1.0
        return latents_concat, mask_concat

    def prepare_cos_sin(self, rope_sizes):
        head_dim = self.config["hidden_size"] // self.config["heads_num"]
        rope_dim_list = self.config.get("rope_dim_list")
        if rope_dim_list is None:
            rope_dim_list = [head_dim // 3] * 3
        table = build_hunyuan_rope_table(rope_sizes, head_dim, rope_dim_list, self.config.get("rope_theta", 256.0), device=self.device)
        if self.seq_p_group is not None:
            world, rank = comm.group_size(self.seq_p_group), comm.group_rank(self.seq_p_group)
            pad = (world - (table.shape[0] % world)) % world
            if pad > 0:
                table = torch.nn.functional.pad(table, (0, 0, 0, pad))
            table = torch.chunk(table, world, dim=0)[rank].contiguous()
        return table

    def step_post(self):
        dt = float(self.sigmas[self.step_index + 1] - self.sigmas[self.step_index])
        v = self.noise_pred.to(torch.float32).reshape(self.latents.shape)  # the prediction comes without the batch axis
        self.latents = hip.lincomb([(1.0, self.latents.to(torch.float32)), (dt, v)])


class HunyuanVideo15StepDistillScheduler(HunyuanVideo15Scheduler):
    """step_distill/scheduler.py:6-33: x0 = x - sigma v, re-noised along v to the next sigma."""

    def __init__(self, config):
        super().__init__(config)
        self.denoising_step_list = config["denoising_step_list"]
        self.infer_steps = len(self.denoising_step_list)
        self.caching_records = [True] * self.infer_steps
        self.sigma_max, self.sigma_min = 1.0, 0.0

    def set_timesteps(self, num_inference_steps, device, shift):
        sigma_start = self.sigma_min + (self.sigma_max - self.sigma_min)
        sigmas = torch.linspace(sigma_start, self.sigma_min, self.num_train_timesteps + 1)[:-1]
        sigmas = self.sample_shift * sigmas / (1 + (self.sample_shift - 1) * sigmas)
        idx = [self.num_train_timesteps - x for x in self.denoising_step_list]
        self.timesteps = (sigmas * self.num_train_timesteps)[idx]
        self.sigmas = sigmas[idx]

    def step_post(self):
        sigma = self.sigmas[self.step_index].item()
        v = self.noise_pred.to(torch.float32).reshape(self.latents.shape)
        terms = [(1.0, self.latents.to(torch.float32)), (-sigma, v)]
        if self.step_index < self.infer_steps - 1:
            terms.append((self.sigmas[self.step_index + 1].item(), v))
        self.latents = hip.lincomb(terms).to(self.latents.dtype)


# ------------------------------------------------------------------------------- model
class HunyuanVideo15Model:
    """pre -> double-stream blocks -> final layer -> unpatchify, with CFG and Ulysses handling
    (model.py:199-278)."""

    def __init__(self, config, weight_dict: Optional[Dict[str, torch.Tensor]] = None):
        self.config = with_defaults(config)
        cfg = self.config
        self.pre_weight = HunyuanVideo15PreWeights(cfg)
        self.transformer_weights = HunyuanVideo15TransformerWeights(cfg)
        self.pre_infer = HunyuanVideo15PreInfer(cfg)
        self.transformer_infer = HunyuanVideo15TransformerInfer(cfg)
        self.post_infer = HunyuanVideo15PostInfer(cfg)
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
        if self.config.get("enable_cfg", False):
            if self.cfg_p_group is not None:
                assert comm.group_size(self.cfg_p_group) == 2, "cfg_p_world_size must be equal to 2"
                mine = self._infer_cond_uncond(inputs, infer_condition=comm.group_rank(self.cfg_p_group) == 0).contiguous()
                parts = [torch.zeros_like(mine) for _ in range(2)]
                comm.transport().all_gather(parts, mine, group=self.cfg_p_group)
                cond, uncond = parts
            else:
                cond = self._infer_cond_uncond(inputs, infer_condition=True)
                uncond = self._infer_cond_uncond(inputs, infer_condition=False)
            sch.noise_pred = hip.cfg_combine(cond, uncond, sch.sample_guide_scale)
        else:
            sch.noise_pred = self._infer_cond_uncond(inputs, infer_condition=True)

    @torch.no_grad()
    def _infer_cond_uncond(self, inputs, infer_condition=True):
        self.scheduler.infer_condition = infer_condition
        pre_infer_out = self.pre_infer.infer(self.pre_weight, inputs)
        if self.seq_p_group is not None:
            pre_infer_out = self._seq_parallel_pre_process(pre_infer_out)
        x = self.transformer_infer.infer(self.transformer_weights, pre_infer_out)
        if self.seq_p_group is not None:
            x = self._seq_parallel_post_process(x)
        return self.post_infer.infer(x, pre_infer_out)[0]

    def _seq_parallel_pre_process(self, pre_infer_out):
        img = pre_infer_out.img
        world, rank = comm.group_size(self.seq_p_group), comm.group_rank(self.seq_p_group)
        pad = (world - (img.shape[1] % world)) % world
        if pad > 0:
            img = torch.nn.functional.pad(img, (0, 0, 0, pad))
        pre_infer_out.img = torch.chunk(img, world, dim=1)[rank].contiguous()
        return pre_infer_out

    def _seq_parallel_post_process(self, x):
        world = comm.group_size(self.seq_p_group)
        gathered = [torch.empty_like(x) for _ in range(world)]
        comm.transport().all_gather(gathered, x.contiguous(), group=self.seq_p_group)
        return torch.cat(gathered, dim=1)


def run_denoise_loop(model, inputs, infer_steps: Optional[int] = None) -> torch.Tensor:
    """default_runner.py:137-169: step_pre -> model.infer -> step_post for every step."""
    sch = model.scheduler
    for step_index in range(sch.infer_steps if infer_steps is None else infer_steps):
        sch.step_pre(step_index=step_index)
        model.infer(inputs)
        sch.step_post()
    return sch.latents
