"""TeaCache and MagCache around the Wan block loop: skip all blocks of a forward and re-use the
previous forward's residual -- when the timestep modulation barely moved (Tea), or while the
calibrated magnitude ratios of the skipped steps stay close to one (Mag).

Mirrors `WanTransformerInferTeaCaching`
(lightx2v/models/networks/wan/infer/feature_caching/transformer_infer.py:25-169): same config keys
(`feature_caching: "Tea"`, `teacache_thresh`, `coefficients`, `use_ret_steps`), same per-CFG-branch
state, same records on the scheduler (`caching_records`, `caching_records_2`).  The decision is host
control logic on a [6, D] (or [1, D]) vector -- it is read back once per forward, as the reference
does with `.cpu().item()`; the token-sized work (residual capture and re-use over [S, D]) runs on
the HIP element-wise kernel.

`WanTransformerInferMagCaching` mirrors the class of that name (same file, :992-1091): config keys
`feature_caching: "Mag"`, `magcache_thresh`, `magcache_K`, `magcache_retention_ratio`,
`magcache_ratios` ([2, infer_steps]: conditional and unconditional branch) and
`magcache_calibration`; its decision reads nothing from the GPU at all.
"""

from __future__ import annotations

import numpy as np
import torch

from . import hip
from .wan_block import WanTransformerInfer


def _residual(x, ori_x, holder):
    """x - ori_x as a new bf16 [S, D] tensor on the element-wise kernel; `holder._minus_one` keeps the gate."""
    if holder._minus_one is None or holder._minus_one.numel() != x.shape[1] or holder._minus_one.device != x.device:
        holder._minus_one = torch.full((x.shape[1],), -1.0, dtype=torch.float32, device=x.device)
    return hip.gate_residual_(x.clone(), ori_x, holder._minus_one)


def _refuse_fp32_stream(runner):
    if runner.fp32_residual:
        raise NotImplementedError("feature caching keeps bf16 residuals of the block stack; it is not combined with sensitive_layer_dtype='fp32'")


class WanTransformerInferTeaCaching(WanTransformerInfer):
    def __init__(self, config):
        super().__init__(config)
        _refuse_fp32_stream(self)
        cfg = self.config
        self.teacache_thresh = cfg["teacache_thresh"]
        self.use_ret_steps = cfg["use_ret_steps"]
        if self.use_ret_steps:
            self.coefficients, self.ret_steps, self.cutoff_steps = cfg["coefficients"][0], 5, cfg["infer_steps"]
        else:
            self.coefficients, self.ret_steps, self.cutoff_steps = cfg["coefficients"][1], 1, cfg["infer_steps"] - 1
        self.must_calc_steps = cfg["changing_resolution_steps"] if cfg.get("changing_resolution", False) else []
        self.clear()

    def clear(self):
        self.accumulated_rel_l1_distance_even = self.accumulated_rel_l1_distance_odd = 0.0
        self.previous_e0_even = self.previous_e0_odd = None            # host fp32 copies of the modulation input
        self.previous_residual_even = self.previous_residual_odd = None  # bf16 [S, D] on the GPU
        self._minus_one = None

    def must_calc(self, step_index):
        return step_index in self.must_calc_steps

    # ---- decision (transformer_infer.py:46-96) ------------------------------------------------
    @torch.no_grad()
    def calculate_should_calc(self, embed, embed0):
        inp = (embed0 if self.use_ret_steps else embed).detach().float().cpu()
        branch = "even" if self.scheduler.infer_condition else "odd"
        acc, prev = getattr(self, f"accumulated_rel_l1_distance_{branch}"), getattr(self, f"previous_e0_{branch}")
        if self.scheduler.step_index < self.ret_steps or self.scheduler.step_index >= self.cutoff_steps:
            should_calc, acc = True, 0.0
        else:
            rel = ((inp - prev).abs().mean() / prev.abs().mean()).item()
            acc += float(np.poly1d(self.coefficients)(rel))
            should_calc = not (acc < self.teacache_thresh)
            if should_calc:
                acc = 0.0
        setattr(self, f"accumulated_rel_l1_distance_{branch}", acc)
        setattr(self, f"previous_e0_{branch}", inp)
        return should_calc

    # ---- block loop with the cache (transformer_infer.py:99-157) -------------------------------
    def infer_main_blocks(self, blocks, pre_infer_out):
        sch = self.scheduler
        index = sch.step_index
        records = sch.caching_records if sch.infer_condition else sch.caching_records_2
        if index <= sch.infer_steps - 1:
            records[index] = self.calculate_should_calc(pre_infer_out.embed, pre_infer_out.embed0)
        if records[index] or self.must_calc(index):
            return self.infer_calculating(blocks, pre_infer_out)
        return self.infer_using_cache(pre_infer_out.x)

    def infer_calculating(self, blocks, pre_infer_out):
        ori_x = pre_infer_out.x.clone()
        x = super().infer_main_blocks(blocks, pre_infer_out)
        residual = _residual(x, ori_x, self)
        if self.scheduler.infer_condition:
            self.previous_residual_even = residual
        else:
            self.previous_residual_odd = residual
        return x

    def infer_using_cache(self, x):
        return hip.gate_residual_(x, self.previous_residual_even if self.scheduler.infer_condition else self.previous_residual_odd, None)


class WanTransformerInferMagCaching(WanTransformerInfer):
    def __init__(self, config):
        super().__init__(config)
        _refuse_fp32_stream(self)
        cfg = self.config
        self.magcache_thresh = cfg["magcache_thresh"]
        self.K = cfg["magcache_K"]
        self.retention_ratio = cfg["magcache_retention_ratio"]
        self.mag_ratios = np.array(cfg["magcache_ratios"])
        self.calibration = bool(cfg.get("magcache_calibration", False))
        if not self.calibration and (self.mag_ratios.ndim != 2 or self.mag_ratios.shape[0] != 2 or self.mag_ratios.shape[1] < cfg["infer_steps"]):
            raise ValueError(f"magcache_ratios must be [2, >= infer_steps ({cfg['infer_steps']})], got {self.mag_ratios.shape}")
        self._minus_one = None
        # calibration output: mean and std of the per-token magnitude ratio of successive residuals,
        # and their mean cosine distance; index 0 = conditional branch, 1 = unconditional
        self.norm_ratio, self.norm_std, self.cos_dis = [[1.0], [1.0]], [[0.0], [0.0]], [[0.0], [0.0]]
        self.clear()

    def clear(self):
        # {True: conditional branch, False: unconditional branch}
        self.accumulated_err = {True: 0.0, False: 0.0}
        self.accumulated_steps = {True: 0, False: 0}
        self.accumulated_ratio = {True: 1.0, False: 1.0}
        self.residual_cache = {True: None, False: None}  # bf16 [S, D] on the GPU

    # ---- decision (transformer_infer.py:1009-1033) ----------------------------------------------
    def should_skip(self, step_index, cond):
        if self.calibration or step_index < int(self.config["infer_steps"] * self.retention_ratio):
            return False
        self.accumulated_ratio[cond] = self.accumulated_ratio[cond] * self.mag_ratios[0 if cond else 1][step_index]
        self.accumulated_steps[cond] += 1
        self.accumulated_err[cond] += np.abs(1 - self.accumulated_ratio[cond])
        if self.accumulated_err[cond] < self.magcache_thresh and self.accumulated_steps[cond] <= self.K:
            return True
        self.accumulated_err[cond], self.accumulated_steps[cond], self.accumulated_ratio[cond] = 0, 0, 1.0
        return False

    def infer_main_blocks(self, blocks, pre_infer_out):
        if self.should_skip(self.scheduler.step_index, self.scheduler.infer_condition):
            return self.infer_using_cache(pre_infer_out.x)
        return self.infer_calculating(blocks, pre_infer_out)

    # ---- block loop with the cache (transformer_infer.py:1045-1079) ------------------------------
    def infer_calculating(self, blocks, pre_infer_out):
        cond = self.scheduler.infer_condition
        ori_x = pre_infer_out.x.clone()
        x = super().infer_main_blocks(blocks, pre_infer_out)
        residual = _residual(x, ori_x, self)
        if self.calibration and self.scheduler.step_index >= 1:
            self._calibrate(residual, self.residual_cache[cond], int(not cond))
        self.residual_cache[cond] = residual
        return x

    def infer_using_cache(self, x):
        return hip.gate_residual_(x, self.residual_cache[self.scheduler.infer_condition], None)

    @torch.no_grad()
    def _calibrate(self, cur, prev, index):
        """Statistics a calibration run prints (transformer_infer.py:1057-1065).  An offline tool, a
        handful of reductions per forward: plain torch on the bf16 residuals, as the reference."""
        ratio = cur.norm(dim=-1) / prev.norm(dim=-1)
        self.norm_ratio[index].append(round(ratio.mean().item(), 5))
        self.norm_std[index].append(round(ratio.std().item(), 5))
        self.cos_dis[index].append(round((1 - torch.nn.functional.cosine_similarity(cur, prev, dim=-1, eps=1e-8)).mean().item(), 5))
