"""TeaCache around the Wan block loop: skip all blocks of a forward when the timestep modulation
barely moved, re-using the previous forward's residual.

Mirrors `WanTransformerInferTeaCaching`
(lightx2v/models/networks/wan/infer/feature_caching/transformer_infer.py:25-169): same config keys
(`feature_caching: "Tea"`, `teacache_thresh`, `coefficients`, `use_ret_steps`), same per-CFG-branch
state, same records on the scheduler (`caching_records`, `caching_records_2`).  The decision is host
control logic on a [6, D] (or [1, D]) vector -- it is read back once per forward, as the reference
does with `.cpu().item()`; the token-sized work (residual capture and re-use over [S, D]) runs on
the HIP element-wise kernel.
"""

from __future__ import annotations

import numpy as np
import torch

from . import hip
from .wan_block import WanTransformerInfer


class WanTransformerInferTeaCaching(WanTransformerInfer):
    def __init__(self, config):
        super().__init__(config)
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
        if self._minus_one is None or self._minus_one.numel() != x.shape[1]:
            self._minus_one = torch.full((x.shape[1],), -1.0, dtype=torch.float32, device=x.device)
        residual = hip.gate_residual_(x.clone(), ori_x, self._minus_one)  # x - ori_x
        if self.scheduler.infer_condition:
            self.previous_residual_even = residual
        else:
            self.previous_residual_odd = residual
        return x

    def infer_using_cache(self, x):
        return hip.gate_residual_(x, self.previous_residual_even if self.scheduler.infer_condition else self.previous_residual_odd, None)
