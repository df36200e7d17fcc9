#!/usr/bin/env python3
"""Seconds per denoising step of the Wan DiT hot path on MI355X (synthetic data, random-init weights).

One "step" = `scheduler.step_pre` -> `WanModel.infer` -> `scheduler.step_post` on real-shaped latents
[16, 21, H/8, W/8]: a conditional and an unconditional DiT forward (patch embedding, timestep / text
MLPs, every transformer block, the head, unpatchify), the classifier-free-guidance combine and the
UniPC update.  At N=1 both
forwards run back to back on the one GPU; at N>1 the ranks form a (cfg_p, seq_p) grid like the
reference's device mesh: cfg_p=2 puts the two forwards on two rank groups (one all_gather of the
prediction), seq_p>1 shards tokens Ulysses-style (one fused q|k|v all-to-all + one return
all-to-all per block, one final all_gather).  Total work is fixed as N grows (strong scaling).

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 2 --warmup 1

Rank 0 prints ONE JSON line (metric s/denoise-step; roofline of the self-attention kernel measured
live with events on the launch stream; at N=1 a CPU baseline from the fp32 oracle on a bounded
sample).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MODELS = {
    # name: dim, heads, ffn, layers
    "wan2.1-t2v-14b": (5120, 40, 13824, 40),
    "wan2.1-t2v-1.3b": (1536, 12, 8960, 30),
    # Wan2.2-I2V-A14B: two experts of the 14B architecture (high-noise / low-noise), in_dim 16 + 20
    "wan2.2-i2v-a14b": (5120, 40, 13824, 40),
    # Wan2.1-I2V-14B (the model of the reference's published end-to-end rows): 14B blocks + 257 CLIP image tokens
    # with their own k / v and a second cross-attention, in_dim 16 + 20
    "wan2.1-i2v-14b": (5120, 40, 13824, 40),
}
CLIP_TOKENS, CLIP_DIM = 257, 1280
MOE_BOUNDARY = 0.9        # i2v switch point (fraction of 1000 training timesteps)
MOE_GUIDE = [3.5, 3.5]
MOE_COND_CHANNELS = 20    # mask (4) + VAE latents of the conditioning image (16)
HY_MODELS = {
    # name: hidden, heads, mlp, double blocks (HunyuanVideo-1.5 transformer)
    "hunyuan-1.5": (2048, 16, 8192, 54),
}
HY_GRIDS = {"720p": (31, 45, 80), "480p": (21, 30, 53)}  # latent tokens, patch (1,1,1), VAE stride (4,16,16): 121 / 81 frames
HY_TEXT_LEN = 256  # valid text tokens after the pre-infer's mask crop (synthetic)
GRIDS = {"720p": (21, 45, 80), "480p": (21, 30, 52)}  # (f, h, w) latent tokens for 81 frames, patch (1,2,2)
TEXT_LEN = 512
TEXT_DIM = 4096  # umT5-xxl features
LATENT_CHANNELS = 16
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense fp8 (block-scaled e4m3)
ATTN_TYPES = {"bf16": "hip_flash_attn", "fp8": "hip_fp8_attn", "fp8-linear": "hip_fp8_attn_linear"}
GUIDE_SCALE = 5.0


def block_flops(S, D, H, F, T=TEXT_LEN):
    d = D // H
    self_attn = 4.0 * S * S * H * d
    cross = 4.0 * S * T * D
    gemm = 8.0 * S * D * D + 4.0 * S * D * D + 4.0 * T * D * D + 4.0 * S * D * F
    return self_attn, cross, gemm


TRAFFIC_FILES = ("r03_attn_traffic.json", "r02_attn_traffic.json", "r01_attn_traffic.json")  # newest first


def measured_traffic(S, heads_local):
    """(bytes, source): HBM-side bytes per self-attention call (main + tail launch) from the newest committed PMC
    collection under profiles/ that matches this launch shape (counters cannot be read from inside
    this process: they need their own rocprofv3 pass), or (None, None)."""
    for name in TRAFFIC_FILES:
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except OSError:
            continue
        w = rec.get("workload", {})
        if w.get("tokens") == S and w.get("heads") == heads_local:
            return rec.get("traffic_bytes_per_call", rec.get("traffic_bytes_per_launch")), "profiles/" + name + " (separate rocprofv3 --pmc pass of this workload, not this run)"
    return None, None


def default_layout(n):
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (2, 4)}.get(n, (1, n))


def build_model(cfg, device, seed):
    """WanModel with random-init weights of the named architecture, generated on the GPU block by block."""
    import lightx2v_amd as L

    D, F = cfg["dim"], cfg["ffn_dim"]
    model = L.WanModel(cfg)
    W = model.transformer_weights
    g = torch.Generator(device=device).manual_seed(seed)

    def rn(*shape, std=1.0, dtype=torch.bfloat16):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        wd = {}
        for name in ("self_attn.q", "self_attn.k", "self_attn.v", "self_attn.o", "cross_attn.q", "cross_attn.k", "cross_attn.v", "cross_attn.o", "ffn.0", "ffn.2"):
            out_f = F if name == "ffn.0" else D
            in_f = F if name == "ffn.2" else D
            wd[p + name + ".weight"] = rn(out_f, in_f, std=0.02)
            wd[p + name + ".bias"] = rn(out_f, std=0.02, dtype=torch.float32)
        for name in ("self_attn.norm_q", "self_attn.norm_k", "cross_attn.norm_q", "cross_attn.norm_k"):
            wd[p + name + ".weight"] = torch.ones(D, device=device)
        wd[p + "norm3.weight"] = torch.ones(D, device=device)
        wd[p + "norm3.bias"] = torch.zeros(D, device=device)
        wd[p + "modulation"] = rn(1, 6, D, std=1.0 / math.sqrt(D), dtype=torch.float32)
        if cfg["task"] == "i2v" and cfg.get("use_image_encoder", True):
            for name in ("cross_attn.k_img", "cross_attn.v_img"):
                wd[p + name + ".weight"] = rn(D, D, std=0.02)
                wd[p + name + ".bias"] = rn(D, std=0.02, dtype=torch.float32)
            wd[p + "cross_attn.norm_k_img.weight"] = torch.ones(D, device=device)
        W.blocks[i].load(wd)
        sa, ca, _ = W.blocks[i].compute_phases
        sa.fused_qkv()
        ca.fused_kv()
    hw = {"head.head.weight": rn(64, D, std=0.02), "head.head.bias": rn(64, std=0.02, dtype=torch.float32),
          "head.modulation": rn(1, 2, D, std=1.0 / math.sqrt(D), dtype=torch.float32)}
    W.head.load(hw)
    W.head_modulation.load(hw)
    pw = {"patch_embedding.weight": rn(D, cfg["in_dim"], 1, 2, 2, std=0.05), "patch_embedding.bias": rn(D, std=0.02, dtype=torch.float32)}
    for name, (o, i) in {"text_embedding.0": (D, TEXT_DIM), "text_embedding.2": (D, D), "time_embedding.0": (D, cfg["freq_dim"]),
                         "time_embedding.2": (D, D), "time_projection.1": (6 * D, D)}.items():
        pw[name + ".weight"] = rn(o, i, std=0.02)
        pw[name + ".bias"] = rn(o, std=0.02, dtype=torch.float32)
    if cfg["task"] == "i2v" and cfg.get("use_image_encoder", True):
        for name, n in (("img_emb.proj.0", CLIP_DIM), ("img_emb.proj.4", D)):
            pw[name + ".weight"], pw[name + ".bias"] = torch.ones(n, device=device), torch.zeros(n, device=device)
        for name, (o, i) in {"img_emb.proj.1": (CLIP_DIM, CLIP_DIM), "img_emb.proj.3": (D, CLIP_DIM)}.items():
            pw[name + ".weight"], pw[name + ".bias"] = rn(o, i, std=0.02), rn(o, std=0.02, dtype=torch.float32)
    model.pre_weight.load(pw)
    return model


class AttnTimer:
    """Wraps the self-attention plugin's apply with events recorded on the stream it launches on, and
    sums the algorithmic FLOPs of what was launched (under Ulysses a block's heads arrive in chunks)."""

    def __init__(self, plugin):
        self.plugin = plugin
        self.inner = plugin.apply
        self.events = []
        self.flops = 0.0
        self.enabled = False
        plugin.apply = self.apply

    def apply(self, q, k, v, *a, **kw):
        if not self.enabled:
            return self.inner(q, k, v, *a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        out = self.inner(q, k, v, *a, **kw)
        e1.record(torch.cuda.current_stream())
        self.events.append((e0, e1))
        self.flops += 4.0 * q.shape[-3] * k.shape[-3] * q.shape[-2] * q.shape[-1]
        return out

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events)


class KernelClassTimer:
    """Event-timed wrappers around the binding's entry points (Linears, norms, every attention call):
    where a step's time goes and what the GEMMs / norms achieve, measured inside this run (events on
    the launch stream; a busy stream makes the span of a call its kernels' duration)."""

    HBM_PEAK_GBS = 8000.0

    def __init__(self):
        from lightx2v_amd import hip

        self.hip, self.enabled, self.rec, self._orig = hip, False, {}, {}
        self._wrap("gemm", "linear_bf16", lambda a, w, *r, **kw: 2.0 * a.shape[0] * w.shape[0] * w.shape[1])
        self._wrap("gemm_fp8", "linear_fp8", lambda a8, a_s, w8, *r, **kw: 2.0 * a8.shape[0] * w8.shape[0] * w8.shape[1])
        self._wrap("ln_modulate", "norm", lambda x, *r, **kw: float(x.shape[0] * x.shape[1] * (x.element_size() + (1 if kw.get("quant") else 2))))
        self._wrap("rmsnorm_rope", "norm", lambda x, *r, **kw: 4.0 * x.shape[0] * x.shape[1])
        self._wrap("attn_fwd", "attention", lambda q, k, *r, **kw: 4.0 * q.shape[0] * k.shape[0] * q.shape[1] * q.shape[2])
        self._wrap("attn_fwd_fp8", "attention", lambda q, k, *r, **kw: 4.0 * q.shape[0] * k.shape[0] * q.shape[1] * q.shape[2])

    def _wrap(self, name, cls, work):
        inner = getattr(self.hip, name)
        self._orig[name] = inner
        slot = self.rec.setdefault(cls, {"events": [], "work": 0.0, "calls": 0})

        def timed(*a, **kw):
            if not self.enabled:
                return inner(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            out = inner(*a, **kw)
            e1.record(torch.cuda.current_stream())
            slot["events"].append((e0, e1))
            slot["work"] += work(*a, **kw)
            slot["calls"] += 1
            return out

        setattr(self.hip, name, timed)

    def summary(self, steps, step_ms, self_attn_ms_per_step):
        """Per step: ms by kernel class (self-attention split out of `attention` by the caller's own timer),
        GEMM TFLOP/s and norm GB/s over the timed steps."""
        ms = {c: sum(a.elapsed_time(b) for a, b in r["events"]) / steps for c, r in self.rec.items()}
        lin_ms = ms.get("linear_bf16", 0.0) + ms.get("linear_fp8", 0.0)
        comp = {"self_attention": self_attn_ms_per_step, "cross_attention": max(0.0, ms.get("attention", 0.0) - self_attn_ms_per_step),
                "linears": lin_ms, "norms": ms.get("norm", 0.0)}
        comp["everything_else"] = max(0.0, step_ms - sum(comp.values()))
        out = {"composition_ms_per_step": {k: round(v, 2) for k, v in comp.items()}}
        for cls, peak, key in (("linear_bf16", MFMA_BF16_PEAK_TFLOPS, "linears_bf16"), ("linear_fp8", MFMA_FP8_PEAK_TFLOPS, "linears_w8a8")):
            r = self.rec[cls]
            if r["calls"]:
                t = r["work"] / (ms[cls] * steps * 1e-3) / 1e12
                out[key] = {"achieved_tflops": t, "frac_of_mfma_peak": t / peak, "calls_timed": r["calls"], "ms_per_step": ms[cls]}
        r = self.rec["norm"]
        if r["calls"]:
            g = r["work"] / (ms["norm"] * steps * 1e-3) / 1e9
            out["norms"] = {"achieved_gbs": g, "frac_of_hbm_peak": g / self.HBM_PEAK_GBS, "calls_timed": r["calls"], "ms_per_step": ms["norm"],
                            "bytes": "rows x dim x (bytes in + bytes out) per call"}
        return out


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(D, H, F, full_S, layers, threads):
    """fp32 oracle block (the reference's torch_sdpa + Default-matmul path restated) on the host
    cores, on a bounded sample: one block of the same architecture at a reduced token count, one
    warm-up run and three timed runs (median); extrapolated to the full step by FLOPs."""
    from oracle import wan_block_ref as R

    torch.set_num_threads(threads)
    grid = (3, 45, 80) if full_S > 40000 else (3, 30, 52)  # a few seconds per run on 16 threads
    S = grid[0] * grid[1] * grid[2]
    w = R.make_block_weights(D, F, seed=0)
    x, context, embed0, _ = R.make_inputs(S, D, TEXT_LEN, seed=0)
    cs = R.build_cos_sin(grid, D // H)
    times = []
    with torch.no_grad():
        for run in range(4):
            t0 = time.time()
            R.block_forward(x, embed0, context, w, cs, H)
            if run > 0:
                times.append(time.time() - t0)
    dt = sorted(times)[1]
    sample_flops = sum(block_flops(S, D, H, F))
    rate = sample_flops / dt  # FLOP/s on this host
    full_flops = sum(block_flops(full_S, D, H, F)) * layers * 2
    out = {
        "value": full_flops / rate, "unit": "s/denoise-step", "cores": threads, "kind": "port", "cpu_model": cpu_model_name(),
        "runs_s": [round(t, 3) for t in times], "extrapolated": True,
        "sample": f"one fp32 oracle block at S={S} tokens ({grid[0]}x{grid[1]}x{grid[2]}), D={D}: 1 warm-up + 3 timed runs, median {dt:.2f} s = {rate / 1e12:.3f} TFLOP/s on {threads} threads; "
                  f"EXTRAPOLATED by algorithmic FLOPs to {layers} blocks x 2 CFG passes at S={full_S}",
    }
    if (D, H, F, full_S) == (1536, 12, 8960, 32760):
        # BASELINE configuration 1 as it stands: one fp32 block of the 1.3B model at all 32 760 tokens, timed, nothing extrapolated
        out.update(cpu_c1_block(threads))
    return out


def cpu_c1_block(threads):
    """BASELINE.json configs[0]: one Wan2.1-T2V-1.3B block forward at 480x832x81f (32 760 tokens), fp32,
    torch_sdpa attention, random-init weights, on the host cores: 1 warm-up + 3 timed runs."""
    from oracle import wan_block_ref as R

    torch.set_num_threads(threads)
    D, H, F, grid = 1536, 12, 8960, GRIDS["480p"]
    S = grid[0] * grid[1] * grid[2]
    w = R.make_block_weights(D, F, seed=0)
    x, context, embed0, _ = R.make_inputs(S, D, TEXT_LEN, seed=0)
    cs = R.build_cos_sin(grid, D // H)
    times = []
    with torch.no_grad():
        for run in range(4):
            t0 = time.time()
            R.block_forward(x, embed0, context, w, cs, H)
            if run > 0:
                times.append(time.time() - t0)
    med = sorted(times)[1]
    return {"c1_block_s": med, "c1_runs_s": [round(t, 3) for t in times],
            "c1_sample": f"BASELINE config 1 unscaled: one fp32 oracle block, Wan2.1-T2V-1.3B dims, S={S}: 1 warm-up + 3 timed runs, median {med:.2f} s = "
                         f"{sum(block_flops(S, D, H, F)) / med / 1e12:.3f} TFLOP/s on {threads} threads"}


def attn_peak(args):
    return MFMA_BF16_PEAK_TFLOPS if args.attn == "bf16" else MFMA_FP8_PEAK_TFLOPS


def arithmetic_label(args):
    """The JSON line's `dtype` and the metric-name suffix for the non-headline arithmetic choices."""
    if args.residual == "fp32":
        base = argparse.Namespace(**dict(vars(args), residual="bf16"))
        d, suffix = arithmetic_label(base)
        return d + " + fp32 residual stream", (suffix[:-1] + ", fp32 residual stream)") if suffix else " (fp32 residual stream)"
    if args.linear == "bf16" and args.attn == "bf16":
        return "bf16", ""
    lin = {"bf16": "bf16 Linears", "fp8": "fp8 Linears (W8A8 e4m3)", "int8": "int8 Linears (W8A8)"}[args.linear]
    att = {"bf16": "bf16 attention", "fp8": "fp8 attention (block-scaled e4m3)", "fp8-linear": "fp8 attention (block-scaled e4m3, exp-free P)"}[args.attn]
    return f"{lin} + {att}", " (" + ", ".join(x for x in (f"{args.linear} Linears" if args.linear != "bf16" else "", {"bf16": "", "fp8": "fp8 attention", "fp8-linear": "fp8 attention with exp-free P"}[args.attn]) if x) + ")"


def emulation_note(args, cfg_p, seq_p):
    return (f" [--as-rank-of: single-GPU emulation of ONE rank of a cfg_p{cfg_p} x seq_p{seq_p} grid, exchanges replaced by device copies of the rank's own data; "
            "compute-only prediction, not a scaling measurement]") if args.as_rank_of else ""


def run_hunyuan(args, dev, world, rank, cfg_p, seq_p, backend, share_gpu, emulate=None):
    """HunyuanVideo-1.5, 4-step distilled schedule, no CFG: one step = scheduler.step_pre ->
    HunyuanVideo15Model.infer (pre-infer with the token refiner, all double-stream blocks, final
    layer, unpatchify) -> scheduler.step_post, on synthetic latents and encoder outputs."""
    import lightx2v_amd as L
    from lightx2v_amd import ulysses

    if args.residual != "bf16":
        raise SystemExit("--residual fp32 is built for the Wan blocks")
    D, H, F, layers = HY_MODELS[args.model]
    if args.layers is not None:
        layers = args.layers
    grid = HY_GRIDS[args.res]
    S = grid[0] * grid[1] * grid[2]
    if cfg_p != 1:
        seq_p, cfg_p = (world if emulate is None else cfg_p * seq_p), 1  # the distilled model runs without CFG: every rank joins the sequence group
    if H % seq_p != 0:
        raise SystemExit(f"heads_num={H} not divisible by seq_p={seq_p}")
    seq_group = None
    if world > 1 or args.sp1:
        _, _, seq_group, _ = ulysses.make_groups(1, seq_p)
    if emulate is not None and seq_p > 1:
        from lightx2v_amd import comm

        seq_group = comm.EmulatedGroup(seq_p)
    C, TEXT_DIM, TXT_TOKENS, BYT5_TOKENS, VIS_TOKENS = 32, 3584, 1000, 256, 729
    cfg = {"mm_double_blocks_depth": layers, "heads_num": H, "hidden_size": D, "task": "t2v", "patch_size": (1, 1, 1), "out_channels": C,
           "dit_quant_scheme": {"bf16": "hip-bf16", "fp8": "hip-fp8", "int8": "hip-int8"}[args.linear],
           "attn_type": ATTN_TYPES[args.attn],
           "seq_parallel": seq_p > 1 or args.sp1, "seq_p_group": seq_group, "parallel": {"seq_p_size": seq_p, "seq_p_attn_type": "hip_ulysses"},
           "enable_cfg": False, "sample_guide_scale": -1.0, "sample_shift": 5.0, "infer_steps": 4, "denoising_step_list": [1000, 750, 500, 250],
           "rope_dim_list": [16, 56, 56], "rope_theta": 256.0, "run_device": str(dev)}
    model = L.HunyuanVideo15Model(cfg)
    W = model.transformer_weights
    g = torch.Generator(device=dev).manual_seed(1234)  # same seed on every rank: weights are replicated

    def rn(*shape, std=1.0, dtype=torch.bfloat16):
        return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * std).to(dtype)

    for i in range(layers):
        p, wd = f"double_blocks.{i}.", {}
        for br in ("img", "txt"):
            for name, (o, k) in {"mod.linear": (6 * D, D), "attn_q": (D, D), "attn_k": (D, D), "attn_v": (D, D), "attn_proj": (D, D),
                                 "mlp.fc1": (F, D), "mlp.fc2": (D, F)}.items():
                wd[f"{p}{br}_{name}.weight"] = rn(o, k, std=0.02 if name != "mod.linear" else 1.0 / math.sqrt(D))
                wd[f"{p}{br}_{name}.bias"] = rn(o, std=0.02, dtype=torch.float32)
            for n in ("q", "k"):
                wd[f"{p}{br}_attn_{n}_norm.weight"] = torch.ones(D // H, device=dev)
        W.double_blocks[i].load(wd)
        W.double_blocks[i].img_branch.fused_qkv()
        W.double_blocks[i].txt_branch.fused_qkv()
    W.final_layer.load({"final_layer.adaLN_modulation.1.weight": rn(2 * D, D, std=1.0 / math.sqrt(D)), "final_layer.adaLN_modulation.1.bias": rn(2 * D, std=0.02, dtype=torch.float32),
                        "final_layer.linear.weight": rn(C, D, std=0.02), "final_layer.linear.bias": rn(C, std=0.02, dtype=torch.float32)})
    pre = {"img_in.proj.weight": rn(D, 2 * C + 1, 1, 1, 1, std=0.05), "img_in.proj.bias": rn(D, std=0.02, dtype=torch.float32),
           "cond_type_embedding.weight": rn(3, D, std=0.1)}
    lin = {"time_in.mlp.0": (D, 256), "time_in.mlp.2": (D, D), "txt_in.t_embedder.mlp.0": (D, 256), "txt_in.t_embedder.mlp.2": (D, D),
           "txt_in.c_embedder.linear_1": (D, TEXT_DIM), "txt_in.c_embedder.linear_2": (D, D), "txt_in.input_embedder": (D, TEXT_DIM)}
    for i in range(2):
        rb = f"txt_in.individual_token_refiner.blocks.{i}."
        lin.update({rb + "self_attn_qkv": (3 * D, D), rb + "self_attn_proj": (D, D), rb + "mlp.fc1": (4 * D, D), rb + "mlp.fc2": (D, 4 * D),
                    rb + "adaLN_modulation.1": (2 * D, D)})
        for n in ("norm1", "norm2"):
            pre[rb + n + ".weight"], pre[rb + n + ".bias"] = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    for name, (o, k) in lin.items():
        pre[name + ".weight"], pre[name + ".bias"] = rn(o, k, std=0.02), rn(o, std=0.02, dtype=torch.float32)
    model.pre_weight.load(pre)
    sch = L.HunyuanVideo15StepDistillScheduler(cfg)
    model.set_scheduler(sch)

    g2 = torch.Generator(device=dev).manual_seed(99)

    def mask(n, valid):
        m = torch.zeros(1, n, dtype=torch.int64, device=dev)
        m[:, :valid] = 1
        return m

    inputs = {"text_encoder_output": {"context": [rn(1, TXT_TOKENS, TEXT_DIM), mask(TXT_TOKENS, HY_TEXT_LEN)], "context_null": [rn(1, TXT_TOKENS, TEXT_DIM), mask(TXT_TOKENS, 1)],
                                      "byt5_features": rn(1, BYT5_TOKENS, D), "byt5_masks": mask(BYT5_TOKENS, 0)},
              "image_encoder_output": {"siglip_output": torch.zeros(1, VIS_TOKENS, D, device=dev, dtype=torch.bfloat16), "siglip_mask": mask(VIS_TOKENS, 0).to(torch.bfloat16),
                                       "cond_latents": None}}
    latents0 = torch.randn(1, C, *grid, generator=g2, device=dev, dtype=torch.float32).to(torch.bfloat16)
    sch.prepare(0, (C, *grid), image_encoder_output=inputs["image_encoder_output"], latents=latents0)
    timers = [AttnTimer(W.double_blocks[i].self_attention) for i in range(layers)]
    classes = None if args.no_breakdown else KernelClassTimer()
    step_no = [0]

    def step():
        i = step_no[0] % sch.infer_steps
        step_no[0] += 1
        if i == 0:
            sch.latents = latents0.clone()
        sch.step_pre(step_index=i)
        model.infer(inputs)
        sch.step_post()
        return sch.latents

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for t in timers:
        t.enabled = True
    if classes is not None:
        classes.enabled = True
    if world > 1 or args.sp1:
        ulysses.COMM_TIMER.enable()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out.float()).all()
    exposed_ms = ulysses.COMM_TIMER.total_ms() / args.steps if (world > 1 or args.sp1) else 0.0
    s_per_step = elapsed / args.steps
    n_launches = sum(len(t.events) for t in timers)
    attn_ms = sum(t.total_ms() for t in timers) / max(1, n_launches)
    attn_flops = sum(t.flops for t in timers) / max(1, n_launches)
    achieved = attn_flops / (attn_ms * 1e-3) / 1e12
    Sj = S + HY_TEXT_LEN
    step_flops = (4.0 * Sj * Sj * D + 2.0 * Sj * D * (4 * D + 2 * F)) * layers
    frames = {31: 121, 21: 81}[grid[0]]
    extra = {} if classes is None else classes.summary(args.steps, s_per_step * 1e3, sum(t.total_ms() for t in timers) / args.steps)
    return {
        **extra,
        "metric": f"s/denoise-step {args.model} {args.res}{frames}f" + arithmetic_label(args)[1] + (f" (one rank of {cfg_p}x{seq_p}, emulated)" if args.as_rank_of else ""),
        "value": s_per_step, "unit": "s/denoise-step", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": s_per_step * 1e3, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": arithmetic_label(args)[0],
        "data": "synthetic latents, text / glyph / image-encoder features and masks, random-init weights of the named architecture"
                + ("" if (backend == "nccl" and not share_gpu) else f" [REHEARSAL: backend={backend} share_gpu={share_gpu}, not a measurement]")
                + (" [--sp1: sequence-parallel code path on an RCCL group of one rank]" if args.sp1 else "") + emulation_note(args, cfg_p, seq_p),
        "config": {"workload": f"{args.model} {args.res} x{frames}f: {S} image + {HY_TEXT_LEN} text tokens, full denoise step (pre-infer with token refiner, {layers} double-stream blocks + final layer, unpatchify, distilled-schedule update), no CFG",
                   "parallelism": f"seq_p{seq_p} (ulysses)", "tokens": S, "dim": D, "heads": H, "ffn_dim": F, "layers": layers},
        "step_tflops": step_flops / 1e12, "achieved_tflops_per_gpu": step_flops / s_per_step / 1e12 / world,
        "exposed_comm_ms_per_step": exposed_ms, "ulysses_chunks": len(ulysses.LAST_WIDTHS) or int(os.environ.get("LX2V_ULYSSES_CHUNKS") or ulysses.default_chunks(seq_p)),
        "ulysses_chunk_heads": list(ulysses.LAST_WIDTHS),
        "roofline": {"kernel": ("attn_fwd_x16_kernel" if args.attn == "bf16" else "attn_fwd_fp8_w4_kernel + quantisation pre-pass") + " (joint image+text attention)",
                     "bound": "mfma", "achieved": achieved, "peak": attn_peak(args),
                     "unit": "TFLOP/s", "frac": achieved / attn_peak(args), "traffic": None,
                     "launch_ms": attn_ms, "flops_per_launch": attn_flops, "launches_timed": n_launches},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default=os.environ.get("LX2V_BENCH_MODEL", "wan2.1-t2v-14b"), choices=sorted(MODELS) + sorted(HY_MODELS))
    ap.add_argument("--res", default=os.environ.get("LX2V_BENCH_RES", "720p"), choices=sorted(GRIDS))
    ap.add_argument("--layout", default=None, help="cfg_p x seq_p, e.g. 2x4 (default: 1x1, 2x1, 2x2, 2x4 for 1/2/4/8 GPUs)")
    ap.add_argument("--layers", type=int, default=None, help="override the layer count (debug only; the JSON line then says so)")
    ap.add_argument("--linear", default="bf16", choices=["bf16", "fp8", "int8"],
                    help="arithmetic of the blocks' Linears: bf16 (the headline configuration), W8A8 e4m3 (the reference's fp8 schemes) or W8A8 int8 (its int8 schemes)")
    ap.add_argument("--attn", default="bf16", choices=["bf16", "fp8", "fp8-linear"],
                    help="self-attention arithmetic: bf16 (the headline configuration), block-scaled e4m3 on the fp8 MFMA (where the reference would run a quantised backend), "
                         "or the same with P taken as 2^x with a linear mantissa instead of by exps")
    ap.add_argument("--residual", default="bf16", choices=["bf16", "fp32"],
                    help="storage of the residual stream between the Linears: bf16 (the headline configuration) or fp32 (the reference's SENSITIVE_LAYER_DTYPE=FP32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true", help="do not time the Linears / norms / attention calls by class (events around each call)")
    ap.add_argument("--sp1", action="store_true",
                    help="one GPU only: run the sequence-parallel code path on an RCCL group of ONE rank (pack kernel, chunked attention, "
                         "asynchronous all-to-alls that move the rank's own data, segmented output projection, final all_gather); "
                         "prices what the N > 1 path costs besides the links")
    ap.add_argument("--as-rank-of", default=None, metavar="CFGxSEQ",
                    help="one GPU only: run what ONE rank of a cfg_p x seq_p grid computes (one CFG branch when cfg_p = 2, token-wise work on S/seq_p rows, "
                         "attention on H/seq_p heads over all S rows in the exchange's real head chunks, segmented output projection); the exchanges are device "
                         "copies of the rank's own data.  A single-GPU emulation that predicts the layout's compute-only s/step -- not a scaling measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`): become the launcher.  Nothing has touched the GPU in
        # this process; the ranks are children of torch.distributed.run, rank 0's JSON line goes to our
        # stdout through the inherited descriptor, and the launcher's exit status is ours.
        import socket
        import subprocess

        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        raise SystemExit(subprocess.call(cmd))

    # stdout carries the one JSON line and nothing else: libraries that print to file descriptor 1 (RCCL's
    # version banner at communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if os.environ.get("LX2V_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["LX2V_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        # a sign of life on stderr every 30 s (the one JSON line comes at the very end, minutes into a 20-step run)
        import threading

        t_start = time.perf_counter()

        def heartbeat():
            while True:
                time.sleep(30)
                print(f"[bench] running, {time.perf_counter() - t_start:.0f} s", file=sys.stderr, flush=True)

        threading.Thread(target=heartbeat, daemon=True).start()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # LX2V_SHARE_GPU=1 + LX2V_DIST_BACKEND=gloo: functional rehearsal of the N>1 path with every rank
    # on GPU 0 (gloo moves CUDA tensors through the host); never used for reported numbers.
    share_gpu = os.environ.get("LX2V_SHARE_GPU", "0") == "1"
    backend = os.environ.get("LX2V_DIST_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if args.sp1:
        if world != 1:
            raise SystemExit("--sp1 is a one-GPU run")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    elif world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
            # rehearsal only: ranks that share one GPU hand gloo host copies (gloo's own CUDA path
            # stalls there, tools/probe_gloo_shared_gpu.py)
            from lightx2v_amd import comm

            comm.set_transport(comm.HostStagedTransport())
    cfg_p, seq_p = default_layout(world) if args.layout is None else tuple(int(v) for v in args.layout.lower().split("x"))
    assert cfg_p * seq_p == world and cfg_p in (1, 2)
    emulate = None
    if args.as_rank_of:
        if world != 1 or args.sp1:
            raise SystemExit("--as-rank-of is a one-GPU run (and excludes --sp1)")
        from lightx2v_amd import comm

        cfg_p, seq_p = (int(v) for v in args.as_rank_of.lower().split("x"))
        assert cfg_p in (1, 2) and seq_p >= 1
        comm.set_transport(comm.EmulatedPeersTransport())
        emulate = (comm.EmulatedGroup(cfg_p) if cfg_p > 1 else None, comm.EmulatedGroup(seq_p) if seq_p > 1 else None)

    import lightx2v_amd as L
    from lightx2v_amd import ulysses

    if args.model in HY_MODELS:
        line = run_hunyuan(args, dev, world, rank, cfg_p, seq_p, backend, share_gpu, emulate)
        if rank == 0:
            print(json.dumps(line), file=json_out, flush=True)
        if world > 1 or args.sp1:
            dist.barrier()
            dist.destroy_process_group()
        return

    D, H, F, layers = MODELS[args.model]
    if args.layers is not None:
        layers = args.layers
    grid = GRIDS[args.res]
    S = grid[0] * grid[1] * grid[2]
    if H % seq_p != 0:
        raise SystemExit(f"num_heads={H} not divisible by seq_p={seq_p}")

    cfg_idx = seq_idx = 0
    seq_group = cfg_group = None
    if world > 1 or args.sp1:
        cfg_idx, seq_idx, seq_group, cfg_group = ulysses.make_groups(cfg_p, seq_p)
    if emulate is not None:
        cfg_group, seq_group = emulate
    cfg = {"num_layers": layers, "num_heads": H, "dim": D, "ffn_dim": F, "task": "t2v", "model_cls": "wan2.1",
           "in_dim": LATENT_CHANNELS, "out_dim": LATENT_CHANNELS, "freq_dim": 256, "infer_steps": 50, "sample_shift": 5.0,
           "sample_guide_scale": GUIDE_SCALE, "enable_cfg": True, "run_device": "cuda",
           "dit_quant_scheme": {"bf16": "hip-bf16", "fp8": "hip-fp8", "int8": "hip-int8"}[args.linear],
           "self_attn_1_type": ATTN_TYPES[args.attn],
           "sensitive_layer_dtype": "fp32" if args.residual == "fp32" else "bf16",
           "cfg_parallel": cfg_p > 1, "cfg_p_group": cfg_group,
           "seq_parallel": seq_p > 1 or args.sp1, "seq_p_group": seq_group, "parallel": {"seq_p_size": seq_p, "cfg_p_size": cfg_p, "seq_p_attn_type": "hip_ulysses"}}
    moe = args.model == "wan2.2-i2v-a14b"
    i2v_clip = args.model == "wan2.1-i2v-14b"
    if moe:
        # a 4-step schedule puts steps 0-1 above the boundary and steps 2-3 below it, so one warm-up
        # and two timed steps run the high-noise expert, then each expert once
        cfg.update({"task": "i2v", "model_cls": "wan2.2_moe", "in_dim": LATENT_CHANNELS + MOE_COND_CHANNELS, "use_image_encoder": False,
                    "sample_guide_scale": MOE_GUIDE, "infer_steps": 4})
        experts = [build_model(cfg, dev, seed=1234), build_model(cfg, dev, seed=4321)]  # same seeds on every rank
        model = L.MultiModelStruct(experts, cfg, boundary=MOE_BOUNDARY)
    else:
        if i2v_clip:
            cfg.update({"task": "i2v", "use_image_encoder": True, "in_dim": LATENT_CHANNELS + MOE_COND_CHANNELS})
        experts = [build_model(cfg, dev, seed=1234)]  # same seed on every rank: weights are replicated
        model = experts[0]
    sch = L.WanScheduler(cfg)
    model.set_scheduler(sch)

    # synthetic step inputs: latents of the clip's shape and umT5-sized text features (cond / uncond)
    S_pad = S + (seq_p - S % seq_p) % seq_p
    latent_shape = (LATENT_CHANNELS, grid[0], 2 * grid[1], 2 * grid[2])
    g = torch.Generator(device=dev).manual_seed(99)
    sch.prepare(0, latent_shape, latents=torch.randn(*latent_shape, generator=g, device=dev, dtype=torch.float32))
    inputs = {"text_encoder_output": {k: torch.randn(1, TEXT_LEN, TEXT_DIM, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
                                      for k in ("context", "context_null")}}
    if moe or i2v_clip:
        inputs["image_encoder_output"] = {"vae_encoder_out": torch.randn(MOE_COND_CHANNELS, *latent_shape[1:], generator=g, device=dev, dtype=torch.float32)}
    if i2v_clip:
        inputs["image_encoder_output"]["clip_encoder_out"] = torch.randn(CLIP_TOKENS, CLIP_DIM, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    timers = [AttnTimer(m.transformer_weights.blocks[i].compute_phases[0].self_attn_1) for m in experts for i in range(layers)]
    classes = None if args.no_breakdown else KernelClassTimer()
    picks = []

    dbg = os.environ.get("LX2V_BENCH_DEBUG") == "1"
    step_no = [0]

    def step():
        i = step_no[0] % cfg["infer_steps"]
        step_no[0] += 1
        if i == 0 and step_no[0] > 1:  # more timed steps than the schedule has: start a new clip
            sch.prepare(0, latent_shape, latents=sch.latents)
        sch.step_pre(step_index=i)
        model.infer(inputs)
        sch.step_post()
        if moe:
            picks.append(model.cur_model_index)
        if dbg:
            torch.cuda.synchronize()
            print(f"[rank {rank}] step {i} done, latents finite={bool(torch.isfinite(sch.latents).all())} mean|x|={float(sch.latents.abs().mean()):.6f} "
                  f"mean|pred|={float(sch.noise_pred.abs().mean()):.6f}", flush=True)
        return sch.latents

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for t in timers:
        t.enabled = True
    if classes is not None:
        classes.enabled = True
    if world > 1 or args.sp1:
        ulysses.COMM_TIMER.enable()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        noise_pred = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(noise_pred).all()
    s_per_step = elapsed / args.steps

    n_launches = sum(len(t.events) for t in timers)
    attn_ms = sum(t.total_ms() for t in timers) / max(1, n_launches)          # mean launch duration on this rank
    attn_flops = sum(t.flops for t in timers) / max(1, n_launches)            # mean algorithmic FLOPs per launch
    achieved = attn_flops / (attn_ms * 1e-3) / 1e12
    sa, ca, ge = block_flops(S, D, H, F)
    step_flops = (sa + ca + ge) * layers * 2

    traffic, traffic_source = measured_traffic(S_pad, H // seq_p) if (args.attn == "bf16" and n_launches == layers * 2 * args.steps // cfg_p) else (None, None)
    exposed_ms = ulysses.COMM_TIMER.total_ms() / args.steps if (world > 1 or args.sp1) else 0.0
    if rank == 0:
        line = {
            "metric": "s/denoise-step Wan2.1-T2V-14B 720p81f" if (args.model, args.res, args.linear, args.attn, args.residual, args.as_rank_of) == ("wan2.1-t2v-14b", "720p", "bf16", "bf16", "bf16", None) else f"s/denoise-step {args.model} {args.res}81f" + arithmetic_label(args)[1] + (f" (one rank of {cfg_p}x{seq_p}, emulated)" if args.as_rank_of else ""),
            "value": s_per_step, "unit": "s/denoise-step", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": s_per_step * 1e3, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": arithmetic_label(args)[0], "data": "synthetic latents and text features, random-init weights of the named architecture"
                    + ("" if (backend == "nccl" and not share_gpu) else f" [REHEARSAL: backend={backend} share_gpu={share_gpu}, not a measurement]")
                + (" [--sp1: sequence-parallel code path on an RCCL group of one rank]" if args.sp1 else "") + emulation_note(args, cfg_p, seq_p),
            "config": {"workload": f"{args.model} {args.res} x81f: {S} tokens, full denoise step (pre-infer, {layers} blocks + head, unpatchify) x 2 CFG forwards + CFG combine + UniPC update, text len {TEXT_LEN}"
                                   + (f"; two resident experts, switch at t = {MOE_BOUNDARY} (experts of the timed steps: {picks[args.warmup:]}), {MOE_COND_CHANNELS} conditioning channels" if moe else "")
                                   + (f"; {CLIP_TOKENS} CLIP image tokens (img_emb MLP + second cross-attention per block), {MOE_COND_CHANNELS} conditioning channels" if i2v_clip else ""),
                       "parallelism": f"cfg_p{cfg_p} x seq_p{seq_p} (ulysses)", "tokens": S, "dim": D, "heads": H, "ffn_dim": F, "layers": layers},
            "step_tflops": step_flops / 1e12,
            "achieved_tflops_per_gpu": step_flops / s_per_step / 1e12 / world,
            # rank 0's compute stream blocked in waits for the Ulysses exchanges, per step (events around each wait)
            "exposed_comm_ms_per_step": exposed_ms, "ulysses_chunks": len(ulysses.LAST_WIDTHS) or int(os.environ.get("LX2V_ULYSSES_CHUNKS") or ulysses.default_chunks(seq_p)),
        "ulysses_chunk_heads": list(ulysses.LAST_WIDTHS),
            "roofline": {"kernel": "attn_fwd_x16_kernel (self-attention: main launch + 64-row tail launch)" if args.attn == "bf16" else "attn_fwd_fp8_w4_kernel + quantisation pre-pass (self-attention)",
                         "bound": "mfma", "achieved": achieved, "peak": attn_peak(args),
                         "unit": "TFLOP/s", "frac": achieved / attn_peak(args),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "launch_ms": attn_ms, "flops_per_launch": attn_flops, "launches_timed": n_launches},
        }
        if classes is not None:
            line.update(classes.summary(args.steps, s_per_step * 1e3, sum(t.total_ms() for t in timers) / args.steps))
        if world == 1 and not args.no_cpu_baseline and not args.as_rank_of:
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
            line["cpu_baseline"] = cpu_baseline(D, H, F, S, layers, threads)
        print(json.dumps(line), file=json_out, flush=True)
    if world > 1 or args.sp1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
