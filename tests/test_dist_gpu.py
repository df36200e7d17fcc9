"""The sequence-parallel and CFG-parallel paths on the real HIP kernels: 2 or 4 ranks share the one
GPU and exchange through gloo on host copies, so the pack kernel, the strided attention reads, the
return all-to-all, the K-segmented output projection and the model-level gathers are exercised end
to end and compared with the unsharded run on the same inputs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn(fn, args, nprocs, deadline_s=150):
    """mp.spawn under a deadline, once: every worker arms a stack dump shortly before it, so a
    stall names the collective each rank is parked in instead of being retried."""
    import time
    ctx = mp.spawn(fn, args=args(_free_port()), nprocs=nprocs, join=False)
    t_end = time.time() + deadline_s
    while time.time() < t_end:
        if ctx.join(timeout=1.0):   # raises if a worker failed
            return
    for pr in ctx.processes:
        if pr.is_alive():
            pr.kill()
    for pr in ctx.processes:
        pr.join(10)
    pytest.fail(f"{fn.__name__}: stalled for {deadline_s} s (stack dumps of the ranks are in the captured stderr)")


def _init(rank, world, port):
    """Process group + transport of a rank that shares GPU 0 with the others.  RCCL refuses two ranks
    on one device and gloo's own CUDA path stalls there (tools/probe_gloo_shared_gpu.py), so gloo is
    handed host copies; the product's issue order and asynchrony are unchanged."""
    import faulthandler

    faulthandler.dump_traceback_later(130, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lightx2v_amd import comm

    comm.set_transport(comm.HostStagedTransport())
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


def _init_rccl(port):
    """A process group of ONE rank on RCCL (backend "nccl" is RCCL on ROCm) with the production
    transport: every collective of the exchange -- asynchronous all-to-alls on RCCL's own stream,
    the waits on the compute stream, list-output all_gathers -- goes through the library that
    carries them on a multi-GPU node, moving this rank's own data."""
    import faulthandler

    faulthandler.dump_traceback_later(130, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from lightx2v_amd import comm

    assert comm.transport().name == "torch.distributed"
    return dev


def _finish():
    import faulthandler

    torch.cuda.synchronize()
    dist.barrier()
    faulthandler.cancel_dump_traceback_later()
    dist.destroy_process_group()


H10_CASE = {"grid": (3, 5, 7), "dim": 1280, "heads": 10, "ffn": 2560, "text_len": 77, "seed": 11}  # 5 heads per rank of 2: chunks of 3 + 2


def _worker(rank, world, port, out_dir, fused, fp8_comm=False, case=None, rccl=False, fp32_stream=False, extra_cfg=None, tag=""):
    dev = _init_rccl(port) if rccl else _init(rank, world, port)
    import lightx2v_amd as L
    from lightx2v_amd import ulysses
    from oracle import wan_block_ref as R
    from tests.util import case_inputs, load_golden

    rec = load_golden("small_d512_g3x5x7") if case is None else case  # S = 105 (odd: exercises the zero padding), H = 4
    c = case_inputs(rec)
    _, seq_idx, seq_group, _ = ulysses.make_groups(1, world)
    cfg = {"num_layers": 1, "num_heads": c["heads"], "dim": c["dim"], "task": "t2v", "seq_parallel": True, "seq_p_group": seq_group,
           "parallel": {"seq_p_size": world, "seq_p_attn_type": "hip_ulysses", "seq_p_fp8_comm": fp8_comm}, "fused_block": fused,
           "sensitive_layer_dtype": "fp32" if fp32_stream else "bf16"}
    cfg.update(extra_cfg or {})
    W = L.WanTransformerWeights(cfg)
    wd = dict(c["w"]); wd.update(c["hw"])
    W.load({k: v.to(dev) for k, v in wd.items()}); W.to_cuda()   # (the W8A8 schemes quantise on the GPU at load time)
    inf = L.WanTransformerInfer(cfg)
    S = c["S"]
    S_pad = S + (world - S % world) % world
    S_loc = S_pad // world
    table = L.build_rope_table(c["grid"], c["dim"] // c["heads"])
    inf.cos_sin = L.shard_rope_table(table, world, seq_idx).to(dev)
    x = torch.zeros(S_pad, c["dim"])
    x[:S] = c["x"]
    xl = x[seq_idx * S_loc:(seq_idx + 1) * S_loc].to(torch.bfloat16).to(dev).contiguous()
    pre = L.WanPreInferModuleOutput(embed=c["embed"].to(dev), grid_sizes=c["grid"], x=xl, embed0=c["embed0"].to(dev),
                                    context=c["context"].to(torch.bfloat16).to(dev))
    out = inf.infer_main_blocks(W.blocks, pre) if fp32_stream else inf.infer_block(W.blocks[0], pre.x, pre)
    assert out.dtype == (torch.float32 if fp32_stream else torch.bfloat16)
    torch.cuda.synchronize()
    torch.save(out.float().cpu(), os.path.join(out_dir, f"o{rank}_{int(fused)}{'_fp8' if fp8_comm else ''}{tag}.pt"))
    _finish()


@pytest.mark.parametrize("fused", [True, False])
def test_sequence_parallel_block_two_ranks_one_gpu(cuda, tmp_path, fused):
    from oracle import wan_block_ref as R
    from tests.util import case_inputs, cos, load_golden

    world = 2
    _spawn(_worker, lambda port: (world, port, str(tmp_path), fused), world)
    rec = load_golden("small_d512_g3x5x7")
    c = case_inputs(rec)
    got = torch.cat([torch.load(os.path.join(str(tmp_path), f"o{r}_{int(fused)}.pt")) for r in range(world)])[:c["S"]]
    # Oracle with the reference's sequence-parallel semantics: the table is zero-padded, so the one
    # padded token has q = k = 0 after RoPE but still takes part in the softmax as a key
    # (lightx2v/models/schedulers/wan/scheduler.py:106-113); real tokens therefore differ from the
    # un-padded run by that extra key, exactly as in the reference.
    S, D = c["S"], c["dim"]
    xp = torch.zeros(S + 1, D); xp[:S] = c["x"]
    cs = torch.cat([c["cos_sin"], torch.zeros(1, 1, c["cos_sin"].shape[-1], dtype=c["cos_sin"].dtype)])
    ref = R.block_forward(xp, c["embed0"], c["context"], c["w"], cs, c["heads"])[:S]
    assert cos(got, ref) >= 0.9999
    assert cos(got - c["x"], ref - c["x"]) >= 0.995


@pytest.mark.parametrize("fused", [True, False])
def test_sequence_parallel_block_uneven_head_chunks(cuda, tmp_path, fused):
    """10 heads on 2 ranks = 5 heads per rank, attended as chunks of 3 and 2 heads (the shape of
    Wan-14B at seq_p = 8): two all-to-alls of different widths and, in the fused block, an output
    projection that reads its operand from segments of two widths."""
    from oracle import wan_block_ref as R
    from tests.util import case_inputs, cos

    world = 2
    _spawn(_worker, lambda port: (world, port, str(tmp_path), fused, False, H10_CASE), world)
    c = case_inputs(H10_CASE)
    got = torch.cat([torch.load(os.path.join(str(tmp_path), f"o{r}_{int(fused)}.pt")) for r in range(world)])[:c["S"]]
    S, D = c["S"], c["dim"]
    xp = torch.zeros(S + 1, D); xp[:S] = c["x"]
    cs = torch.cat([c["cos_sin"], torch.zeros(1, 1, c["cos_sin"].shape[-1], dtype=c["cos_sin"].dtype)])
    ref = R.block_forward(xp, c["embed0"], c["context"], c["w"], cs, c["heads"])[:S]
    assert cos(got, ref) >= 0.9999
    assert cos(got - c["x"], ref - c["x"]) >= 0.995


@pytest.mark.parametrize("fused", [True, False])
def test_sequence_parallel_block_fp8_payload(cuda, tmp_path, fused):
    """`seq_p_fp8_comm`: q, k, v and the attention output cross the exchange as e4m3 + per-(token, head)
    scales.  Against the fp32 oracle of the unsharded block the result stays inside the fp8 payload's
    error (and differs from the bf16-payload run, i.e. the switch is live)."""
    from oracle import wan_block_ref as R
    from tests.util import case_inputs, cos, load_golden

    world = 2
    _spawn(_worker, lambda port: (world, port, str(tmp_path), fused, True), world)
    _spawn(_worker, lambda port: (world, port, str(tmp_path), fused, False), world)
    c = case_inputs(load_golden("small_d512_g3x5x7"))
    got8 = torch.cat([torch.load(os.path.join(str(tmp_path), f"o{r}_{int(fused)}_fp8.pt")) for r in range(world)])[:c["S"]]
    got = torch.cat([torch.load(os.path.join(str(tmp_path), f"o{r}_{int(fused)}.pt")) for r in range(world)])[:c["S"]]
    S, D = c["S"], c["dim"]
    xp = torch.zeros(S + 1, D); xp[:S] = c["x"]
    cs = torch.cat([c["cos_sin"], torch.zeros(1, 1, c["cos_sin"].shape[-1], dtype=c["cos_sin"].dtype)])
    ref = R.block_forward(xp, c["embed0"], c["context"], c["w"], cs, c["heads"])[:S]
    assert not torch.equal(got8, got)
    assert cos(got8, ref) >= 0.9999
    assert cos(got8 - c["x"], ref - c["x"]) >= 0.99


def _hy_worker(rank, world, port, out_dir, rccl=False, extra_cfg=None):
    dev = _init_rccl(port) if rccl else _init(rank, world, port)
    import lightx2v_amd as L
    from lightx2v_amd import ulysses
    from tests.test_hy_oracle import hy_case_inputs
    from tests.util import load_golden

    c = hy_case_inputs(load_golden("hy_tiny_d256"))  # 48 image tokens + 19 text tokens, H = 2
    _, seq_idx, seq_group, _ = ulysses.make_groups(1, world)
    cfg = {"mm_double_blocks_depth": c["depth"], "heads_num": c["heads"], "hidden_size": c["dim"], "seq_parallel": True, "seq_p_group": seq_group,
           "parallel": {"seq_p_size": world, "seq_p_attn_type": "hip_ulysses"}}
    cfg.update(extra_cfg or {})
    W = L.HunyuanVideo15TransformerWeights(cfg)
    wd = {}
    for b in c["blocks"]:
        wd.update(b)
    wd.update(c["fw"])
    W.load({k: v.to(dev) for k, v in wd.items()})
    inf = L.HunyuanVideo15TransformerInfer(cfg)
    sl = c["L"] // world
    rows = slice(seq_idx * sl, (seq_idx + 1) * sl)
    inf.scheduler.cos_sin = c["cos_sin"][rows].contiguous().to(dev)
    bf = lambda t: t.to(torch.bfloat16).to(dev)
    io = L.HunyuanVideo15InferModuleOutput(img=bf(c["img"][:, rows]).contiguous(), txt=bf(c["txt"]), vec=bf(c["vec"]), grid_sizes=c["grid"])
    out = inf.infer(W, io)  # both blocks + final layer on this rank's image rows
    torch.cuda.synchronize()
    torch.save({"final": out.float().cpu(), "txt": io.txt.float().cpu()}, os.path.join(out_dir, f"hy{rank}.pt"))
    _finish()


def test_hunyuan_double_blocks_sequence_parallel_two_ranks_one_gpu(cuda, tmp_path):
    """Image rows sharded over 2 ranks, text rows replicated: the gathered final-layer output
    equals the reference's unsharded run, and both ranks end with the same text stream."""
    from tests.util import cos, load_golden

    world = 2
    _spawn(_hy_worker, lambda port: (world, port, str(tmp_path)), world)
    res = [torch.load(os.path.join(str(tmp_path), f"hy{r}.pt")) for r in range(world)]
    rec = load_golden("hy_tiny_d256")
    got = torch.cat([r["final"][0] for r in res])
    assert cos(got, torch.from_numpy(rec["final_out"])) >= 0.9995
    assert cos(res[0]["txt"], res[1]["txt"]) >= 0.99999


# ---- WanModel.infer with cfg_parallel / seq_parallel (lightx2v/models/networks/wan/model.py:412-433,468-499) ----
def _model_worker(rank, world, port, out_dir, cfg_p, seq_p, chunks, rccl=False):
    os.environ["LX2V_ULYSSES_CHUNKS"] = str(chunks)
    dev = _init_rccl(port) if rccl else _init(rank, world, port)
    import lightx2v_amd as L
    from lightx2v_amd import ulysses
    from tests.test_step_oracle import step_case

    rec, meta, model = step_case("step_small_d512")   # 105 tokens: odd, so every seq_p > 1 pads
    cfg_idx, seq_idx, seq_group, cfg_group = ulysses.make_groups(cfg_p, seq_p)
    cfg = {"dim": meta["dim"], "num_heads": meta["heads"], "num_layers": meta["layers"], "ffn_dim": meta["ffn"], "in_dim": 16, "out_dim": 16,
           "freq_dim": 256, "task": "t2v", "infer_steps": meta["steps"], "sample_shift": meta["shift"], "sample_guide_scale": meta["guide"],
           "enable_cfg": True, "run_device": "cuda", "keep_latents_dtype_in_scheduler": True,
           "cfg_parallel": cfg_p > 1, "cfg_p_group": cfg_group, "seq_parallel": seq_p > 1 or rccl, "seq_p_group": seq_group,
           "parallel": {"seq_p_size": seq_p, "cfg_p_size": cfg_p, "seq_p_attn_type": "hip_ulysses"}}
    m = L.WanModel(cfg, weight_dict=dict(model))
    m.to_cuda()
    sch = L.WanScheduler(cfg)
    m.set_scheduler(sch)
    sch.prepare(meta["seed"], meta["latent_shape"], latents=torch.from_numpy(rec["latents0"]))
    inputs = {"text_encoder_output": {k: torch.from_numpy(rec[k]).to(dev) for k in ("context", "context_null")}}
    preds = []
    for i in range(2):   # two steps: the second runs on latents the first produced
        sch.step_pre(i)
        m.infer(inputs)
        preds.append(sch.noise_pred.float().cpu())
        sch.step_post()
    torch.save({"preds": preds, "latents": sch.latents.float().cpu()}, os.path.join(out_dir, f"m{rank}.pt"))
    _finish()


@pytest.mark.parametrize("cfg_p,seq_p,chunks", [(2, 1, 2), (1, 2, 2), (2, 2, 2), (2, 2, 1)])
def test_model_infer_with_cfg_and_sequence_parallel_groups(cuda, tmp_path, cfg_p, seq_p, chunks):
    """`WanModel.infer` on a (cfg_p, seq_p) grid of ranks: the CFG all_gather, the pad / chunk of the
    tokens, Ulysses inside every block and the final all_gather.  Every rank must end with the same
    noise prediction, equal to the reference's golden trajectory within the step tolerance (with
    seq_p > 1 the one padded token takes part as a key with q = k = 0, as in the reference)."""
    import numpy as np
    from tests.test_step_oracle import step_case
    from tests.util import cos

    world = cfg_p * seq_p
    _spawn(_model_worker, lambda port: (world, port, str(tmp_path), cfg_p, seq_p, chunks), world)
    rec, meta, _ = step_case("step_small_d512")
    res = [torch.load(os.path.join(str(tmp_path), f"m{r}.pt")) for r in range(world)]
    for i in range(2):
        gold = torch.from_numpy(np.asarray(rec[f"noise_pred_{i}"]))
        for r in range(world):
            assert cos(res[r]["preds"][i], gold) >= 0.999, (i, r, cos(res[r]["preds"][i], gold))
            assert torch.equal(res[r]["preds"][i], res[0]["preds"][i]), (i, r)   # all ranks hold the combined prediction
    assert cos(res[0]["latents"], torch.from_numpy(np.asarray(rec["latents_2"]))) >= 0.9995


def _hy_model_worker(rank, world, port, out_dir, name, cfg_p, seq_p):
    dev = _init(rank, world, port)
    import lightx2v_amd as L
    from lightx2v_amd import ulysses
    from tests.test_hy_step_gpu import _model, _to_cuda
    from tests.test_hy_step_oracle import hy_step_case

    rec, meta, pre_w, blocks, fw, inputs = hy_step_case(name)
    _, _, seq_group, cfg_group = ulysses.make_groups(cfg_p, seq_p)
    m, sch = _model(meta, pre_w, blocks, fw, dev, cfg_parallel=cfg_p > 1, cfg_p_group=cfg_group, seq_parallel=seq_p > 1, seq_p_group=seq_group,
                    parallel={"seq_p_size": seq_p, "cfg_p_size": cfg_p, "seq_p_attn_type": "hip_ulysses"})
    dev_inputs = _to_cuda(inputs, dev)
    sch.prepare(meta["seed"], meta["latent_shape"], image_encoder_output=dev_inputs["image_encoder_output"], latents=torch.from_numpy(rec["latents0"]))
    preds = []
    for i in range(2):
        sch.step_pre(i)
        m.infer(dev_inputs)
        preds.append(sch.noise_pred.float().cpu())
        sch.step_post()
    torch.save({"preds": preds, "latents": sch.latents.float().cpu()}, os.path.join(out_dir, f"h{rank}.pt"))
    _finish()


@pytest.mark.parametrize("name,cfg_p,seq_p", [("hy_step_distill_d256", 1, 2), ("hy_step_cfg_i2v_d256", 2, 1), ("hy_step_cfg_i2v_d256", 2, 2)])
def test_hunyuan_model_infer_with_cfg_and_sequence_parallel_groups(cuda, tmp_path, name, cfg_p, seq_p):
    """`HunyuanVideo15Model.infer` on a (cfg_p, seq_p) grid: the image tokens are chunked over the
    sequence group while every rank keeps the whole text, Ulysses runs inside every double-stream
    block, the final layer's rows are gathered, the two CFG branches are gathered and combined."""
    import numpy as np
    from tests.test_hy_step_oracle import hy_step_case
    from tests.util import cos

    world = cfg_p * seq_p
    _spawn(_hy_model_worker, lambda port: (world, port, str(tmp_path), name, cfg_p, seq_p), world)
    rec, meta, *_ = hy_step_case(name)
    res = [torch.load(os.path.join(str(tmp_path), f"h{r}.pt")) for r in range(world)]
    for i in range(2):
        gold = torch.from_numpy(np.asarray(rec[f"noise_pred_{i}"]))
        for r in range(world):
            assert cos(res[r]["preds"][i], gold) >= 0.999, (i, r, cos(res[r]["preds"][i], gold))
            assert torch.equal(res[r]["preds"][i], res[0]["preds"][i]), (i, r)
    assert cos(res[0]["latents"], torch.from_numpy(np.asarray(rec["latents_2"]))) >= 0.9995


# ---- the same paths over RCCL (one rank: the one-GPU box cannot hold two) ----
@pytest.mark.parametrize("fused,fp8_comm,fp32_stream", [(True, False, False), (False, False, False), (True, True, False), (True, False, True)])
def test_sequence_parallel_block_over_rccl_one_rank(cuda, tmp_path, fused, fp8_comm, fp32_stream):
    """The block with `seq_parallel` on a sequence group of one rank whose collectives are RCCL's:
    the head chunks (the one rank's 10 heads in chunks of 5 and 5), the asynchronous all-to-alls,
    their waits and the segmented output projection run against the production transport.  A group of one pads nothing, so the result is the unsharded block's."""
    from oracle import wan_block_ref as R
    from tests.util import case_inputs, cos

    _spawn(_worker, lambda port: (1, port, str(tmp_path), fused, fp8_comm, H10_CASE, True, fp32_stream), 1)   # fp32_stream: the segmented output projection feeds an fp32 residual
    c = case_inputs(H10_CASE)
    got = torch.load(os.path.join(str(tmp_path), f"o0_{int(fused)}{'_fp8' if fp8_comm else ''}.pt"))
    ref = R.block_forward(c["x"], c["embed0"], c["context"], c["w"], c["cos_sin"], c["heads"])
    assert got.shape == ref.shape
    assert cos(got, ref) >= 0.9999
    assert cos(got - c["x"], ref - c["x"]) >= (0.99 if fp8_comm else 0.995)


def test_model_infer_over_rccl_one_rank(cuda, tmp_path):
    """`WanModel.infer` with `seq_parallel` on an RCCL group of one rank: pad / chunk, Ulysses in every
    block and the final all_gather on the production transport, against the golden trajectory."""
    import numpy as np
    from tests.test_step_oracle import step_case
    from tests.util import cos

    _spawn(_model_worker, lambda port: (1, port, str(tmp_path), 1, 1, 3, True), 1)
    rec, meta, _ = step_case("step_small_d512")
    res = torch.load(os.path.join(str(tmp_path), "m0.pt"))
    for i in range(2):
        gold = torch.from_numpy(np.asarray(rec[f"noise_pred_{i}"]))
        assert cos(res["preds"][i], gold) >= 0.999, (i, cos(res["preds"][i], gold))
    assert cos(res["latents"], torch.from_numpy(np.asarray(rec["latents_2"]))) >= 0.9995


def test_hunyuan_double_blocks_over_rccl_one_rank(cuda, tmp_path):
    """The HunyuanVideo layout (image rows exchanged, text rows gathered along the head axis with a
    list-output all_gather) on an RCCL group of one rank, against the reference's unsharded run."""
    from tests.util import cos, load_golden

    _spawn(_hy_worker, lambda port: (1, port, str(tmp_path), True), 1)
    res = torch.load(os.path.join(str(tmp_path), "hy0.pt"))
    rec = load_golden("hy_tiny_d256")
    assert cos(res["final"][0], torch.from_numpy(rec["final_out"])) >= 0.9995


def _full_size_worker(rank, world, port, out_dir):
    dev = _init_rccl(port)
    from lightx2v_amd import comm, ulysses
    from lightx2v_amd.registry_factory import ATTN_WEIGHT_REGISTER

    S, H, d = 75600, 40, 128
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.randn(S, H, d, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
    plain, uly = ATTN_WEIGHT_REGISTER["hip_flash_attn"](), ATTN_WEIGHT_REGISTER["hip_ulysses"]()
    group = dist.new_group(ranks=[0])
    ref = plain.apply(q, k, v, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=S, max_seqlen_kv=S)
    got = uly.apply(q, k, v, img_qkv_len=S, cu_seqlens_qkv=[0, S], attention_module=plain, seq_p_group=group)
    torch.cuda.synchronize()
    refused = False
    big = torch.empty((1 << 30) + 4096, dtype=torch.uint8, device=dev)
    try:
        comm.transport().all_to_all_single(torch.empty_like(big), big, group=group)
    except ValueError:
        refused = True
    torch.save({"max_diff": float((got.float() - ref.float()).abs().max()), "max_ref": float(ref.float().abs().max()),
                "widths": list(ulysses.LAST_WIDTHS), "refused": refused}, os.path.join(out_dir, "full.pt"))
    _finish()


def test_full_size_exchange_over_rccl_one_rank(cuda, tmp_path):
    """75 600 tokens x 40 heads through the exchange on an RCCL group of one rank: 2.3 GB of q|k|v, which
    two head chunks would hand RCCL as 1.16 GB per peer -- a size it moves only in part (the first half
    arrives, no error).  The exchange must split into chunks below the limit and return the plain
    attention's result; the transport must refuse a payload above 1 GiB outright."""
    _spawn(_full_size_worker, lambda port: (1, port, str(tmp_path)), 1)
    res = torch.load(os.path.join(str(tmp_path), "full.pt"))
    assert res["widths"] == [8] * 5
    # launches of 8 heads and one launch of 40 differ at most in the last bf16 bit of a few outputs
    assert res["max_diff"] <= 2.0 ** -8 * res["max_ref"], res
    assert res["refused"]


# ---- BASELINE config 5's composition: W8A8 fp8 Linears and the fp8 attention under sequence parallelism ----
# (lightx2v/common/ops/attn/ulysses_attn.py:80-118 around a quantised attention module; lightx2v/common/ops/mm/mm_weight.py:901-928
# behind the output projection that consumes the exchanged attention result)
FP8_MODES = {"fp8_linears": {"dit_quant_scheme": "hip-fp8"}, "fp8_attn": {"self_attn_1_type": "hip_fp8_attn"},
             "fp8_both": {"dit_quant_scheme": "hip-fp8", "self_attn_1_type": "hip_fp8_attn"}}


def _unsharded_wan_block(c, dev, extra_cfg, pad_to):
    """The same block in this process without seq_parallel, on the token rows zero-padded to `pad_to`
    (the padded rows' RoPE entries are zero, like the sharded table's)."""
    import lightx2v_amd as L

    cfg = {"num_layers": 1, "num_heads": c["heads"], "dim": c["dim"], "task": "t2v", "fused_block": True}
    cfg.update(extra_cfg)
    W = L.WanTransformerWeights(cfg)
    W.load({k: v.to(dev) for k, v in {**c["w"], **c["hw"]}.items()})
    W.to_cuda()
    inf = L.WanTransformerInfer(cfg)
    table = L.build_rope_table(c["grid"], c["dim"] // c["heads"])
    table = torch.cat([table, table.new_zeros(pad_to - c["S"], table.shape[1])]) if pad_to > c["S"] else table
    inf.cos_sin = table.contiguous().to(dev)
    x = torch.zeros(pad_to, c["dim"])
    x[:c["S"]] = c["x"]
    pre = L.WanPreInferModuleOutput(embed=c["embed"].to(dev), grid_sizes=c["grid"], x=x.to(torch.bfloat16).to(dev), embed0=c["embed0"].to(dev),
                                    context=c["context"].to(torch.bfloat16).to(dev))
    out = inf.infer_block(W.blocks[0], pre.x, pre)
    torch.cuda.synchronize()
    return out.float().cpu()[:c["S"]]


@pytest.mark.parametrize("mode", sorted(FP8_MODES))
@pytest.mark.parametrize("rccl", [False, True])
def test_sequence_parallel_block_with_fp8_linears_and_fp8_attention(cuda, tmp_path, mode, rccl):
    """The Wan block with `hip-fp8` Linears and / or `hip_fp8_attn` under `seq_parallel`: two ranks on
    the host-staged transport (10 heads -> chunks of 3 + 2 heads per rank, one padded token), and one
    rank on RCCL (chunks of 5 + 5).  The fp8 attention plugin re-sizes its workspace per chunk shape;
    the fp8 output projection takes the gathered operand (its activation quantiser needs whole rows)
    and the LayerNorms hand e4m3 rows to sharded consumers.  Held to the unsharded run of the SAME
    arithmetic (head chunks change the launch plan, not the per-head arithmetic: cosine >= 0.99999,
    delta cosine >= 0.9995) and to the fp32 oracle within the fp8 gate of the unsharded tests."""
    from oracle import wan_block_ref as R
    from tests.util import case_inputs, cos

    extra = FP8_MODES[mode]
    world = 1 if rccl else 2
    _spawn(_worker, lambda port: (world, port, str(tmp_path), True, False, H10_CASE, rccl, False, extra, "_" + mode), world)
    c = case_inputs(H10_CASE)
    S, D = c["S"], c["dim"]
    got = torch.cat([torch.load(os.path.join(str(tmp_path), f"o{r}_1_{mode}.pt")) for r in range(world)])[:S]
    pad_to = S + (world - S % world) % world
    same = _unsharded_wan_block(c, cuda, extra, pad_to)
    c_same, c_same_delta = cos(got, same), cos(got - c["x"], same - c["x"])
    xp = torch.zeros(pad_to, D); xp[:S] = c["x"]
    cs = torch.cat([c["cos_sin"], torch.zeros(pad_to - S, 1, c["cos_sin"].shape[-1], dtype=c["cos_sin"].dtype)])
    ref = R.block_forward(xp, c["embed0"], c["context"], c["w"], cs, c["heads"])[:S]
    c_out, c_delta = cos(got, ref), cos(got - c["x"], ref - c["x"])
    print(f"\nseq-parallel block, {mode}, {'RCCL x1' if rccl else 'host-staged x2'}: vs unsharded same arithmetic {c_same:.7f} / delta {c_same_delta:.6f}; vs fp32 oracle {c_out:.6f} / delta {c_delta:.6f}")
    assert c_same >= 0.99999 and c_same_delta >= 0.9995
    assert c_out >= 0.999 and c_delta >= 0.99


@pytest.mark.parametrize("mode", ["fp8_linears", "fp8_attn", "fp8_both"])
@pytest.mark.parametrize("rccl", [False, True])
def test_hunyuan_double_blocks_sequence_parallel_with_fp8(cuda, tmp_path, mode, rccl):
    """BASELINE config 5's arithmetic on the HunyuanVideo layout: image rows exchanged, the rank's text
    rows appended to every head chunk the fp8 attention attends, text outputs gathered along the head
    axis, fp8 Linears on both streams.  Against the unsharded run of the same arithmetic and the
    reference's fp32 golden."""
    import lightx2v_amd as L
    from tests.test_hy_gpu import _build, _io
    from tests.test_hy_oracle import hy_case_inputs
    from tests.util import cos, load_golden

    extra = {"fp8_linears": {"dit_quant_scheme": "hip-fp8"}, "fp8_attn": {"attn_type": "hip_fp8_attn"},
             "fp8_both": {"dit_quant_scheme": "hip-fp8", "attn_type": "hip_fp8_attn"}}[mode]
    world = 1 if rccl else 2
    _spawn(_hy_worker, lambda port: (world, port, str(tmp_path), rccl, extra), world)
    res = [torch.load(os.path.join(str(tmp_path), f"hy{r}.pt")) for r in range(world)]
    rec = load_golden("hy_tiny_d256")
    got = torch.cat([r["final"][0] for r in res])
    c = hy_case_inputs(rec)
    W, inf = _build(c, cuda, **extra)
    same = inf.infer(W, _io(c, cuda))[0].float().cpu()
    gold = torch.from_numpy(rec["final_out"])
    print(f"\nHunyuanVideo seq-parallel, {mode}, {'RCCL x1' if rccl else 'host-staged x2'}: vs unsharded same arithmetic {cos(got, same):.6f}; vs reference golden {cos(got, gold):.6f} (unsharded: {cos(same, gold):.6f})")
    assert cos(got, same) >= 0.9999
    assert cos(got, gold) >= 0.995
    if world > 1:
        assert cos(res[0]["txt"], res[1]["txt"]) >= 0.9999


# ---- per-token modulation under sequence parallelism (lightx2v/models/networks/wan/model.py:480-489) ----
def _ti2v_worker(rank, world, port, out_dir):
    dev = _init(rank, world, port)
    import lightx2v_amd as L
    from lightx2v_amd import ulysses
    from tests.test_step_oracle import plain_case

    rec, meta, (model,) = plain_case("step_ti2v_d256")
    _, _, seq_group, _ = ulysses.make_groups(1, world)
    cfg = {"dim": meta["dim"], "num_heads": meta["heads"], "num_layers": meta["layers"], "ffn_dim": meta["ffn"], "in_dim": 16, "out_dim": 16,
           "freq_dim": 256, "task": "i2v", "model_cls": "wan2.2", "use_image_encoder": False, "infer_steps": meta["steps"], "sample_shift": meta["shift"],
           "sample_guide_scale": meta["guide"], "enable_cfg": True, "run_device": "cuda", "keep_latents_dtype_in_scheduler": True,
           "seq_parallel": True, "seq_p_group": seq_group, "parallel": {"seq_p_size": world, "seq_p_attn_type": "hip_ulysses"}}
    m = L.WanModel(cfg, weight_dict=dict(model))
    m.to_cuda()
    sch = L.WanScheduler(cfg)
    m.set_scheduler(sch)
    z = torch.from_numpy(rec["z"])
    inputs = {"text_encoder_output": {k: torch.from_numpy(rec[k]).to(dev) for k in ("context", "context_null")},
              "image_encoder_output": {"vae_encoder_out": z.to(dev)}}
    sch.prepare(meta["seed"], meta["latent_shape"], image_encoder_output=inputs["image_encoder_output"], latents=torch.from_numpy(rec["noise0"]))
    inputs["image_encoder_output"]["vae_encoder_out"] = None
    shapes = []
    real = m._seq_parallel_pre_process

    def spy(pre):
        out = real(pre)
        shapes.append((tuple(out.x.shape), tuple(out.embed.shape), tuple(out.embed0.shape)))
        return out

    m._seq_parallel_pre_process = spy
    preds = []
    for i in range(meta["steps"]):
        sch.step_pre(i)
        m.infer(inputs)
        preds.append(sch.noise_pred.float().cpu())
        sch.step_post()
    torch.save({"preds": preds, "latents": sch.latents.float().cpu(), "shapes": shapes}, os.path.join(out_dir, f"t{rank}.pt"))
    _finish()


def test_ti2v_per_token_modulation_is_sharded_with_the_tokens(cuda, tmp_path):
    """Wan2.2 dense ti2v with an image runs a per-token timestep: embed [S, D] and embed0 [S, 6, D] must
    be padded and chunked over the sequence group together with x, or every rank would hand S-row
    modulation tensors to kernels working on S/N rows.  Two ranks against the reference's golden trajectory."""
    import numpy as np
    from tests.test_step_oracle import plain_case
    from tests.util import cos, rel_err

    world = 2
    _spawn(_ti2v_worker, lambda port: (world, port, str(tmp_path)), world)
    rec, meta, _ = plain_case("step_ti2v_d256")
    res = [torch.load(os.path.join(str(tmp_path), f"t{r}.pt")) for r in range(world)]
    for xs, es, e0s in res[0]["shapes"]:
        assert es[0] == xs[0] and e0s == (xs[0], 6, xs[1]), (xs, es, e0s)
    for i in range(meta["steps"]):
        gold = torch.from_numpy(np.asarray(rec[f"noise_pred_{i}"]))
        for r in range(world):
            assert cos(res[r]["preds"][i], gold) >= 0.999, (i, r, cos(res[r]["preds"][i], gold))
            assert torch.equal(res[r]["preds"][i], res[0]["preds"][i]), (i, r)
    final = torch.from_numpy(np.asarray(rec[f"latents_{meta['steps']}"]))
    assert rel_err(res[0]["latents"], final) <= 3e-2 and cos(res[0]["latents"], final) >= 0.9995
