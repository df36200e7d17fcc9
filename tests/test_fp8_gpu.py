"""GPU: fp8 (OCP e4m3) W8A8 Linear -- row-wise quantisation kernel and the fp8-MFMA GEMM -- against
the restated formula of the reference's "fp8-sgl" scheme (oracle.wan_block_ref.quant_fp8_rowwise /
linear_fp8; parity unpinned: the reference's kernels for it live in absent wheels)."""
import pytest
import torch

import lightx2v_amd as L
from lightx2v_amd import hip
from oracle import wan_block_ref as R
from tests.util import case_inputs, cos, load_golden

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("rows,dim", [(1, 128), (7, 256), (130, 1536), (33, 5120), (5, 13824), (9, 520)])
def test_quant_fp8_rowwise_bit_exact(cuda, rows, dim):
    x = _rand((rows, dim), 70 + rows)
    x[0, :3] = torch.tensor([448.0, -1000.0, 3.0e-3])  # a row whose maximum is far above the rest
    if rows > 2:
        x[2] = 0  # all-zero row: scale floor, zeros out
    q_ref, s_ref = R.quant_fp8_rowwise(x)
    q, s = hip.quant_fp8_rowwise(x.to(cuda))
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))


@pytest.mark.parametrize("rows,dim", [(1000, 128), (37, 64), (300, 256), (5, 8), (129, 1536)])
def test_short_row_quant_and_dequant_bit_exact(cuda, rows, dim):
    """Rows of one head (the fp8 Ulysses payload): several rows share a wave; the inverse widens
    with the bf16-rounded scale."""
    x = _rand((rows, dim), 300 + rows)
    x[rows // 2] = 0
    q_ref, s_ref = R.quant_fp8_rowwise(x)
    q, s = hip.quant_fp8_rowwise(x.to(cuda))
    assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))
    back = hip.dequant_fp8_rowwise(q, s)
    assert torch.equal(back.cpu(), R.dequant_fp8_rowwise(q_ref, s_ref, torch.bfloat16))
    # into a strided destination (a column range of a wider buffer)
    buf = torch.zeros(rows, dim + 16, dtype=torch.bfloat16, device=cuda)
    hip.dequant_fp8_rowwise(q, s, out=buf[:, :dim])
    assert torch.equal(buf[:, :dim], back) and buf[:, dim:].abs().max().item() == 0


@pytest.mark.parametrize("M,N,K", [(16, 64, 128), (300, 512, 256), (1000, 1536, 1536), (257, 264, 384), (64, 5120, 13824)])
def test_gemm_fp8_matches_quantised_domain_product(cuda, M, N, K):
    a, w = _rand((M, K), 80), _rand((N, K), 81, 0.05)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(82)) * 0.1
    aq, a_s = R.quant_fp8_rowwise(a)
    wq, w_s = R.quant_fp8_rowwise(w)
    ref = (aq.double() @ wq.double().t()) * a_s.double()[:, None] * w_s.double()[None, :] + bias.double()
    out = hip.gemm_fp8(aq.to(cuda), a_s.to(cuda), wq.to(cuda), w_s.to(cuda), bias.to(cuda))
    # products of e4m3 values are exact in fp32; what separates the two is the fp32 accumulation
    # order and the final bf16 rounding
    tol = 2.0 ** -8 * ref.abs().max().item() + 1e-3
    assert (out.double().cpu() - ref).abs().max().item() <= tol
    # and the whole scheme (quantise + multiply) stays close to the unquantised product
    assert cos(out, R.linear(a.float(), w.float(), bias)) >= 0.999


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("M,N,K,epi", [(3000, 2048, 2048, "none"), (1111, 1536, 8960, "gelu"), (2000, 5120, 5120, "resid"), (700, 512, 384, "none"),
                                       (257, 264, 128, "resid"),        # one K-tile per output tile, ragged M and N
                                       (33, 8, 256, "gelu"),            # a single, mostly empty tile
                                       (20000, 1536, 640, "none"),      # odd number of K-tiles (5), 474 tiles: two per workgroup, the last round partial
                                       (75600, 5120, 5120, "resid"),    # the 14B output projection: 5920 tiles, 23-24 per workgroup of the persistent kernel
                                       (40000, 13824, 2048, "gelu")])   # many tiles of few K-tiles each (16): epilogue-heavy
def test_gemm_w8a8_schedules_agree_bitwise(cuda, M, N, K, epi, int8, options):
    """The persistent one-wave-per-SIMD kernel (the default), the one-barrier 8-wave kernel and (fp8) the
    ping-pong schedule accumulate each output in the same order and share the epilogue arithmetic:
    bit-identical results, across repeated launches with other traffic in flight."""
    g = torch.Generator(device=cuda).manual_seed(60 + M % 97)
    a = torch.randn(M, K, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=cuda, dtype=torch.float32) * 0.05).to(torch.bfloat16)
    quant = hip.quant_int8_rowwise if int8 else hip.quant_fp8_rowwise
    gemm = hip.gemm_int8 if int8 else hip.gemm_fp8
    aq, a_s = quant(a)
    wq, w_s = quant(w)
    bias = torch.randn(N, generator=g, device=cuda) * 0.1
    gate = torch.randn(N, generator=g, device=cuda)
    resid = torch.randn(M, N, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16) if epi == "resid" else None
    del a, w
    args = (aq, a_s, wq, w_s, bias)

    def run():
        if epi == "gelu":
            return gemm(*args, epilogue=hip.EPI_GELU_TANH)
        if epi == "resid":
            return gemm(*args, epilogue=hip.EPI_RESIDUAL, resid=resid.clone(), gate=gate)
        return gemm(*args)

    options(gemm_kernel=1)  # one barrier per K-tile, 8 waves
    ref = run()
    assert bool(torch.isfinite(ref).all())
    noise = torch.empty(64 << 20, dtype=torch.uint8, device=cuda)
    for kernel in (0, 2) if not int8 else (0,):   # persistent (default); ping-pong (fp8 only)
        options(gemm_kernel=kernel)
        for i in range(4):
            if i % 2:
                noise.fill_(i)
            got = run()
            assert torch.equal(got, ref), (kernel, i, int((got != ref).sum()), float((got.float() - ref.float()).abs().max()))


def test_gemm_fp8_epilogues_and_plugin(cuda):
    M, N, K = 300, 512, 256
    a, w = _rand((M, K), 90), _rand((N, K), 91, 0.05)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(92)) * 0.1
    gate = torch.randn(N, generator=torch.Generator().manual_seed(93))
    resid = _rand((M, N), 94)
    mm = L.MM_WEIGHT_REGISTER["hip-fp8"]("w.weight", "w.bias")
    mm.load({"w.weight": w.to(cuda), "w.bias": bias.to(cuda)})
    wq, w_s = R.quant_fp8_rowwise(w)
    assert torch.equal(mm.weight_nk.cpu().view(torch.uint8), wq.view(torch.uint8)) and torch.equal(mm.weight_scale.cpu(), w_s)
    ref = R.linear_fp8(a, wq, w_s, bias)
    assert (mm.apply(a.to(cuda)).float().cpu() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-3
    g = torch.nn.functional.gelu(ref, approximate="tanh")
    assert (mm.apply_gelu(a.to(cuda)).float().cpu() - g).abs().max().item() <= 2.0 ** -7 * g.abs().max().item() + 1e-3
    r = resid.to(cuda).clone()
    mm.apply_residual_(a.to(cuda), r, gate=gate.to(cuda))
    rr = resid.float() + ref * gate
    assert (r.float().cpu() - rr).abs().max().item() <= 2.0 ** -7 * rr.abs().max().item() + 1e-3
    # a pre-quantised checkpoint loads as is
    mm2 = L.MM_WEIGHT_REGISTER["hip-fp8"]("w.weight", "w.bias")
    mm2.load({"w.weight": wq.to(cuda), "w.weight_scale": w_s.to(cuda), "w.bias": bias.to(cuda)})
    assert torch.equal(mm2.apply(a.to(cuda)), mm.apply(a.to(cuda)))


@pytest.mark.parametrize("fused", [False, True])
def test_block_with_fp8_linears(cuda, fused):
    """The whole block with every Linear in W8A8 fp8 (both runners) against the fp32 oracle."""
    rec = load_golden("wan1p3b_g3x6x8")
    c = case_inputs(rec)
    cfg = {"num_layers": 1, "num_heads": c["heads"], "dim": c["dim"], "task": "t2v", "dit_quant_scheme": "hip-fp8", "fused_block": fused}
    W = L.WanTransformerWeights(cfg)
    wd = {k: v.to(cuda) for k, v in {**c["w"], **c["hw"]}.items()}
    W.load(wd)
    W.to_cuda()
    inf = L.WanTransformerInfer(cfg)
    inf.cos_sin = R.cos_sin_to_real(c["cos_sin"]).contiguous().to(cuda)
    pre = L.WanPreInferModuleOutput(embed=c["embed"].to(cuda), grid_sizes=c["grid"], x=c["x"].to(torch.bfloat16).to(cuda),
                                    embed0=c["embed0"].to(cuda), context=c["context"].to(torch.bfloat16).to(cuda))
    inf.get_scheduler_values = lambda: None
    out = inf.infer_main_blocks(W.blocks, pre)
    ref = R.block_forward(c["x"], c["embed0"], c["context"], c["w"], c["cos_sin"], c["heads"])
    c_out, c_delta = cos(out, ref), cos(out.float().cpu() - c["x"], ref - c["x"])
    print(f"\nfp8 block: output cosine {c_out:.6f}, delta cosine {c_delta:.6f}")
    assert c_out >= 0.999 and c_delta >= 0.99


@pytest.mark.parametrize("rows,dim", [(1, 128), (37, 1536), (130, 2048), (21, 5120), (9, 8192)])
@pytest.mark.parametrize("mode", ["plain", "affine", "modulated", "per_token"])
def test_ln_modulate_quantised_output_is_the_quantised_bf16_output(cuda, rows, dim, mode):
    """The LayerNorm that hands e4m3 rows + row scales to a W8A8 Linear: byte for byte what
    quantising its bf16 output gives (and so, through the bit-exact quantisation test above, what the
    restated reference formula gives for that output)."""
    g = torch.Generator().manual_seed(rows * 31 + dim)
    x = (torch.randn(rows, dim, generator=g) * 3 + 0.5).to(torch.bfloat16).to(cuda)
    ln_w = ln_b = scale = shift = None
    if mode == "affine":
        ln_w, ln_b = (1 + 0.1 * torch.randn(dim, generator=g)).to(cuda), (0.1 * torch.randn(dim, generator=g)).to(cuda)
    if mode in ("modulated", "per_token"):
        scale, shift = (0.3 * torch.randn(dim, generator=g)).to(cuda), (0.3 * torch.randn(dim, generator=g)).to(cuda)
    if mode == "per_token":
        rows6 = (0.2 * torch.randn(rows, 6 * dim, generator=g)).to(torch.bfloat16).to(cuda)
        scale, shift = hip.RowMod(scale, rows6[:, dim:2 * dim]), hip.RowMod(shift, rows6[:, :dim])
    plain = hip.ln_modulate(x, ln_w, ln_b, scale, shift)
    q_ref, s_ref = hip.quant_fp8_rowwise(plain)
    got = hip.ln_modulate(x, ln_w, ln_b, scale, shift, quant=True)
    assert isinstance(got, hip.QuantRows) and got.q.dtype == torch.float8_e4m3fn and tuple(got.shape) == (rows, dim)
    assert torch.equal(got.scale, s_ref)
    assert torch.equal(got.q.view(torch.uint8), q_ref.view(torch.uint8))
    with pytest.raises(hip.Lx2vError):
        hip.ln_modulate(x, ln_w, ln_b, scale, shift, quant=True, out=plain)


def test_fp8_block_is_unchanged_by_quantising_inside_the_layernorm(cuda, monkeypatch):
    """The fused runner lets the three LayerNorms quantise for their W8A8 consumers; the block
    output is bit-identical to the one with separate quantisation passes, and the LayerNorms really
    hand over quantised rows."""
    rec = load_golden("wan1p3b_g3x6x8")
    c = case_inputs(rec)
    cfg = {"num_layers": 1, "num_heads": c["heads"], "dim": c["dim"], "task": "t2v", "dit_quant_scheme": "hip-fp8", "fused_block": True}
    W = L.WanTransformerWeights(cfg)
    W.load({k: v.to(cuda) for k, v in {**c["w"], **c["hw"]}.items()})
    W.to_cuda()
    inf = L.WanTransformerInfer(cfg)
    inf.cos_sin = R.cos_sin_to_real(c["cos_sin"]).contiguous().to(cuda)
    inf.get_scheduler_values = lambda: None

    def run():
        pre = L.WanPreInferModuleOutput(embed=c["embed"].to(cuda), grid_sizes=c["grid"], x=c["x"].to(torch.bfloat16).to(cuda),
                                        embed0=c["embed0"].to(cuda), context=c["context"].to(torch.bfloat16).to(cuda))
        return inf.infer_main_blocks(W.blocks, pre).clone()

    seen = []
    real = hip.ln_modulate

    def spy(*a, **k):
        out = real(*a, **k)
        seen.append(type(out).__name__)
        return out

    monkeypatch.setattr(hip, "ln_modulate", spy)
    folded = run()
    assert seen == ["QuantRows"] * 3
    monkeypatch.setattr(L.MM_WEIGHT_REGISTER["hip-fp8"], "accepts_quantized", False)
    seen.clear()
    separate = run()
    assert seen == ["Tensor"] * 3
    assert torch.equal(folded, separate)


@pytest.mark.parametrize("widths,world", [([2], 2), ([3, 2], 2), ([1, 1, 1], 4), ([5], 1)])
def test_rowwise_quantisers_read_a_segmented_source(cuda, widths, world):
    """The sequence-parallel attention result in its receive layout (per head chunk [source rank][rows][w*d],
    chunks of up to two widths) quantised row-wise straight from the segments: byte for byte and scale for
    scale what quantising the gathered [rows, H*d] copy gives -- fp8 and int8 -- and the W8A8 output
    projection fed with the segmented operand equals the one fed with the copy."""
    from lightx2v_amd.ulysses import SegmentedOperand

    sl, d = 37, 128
    dim = world * sum(widths) * d
    flat = _rand((sl * dim,), 500 + len(widths)).to(cuda)
    op = SegmentedOperand(flat, sl, world, widths, d)
    plain = op.gathered().contiguous()
    assert plain.shape == (sl, dim)
    for quant in (hip.quant_fp8_rowwise, hip.quant_int8_rowwise):
        q_ref, s_ref = quant(plain)
        q, s = quant(op.first, kseg=op.kseg, dim=dim)
        assert torch.equal(s, s_ref)
        assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8))
    for name in ("hip-fp8", "hip-int8"):
        mm = L.MM_WEIGHT_REGISTER[name]("w.weight", "w.bias")
        mm.load({"w.weight": _rand((256, dim), 77, 0.05).to(cuda), "w.bias": torch.randn(256, generator=torch.Generator().manual_seed(5)).to(cuda)})
        assert mm.supports_kseg
        x0 = _rand((sl, 256), 78).to(cuda)
        a, b = x0.clone(), x0.clone()
        mm.apply_residual_(plain, a, None)
        mm.apply_residual_(op.first, b, None, kseg=op.kseg)
        assert torch.equal(a, b)
    with pytest.raises(hip.Lx2vError):
        hip.quant_fp8_rowwise(op.first, kseg=op.kseg, dim=dim + 8)   # the segments do not tile this many columns


def test_fp8_weight_quantised_from_fp32_rows(cuda):
    """An fp32 (or fp16) checkpoint weight goes into the load-time quantiser with its own values (the reference hands
    its quantiser `weight.to(torch.float32)`, mm_weight.py:312-320): bit for bit the restated formula on fp32 input, and
    not what a bf16 rounding in front would give."""
    w = 0.02 * torch.randn(48, 384, generator=torch.Generator().manual_seed(12))
    q_ref, s_ref = R.quant_fp8_rowwise(w)
    q, s = hip.quant_fp8_rowwise(w.to(cuda))
    assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))
    assert not torch.equal(R.quant_fp8_rowwise(w.to(torch.bfloat16))[1], s_ref)
    mm = L.MM_WEIGHT_REGISTER["hip-fp8"]("w.weight", None)
    mm.load({"w.weight": w.to(cuda)})
    assert torch.equal(mm.weight_scale.cpu(), s_ref) and torch.equal(mm.weight_nk.cpu().view(torch.uint8), q_ref.view(torch.uint8))


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("M,N,K,epi", [(4352, 4096, 8192, "none"),      # 272 tiles: one round + 16, split 16 ways (4 K-tiles each)
                                       (4300, 4104, 8320, "gelu"),      # ragged last row panel / column, 65 K-tiles -> 13 parts of 5
                                       (75600, 5120, 13824, "resid")])  # 14B FFN-2: 23 rounds + 32 tiles, 8 parts of 14 K-tiles (the last of 10)
def test_gemm_w8a8_k_split_of_a_sparse_last_round(cuda, M, N, K, epi, int8, options):
    """With a workspace the tiles of a sparsely filled last round are split along K over all CUs and a fix-up launch
    adds the raw partial sums in a fixed order before scales and epilogue.  int8: the partial sums are exact, so the
    whole output equals the plain call's bit for bit.  fp8: the tail tiles differ by fp32 summation order only."""
    g = torch.Generator(device=cuda).manual_seed(80 + M % 89)
    a = torch.randn(M, K, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=cuda, dtype=torch.float32) * 0.05).to(torch.bfloat16)
    quant = hip.quant_int8_rowwise if int8 else hip.quant_fp8_rowwise
    gemm = hip.gemm_int8 if int8 else hip.gemm_fp8
    aq, a_s = quant(a)
    wq, w_s = quant(w)
    del a, w
    bias = torch.randn(N, generator=g, device=cuda) * 0.1
    gate = torch.randn(N, generator=g, device=cuda)
    resid = torch.randn(M, N, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16) if epi == "resid" else None
    options(gemm_kernel=0)
    need = hip.gemm_w8a8_workspace_bytes(M, N, K)
    rem = (-(-M // 256) * -(-N // 256)) % 256
    assert need > 0 and need % (rem * 256 * 256 * 4) == 0
    ws = torch.empty(need, dtype=torch.uint8, device=cuda)

    def run(workspace):
        if epi == "gelu":
            return gemm(aq, a_s, wq, w_s, bias, epilogue=hip.EPI_GELU_TANH, workspace=workspace)
        if epi == "resid":
            return gemm(aq, a_s, wq, w_s, bias, epilogue=hip.EPI_RESIDUAL, resid=resid.clone(), gate=gate, workspace=workspace)
        return gemm(aq, a_s, wq, w_s, bias, workspace=workspace)

    plain = run(None)
    ws.fill_(0x5A)
    split = run(ws)
    assert bool((ws != 0x5A).any())                          # the split is live: partial sums were written
    assert torch.equal(run(ws), split)                       # deterministic
    ws.fill_(0x5A)
    assert torch.equal(run(ws[:need - 16]), plain)           # too small a workspace only switches the split off ...
    assert bool((ws == 0x5A).all())                          # ... and is not touched
    if int8:
        assert torch.equal(split, plain)
        return
    # (sums of e4m3 products are often exact in fp32, so the two orders can agree in every bit)
    diff = plain != split
    pad = torch.zeros(-(-M // 256) * 256, -(-N // 256) * 256, dtype=torch.bool, device=cuda)
    pad[:M, :N] = diff
    touched = pad.view(-(-M // 256), 256, -(-N // 256), 256).any(dim=3).any(dim=1)
    assert int(touched.sum()) <= rem, (int(touched.sum()), rem)
    assert (plain.float() - split.float()).abs().max().item() <= 2.0 ** -7 * plain.float().abs().max().item()   # one bf16 step
    assert int(diff.sum()) <= 0.02 * rem * 65536


def test_gemm_w8a8_workspace_query(cuda, options):
    options(gemm_kernel=0)
    assert hip.gemm_w8a8_workspace_bytes(75600, 5120, 13824) == 32 * 8 * 256 * 256 * 4
    assert hip.gemm_w8a8_workspace_bytes(75600, 15360, 13824) == 0     # last round 96 tiles
    assert hip.gemm_w8a8_workspace_bytes(75600, 5120, 5120) == 0       # K too short: a round of 40 K-tiles is over before two more launches pay
    options(gemm_kernel=1)
    assert hip.gemm_w8a8_workspace_bytes(75600, 5120, 13824) == 0
