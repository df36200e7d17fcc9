"""GPU: int8 W8A8 Linear -- row-wise quantisation kernel and the int8-MFMA GEMM.  Integer work is held
to bit-exactness: the weight quantiser against the reference's own IntegerQuantizer (fixture written
by oracle/gen_golden_int8.py), the activation quantiser against the published vLLM arithmetic
(oracle.wan_block_ref.quant_int8_rowwise; parity unpinned, vLLM is absent), the int32 accumulators
against an exact integer product on the CPU."""
import pytest
import torch

import lightx2v_amd as L
from lightx2v_amd import hip
from oracle import wan_block_ref as R
from tests.util import case_inputs, cos, load_golden

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(torch.bfloat16)


def test_weight_quantiser_equals_the_references(cuda):
    rec = load_golden("int8_weight_quant")
    w = torch.from_numpy(rec["w_bf16_bits"]).view(torch.bfloat16)
    q, s = hip.quant_int8_rowwise(w.to(cuda), weight_mode=True)
    assert torch.equal(q.cpu(), torch.from_numpy(rec["q"])) and torch.equal(s.cpu(), torch.from_numpy(rec["scale"]))
    mm = L.MM_WEIGHT_REGISTER["hip-int8"]("w.weight", None)
    mm.load({"w.weight": w.to(cuda)})                       # weight_auto_quant: quantised at load time
    assert torch.equal(mm.weight_nk.cpu(), torch.from_numpy(rec["q"])) and torch.equal(mm.weight_scale.cpu(), torch.from_numpy(rec["scale"]))
    # an fp32 checkpoint weight is quantised from its own values, as the reference's `weight.to(torch.float32)` does
    w32 = torch.from_numpy(rec["w_f32"])
    q, s = hip.quant_int8_rowwise(w32.to(cuda), weight_mode=True)
    assert torch.equal(q.cpu(), torch.from_numpy(rec["q_f32"])) and torch.equal(s.cpu(), torch.from_numpy(rec["scale_f32"]))
    mm = L.MM_WEIGHT_REGISTER["hip-int8"]("w.weight", None)
    mm.load({"w.weight": w32.to(cuda)})
    assert torch.equal(mm.weight_nk.cpu(), torch.from_numpy(rec["q_f32"])) and torch.equal(mm.weight_scale.cpu(), torch.from_numpy(rec["scale_f32"]))
    mm.load({"w.weight": w32.to(torch.float16).to(cuda)})   # fp16 is widened, not rounded to bf16
    assert torch.equal(mm.weight_nk.cpu(), R.quant_int8_weight(w32.to(torch.float16))[0])


@pytest.mark.parametrize("rows,dim", [(1, 128), (7, 256), (130, 1536), (33, 5120), (5, 13824), (9, 520), (1000, 128)])
def test_activation_quantiser_bit_exact(cuda, rows, dim):
    x = _rand((rows, dim), 60 + rows, 3.0)
    x[0, : min(dim, 64)] = 0
    if rows > 2:
        x[2] = 0                                             # a row of zeros: scale 0, zeros
    q, s = hip.quant_int8_rowwise(x.to(cuda))
    rq, rs = R.quant_int8_rowwise(x)
    assert q.dtype == torch.int8 and torch.equal(q.cpu(), rq) and torch.equal(s.cpu(), rs)
    assert int(q.abs().max()) == 127


@pytest.mark.parametrize("M,N,K", [(16, 64, 128), (300, 512, 256), (1000, 1536, 1536), (257, 264, 384), (64, 5120, 13824)])
def test_gemm_int8_accumulates_exactly(cuda, M, N, K):
    """Unit scales and no bias: the bf16 output is the rounding of the exact integer product."""
    g = torch.Generator().manual_seed(70 + M)
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8)
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    ones_m, ones_n = torch.ones(M), torch.ones(N)
    out = hip.gemm_int8(a.to(cuda), ones_m.to(cuda), w.to(cuda), ones_n.to(cuda))
    exact = a.to(torch.int64) @ w.to(torch.int64).t()
    assert int(exact.abs().max()) < 2 ** 31
    assert torch.equal(out.cpu(), exact.to(torch.int32).float().to(torch.bfloat16))
    # scaled: fp32 (acc * a_scale * w_scale + bias), one bf16 rounding
    sa, sw = torch.rand(M, generator=g) * 1e-3 + 1e-4, torch.rand(N, generator=g) * 1e-3 + 1e-4
    bias = torch.randn(N, generator=g)
    out = hip.gemm_int8(a.to(cuda), sa.to(cuda), w.to(cuda), sw.to(cuda), bias.to(cuda)).float().cpu()
    ref = exact.to(torch.int32).float() * sa[:, None] * sw[None, :] + bias
    assert (out - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item()


def test_gemm_int8_epilogues_and_plugin(cuda):
    M, N, K = 300, 512, 256
    a, w = _rand((M, K), 90), _rand((N, K), 91, 0.05)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(92)) * 0.1
    gate = torch.randn(N, generator=torch.Generator().manual_seed(93))
    resid = _rand((M, N), 94)
    mm = L.MM_WEIGHT_REGISTER["hip-int8"]("w.weight", "w.bias")
    mm.load({"w.weight": w.to(cuda), "w.bias": bias.to(cuda)})
    wq, w_s = R.quant_int8_weight(w)
    assert torch.equal(mm.weight_nk.cpu(), wq) and torch.equal(mm.weight_scale.cpu(), w_s)
    ref = R.linear_int8(a, wq, w_s, bias)
    exact = R.linear(a.float(), w.float(), bias)
    assert cos(ref, exact) >= 0.9998                          # what the scheme itself costs
    assert (mm.apply(a.to(cuda)).float().cpu() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-3
    g = torch.nn.functional.gelu(ref, approximate="tanh")
    assert (mm.apply_gelu(a.to(cuda)).float().cpu() - g).abs().max().item() <= 2.0 ** -7 * g.abs().max().item() + 1e-3
    r = resid.to(cuda).clone()
    mm.apply_residual_(a.to(cuda), r, gate=gate.to(cuda))
    rr = resid.float() + ref * gate
    assert (r.float().cpu() - rr).abs().max().item() <= 2.0 ** -7 * rr.abs().max().item() + 1e-3
    # a pre-quantised checkpoint (the reference's load_quantized layout: int8 weight + [N, 1] fp32 scale) loads as is
    mm2 = L.MM_WEIGHT_REGISTER["hip-int8"]("w.weight", "w.bias")
    mm2.load({"w.weight": wq.to(cuda), "w.weight_scale": w_s.reshape(-1, 1).to(cuda), "w.bias": bias.to(cuda)})
    assert torch.equal(mm2.apply(a.to(cuda)), mm.apply(a.to(cuda)))
    with pytest.raises(hip.Lx2vError):
        hip.gemm_int8(wq.to(cuda).float(), torch.ones(N).to(cuda), wq.to(cuda), w_s.to(cuda))


@pytest.mark.parametrize("fused", [False, True])
def test_block_with_int8_linears(cuda, fused):
    """The whole block with every Linear in W8A8 int8 (both runners) against the fp32 oracle."""
    rec = load_golden("wan1p3b_g3x6x8")
    c = case_inputs(rec)
    cfg = {"num_layers": 1, "num_heads": c["heads"], "dim": c["dim"], "task": "t2v", "dit_quant_scheme": "hip-int8", "fused_block": fused}
    W = L.WanTransformerWeights(cfg)
    W.load({k: v.to(cuda) for k, v in {**c["w"], **c["hw"]}.items()})
    W.to_cuda()
    inf = L.WanTransformerInfer(cfg)
    inf.cos_sin = R.cos_sin_to_real(c["cos_sin"]).contiguous().to(cuda)
    pre = L.WanPreInferModuleOutput(embed=c["embed"].to(cuda), grid_sizes=c["grid"], x=c["x"].to(torch.bfloat16).to(cuda),
                                    embed0=c["embed0"].to(cuda), context=c["context"].to(torch.bfloat16).to(cuda))
    out = inf.infer_main_blocks(W.blocks, pre)
    ref = R.block_forward(c["x"], c["embed0"], c["context"], c["w"], c["cos_sin"], c["heads"])
    c_out, c_delta = cos(out, ref), cos(out.float().cpu() - c["x"], ref - c["x"])
    print(f"\nint8 block: output cosine {c_out:.6f}, delta cosine {c_delta:.6f}")
    assert c_out >= 0.999 and c_delta >= 0.99
