"""GPU: block-scaled fp8 attention (lx2v_attn_fwd_fp8) against the fp32 softmax and against the CPU
emulation of its quantisation scheme (oracle/attn_fp8_ref.py; parity unpinned, see there).

Tolerances: e4m3 carries 3 mantissa bits (2.2 % rms relative rounding error); with q, k, P and v all
rounded, an attention output on N(0,1) data is off by ~5 % rms of its own scale whatever the number of
keys -- cosine ~0.9986 against the fp32 softmax, which is what the scheme gives (the emulation shows the
same figure), so the attention-level bound is cosine >= 0.998 / rel rms <= 0.06; against the emulation
(same quantised operands; P rounded after a different per-row offset) cosine >= 0.999.  The north-star
gate (block output cosine >= 0.999) is checked at block level, where it holds with a wide margin."""
import math

import pytest
import torch

import lightx2v_amd as L
from lightx2v_amd import hip
from oracle import attn_fp8_ref as A8
from oracle import wan_block_ref as R
from tests.util import cos

pytestmark = pytest.mark.gpu


def _qkv(sq, skv, H, seed, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn(sq, H, 128, generator=g) * spread).to(torch.bfloat16)
    k = (torch.randn(skv, H, 128, generator=g) * spread).to(torch.bfloat16)
    v = torch.randn(skv, H, 128, generator=g).to(torch.bfloat16)
    return q, k, v


def test_e8m0_choice_is_the_smallest_power_of_two_that_fits():
    amax = torch.tensor([448.0, 448.1, 447.9, 1.0, 1.75, 1.7500001, 3.0e-5, 0.0, 7.0e4])
    e = A8.e8m0_for(amax) - 127
    ok = amax / torch.pow(2.0, e.float()) <= 448.0
    tight = amax / torch.pow(2.0, (e - 1).float()) > 448.0
    assert bool(ok.all()) and bool(tight[:7].all()) and bool(tight[8])


@pytest.mark.parametrize("sq,skv,H", [(64, 64, 1), (300, 300, 2), (257, 512, 3), (1000, 1000, 2), (96, 77, 2)])
def test_fp8_attention_matches_fp32_softmax_and_emulation(cuda, sq, skv, H):
    q, k, v = _qkv(sq, skv, H, 500 + sq)
    out = hip.attn_fwd_fp8(q.to(cuda), k.to(cuda), v.to(cuda)).float().cpu()
    ref = R.sdpa(q.float(), k.float(), v.float())
    emu = A8.attention_fp8(q, k, v)
    rel_rms = ((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"\nfp8 attn sq={sq} skv={skv} H={H}: cos vs fp32 {cos(out, ref):.6f} (emulation vs fp32 {cos(emu, ref):.6f}), cos vs emulation {cos(out, emu):.6f}, rel rms {rel_rms:.4f}")
    assert cos(out, ref) >= 0.998 and rel_rms <= 0.06
    assert cos(out, emu) >= 0.999


@pytest.mark.parametrize("sq,skv,H", [(64, 64, 1), (257, 512, 3), (1000, 1000, 2), (96, 77, 2), (300, 1700, 1)])
def test_fp8_attention_kernels_agree(cuda, sq, skv, H):
    """The one-wave-per-SIMD kernel (default) and the 8-wave kernel compute the same thing from the
    same quantised operands, but a wave of the former holds 64 query rows and of the latter 32, and
    a wave moves the running maxima of all its rows when one of them calls for it: the per-row
    offset under which P is rounded to e4m3 differs between the two, so they agree to the rounding
    noise of P (the figure both show against the emulation), not bit for bit."""
    q, k, v = (t.to(cuda) for t in _qkv(sq, skv, H, 900 + skv, spread=1.5))
    assert hip.get_option("attn_fp8_kernel") == 4
    a = hip.attn_fwd_fp8(q, k, v).float()
    hip.set_option("attn_fp8_kernel", 8)
    try:
        b = hip.attn_fwd_fp8(q, k, v).float()
    finally:
        hip.set_option("attn_fp8_kernel", 4)
    assert torch.isfinite(a).all()
    rel_rms = ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    print(f"\nfp8 kernels sq={sq} skv={skv} H={H}: cos {cos(a.cpu(), b.cpu()):.6f}, rel rms {rel_rms:.4f}")
    assert cos(a.cpu(), b.cpu()) >= 0.9993 and rel_rms <= 0.04


def test_fp8_attention_peaked_scores_and_strided_inputs(cuda):
    """Large score spread (the online-softmax rescale path runs) with q, k, v as thirds of one
    [S, 3*H*128] buffer (the fused block's layout)."""
    S, H = 700, 2
    q, k, v = _qkv(S, S, H, 77, spread=2.5)
    buf = torch.cat([q.reshape(S, -1), k.reshape(S, -1), v.reshape(S, -1)], dim=1).to(cuda)
    D = H * 128
    qv, kv, vv = (buf[:, i * D:(i + 1) * D].unflatten(1, (H, 128)) for i in range(3))
    out = hip.attn_fwd_fp8(qv, kv, vv).float().cpu()
    ref = R.sdpa(q.float(), k.float(), v.float())
    emu = A8.attention_fp8(q, k, v)
    print(f"\npeaked scores: cos vs fp32 {cos(out, ref):.6f} (emulation vs fp32 {cos(emu, ref):.6f}), cos vs emulation {cos(out, emu):.6f}")
    # scores with a standard deviation of 6 carry the 3 % rounding error of q and k as an ABSOLUTE
    # error of ~0.2 in the exponent: the scheme itself (emulation) is at ~0.994 here
    assert cos(out, ref) >= 0.99 and cos(out, emu) >= 0.998
    assert torch.isfinite(out).all()


@pytest.fixture
def linear_p():
    """`attn_fp8_p = 1` for the test: P = 2^x with a linear mantissa (no exps), row sums over the same codes."""
    hip.set_option("attn_fp8_p", 1)
    try:
        yield
    finally:
        hip.set_option("attn_fp8_p", 0)


@pytest.mark.parametrize("sq,skv,H", [(64, 64, 1), (300, 300, 2), (257, 512, 3), (1000, 1000, 2), (96, 77, 2), (300, 1700, 1), (64, 896, 1)])
def test_fp8_attention_with_linear_p(cuda, linear_p, sq, skv, H):
    """The exp-free P costs accuracy against the fp32 softmax in the fourth digit of the cosine (the linear
    mantissa is off by up to 0.35 code beside the 0.5 code of e4m3 rounding).  Against the emulation of the
    same rule the kernel agrees to the last digit while its running maximum is the true one (one tile);
    with more tiles the maximum it rounds under lags, and both the rounding and the linear-mantissa error
    fall differently (cosine ~0.999)."""
    q, k, v = _qkv(sq, skv, H, 500 + sq)
    out = hip.attn_fwd_fp8(q.to(cuda), k.to(cuda), v.to(cuda)).float().cpu()
    ref = R.sdpa(q.float(), k.float(), v.float())
    emu = A8.attention_fp8(q, k, v, p_mode="linear")
    rel_rms = ((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"\nfp8 attn, linear P, sq={sq} skv={skv} H={H}: cos vs fp32 {cos(out, ref):.6f} (emulation vs fp32 {cos(emu, ref):.6f}), cos vs emulation {cos(out, emu):.6f}, rel rms {rel_rms:.4f}")
    assert torch.isfinite(out).all()
    assert cos(out, ref) >= 0.998 and rel_rms <= 0.065
    # one tile: the kernel's maximum is the emulation's.  More tiles: the emulation, rounding under the TRUE row maximum,
    # pushes more of a long row's small P into the subnormal codes (where the line is no longer 2^x) than the kernel does
    # under its lagging maximum, and is the less accurate of the two against the fp32 softmax
    assert cos(out, emu) >= (0.99999 if skv <= 64 else 0.997)
    assert cos(out, ref) >= cos(emu, ref) - 2e-4


@pytest.mark.parametrize("sq,skv,H", [(5, 1, 1), (130, 33, 2), (64, 65, 1), (1, 700, 1), (1000, 512, 2)])
def test_fp8_attention_with_linear_p_edge_shapes(cuda, linear_p, sq, skv, H):
    """One key, ragged last tiles of one to 63 keys, a single query row, the cross-attention shape (the
    kernel's short-sequence instance): against the fp32 softmax and against the exp formulation."""
    q, k, v = _qkv(sq, skv, H, 40 + sq + skv)
    out = hip.attn_fwd_fp8(q.to(cuda), k.to(cuda), v.to(cuda)).float().cpu()
    hip.set_option("attn_fp8_p", 0)
    exact = hip.attn_fwd_fp8(q.to(cuda), k.to(cuda), v.to(cuda)).float().cpu()
    hip.set_option("attn_fp8_p", 1)
    ref = R.sdpa(q.float(), k.float(), v.float())
    print(f"\nlinear P, sq={sq} skv={skv} H={H}: cos vs fp32 {cos(out, ref):.6f} (exp formulation {cos(exact, ref):.6f}), cos between the two {cos(out, exact):.6f}")
    assert torch.isfinite(out).all()
    assert cos(out, ref) >= (0.9975 if skv > 1 else 0.9995)   # one key: P = 1 whatever the rule, what is left is the rounding of v
    assert cos(out, exact) >= 0.998


def test_fp8_attention_with_linear_p_degenerate_scores(cuda, linear_p):
    """All scores equal (q = 0: every P is the code of 1.0 and a 64-key tile adds 64 to a row's sum, well below the
    check's 2^8), and scores so far apart that one key carries a row alone."""
    S, H = 300, 1
    g = torch.Generator().manual_seed(3)
    v = torch.randn(S, H, 128, generator=g).to(torch.bfloat16)
    k = torch.randn(S, H, 128, generator=g).to(torch.bfloat16)
    q0 = torch.zeros(S, H, 128, dtype=torch.bfloat16)
    out = hip.attn_fwd_fp8(q0.to(cuda), k.to(cuda), v.to(cuda)).float().cpu()
    ref = R.sdpa(q0.float(), k.float(), v.float())
    assert torch.isfinite(out).all() and cos(out, ref) >= 0.999
    q1 = (k.float() * 6.0).to(torch.bfloat16)   # row i scores key i with 6 |k_i|^2 / sqrt(128) ~ 68: a one-hot softmax
    out = hip.attn_fwd_fp8(q1.to(cuda), k.to(cuda), v.to(cuda)).float().cpu()
    ref = R.sdpa(q1.float(), k.float(), v.float())
    print(f"\none-hot rows, linear P: cos vs fp32 {cos(out, ref):.6f}")
    assert torch.isfinite(out).all() and cos(out, ref) >= 0.999


@pytest.mark.parametrize("jump", [8.875, 8.0, 9.5, 20.0])
def test_fp8_attention_with_linear_p_catches_every_kind_of_overflow(cuda, linear_p, jump):
    """One key of the second tile scores `jump` (in log2 units) above everything before it, with operands that the
    quantisation keeps exactly: 8.875 lands on code 127 (e4m3's NaN), 8.0 on 120 (a finite 256: caught by the size of the
    row-sum increase), 9.5 and 20 saturate into the codes with the sign bit.  The rare path must take all of them."""
    S, H = 128, 1
    q = torch.zeros(S, H, 128)
    q[:, 0, 0] = 1.0
    q[:, 0, 1] = 1.0
    k = torch.zeros(S, H, 128)
    hi, lo = (8.0, jump - 8.0) if jump < 16 else (16.0, jump - 16.0)
    k[100, 0, 0], k[100, 0, 1] = hi, lo      # score = hi + lo = jump with scale * log2(e) = 1
    g = torch.Generator().manual_seed(8)
    v = torch.randn(S, H, 128, generator=g)
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    out = hip.attn_fwd_fp8(qb.to(cuda), kb.to(cuda), vb.to(cuda), scale=1.0 / math.log2(math.e)).float().cpu()
    w = torch.exp2(torch.einsum("qhd,khd->hqk", qb.double(), kb.double()))
    ref = torch.einsum("hqk,khd->qhd", w / w.sum(-1, keepdim=True), vb.double()).reshape(S, H * 128).float()
    print(f"\njump {jump}: cos vs fp32 {cos(out, ref):.6f}")
    assert torch.isfinite(out).all()
    assert cos(out, ref) >= 0.998


def test_fp8_attention_static_entry_point_with_linear_p(cuda, linear_p):
    """The entry point that allocates nothing launches the same kernels: same option, same bits."""
    q, k, v = (t.to(cuda) for t in _qkv(300, 500, 2, 12))
    a = hip.attn_fwd_fp8(q, k, v)
    b = hip.attn_fwd_fp8_static(q, k, v, torch.empty_like(a))
    assert torch.equal(a, b)


def test_fp8_attention_with_linear_p_peaked_scores(cuda, linear_p):
    """Score spread of 6: later tiles exceed the running maximum by more than 2^8 for some rows, so the
    check on the row-sum increase fires and the rare path re-bases (and re-adds the tile's row sums)."""
    S, H = 700, 2
    q, k, v = _qkv(S, S, H, 77, spread=2.5)
    out = hip.attn_fwd_fp8(q.to(cuda), k.to(cuda), v.to(cuda)).float().cpu()
    ref = R.sdpa(q.float(), k.float(), v.float())
    emu = A8.attention_fp8(q, k, v, p_mode="linear")
    print(f"\npeaked scores, linear P: cos vs fp32 {cos(out, ref):.6f} (emulation vs fp32 {cos(emu, ref):.6f}), cos vs emulation {cos(out, emu):.6f}")
    assert torch.isfinite(out).all()
    assert cos(out, ref) >= 0.99 and cos(out, emu) >= 0.998


def test_fp8_attention_with_linear_p_rising_scores(cuda, linear_p):
    """Keys whose scores rise along the sequence (each tile beats the last by a wide margin): the rare
    path runs again and again, with increases far beyond e4m3's range in between."""
    S, H = 640, 1
    g = torch.Generator().manual_seed(5)
    q = torch.randn(S, H, 128, generator=g)
    k = torch.randn(S, H, 128, generator=g)
    ramp = torch.linspace(0.2, 6.0, S)[:, None, None]
    k = (k * 0.3 + q.mean(0, keepdim=True) * ramp).to(torch.bfloat16)
    q = q.to(torch.bfloat16)
    v = torch.randn(S, H, 128, generator=g).to(torch.bfloat16)
    out = hip.attn_fwd_fp8(q.to(cuda), k.to(cuda), v.to(cuda)).float().cpu()
    ref = R.sdpa(q.float(), k.float(), v.float())
    emu = A8.attention_fp8(q, k, v, p_mode="linear")
    print(f"\nrising scores, linear P: cos vs fp32 {cos(out, ref):.6f} (emulation vs fp32 {cos(emu, ref):.6f}), cos vs emulation {cos(out, emu):.6f}")
    assert torch.isfinite(out).all()
    assert cos(out, ref) >= 0.99 and cos(out, emu) >= 0.995


def test_fp8_attention_plugin_and_workspace_checks(cuda):
    q, k, v = _qkv(128, 128, 2, 9)
    plug = L.ATTN_WEIGHT_REGISTER["hip_fp8_attn"]()
    a = plug.apply(q=q.to(cuda), k=k.to(cuda), v=v.to(cuda), cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=128, max_seqlen_kv=128)
    b = plug.apply(q=q.to(cuda)[:64], k=k.to(cuda), v=v.to(cuda))  # smaller call re-uses the workspace
    assert a.shape == (128, 256) and b.shape == (64, 256) and torch.equal(a[:64], b)
    need = hip.attn_fp8_workspace_bytes(128, 128, 2)
    assert need == 128 * 2 * 132 + 2 * 2 * 17152
    with pytest.raises(hip.Lx2vError):
        hip.attn_fwd_fp8(q.to(cuda), k.to(cuda), v.to(cuda), workspace=torch.empty(need - 1, dtype=torch.uint8, device=cuda))
    with pytest.raises(NotImplementedError):
        plug.apply(q=q.to(cuda), k=k.to(cuda), v=v.to(cuda), causal=True)


@pytest.mark.parametrize("attn_type", ["hip_fp8_attn", "hip_fp8_attn_linear"])
def test_block_with_fp8_attention_meets_the_block_gate(cuda, attn_type):
    """Wan block with `self_attn_1_type = hip_fp8_attn` (and its exp-free variant) against the reference's
    fp32 run: output cosine >= 0.999 (north-star gate); the delta cosine is printed and bounded."""
    from tests.util import case_inputs, load_golden

    rec = load_golden("wan1p3b_g3x30x52")
    c = case_inputs(rec)
    cfg = {"num_layers": 1, "num_heads": c["heads"], "dim": c["dim"], "task": "t2v", "self_attn_1_type": attn_type}
    W = L.WanTransformerWeights(cfg)
    W.load({k: v.to(cuda) for k, v in {**c["w"], **c["hw"]}.items()})
    inf = L.WanTransformerInfer(cfg)
    inf.cos_sin = R.cos_sin_to_real(c["cos_sin"]).contiguous().to(cuda)
    inf.get_scheduler_values = lambda: None
    pre = L.WanPreInferModuleOutput(embed=c["embed"].to(cuda), grid_sizes=c["grid"], x=c["x"].to(torch.bfloat16).to(cuda),
                                    embed0=c["embed0"].to(cuda), context=c["context"].to(torch.bfloat16).to(cuda))
    out = inf.infer_main_blocks(W.blocks, pre).float().cpu()
    rows = torch.from_numpy(rec["rows"]).long()
    ref = torch.from_numpy(rec["x_out"])
    c_out, c_delta = cos(out[rows], ref), cos(out[rows] - c["x"][rows], ref - c["x"][rows])
    print(f"\nblock with {attn_type} self-attention (S={c['S']}): output cosine {c_out:.6f}, delta cosine {c_delta:.6f}")
    assert c_out >= 0.999 and c_delta >= 0.99
    assert hip.get_option("attn_fp8_p") == 0   # the linear plugin sets the option around its launch only
