"""Each HIP kernel (through the C ABI) against the CPU oracle's op on the same seeded inputs.
Inputs are rounded to bf16 first, so the only differences are the kernel's own arithmetic and the
single bf16 rounding of its output.  Tolerances are stated per test."""
import math

import pytest
import torch

from lightx2v_amd import hip
from oracle import wan_block_ref as R
from tests.util import cos, rel_err

pytestmark = pytest.mark.gpu

BF16_EPS = 2.0 ** -8  # one bf16 rounding: relative error <= 2^-9, allow 2x


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(*shape, generator=g)).to(torch.bfloat16)


@pytest.mark.parametrize("rows,dim", [(1, 64), (7, 256), (130, 1536), (33, 5120), (5, 8192), (9, 520)])
@pytest.mark.parametrize("affine,mod", [(False, True), (True, False), (True, True), (False, False)])
def test_ln_modulate(cuda, rows, dim, affine, mod):
    x = _rand((rows, dim), 1, 2.0) + 0.5
    g = torch.Generator().manual_seed(2)
    w = 1 + 0.1 * torch.randn(dim, generator=g) if affine else None
    b = 0.1 * torch.randn(dim, generator=g) if affine else None
    sc = 0.2 * torch.randn(dim, generator=g) if mod else None
    sh = 0.2 * torch.randn(dim, generator=g) if mod else None
    ref = R.layer_norm(x.float(), w, b)
    if mod:
        ref = R.modulate(ref, sc, sh)
    out = hip.ln_modulate(x.to(cuda), w, b, sc, sh)
    assert out.dtype == torch.bfloat16 and out.shape == (rows, dim)
    assert (out.float().cpu() - ref).abs().max().item() <= BF16_EPS * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("rows,dim,hd", [(3, 256, 128), (130, 1536, 128), (19, 5120, 128), (4, 512, 64)])
@pytest.mark.parametrize("rope", [False, True])
def test_rmsnorm_rope(cuda, rows, dim, hd, rope):
    x = _rand((rows, dim), 3, 1.5)
    g = torch.Generator().manual_seed(4)
    w = 1 + 0.1 * torch.randn(dim, generator=g)
    ref = R.rms_norm(x.float(), w)
    table = None
    if rope:
        # a table with all three axes active
        cs = R.build_cos_sin((rows, 2, 2), hd)[3::4][:rows]
        ref, _ = R.rope_apply(ref.view(rows, dim // hd, hd), ref.view(rows, dim // hd, hd), cs)
        ref = ref.reshape(rows, dim)
        table = R.cos_sin_to_real(cs).contiguous().to(cuda)
    out = hip.rmsnorm_rope(x.to(cuda), w, table, head_dim=hd)
    assert (out.float().cpu() - ref).abs().max().item() <= BF16_EPS * ref.abs().max().item() + 1e-3
    # in-place on a strided view (the q part of a fused q|k|v buffer)
    big = torch.zeros(rows, 3 * dim, dtype=torch.bfloat16, device=cuda)
    big[:, dim:2 * dim] = x.to(cuda)
    view = big[:, dim:2 * dim]
    hip.rmsnorm_rope(view, w, table, head_dim=hd, out=view)
    assert torch.equal(view, out) and torch.count_nonzero(big[:, :dim]) == 0 and torch.count_nonzero(big[:, 2 * dim:]) == 0


def test_rope_only_matches_oracle(cuda):
    S, H, d = 96, 3, 128
    q = _rand((S, H * d), 5)
    cs = R.build_cos_sin((4, 4, 6), d)
    ref, _ = R.rope_apply(q.float().view(S, H, d), q.float().view(S, H, d), cs)
    out = hip.rmsnorm_rope(q.to(cuda), None, R.cos_sin_to_real(cs).to(cuda), head_dim=d, do_norm=False)
    assert (out.float().cpu() - ref.reshape(S, -1)).abs().max().item() <= BF16_EPS * ref.abs().max().item()


@pytest.mark.parametrize("gated", [True, False])
def test_gate_residual_and_gelu(cuda, gated):
    x, y = _rand((37, 1536), 6), _rand((37, 1536), 7)
    gate = 0.3 * torch.randn(1536, generator=torch.Generator().manual_seed(8)) if gated else None
    ref = x.float() + (y.float() * gate if gated else y.float())
    xd = x.to(cuda)
    out = hip.gate_residual_(xd, y.to(cuda), gate)
    assert out.data_ptr() == xd.data_ptr()
    assert (out.float().cpu() - ref).abs().max().item() <= BF16_EPS * ref.abs().max().item()
    h = _rand((64, 8960), 9, 2.0)
    gref = torch.nn.functional.gelu(h.float(), approximate="tanh")
    gout = hip.gelu_tanh(h.to(cuda))
    assert (gout.float().cpu() - gref).abs().max().item() <= BF16_EPS * gref.abs().max().item() + 1e-3


# ---- the residual stream kept in fp32 (the reference's SENSITIVE_LAYER_DTYPE=FP32) ----
@pytest.mark.parametrize("rows,dim", [(1, 64), (7, 256), (130, 1536), (33, 5120), (5, 8192)])
@pytest.mark.parametrize("affine,mod", [(False, True), (True, False), (True, True)])
def test_ln_modulate_on_fp32_rows(cuda, rows, dim, affine, mod):
    g = torch.Generator().manual_seed(12)
    x = 2.0 * torch.randn(rows, dim, generator=g) + 0.5          # fp32 values no bf16 tensor holds
    w = 1 + 0.1 * torch.randn(dim, generator=g) if affine else None
    b = 0.1 * torch.randn(dim, generator=g) if affine else None
    sc = 0.2 * torch.randn(dim, generator=g) if mod else None
    sh = 0.2 * torch.randn(dim, generator=g) if mod else None
    ref = R.layer_norm(x, w, b)
    if mod:
        ref = R.modulate(ref, sc, sh)
    out = hip.ln_modulate(x.to(cuda), w, b, sc, sh)
    assert out.dtype == torch.bfloat16 and out.shape == (rows, dim)
    assert (out.float().cpu() - ref).abs().max().item() <= BF16_EPS * ref.abs().max().item() + 1e-3
    # the fp32 rows matter: the same call on their bf16 rounding gives another result
    assert not torch.equal(out, hip.ln_modulate(x.to(torch.bfloat16).to(cuda), w, b, sc, sh)) or rows * dim < 512
    # strided rows (a column range of a wider fp32 buffer)
    wide = torch.zeros(rows, dim + 64, device=cuda)
    wide[:, 32:32 + dim] = x.to(cuda)
    if (32 * 4) % 16 == 0:
        assert torch.equal(hip.ln_modulate(wide[:, 32:32 + dim], w, b, sc, sh), out)
    with pytest.raises(hip.Lx2vError):
        hip.ln_modulate(x.to(cuda), w, b, sc, sh, quant=True)


@pytest.mark.parametrize("gated", [True, False])
def test_gate_residual_on_fp32_rows(cuda, gated):
    g = torch.Generator().manual_seed(13)
    x = torch.randn(37, 1536, generator=g)
    y = _rand((37, 1536), 7)
    gate = 0.3 * torch.randn(1536, generator=g) if gated else None
    ref = x + (y.float() * gate if gated else y.float())
    xd = x.to(cuda)
    out = hip.gate_residual_(xd, y.to(cuda), gate)
    assert out.data_ptr() == xd.data_ptr() and out.dtype == torch.float32
    # fp32 arithmetic: product and sum (a fused multiply-add differs from the two roundings by at most one unit in the last place)
    assert (out.cpu() - ref).abs().max().item() <= 2.0 ** -22 * ref.abs().max().item()
    # per-token gate
    rowsmod = hip.RowMod(gate if gated else torch.zeros(1536), _rand((37, 1536), 14, 0.2).to(cuda))
    x2 = x.to(cuda)
    hip.gate_residual_(x2, y.to(cuda), rowsmod)
    ref2 = x + y.float() * ((gate if gated else torch.zeros(1536)) + rowsmod.rows.float().cpu())
    assert (x2.cpu() - ref2).abs().max().item() <= 2.0 ** -21 * ref2.abs().max().item()
    with pytest.raises(hip.Lx2vError):
        hip.gate_residual_(xd, x.to(cuda), gate)   # y must be bf16


GEMM_SHAPES = [
    (256, 256, 64), (1, 256, 64), (300, 512, 128), (257, 264, 192), (1000, 1536, 1536), (520, 64, 1536),
    (512, 3072, 5120), (777, 8960, 1536), (333, 1536, 8960),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_bias(cuda, M, N, K):
    a, w = _rand((M, K), 10), _rand((N, K), 11, 0.05)
    b = 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(12))
    ref = R.linear(a.float(), w.float(), b)
    out = hip.gemm(a.to(cuda), w.to(cuda), b)
    # fp32 accumulation of exact bf16 products in a different order than the CPU: ~sqrt(K) * 2^-24
    # relative, far below the output's own bf16 rounding
    assert (out.float().cpu() - ref).abs().max().item() <= BF16_EPS * ref.abs().max().item() + 1e-3
    out2 = hip.gemm(a.to(cuda), w.to(cuda), None)
    assert (out2.float().cpu() - R.linear(a.float(), w.float(), None)).abs().max().item() <= BF16_EPS * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("M,N,K,epi", [(4000, 2048, 2048, "none"), (1111, 1536, 8960, "gelu"), (3000, 5120, 5120, "resid"), (700, 512, 192, "none")])
def test_gemm_schedules_agree_bitwise(cuda, M, N, K, epi, options):
    """The ping-pong schedule accumulates every output in the same order as the one-barrier
    schedule, so the two must agree bit for bit -- a stale or early-read LDS tile would not.
    Repeated launches with other work in flight in between (different timing each time)."""
    a, w = _rand((M, K), 40), _rand((N, K), 41, 0.05)
    b = 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(42))
    gate = torch.randn(N, generator=torch.Generator().manual_seed(43))
    resid = _rand((M, N), 44).to(cuda)
    ad, wd, bd = a.to(cuda), w.to(cuda), b.to(cuda)

    def run():
        if epi == "gelu":
            return hip.gemm(ad, wd, bd, epilogue=hip.EPI_GELU_TANH)
        if epi == "resid":
            return hip.gemm(ad, wd, bd, epilogue=hip.EPI_RESIDUAL, resid=resid, gate=gate.to(cuda))
        return hip.gemm(ad, wd, bd)

    options(gemm_kernel=1)  # the simple schedule
    ref = run()
    options(gemm_kernel=0)
    noise = torch.empty(64 << 20, dtype=torch.uint8, device=cuda)
    for i in range(6):
        if i % 2:
            noise.fill_(i)  # HBM traffic in flight while the GEMM starts
        assert torch.equal(run(), ref), i


@pytest.mark.parametrize("M,N,K,epi", [(75600, 5120, 5120, "resid"), (20000, 13824, 5120, "gelu"), (75600, 5120, 13824, "none")])
def test_gemm_many_tiles_per_workgroup(cuda, M, N, K, epi, options):
    """The 14B / 720p Linear shapes: thousands of output tiles, so every workgroup of the persistent
    kernel walks some twenty of them back to back (loads of the next tile in flight during an
    epilogue, partial last row panel, a last round that a few workgroups take alone).  Bit for bit
    against the one-barrier schedule, and sampled rows against the fp32 product on the CPU."""
    g = torch.Generator(device=cuda).manual_seed(50)
    a = torch.randn(M, K, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16)
    w = (0.05 * torch.randn(N, K, generator=g, device=cuda, dtype=torch.float32)).to(torch.bfloat16)
    b = 0.1 * torch.randn(N, generator=g, device=cuda, dtype=torch.float32)
    gate = torch.randn(N, generator=g, device=cuda, dtype=torch.float32)
    resid = torch.randn(M, N, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16) if epi == "resid" else None

    def run():
        if epi == "gelu":
            return hip.gemm(a, w, b, epilogue=hip.EPI_GELU_TANH)
        if epi == "resid":
            return hip.gemm(a, w, b, epilogue=hip.EPI_RESIDUAL, resid=resid, gate=gate)   # a new output tensor; resid is read only
        return hip.gemm(a, w, b)

    options(gemm_kernel=1)
    ref = run()
    options(gemm_kernel=0)
    out = run()
    assert torch.equal(out, ref)
    assert torch.equal(run(), ref)
    rows = torch.tensor(sorted({0, 1, 255, 256, 257, M - 1, M - 2, (M // 256) * 256 - 1, (M // 256) * 256} | set(torch.randint(0, M, (40,), generator=torch.Generator().manual_seed(51)).tolist())))
    want = a[rows.to(cuda)].float().cpu() @ w.float().cpu().t() + b.cpu()
    if epi == "gelu":
        want = torch.nn.functional.gelu(want, approximate="tanh")
    if epi == "resid":
        want = resid[rows.to(cuda)].float().cpu() + want.to(torch.bfloat16).float() * gate.cpu()
    got = out[rows.to(cuda)].float().cpu()
    assert (got - want).abs().max().item() <= 2 * BF16_EPS * want.abs().max().item()


@pytest.mark.parametrize("M,N,K,epi", [(4352, 4096, 4096, "none"),      # 17 x 16 = 272 tiles: one round + 16, split 16 ways (4 K-tiles each)
                                       (4300, 4104, 4160, "resid"),     # the same grid with a ragged last row panel / column, 65 K-tiles -> 13 parts of 5
                                       (75600, 5120, 5120, "gelu"),     # 14B output projection shape: 23 rounds + 32 tiles, 8 parts of 10 K-tiles
                                       (75600, 5120, 13824, "resid")])  # 14B FFN-2: 8 parts of 27 K-tiles
def test_gemm_k_split_of_a_sparse_last_round(cuda, M, N, K, epi, options):
    """With a workspace the tiles of a sparsely filled last round are split along K over all CUs (raw fp32 partial
    sums, summed in a fixed order by a fix-up launch that applies the epilogue).  Everything outside those tiles is
    bit-identical to the plain call; inside, the outputs differ by fp32 summation order only; both match the
    fp32 product on the CPU on sampled rows."""
    g = torch.Generator(device=cuda).manual_seed(70)
    a = torch.randn(M, K, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16)
    w = (0.05 * torch.randn(N, K, generator=g, device=cuda, dtype=torch.float32)).to(torch.bfloat16)
    b = 0.1 * torch.randn(N, generator=g, device=cuda, dtype=torch.float32)
    gate = torch.randn(N, generator=g, device=cuda, dtype=torch.float32)
    resid = torch.randn(M, N, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16) if epi == "resid" else None
    options(gemm_kernel=0)
    need = hip.gemm_workspace_bytes(M, N, K)
    tiles = -(-M // 256) * -(-N // 256)
    rem = tiles % 256
    assert need > 0 and need % (rem * 256 * 256 * 4) == 0
    ws = torch.empty(need, dtype=torch.uint8, device=cuda)

    def run(workspace):
        if epi == "gelu":
            return hip.gemm(a, w, b, epilogue=hip.EPI_GELU_TANH, workspace=workspace)
        if epi == "resid":
            return hip.gemm(a, w, b, epilogue=hip.EPI_RESIDUAL, resid=resid, gate=gate, workspace=workspace)
        return hip.gemm(a, w, b, workspace=workspace)

    plain = run(None)
    split = run(ws)
    assert torch.equal(run(ws), split)                       # deterministic
    assert torch.equal(run(ws[:need - 16]), plain)           # too small a workspace only switches the split off
    if epi == "resid":                                       # in place (the Linear plugin's call: out is resid)
        x = resid.clone()
        assert hip.gemm(a, w, b, epilogue=hip.EPI_RESIDUAL, resid=x, gate=gate, out=x, workspace=ws) is x
        assert torch.equal(x, split)
    diff = plain != split
    assert bool(diff.any())                                  # the split is live
    # the differing outputs lie in at most `rem` tiles ...
    pad = torch.zeros(-(-M // 256) * 256, -(-N // 256) * 256, dtype=torch.bool, device=cuda)
    pad[:M, :N] = diff
    touched = pad.view(-(-M // 256), 256, -(-N // 256), 256).any(dim=3).any(dim=1)
    assert int(touched.sum()) <= rem, (int(touched.sum()), rem)
    # ... and are roundings of nearly equal fp32 values: one bf16 step at most
    scale = plain.float().abs().max().item()
    assert (plain.float() - split.float()).abs().max().item() <= 2 * BF16_EPS * scale
    assert int(diff.sum()) <= 0.02 * rem * 65536
    tm, tn = (int(v) for v in touched.nonzero()[0])
    rows = torch.tensor(sorted({0, M - 1, tm * 256, min(tm * 256 + 255, M - 1)} | set(torch.randint(0, M, (24,), generator=torch.Generator().manual_seed(71)).tolist())))
    want = a[rows.to(cuda)].float().cpu() @ w.float().cpu().t() + b.cpu()
    if epi == "gelu":
        want = torch.nn.functional.gelu(want, approximate="tanh")
    if epi == "resid":
        want = resid[rows.to(cuda)].float().cpu() + want.to(torch.bfloat16).float() * gate.cpu()
    got = split[rows.to(cuda)].float().cpu()
    assert (got - want).abs().max().item() <= 2 * BF16_EPS * want.abs().max().item() + 1e-3


def test_gemm_workspace_query_says_when_a_split_pays(cuda, options):
    options(gemm_kernel=0)
    assert hip.gemm_workspace_bytes(75600, 5120, 5120) == 32 * 8 * 256 * 256 * 4      # 5920 tiles on 256 CUs
    assert hip.gemm_workspace_bytes(75600, 15360, 5120) == 0                          # last round 96 tiles: more than a quarter of the CUs
    assert hip.gemm_workspace_bytes(4352, 4096, 2048) == 0                            # K too short for two more launches to pay
    assert hip.gemm_workspace_bytes(2048, 2048, 8192) == 0                            # a single round
    options(gemm_kernel=1)
    assert hip.gemm_workspace_bytes(75600, 5120, 5120) == 0                           # only the persistent kernel splits


def test_gemm_mfma_operand_orientation_identity(cuda):
    """A = I against an ASYMMETRIC W catches a transposed accumulator write-back."""
    n = 256
    a = torch.eye(n, dtype=torch.bfloat16)
    w = (torch.arange(n * n, dtype=torch.float32).view(n, n) % 251 - 125).to(torch.bfloat16)  # exactly representable
    out = hip.gemm(a.to(cuda), w.to(cuda), None)
    assert torch.equal(out.float().cpu(), w.float().t())


@pytest.mark.parametrize("M,N,K", [(300, 512, 128), (1000, 1536, 1536)])
def test_gemm_epilogues(cuda, M, N, K):
    a, w = _rand((M, K), 13), _rand((N, K), 14, 0.05)
    g = torch.Generator().manual_seed(15)
    b = 0.1 * torch.randn(N, generator=g)
    gate = 0.5 * torch.randn(N, generator=g)
    resid = _rand((M, N), 16)
    lin = R.linear(a.float(), w.float(), b)
    gout = hip.gemm(a.to(cuda), w.to(cuda), b, epilogue=hip.EPI_GELU_TANH)
    gref = torch.nn.functional.gelu(lin, approximate="tanh")
    assert (gout.float().cpu() - gref).abs().max().item() <= BF16_EPS * gref.abs().max().item() + 1e-3
    for gt in (gate, None):
        r = resid.to(cuda).clone()
        out = hip.gemm(a.to(cuda), w.to(cuda), b, epilogue=hip.EPI_RESIDUAL, resid=r, gate=gt, out=r)
        assert out.data_ptr() == r.data_ptr()
        ref = resid.float() + (lin * gt if gt is not None else lin)
        # the product term is rounded to bf16 once before the add, then the sum once more
        assert (out.float().cpu() - ref).abs().max().item() <= 2 * BF16_EPS * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("schedule", [0, 1])
def test_gemm_k_segmented_operand(cuda, options, schedule):
    """A given as [segments][M][kseg] (the Ulysses receive layout) equals the gathered matrix."""
    options(gemm_kernel=schedule)
    M, N, kseg, nseg = 130, 256, 128, 4
    segs = _rand((nseg, M, kseg), 17)
    w = _rand((N, kseg * nseg), 18, 0.05)
    a_full = segs.permute(1, 0, 2).reshape(M, nseg * kseg)
    ref = R.linear(a_full.float(), w.float(), None)
    sd = segs.to(cuda)
    out = hip.gemm(sd.view(nseg * M, kseg)[:M], w.to(cuda), None, kseg=hip.KSeg(seg=kseg, stride=M * kseg))
    assert (out.float().cpu() - ref).abs().max().item() <= BF16_EPS * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("schedule", [0, 1])
@pytest.mark.parametrize("M", [130, 700])
def test_gemm_k_segmented_operand_two_regions(cuda, options, schedule, M):
    """Head chunks of two widths (5 heads per rank -> 3 + 2): region 1 = `world` segments of
    3*128 columns, region 2 = `world` segments of 2*128 columns with their own row stride, stored
    one after the other in one buffer.  Bit-identical to the GEMM on the gathered matrix."""
    options(gemm_kernel=schedule)
    world, N = 2, 256
    w0, w1 = 3 * 128, 2 * 128
    r1, r2 = _rand((world, M, w0), 27), _rand((world, M, w1), 28)
    w = _rand((N, world * (w0 + w1)), 29, 0.05)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(30)) * 0.1
    a_full = torch.cat([r1.permute(1, 0, 2).reshape(M, world * w0), r2.permute(1, 0, 2).reshape(M, world * w1)], dim=1)
    flat = torch.cat([r1.reshape(-1), r2.reshape(-1)]).to(cuda)
    ks = hip.KSeg(seg=w0, stride=M * w0, split=world * w0, seg2=w1, stride2=M * w1, base2=world * M * w0, lda2=w1)
    out = hip.gemm(flat[:M * w0].view(M, w0), w.to(cuda), bias.to(cuda), kseg=ks)
    plain = hip.gemm(a_full.to(cuda), w.to(cuda), bias.to(cuda))
    assert torch.equal(out, plain)
    ref = R.linear(a_full.float(), w.float(), bias)
    assert (out.float().cpu() - ref).abs().max().item() <= BF16_EPS * ref.abs().max().item() + 1e-3


@pytest.fixture(params=["x16-4", "x16-5", "w4", "w8"])
def attn_kernel(request, options):
    """Every attention kernel of the library behind the one C entry point: the 16x16x32-MFMA
    kernel (one wave per SIMD) with 4 and with 5 query blocks per wave, the 32x32x16 one and the
    8-wave one (options attn_kernel / attn_qb)."""
    name, _, qb = request.param.partition("-")
    options(attn_kernel={"x16": 16, "w4": 4, "w8": 8}[name])
    if qb:
        options(attn_qb=int(qb))
    return request.param


ATTN_SHAPES = [(64, 64, 1), (256, 64, 2), (100, 100, 1), (300, 520, 2), (257, 1, 1), (1, 257, 1), (1040, 1040, 3), (2000, 512, 2)]


@pytest.mark.parametrize("sq,skv,H", ATTN_SHAPES)
def test_attention_matches_sdpa(cuda, attn_kernel, sq, skv, H):
    q, k, v = _rand((sq, H, 128), 20), _rand((skv, H, 128), 21), _rand((skv, H, 128), 22)
    ref = R.sdpa(q.float(), k.float(), v.float())
    out = hip.attn_fwd(q.to(cuda), k.to(cuda), v.to(cuda))
    assert out.shape == (sq, H * 128)
    # P is rounded to bf16 before the PV product (relative 2^-9 per term, averaged by the sum) and
    # the output once more; cosine is the north-star metric, the abs bound catches layout bugs.
    assert cos(out, ref) >= 0.9999
    assert (out.float().cpu() - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("qb,H", [("4", 4), ("5", 5)])
def test_attention_short_last_round_goes_to_a_second_launch(cuda, options, qb, H):
    """260 workgroups on a 256-CU part: the four left over after the last full round (the end of
    the last head) are computed by a second launch with 64-row workgroups; every row must come
    out the same as in the single-launch run and match the reference."""
    options(attn_kernel=16, attn_qb=int(qb))
    sq, skv = 16640 - 37, 200
    q, k, v = _rand((sq, H, 128), 40), _rand((skv, H, 128), 41), _rand((skv, H, 128), 42)
    ref = R.sdpa(q.float(), k.float(), v.float())
    qd, kd, vd = q.to(cuda), k.to(cuda), v.to(cuda)
    out = hip.attn_fwd(qd, kd, vd)
    options(attn_tail=0)
    one = hip.attn_fwd(qd, kd, vd)
    assert torch.equal(out[: sq - 2048], one[: sq - 2048])
    assert cos(out, ref) >= 0.9999 and cos(one, ref) >= 0.9999
    assert (out.float().cpu() - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1.0)
    assert (out.float() - one.float()).abs().max().item() <= 2e-2


def test_attention_operand_layout_exact_integers(cuda, attn_kernel):
    """One-hot softmax (a single huge score per row) turns attention into a row gather of V with
    asymmetric integer data: any key/dim permutation mistake in the MFMA operand maps shows up."""
    S, H = 192, 2
    q = torch.zeros(S, H, 128)
    k = torch.zeros(S, H, 128)
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(3))
    for i in range(S):
        q[i, :, i % 128] = 16.0
        q[i, :, (i // 128) + 64] = 16.0 if i >= 128 else 0.0
    # key j matches query i iff j == perm[i]
    k[perm] = q * 16.0
    v = (torch.arange(S * H * 128, dtype=torch.float32).view(S, H, 128) % 199 - 99)
    ref = R.sdpa(q, k, v)
    out = hip.attn_fwd(q.to(torch.bfloat16).to(cuda), k.to(torch.bfloat16).to(cuda), v.to(torch.bfloat16).to(cuda))
    assert (out.float().cpu() - ref).abs().max().item() <= 0.51  # bf16 rounding of |v| <= 99


def test_attention_online_softmax_rescale_branch(cuda, attn_kernel):
    """Spike one key per query block late in the sequence so the running maximum jumps at a chosen
    tile (forces the rescale path after many no-rescale tiles)."""
    S, H = 640, 1
    q, k, v = _rand((S, H, 128), 30, 0.3), _rand((S, H, 128), 31, 0.3), _rand((S, H, 128), 32)
    k = k.float()
    k[500] = q[5].float() * 40.0    # huge score for row 5 at tile 7
    k[70] = q[300].float() * 25.0   # and for row 300 at tile 1
    k = k.to(torch.bfloat16)
    ref = R.sdpa_naive(q.float(), k.float(), v.float())
    out = hip.attn_fwd(q.to(cuda), k.to(cuda), v.to(cuda))
    assert (out.double().cpu() - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1.0)
    assert cos(out, ref.float()) >= 0.9999


@pytest.mark.parametrize("sq,skv", [(300, 100), (256, 257), (300, 1000), (520, 2200)])
def test_attention_never_consumes_memory_past_kv(cuda, attn_kernel, sq, skv):
    """K and V sit in the middle of one allocation whose every other element is NaN: the rows the
    kernel prefetches past the last key must come back as zeros, not as whatever follows the tensor."""
    H, pad = 1, 512
    q = _rand((sq, H, 128), 60)
    k, v = _rand((skv, H, 128), 61), _rand((skv, H, 128), 62)
    ref = R.sdpa(q.float(), k.float(), v.float())
    arena = torch.full((2, pad + skv + pad, H, 128), float("nan"), dtype=torch.bfloat16, device=cuda)
    arena[0, pad:pad + skv] = k.to(cuda)
    arena[1, pad:pad + skv] = v.to(cuda)
    out = hip.attn_fwd(q.to(cuda), arena[0, pad:pad + skv], arena[1, pad:pad + skv])
    assert torch.isfinite(out).all()
    assert cos(out, ref) >= 0.9999


def test_attention_strided_inputs_and_fused_rope(cuda):
    S, H, d = 360, 2, 128
    qkv = _rand((S, 3 * H * d), 40)
    cs = R.build_cos_sin((3, 10, 12), d)
    q, k, v = (qkv[:, i * H * d:(i + 1) * H * d].float().view(S, H, d) for i in range(3))
    qr, kr = R.rope_apply(q, k, cs)
    ref = R.sdpa(qr, kr, v)
    dev = qkv.to(cuda)
    views = [dev[:, i * H * d:(i + 1) * H * d].unflatten(1, (H, d)) for i in range(3)]
    out = hip.attn_fwd(*views, rope_cos_sin=R.cos_sin_to_real(cs).to(cuda))
    assert cos(out, ref) >= 0.9999
    assert (out.float().cpu() - ref).abs().max().item() <= 2e-2
    # pre-rotated path agrees with the fused one up to the extra bf16 rounding of rotated q/k
    table = R.cos_sin_to_real(cs).to(cuda)
    q2 = hip.rmsnorm_rope(dev[:, :H * d], None, table, head_dim=d, do_norm=False)
    k2 = hip.rmsnorm_rope(dev[:, H * d:2 * H * d], None, table, head_dim=d, do_norm=False)
    out2 = hip.attn_fwd(q2.view(S, H, d), k2.view(S, H, d), views[2])
    assert cos(out2, ref) >= 0.9999


def test_ulysses_pack_layout(cuda):
    rows, dim, world = 24, 512, 4
    q, k, v = _rand((rows, dim), 50), _rand((rows, dim), 51), _rand((rows, dim), 52)
    out = hip.ulysses_pack_qkv(q.to(cuda), k.to(cuda), v.to(cuda), world).cpu()
    seg = dim // world
    exp = torch.stack([q, k, v], dim=1).view(rows, 3, world, seg).permute(2, 0, 1, 3)
    assert torch.equal(out, exp)


def test_empty_inputs_are_no_ops(cuda):
    """Zero rows / tokens: every entry point returns an empty result without launching anything
    (a sequence-parallel rank can hold no text rows, a caller can pass an empty batch)."""
    e = lambda *shape, dtype=torch.bfloat16: torch.empty(*shape, dtype=dtype, device=cuda)
    w, b = _rand((64, 128), 1).to(cuda), torch.zeros(64, device=cuda)
    assert hip.gemm(e(0, 128), w, b).shape == (0, 64)
    assert hip.ln_modulate(e(0, 128)).shape == (0, 128)
    assert hip.rmsnorm_rope(e(0, 128), torch.ones(128, device=cuda), None).shape == (0, 128)
    assert hip.gate_residual_(e(0, 128), e(0, 128), None).shape == (0, 128)
    assert hip.gelu_tanh(e(0, 128)).shape == (0, 128)
    k = _rand((70, 2, 128), 2).to(cuda)
    assert hip.attn_fwd(e(0, 2, 128), k, k).shape == (0, 256)
    assert hip.attn_fwd_fp8(e(0, 2, 128), k, k).shape == (0, 256)
    q8, sc = hip.quant_fp8_rowwise(e(0, 128))
    assert q8.shape == (0, 128) and sc.shape == (0,)
    assert hip.dequant_fp8_rowwise(q8, sc).shape == (0, 128)
    w8, ws = hip.quant_fp8_rowwise(w)
    assert hip.gemm_fp8(q8, sc, w8, ws, b).shape == (0, 64)
    with pytest.raises(hip.Lx2vError):
        hip.attn_fwd(_rand((4, 2, 128), 3).to(cuda), e(0, 2, 128), e(0, 2, 128))  # queries but no keys
