"""Self-attention at the 720p x 81-frame token count (75 600), through the launch plan the headline
configuration uses: many rounds of one workgroup per CU with round alignment, 4 or 5 query blocks
per wave, and the short last round moved to a second launch of 64-row workgroups.  q, k and v are
strided views of one [S, 3*H*128] buffer, as the fused block hands them over.

A full fp32 reference at this size is hours of CPU time, so a few hundred query rows (workgroup
and launch boundaries, first and last rows, random others) are checked against the CPU oracle,
which is exact per row; whole outputs are compared between launch plans bit for bit where the
arithmetic is the same."""
import pytest
import torch

from lightx2v_amd import hip
from oracle import wan_block_ref as R
from tests.util import cos

pytestmark = pytest.mark.gpu

S, H, D = 75600, 13, 128   # 13 heads: 237 * 13 = 12 rounds of 256 + 9 and 296 * 13 = 15 rounds + 8 -> both plans have a tail


def _rows(sample_seed=5):
    edge = [0, 1, 15, 16, 63, 64, 255, 256, 319, 320, 321, 1279, 1280, S - 1, S - 2, S - 64, S - 65, S - 320, S - 321]
    for first_tail_row in ((237 - 9) * 320, (296 - 8) * 256):  # where the second launch starts (5 / 4 blocks per wave)
        edge += [first_tail_row - 1, first_tail_row, first_tail_row + 63, first_tail_row + 64]
    g = torch.Generator().manual_seed(sample_seed)
    rnd = torch.randint(0, S, (230,), generator=g).tolist()
    return torch.tensor(sorted(set(edge + rnd)))


@pytest.fixture(scope="module")
def case(cuda):
    g = torch.Generator(device=cuda).manual_seed(77)
    qkv = torch.randn(S, 3 * H * D, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16)
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D].unflatten(1, (H, D)) for i in range(3))
    assert q.stride(0) == 3 * H * D and not q.is_contiguous()
    rows = _rows()
    kf, vf = k.float().cpu(), v.float().cpu()
    ref = R.sdpa(q[rows.to(cuda)].float().cpu(), kf, vf)   # exact per query row: [rows, H*D]
    return q, k, v, rows, ref


@pytest.mark.parametrize("qb,tail", [(5, 1), (5, 0), (4, 1), (4, 0)])
def test_sampled_rows_match_the_cpu_oracle(case, options, qb, tail):
    q, k, v, rows, ref = case
    options(attn_kernel=16, attn_qb=qb, attn_tail=tail)
    out = hip.attn_fwd(q, k, v)
    assert out.shape == (S, H * D) and bool(torch.isfinite(out).all())
    got = out[rows.to(out.device)].float().cpu()
    assert cos(got, ref) >= 0.9999, cos(got, ref)
    # softmax over 75 600 N(0,1)-keyed scores averages V down to ~1e-2: the bound is relative to that scale
    assert (got - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1e-3)
    per_row = torch.nn.functional.cosine_similarity(got, ref, dim=1)
    assert per_row.min().item() >= 0.999, (rows[per_row.argmin()].item(), per_row.min().item())


def test_round_alignment_and_launch_split_do_not_change_results(case, options):
    """Round alignment only delays workgroups; the split of the last round changes which kernel
    instance computes the last rows of the last head and nothing else."""
    q, k, v, _, _ = case
    options(attn_kernel=16, attn_qb=0, attn_tail=1, attn_round_sync=1)
    base = hip.attn_fwd(q, k, v)
    again = hip.attn_fwd(q, k, v)
    assert torch.equal(base, again)                       # run-to-run deterministic
    options(attn_round_sync=0)
    assert torch.equal(hip.attn_fwd(q, k, v), base)       # a locality hint, not arithmetic
    options(attn_round_sync=1, attn_tail=0)
    one = hip.attn_fwd(q, k, v)
    first_tail_row = (237 - 9) * 320
    assert torch.equal(one[:first_tail_row], base[:first_tail_row])
    assert torch.equal(one[:, :(H - 1) * D], base[:, :(H - 1) * D])
    assert (one.float() - base.float()).abs().max().item() <= 2e-3   # same rows by 64-row workgroups: other rescale points only


def test_row_split_of_a_launch_with_few_heads(case, options):
    """A head chunk of a sequence-parallel rank: 5 heads x 75 600 rows = 4.63 rounds of 320-row
    workgroups.  The launch plan runs rows [0, 65 280) of every head in four full rounds and the other
    10 320 rows of every head as 256-row workgroups in a second launch; one launch (attn_tail = 0)
    computes the same rows with 320-row workgroups."""
    q, k, v, rows, ref = case
    q5, k5, v5 = q[:, :5], k[:, :5], v[:, :5]
    row0 = 204 * 320
    options(attn_kernel=16, attn_qb=0, attn_tail=1)
    base = hip.attn_fwd(q5, k5, v5)
    options(attn_tail=0)
    one = hip.attn_fwd(q5, k5, v5)
    assert torch.equal(one[:row0], base[:row0])           # the main launch is the same arithmetic
    assert not torch.equal(one[row0:], base[row0:])       # the second launch is live (other block count, other rescale points)
    assert (one.float() - base.float()).abs().max().item() <= 2e-3
    pick = torch.cat([rows, torch.tensor([row0 - 1, row0, row0 + 255, row0 + 256, S - 1])]).unique()
    want = R.sdpa(q5[pick.to(q.device)].float().cpu(), k5.float().cpu(), v5.float().cpu())
    got = base[pick.to(base.device)].float().cpu()
    assert cos(got, want) >= 0.9999
    assert torch.nn.functional.cosine_similarity(got, want, dim=1).min().item() >= 0.999


def test_w4_kernel_at_full_length(case, options):
    q, k, v, rows, ref = case
    options(attn_kernel=4)
    out = hip.attn_fwd(q[:, :2], k[:, :2], v[:, :2])   # 2 heads: 592 workgroups of 256 rows
    got = out[rows.to(out.device)].float().cpu()
    assert cos(got, ref[:, :2 * D]) >= 0.9999


def test_a_wave_held_back_behind_the_tile_barrier_changes_nothing(case, options):
    """The kernel reuses a K / V ring slot for LDS-DMA one barrier after its last reads were issued,
    without waiting for those reads.  That is safe when every read of the slot precedes the barrier
    in program order; the test build `build/x16_lag/liblx2v_hip.so` (make -C lightx2v_amd/csrc
    lagtest) makes one wave per tile sleep 512 cycles directly behind the barrier -- five steps of
    the others' progress, enough for their first DMA into the slot to be issued -- so a read of the
    slot that came after the barrier would now see overwritten data.  Same shape, same launch plan:
    the output must be bit-identical to the product library's."""
    import ctypes
    import os

    from lightx2v_amd.hip import _SIGNATURES

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "x16_lag", "liblx2v_hip.so")
    if not os.path.exists(path):
        pytest.skip("build/x16_lag/liblx2v_hip.so not built (make -C lightx2v_amd/csrc lagtest)")
    q, k, v, rows, ref = case
    options(attn_kernel=16, attn_qb=5, attn_tail=1)
    want = hip.attn_fwd(q, k, v)
    lag = ctypes.CDLL(path)
    lag.lx2v_attn_fwd_bf16.argtypes = _SIGNATURES["lx2v_attn_fwd_bf16"]
    lag.lx2v_attn_fwd_bf16.restype = ctypes.c_int
    lag.lx2v_last_error.restype = ctypes.c_char_p
    got = torch.empty_like(want)
    st = lag.lx2v_attn_fwd_bf16(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), got.data_ptr(), got.stride(0),
                                S, S, H, D, float(D ** -0.5), None, torch.cuda.current_stream().cuda_stream)
    assert st == 0, lag.lx2v_last_error()
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_no_key_split_where_the_row_split_already_fills_the_last_round(case, options):
    """3 heads x 75 600 rows: two full rounds of 320-row workgroups, then 249 workgroups of 256 rows -- 97 % of a round.
    No key split pays there, and the workspace query says so."""
    q, k, v, _, _ = case
    options(attn_kernel=16, attn_qb=0, attn_tail=1)
    assert hip.attn_workspace_bytes(S, S, 3) == 0
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=q.device)
    assert torch.equal(hip.attn_fwd(q[:, :3], k[:, :3], v[:, :3], workspace=ws), hip.attn_fwd(q[:, :3], k[:, :3], v[:, :3]))


@pytest.mark.parametrize("heads", [5, 2])
def test_key_split_of_the_last_round_of_a_launch_with_few_heads(case, options, heads):
    """Head chunks of a sequence-parallel rank (5 heads: Wan-14B at seq_p = 4 in two chunks; 2: the narrow chunk at seq_p = 8) are a
    few rounds of one workgroup per CU long.  Given a workspace, the launch plan splits the KEYS of the last round's
    query rows into ranges, whose partial results a merge kernel combines: the rows of the full rounds are computed
    exactly as without it, the merged rows agree up to the summation order, and both match the CPU oracle."""
    import time

    q, k, v, rows, ref = case
    qh, kh, vh = q[:, :heads], k[:, :heads], v[:, :heads]
    options(attn_kernel=16, attn_qb=0, attn_tail=1)
    need = hip.attn_workspace_bytes(S, S, heads)
    assert need > 0, "the plan for this shape should hold a key split"
    ws = torch.empty(need, dtype=torch.uint8, device=q.device)
    plain = hip.attn_fwd(qh, kh, vh)
    split = hip.attn_fwd(qh, kh, vh, workspace=ws)
    assert torch.equal(hip.attn_fwd(qh, kh, vh, workspace=ws), split)     # deterministic
    differ = (plain != split).any(dim=1).nonzero().flatten()
    assert differ.numel() > 0                                              # the split is live
    row0 = int(differ.min())
    if heads == 5:
        # both plans run four full rounds of 320-row workgroups first: those rows are the same arithmetic
        assert row0 == 204 * 320 and torch.equal(plain[:row0], split[:row0])
    # (2 heads: without a workspace the cheaper plan has 256-row workgroups, with it 320-row ones and a key split --
    # other rescale points in every row, the last bf16 bit of a few outputs)
    assert (plain.float() - split.float()).abs().max().item() <= 2e-3
    tail0 = 204 * 320 if heads == 5 else 128 * 320   # first row of the key-split launch
    pick = torch.cat([rows, torch.tensor([tail0 - 1, tail0, tail0 + 1, tail0 + 319, tail0 + 320, S - 1])]).unique()
    want = R.sdpa(qh[pick.to(q.device)].float().cpu(), kh.float().cpu(), vh.float().cpu())
    got = split[pick.to(split.device)].float().cpu()
    assert cos(got, want) >= 0.9999
    assert torch.nn.functional.cosine_similarity(got, want, dim=1).min().item() >= 0.999
    # a workspace that is too small only limits the plan
    small = hip.attn_fwd(qh, kh, vh, workspace=ws[:4096])
    assert torch.equal(small, plain)
    t = {}
    for name, w in (("plain", None), ("split", ws)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            hip.attn_fwd(qh, kh, vh, workspace=w)
        torch.cuda.synchronize()
        t[name] = (time.perf_counter() - t0) / 5 * 1e3
    fl = 4.0 * S * S * heads * D
    print(f"\n{heads} heads x {S} rows: rows split {t['plain']:.2f} ms ({fl / t['plain'] * 1e-9:.0f} TFLOP/s), with key split {t['split']:.2f} ms ({fl / t['split'] * 1e-9:.0f} TFLOP/s)")


@pytest.mark.parametrize("side", ["few_cus", "all_cus"])
def test_round_alignment_while_another_stream_takes_cus(case, options, side):
    """Round alignment assumes one workgroup per CU and round; a collective's kernels (or any other stream's) take CUs
    away while the attention runs, so rounds no longer line up and workgroups meet the wait's time limit instead of
    their round-mates.  The result must not change and the launch must not take longer than the two workloads
    back to back (plus slack): the wait is bounded, a missed alignment costs locality only."""
    q, k, v, _, _ = case
    options(attn_kernel=16, attn_qb=0, attn_tail=1, attn_round_sync=1)
    base = hip.attn_fwd(q, k, v)
    n = 256 if side == "few_cus" else 8192            # 256^2 outputs: a handful of workgroups per launch, like a collective's kernels
    reps = 3000 if side == "few_cus" else 40
    a = torch.randn(n, 8192, device=q.device, dtype=torch.bfloat16)
    b = torch.randn(8192, n, device=q.device, dtype=torch.bfloat16)
    other = torch.cuda.Stream()

    def side_work():
        with torch.cuda.stream(other):
            for _ in range(reps):
                torch.mm(a, b)

    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        return out, e0.elapsed_time(e1)

    side_work()                                        # warm-up (rocBLAS picks and loads its kernel)
    _, t_attn = timed(lambda: hip.attn_fwd(q, k, v))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(other):
        e0.record()
    side_work()
    with torch.cuda.stream(other):
        e1.record()
    torch.cuda.synchronize()
    t_side = e0.elapsed_time(e1)
    side_work()                                        # now both at once
    got, t_both = timed(lambda: hip.attn_fwd(q, k, v))
    torch.cuda.synchronize()
    print(f"\nattention alone {t_attn:.1f} ms, side work ({side}) alone {t_side:.1f} ms, attention beside it {t_both:.1f} ms")
    assert torch.equal(got, base)
    assert t_both <= 1.25 * (t_attn + t_side) + 5.0


def test_key_split_of_the_last_heads_tail(case, options):
    """Many heads: the last, sparse round is a few workgroups of the last head (here 13 x 237 = 12 rounds of 256 + 9).
    Without a workspace those rows go to 64-row workgroups over all keys; with one, to 320-row workgroups over 1 / 28
    of the keys each (9 x 28 = 252 workgroups: one short round) and a merge.  Everything else is the same arithmetic."""
    import time

    q, k, v, rows, ref = case
    options(attn_kernel=16, attn_qb=5, attn_tail=1)
    need = hip.attn_workspace_bytes(S, S, H)
    first_tail_row = (237 - 9) * 320
    assert need >= (S - first_tail_row) * 28 * (D + 2) * 4
    ws = torch.empty(need, dtype=torch.uint8, device=q.device)
    plain = hip.attn_fwd(q, k, v)
    split = hip.attn_fwd(q, k, v, workspace=ws)
    assert torch.equal(hip.attn_fwd(q, k, v, workspace=ws), split)         # deterministic
    assert torch.equal(split[:first_tail_row], plain[:first_tail_row])
    assert torch.equal(split[:, :(H - 1) * D], plain[:, :(H - 1) * D])
    assert not torch.equal(split[first_tail_row:, (H - 1) * D:], plain[first_tail_row:, (H - 1) * D:])   # the split is live
    assert (split.float() - plain.float()).abs().max().item() <= 2e-3
    got = split[rows.to(split.device)].float().cpu()
    assert cos(got, ref) >= 0.9999
    assert torch.nn.functional.cosine_similarity(got, ref, dim=1).min().item() >= 0.999
    assert torch.equal(hip.attn_fwd(q, k, v, workspace=ws[:4096]), plain)   # a workspace that is too small only limits the plan
    t = {}
    for name, w in (("plain", None), ("split", ws)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            hip.attn_fwd(q, k, v, workspace=w)
        torch.cuda.synchronize()
        t[name] = (time.perf_counter() - t0) / 5 * 1e3
    print(f"\n13 heads x {S}: 64-row tail {t['plain']:.2f} ms, key-split tail {t['split']:.2f} ms")
