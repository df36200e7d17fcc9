"""The CPU oracle (oracle/wan_block_ref.py) against the vectors the reference itself produced
(tests/golden/*.npz, written by oracle/gen_golden.py from the reference's torch_sdpa fp32 path)."""
import numpy as np
import pytest
import torch

from oracle import wan_block_ref as R
from tests.util import BLOCK_CASES, case_inputs, golden_rows, load_golden

TOL = 2e-5  # fp32 vs fp32: same ops, only thread-count dependent summation order may differ


@pytest.mark.parametrize("name", BLOCK_CASES)
def test_block_matches_reference_golden(name):
    rec = load_golden(name)
    c = case_inputs(rec)
    trace = R.BlockTrace()
    with torch.no_grad():
        out = R.block_forward(c["x"], c["embed0"], c["context"], c["w"], c["cos_sin"], c["heads"], trace=trace)
        head = R.head_forward(out, c["embed"], c["hw"])
    rows = golden_rows(rec, c["S"])
    ref = torch.from_numpy(rec["x_out"])
    assert (out[rows] - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    href = torch.from_numpy(rec["head_out"])
    assert (head[rows] - href).abs().max().item() <= TOL * max(1.0, href.abs().max().item())
    checked = 0
    for key in rec.files:
        if not key.startswith("stage_"):
            continue
        st = trace.stages[key[len("stage_"):]]
        g = torch.from_numpy(rec[key])
        mine = st[rows] if st.shape[0] == c["S"] and g.shape[0] == len(rows) else st
        assert mine.shape == g.shape, key
        assert (mine - g).abs().max().item() <= TOL * max(1.0, g.abs().max().item()), key
        checked += 1
    assert checked >= 7  # the 14B-dim fixture keeps 7 stages, the others all 18


def test_sdpa_against_float64_softmax():
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(s, 3, 128, generator=g) for s in (37, 53, 53))
    a = R.sdpa(q, k, v)
    b = R.sdpa_naive(q, k, v)
    assert (a.double() - b).abs().max().item() < 1e-5


def test_rope_is_interleaved_pair_rotation():
    grid = (2, 3, 4)
    cs = R.build_cos_sin(grid, 128)
    S = 24
    g = torch.Generator().manual_seed(5)
    q = torch.randn(S, 2, 128, generator=g)
    qr, _ = R.rope_apply(q, q, cs)
    real = R.cos_sin_to_real(cs)
    co, si = real[:, :64], real[:, 64:]
    a, b = q[..., 0::2], q[..., 1::2]
    exp = torch.stack([a * co[:, None] - b * si[:, None], a * si[:, None] + b * co[:, None]], dim=-1).flatten(2)
    assert (qr - exp).abs().max().item() < 1e-6
    # slot split 22 | 21 | 21, temporal slots depend only on f
    ang = torch.atan2(si, co)
    assert torch.allclose(ang[0, :22], ang[11, :22])      # same frame 0
    assert not torch.allclose(ang[0, :22], ang[12, :22])  # next frame


@pytest.mark.parametrize("world", [2, 4])
def test_ulysses_model_matches_reference_gloo_run(world):
    rec = load_golden(f"ulysses_n{world}")
    S, H, d, seed = (int(rec[k]) for k in ("S", "H", "d", "seed"))
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(S, H, d, generator=g) for _ in range(3))
    outs = R.ulysses_self_attention(list(q.chunk(world)), list(k.chunk(world)), list(v.chunk(world)))
    assert (torch.cat(outs, 0) - torch.from_numpy(rec["out"])).abs().max().item() < 1e-5
    s2h = R.seq2head_reference(list(q.chunk(world)))
    assert torch.equal(s2h[1], torch.from_numpy(rec["s2h_rank1"]))
    assert (torch.cat(outs, 0) - R.sdpa(q, k, v)).abs().max().item() < 1e-5


def test_int8_weight_quantiser_restatement_matches_the_references_own():
    """oracle.wan_block_ref.quant_int8_weight against the fixture the reference's IntegerQuantizer wrote."""
    import numpy as np
    import torch

    from oracle import wan_block_ref as R
    from tests.util import load_golden

    rec = load_golden("int8_weight_quant")
    w = torch.from_numpy(rec["w_bf16_bits"]).view(torch.bfloat16)
    q, s = R.quant_int8_weight(w)
    assert torch.equal(q, torch.from_numpy(rec["q"])) and torch.equal(s, torch.from_numpy(rec["scale"]))
    assert int(q[3].abs().max()) == 0 and float(s[3]) == np.float32(1e-5) / np.float32(127.0)   # the all-zero channel sits on the scale floor
    # the activation form: a row of zeros keeps scale 0; full range is used
    xq, xs = R.quant_int8_rowwise(torch.cat([w.float(), torch.zeros(1, w.shape[1])]))
    assert float(xs[-1]) == 0.0 and int(xq[-1].abs().max()) == 0 and int(xq.abs().max()) == 127
