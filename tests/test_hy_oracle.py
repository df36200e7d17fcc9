"""CPU: the HunyuanVideo-1.5 restatement (oracle/hy_block_ref.py) against the fixtures that
oracle/gen_golden_hy.py produced by running the reference's own classes, plus host-side checks of
the product's containers (no GPU compute)."""
import numpy as np
import pytest
import torch

import lightx2v_amd as L
from oracle import hy_block_ref as HY
from oracle import wan_block_ref as R
from tests.util import load_golden

HY_CASES = ["hy_tiny_d256", "hy_d2048_g3x6x8"]


def hy_case_inputs(rec):
    grid = tuple(int(v) for v in rec["grid"])
    dim, heads, mlp, txt_len, depth, seed = (int(rec[k]) for k in ("dim", "heads", "mlp", "txt_len", "depth", "seed"))
    L_ = grid[0] * grid[1] * grid[2]
    blocks = [HY.make_double_block_weights(dim, heads, mlp, seed=seed + i, block_index=i) for i in range(depth)]
    fw = HY.make_final_weights(dim, 32, seed=seed)
    img, txt, vec = HY.make_inputs(L_, txt_len, dim, seed=seed)
    cos_sin = HY.build_cos_sin(grid, dim // heads, [int(v) for v in rec["rope_dim_list"]])
    return dict(grid=grid, dim=dim, heads=heads, mlp=mlp, depth=depth, L=L_, Lt=txt_len, blocks=blocks, fw=fw, img=img, txt=txt, vec=vec, cos_sin=cos_sin)


@pytest.mark.parametrize("name", HY_CASES)
def test_restatement_matches_reference_run(name):
    rec = load_golden(name)
    c = hy_case_inputs(rec)
    trace = HY.DoubleBlockTrace()
    img, txt = HY.double_block_forward(c["img"][0], c["txt"][0], c["vec"], c["blocks"][0], c["cos_sin"], c["heads"], trace=trace)
    assert (img - torch.from_numpy(rec["img_out"])).abs().max().item() <= 2e-5
    assert (txt - torch.from_numpy(rec["txt_out"])).abs().max().item() <= 2e-5
    out = HY.forward(c["img"][0], c["txt"][0], c["vec"], c["blocks"], c["fw"], c["cos_sin"], c["heads"])
    assert (out - torch.from_numpy(rec["final_out"])).abs().max().item() <= 2e-5
    if "cos_sin" in rec.files:
        assert torch.equal(c["cos_sin"], torch.from_numpy(rec["cos_sin"]))
        for k, v in trace.stages.items():
            assert (v - torch.from_numpy(rec["stage_" + k])).abs().max().item() <= 2e-5, k


def test_rope_matches_complex_form():
    """The interleaved-pair rotation is the same map as Wan's complex multiplication."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(24, 3, 128, generator=g)
    cs = HY.build_cos_sin((2, 3, 4))
    cplx = torch.complex(cs[:, :64], cs[:, 64:]).unsqueeze(1)
    a, _ = R.rope_apply(x, x, cplx)
    assert (HY.rope_apply(x, cs) - a.view(24, 3, 128)).abs().max().item() < 1e-6


@pytest.mark.parametrize("world", [2, 4])
def test_joint_ulysses_model_matches_reference_run(world):
    rec = load_golden(f"hy_ulysses_txt_n{world}")
    li, lt, H, d, seed = (int(rec[k]) for k in ("li", "lt", "H", "d", "seed"))
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(li + lt, H, d, generator=g) for _ in range(3))
    sl = li // world
    shards = [[torch.cat([t[r * sl:(r + 1) * sl], t[li:]]) for r in range(world)] for t in (q, k, v)]
    mine = HY.ulysses_joint_attention(*shards, img_len_local=sl)
    for r in range(world):
        assert (mine[r] - torch.from_numpy(rec["out"][r])).abs().max().item() < 1e-5


def test_container_names_and_roundtrip():
    cfg = {"mm_double_blocks_depth": 2, "heads_num": 2, "hidden_size": 256}
    W = L.HunyuanVideo15TransformerWeights(cfg)
    wd = {}
    for i in range(2):
        wd.update(HY.make_double_block_weights(256, 2, 1024, seed=i, block_index=i))
    wd.update(HY.make_final_weights(256, 32))
    W.load(wd)
    sd = W.state_dict()
    assert set(sd) == set(wd)
    ib = W.double_blocks[1].img_branch
    assert ib.img_attn_q.weight_nk.shape == (256, 256) and ib.img_mod.weight_nk.shape == (6 * 256, 256)
    assert ib.img_attn_q_norm.weight.shape == (128,) and ib.img_norm1.weight is None
    fq = ib.fused_qkv()
    assert fq.weight_nk.shape == (768, 256) and ib.img_attn_v.weight_nk.data_ptr() == fq.weight_nk[512:].data_ptr()
    with pytest.raises(KeyError):
        L.HunyuanVideo15TransformerWeights(dict(cfg, dit_quant_scheme="nope"))
    with pytest.raises(NotImplementedError):
        L.HunyuanVideo15TransformerWeights(dict(cfg, cpu_offload=True))


def test_product_rope_table_equals_reference_table():
    from lightx2v_amd.rope_table import build_hunyuan_rope_table

    rec = load_golden("hy_tiny_d256")
    assert torch.equal(build_hunyuan_rope_table((2, 4, 6)), torch.from_numpy(rec["cos_sin"]))
    assert torch.equal(build_hunyuan_rope_table((3, 5, 7)), HY.build_cos_sin((3, 5, 7)))
