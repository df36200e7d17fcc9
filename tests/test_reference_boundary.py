"""The drop-in boundary against the REFERENCE's own objects (build container only: skipped where
/root/reference is absent, e.g. on the GPU box).

The reference's registries adopt this package's plugins, then the reference's own container classes
are constructed with the hip keys -- exactly what its JSON configs would do -- and its own
WeightModule tree drives `load` / `state_dict` / `to_cpu` on them.  Nothing is computed here: on CPU
tensors every `apply` of this package raises (no fallback), which is itself checked through the
reference's call sites, with the reference's argument conventions."""
import os

import pytest
import torch

REF = os.environ.get("LX2V_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lightx2v")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    from oracle.gen_golden import _install_reference

    _install_reference(REF)
    import lightx2v.common.ops  # noqa: F401  (registers the reference's own plugins)
    import lightx2v.utils.registry_factory as ref_reg

    from lightx2v_amd import registry_factory

    added = registry_factory.adopt_into(ref_reg)
    return ref_reg, added


def test_reference_registries_adopt_the_hip_plugins(ref):
    ref_reg, added = ref
    import lightx2v_amd as L

    assert ref_reg.ATTN_WEIGHT_REGISTER["hip_flash_attn"] is L.ATTN_WEIGHT_REGISTER["hip_flash_attn"]
    assert ref_reg.ATTN_WEIGHT_REGISTER["hip_ulysses"] is L.ATTN_WEIGHT_REGISTER["hip_ulysses"]
    assert ref_reg.MM_WEIGHT_REGISTER["hip-bf16"] is L.MM_WEIGHT_REGISTER["hip-bf16"] and "hip-fp8" in ref_reg.MM_WEIGHT_REGISTER
    assert "hip" in ref_reg.LN_WEIGHT_REGISTER and "hip" in ref_reg.RMS_WEIGHT_REGISTER
    assert "torch_sdpa" in ref_reg.ATTN_WEIGHT_REGISTER and "Default" in ref_reg.MM_WEIGHT_REGISTER  # theirs untouched
    # a second adoption finds nothing to add, and the reference's duplicate rule still holds
    from lightx2v_amd import registry_factory

    assert registry_factory.adopt_into(ref_reg) == {}
    with pytest.raises(Exception, match="already exists"):
        ref_reg.MM_WEIGHT_REGISTER.register(L.MM_WEIGHT_REGISTER["hip-bf16"], key="hip-bf16")


def test_reference_wan_block_built_with_hip_keys(ref):
    """lightx2v/models/networks/wan/weights/transformer_weights.py:80-490 with mm_type "hip-bf16" and
    attention types "hip_flash_attn" / "hip_ulysses": the Linears and attention objects are this
    package's; the reference's own constructor arguments, `load`, `state_dict` and key names work."""
    import lightx2v_amd as L
    from lightx2v.models.networks.wan.weights.transformer_weights import WanTransformerAttentionBlock
    from oracle import wan_block_ref as R

    D, F, H = 256, 512, 2
    cfg = {"task": "t2v", "model_cls": "wan2.1", "num_layers": 1, "num_heads": H, "dim": D, "self_attn_1_type": "hip_flash_attn",
           "cross_attn_1_type": "hip_flash_attn", "cross_attn_2_type": "hip_flash_attn", "seq_parallel": True,
           "parallel": {"seq_p_attn_type": "hip_ulysses"}, "cpu_offload": False}
    blk = WanTransformerAttentionBlock(0, "t2v", "hip-bf16", cfg)
    sa, ca, ff = blk.compute_phases
    mm_cls, attn_cls = L.MM_WEIGHT_REGISTER["hip-bf16"], L.ATTN_WEIGHT_REGISTER["hip_flash_attn"]
    linears = [sa.self_attn_q, sa.self_attn_k, sa.self_attn_v, sa.self_attn_o, ca.cross_attn_q, ca.cross_attn_k, ca.cross_attn_v, ca.cross_attn_o, ff.ffn_0, ff.ffn_2]
    assert all(type(m) is mm_cls for m in linears)
    assert type(sa.self_attn_1) is attn_cls and type(ca.cross_attn_1) is attn_cls
    assert type(sa.self_attn_1_parallel) is L.ATTN_WEIGHT_REGISTER["hip_ulysses"]
    w = R.make_block_weights(D, F, seed=3)
    for m in linears:   # the reference's own tensor / norm classes pin host memory in load(), which needs a GPU; ours do not
        m.load(w)
    sd = {}
    for m in linears:
        m.state_dict(sd)
    assert set(sd) == {k for k in w if k.endswith((".weight", ".bias")) and k.split(".")[-2] in ("q", "k", "v", "o", "0", "2")}
    assert sa.self_attn_q.weight.shape == (D, D) and ff.ffn_0.weight.shape == (D, F)   # the reference's [in, out] view
    assert torch.equal(sd["blocks.0.ffn.2.weight"].float(), w["blocks.0.ffn.2.weight"].to(torch.bfloat16).float())
    # the reference's call conventions bind to the plugins' signatures (a CPU tensor is refused, loudly)
    from lightx2v_amd.hip import Lx2vError

    x = torch.zeros(8, D, dtype=torch.bfloat16)
    with pytest.raises(Lx2vError, match="GPU tensor"):
        sa.self_attn_q.apply(x)                                                      # mm_weight.py:163
    q = torch.zeros(8, H, D // H, dtype=torch.bfloat16)
    cu = torch.tensor([0, 8], dtype=torch.int32)
    with pytest.raises(Lx2vError, match="GPU tensor"):                               # transformer_infer.py:190-199
        sa.self_attn_1.apply(q=q, k=q, v=q, cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=8, max_seqlen_kv=8, model_cls="wan2.1")
    # the block tree's own device moves reach the plugins
    for m in linears:
        m.to_cpu()
    assert not sa.self_attn_q.weight_nk.is_cuda


def test_reference_hunyuan_block_built_with_hip_keys(ref):
    """lightx2v/models/networks/hunyuan_video/weights/transformer_weights.py:10-357 reads every op type
    from its config, so with the hip keys the whole double-stream block is made of this package's
    plugins and the reference's WeightModule.load / state_dict run on them end to end."""
    import lightx2v_amd as L
    from lightx2v.models.networks.hunyuan_video.weights.transformer_weights import HunyuanVideo15TransformerWeights
    from oracle import hy_block_ref as HR

    D, H, F = 256, 2, 512
    cfg = {"task": "t2v", "mm_double_blocks_depth": 1, "heads_num": H, "hidden_size": D, "dit_quant_scheme": "hip-bf16", "ln_type": "hip",
           "rms_type": "hip", "attn_type": "hip_flash_attn", "seq_parallel": False, "cpu_offload": False}
    W = HunyuanVideo15TransformerWeights(cfg)
    blk = W.double_blocks[0]
    assert type(blk.self_attention) is L.ATTN_WEIGHT_REGISTER["hip_flash_attn"]
    assert type(blk.img_branch.img_attn_q) is L.MM_WEIGHT_REGISTER["hip-bf16"] and type(blk.txt_branch.txt_attn_k_norm) is L.RMS_WEIGHT_REGISTER["hip"]
    assert type(blk.img_branch.img_norm1) is L.LN_WEIGHT_REGISTER["hip"]
    wd = HR.make_double_block_weights(D, H, F, seed=4, block_index=0)
    blk.load(dict(wd))                       # the reference's WeightModule.load fanning out to the hip plugins
    sd = blk.state_dict()
    assert set(sd) == set(wd)
    assert blk.img_branch.img_mlp_fc1.weight_nk.shape == (F, D)
