"""Host-side behaviour that needs no GPU: registry contract, module tree, RoPE table, the C ABI's
exported symbols, and loud failure (never a fallback) when an op is handed a CPU tensor."""
import ctypes
import os
import re

import pytest
import torch

import lightx2v_amd as L
from lightx2v_amd import hip, plugins, registry_factory, rope_table, wan_block
from oracle import wan_block_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_contract():
    reg = registry_factory.Register()

    @reg("a")
    class A:
        pass

    @reg
    class B:
        pass

    assert reg["a"] is A and reg["B"] is B and "a" in reg and "zzz" not in reg
    with pytest.raises(Exception, match="already exists"):
        reg.register(A, key="a")
    with pytest.raises(KeyError):
        reg["missing"]
    with pytest.raises(Exception, match="must be callable"):
        reg.register(3, key="x")
    assert set(reg.keys()) == {"a", "B"}


def test_plugins_registered_under_reference_registry_names():
    assert "hip_flash_attn" in L.ATTN_WEIGHT_REGISTER and "hip_ulysses" in L.ATTN_WEIGHT_REGISTER
    assert "hip-bf16" in L.MM_WEIGHT_REGISTER and "hip" in L.LN_WEIGHT_REGISTER and "hip" in L.RMS_WEIGHT_REGISTER
    assert "Default" in L.TENSOR_REGISTER


def test_adopt_into_foreign_registries():
    class Foreign:
        MM_WEIGHT_REGISTER = registry_factory.Register()
        ATTN_WEIGHT_REGISTER = registry_factory.Register()

    added = registry_factory.adopt_into(Foreign)
    assert "hip-bf16" in Foreign.MM_WEIGHT_REGISTER and "hip_flash_attn" in Foreign.ATTN_WEIGHT_REGISTER
    assert added["ATTN_WEIGHT_REGISTER"] == ["hip_flash_attn", "hip_fp8_attn", "hip_fp8_attn_linear", "hip_ulysses"]
    assert registry_factory.adopt_into(Foreign) == {}  # idempotent


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lx2v_hip.h")).read()
    declared = set(re.findall(r"\b(lx2v_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    lib = ctypes.CDLL(hip.lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.load().lx2v_abi_version() == hip.ABI_VERSION == 4


def test_ops_refuse_cpu_tensors_loudly():
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(hip.Lx2vError, match="GPU tensor"):
        hip.ln_modulate(x)
    with pytest.raises(hip.Lx2vError, match="GPU tensor"):
        hip.gemm(x, x)
    with pytest.raises(hip.Lx2vError):
        L.ATTN_WEIGHT_REGISTER["hip_flash_attn"]().apply(torch.zeros(4, 1, 128, dtype=torch.bfloat16), torch.zeros(4, 1, 128, dtype=torch.bfloat16), torch.zeros(4, 1, 128, dtype=torch.bfloat16))


def test_c_abi_argument_validation_without_launching():
    lib = hip.load()
    # null pointers / bad shapes are rejected before any HIP call
    assert lib.lx2v_gemm_bf16(None, 64, None, None, 64, None, None, 64, 8, 8, 64, 0, None, 0, None, None) == -1
    assert b"null" in lib.lx2v_last_error()
    buf = ctypes.create_string_buffer(4096 + 16)
    p = (ctypes.addressof(buf) + 15) & ~15
    assert lib.lx2v_gemm_bf16(p, 64, None, p, 64, None, p, 64, 8, 8, 63, 0, None, 0, None, None) == -2   # K % 64
    bad = hip.KSeg(seg=64, stride=64, split=32, seg2=64, stride2=64, base2=0, lda2=64)
    assert lib.lx2v_gemm_bf16(p, 64, ctypes.byref(bad), p, 64, None, p, 64, 8, 8, 128, 0, None, 0, None, None) == -1  # region 1 not a whole number of segments
    assert lib.lx2v_attn_fwd_bf16(p, 128, p, 128, p, 128, p, 128, 8, 8, 1, 64, 0.1, None, None) == -2      # head_dim != 128
    assert lib.lx2v_ln_modulate_bf16(p, 64, p, None, None, None, p, 64, 4, 64, 1e-6, None) == -1          # ln_w without ln_b
    assert lib.lx2v_ln_modulate_bf16(p + 2, 64, None, None, None, None, p, 64, 4, 64, 1e-6, None) == -1   # misaligned
    assert lib.lx2v_rmsnorm_rope_bf16(p, 64, None, 1, None, 128, p, 64, 4, 9000, 1e-6, None) == -2        # dim too large / not multiple of 8
    assert lib.lx2v_ulysses_pack_qkv_bf16(p, 64, p, 64, p, 64, p, 4, 64, 3, None) == -1                  # dim % (8*world)
    # zero rows is a no-op success
    assert lib.lx2v_ln_modulate_bf16(p, 64, None, None, None, None, p, 64, 0, 64, 1e-6, None) == 0


@pytest.mark.parametrize("grid", [(2, 4, 4), (3, 6, 8), (21, 30, 52)])
def test_rope_table_matches_oracle(grid):
    mine = rope_table.build_rope_table(grid, 128)
    ref = R.cos_sin_to_real(R.build_cos_sin(grid, 128))
    assert mine.shape == ref.shape == (grid[0] * grid[1] * grid[2], 128)
    assert (mine - ref).abs().max().item() < 2e-6
    assert rope_table.axis_slots(128) == (22, 21, 21)


def test_rope_table_sharding_pads_with_zeros():
    t = rope_table.build_rope_table((1, 3, 3), 128)  # 9 tokens
    parts = [rope_table.shard_rope_table(t, 4, r) for r in range(4)]
    assert all(p.shape == (3, 128) for p in parts)
    full = torch.cat(parts)
    assert torch.equal(full[:9], t) and torch.count_nonzero(full[9:]) == 0
    ref = torch.cat([R.cos_sin_to_real(R.pad_and_shard_cos_sin(R.build_cos_sin((1, 3, 3), 128), 4, r)) for r in range(4)])
    assert (full - ref).abs().max().item() < 2e-6


def _cfg(**kw):
    c = {"num_layers": 2, "num_heads": 2, "dim": 256, "task": "t2v"}
    c.update(kw)
    return wan_block.with_defaults(c)


def test_weight_tree_names_and_load_roundtrip():
    cfg = _cfg()
    W = L.WanTransformerWeights(cfg)
    wd = {}
    for i in range(2):
        wd.update(R.make_block_weights(256, 512, seed=7, block_index=i))
    wd.update(R.make_head_weights(256, 64, seed=7))
    expected_keys = set(wd)
    W.load(dict(wd))
    sd = W.state_dict()
    assert set(sd) == expected_keys
    sa, ca, ff = W.blocks[1].compute_phases
    assert sa.self_attn_q.weight_nk.dtype == torch.bfloat16 and sa.self_attn_q.weight_nk.shape == (256, 256)
    assert sa.self_attn_q.weight.shape == (256, 256) and sa.self_attn_q.bias.dtype == torch.float32
    assert ff.ffn_0.weight_nk.shape == (512, 256) and ff.ffn_2.weight_nk.shape == (256, 512)
    assert ca.norm3.weight is not None and sa.norm1.weight is None and ff.norm2.weight is None
    assert torch.equal(sd["blocks.1.self_attn.q.weight"].float(), wd["blocks.1.self_attn.q.weight"].to(torch.bfloat16).float())
    names = [n for n, _ in W.named_parameters()]
    assert "blocks.0.compute_phases.0.self_attn_q" in names and "head_modulation" in names
    assert W.calculate_size() > 0
    # fused q|k|v aliases the same storage as the per-projection plugins afterwards
    fq = sa.fused_qkv()
    w, b = fq.weight_nk, fq.bias
    assert w.shape == (768, 256) and sa.self_attn_k.weight_nk.data_ptr() == w[256:512].data_ptr()
    assert torch.equal(sa.self_attn_v.bias, b[512:])


def test_unknown_plugin_names_fail():
    with pytest.raises(KeyError):
        L.WanTransformerWeights(_cfg(dit_quant_scheme="nope"))
    with pytest.raises(KeyError):
        L.WanTransformerWeights(_cfg(self_attn_1_type="flash_attn9"))
    with pytest.raises(ValueError):
        L.WanTransformerInfer(_cfg(rope_type="flashinfer"))


def test_pre_process_chunk_order():
    inf = L.WanTransformerInfer(_cfg())
    mod = plugins.DefaultTensor("blocks.0.modulation")
    mod.tensor = torch.arange(6 * 256, dtype=torch.float32).view(1, 6, 256)
    e0 = torch.ones(6, 256)
    parts = inf.pre_process(mod, e0)
    ref = (mod.tensor + e0).chunk(6, dim=1)
    assert len(parts) == 6
    for a, b in zip(parts, ref):
        assert torch.equal(a, b.reshape(-1))


def test_fused_qkv_follows_reloads_and_moves():
    """The fused q|k|v plugin is a cache of the members' weights: a second load() must replace it,
    and a member without a bias must not break the fusion."""
    cfg = _cfg(num_layers=1)
    W = L.WanTransformerWeights(cfg)
    w1 = R.make_block_weights(256, 512, seed=7)
    w1.update(R.make_head_weights(256, 64, seed=7))
    W.load(dict(w1))
    sa, ca, _ = W.blocks[0].compute_phases
    f1 = sa.fused_qkv()
    assert sa.fused_qkv() is f1  # cached while the members alias it
    kv1 = ca.fused_kv()
    w2 = R.make_block_weights(256, 512, seed=8)
    w2.update(R.make_head_weights(256, 64, seed=8))
    W.load(dict(w2))
    f2 = sa.fused_qkv()
    assert f2 is not f1 and ca.fused_kv() is not kv1
    assert torch.equal(f2.weight_nk[256:512].float(), w2["blocks.0.self_attn.k.weight"].to(torch.bfloat16).float())
    assert torch.equal(f2.bias[:256], w2["blocks.0.self_attn.q.bias"].float())
    # a member swapped by hand (e.g. a LoRA merge that re-assigns weight_nk) is noticed too
    sa.self_attn_v.weight_nk = sa.self_attn_v.weight_nk.clone()
    assert sa.fused_qkv() is not f2
    # bias-less members: zeros stand in, all-None stays None
    sa.self_attn_k.bias = None
    f3 = sa.fused("qkv_nobias_k", [sa.self_attn_q, sa.self_attn_k, sa.self_attn_v])
    assert torch.count_nonzero(f3.bias[256:512]) == 0 and sa.self_attn_k.bias is None and sa.self_attn_q.bias is not None
    for m in (sa.self_attn_q, sa.self_attn_k, sa.self_attn_v):
        m.bias = None
    assert sa.fused("qkv_nobias", [sa.self_attn_q, sa.self_attn_k, sa.self_attn_v]).bias is None
    # device moves drop the cache (checked with the CPU move, which needs no GPU)
    f4 = ca.fused_kv()
    W.to_cpu()
    assert ca.fused_kv() is not f4


def test_attention_plugins_refuse_varlen_descriptions_they_would_ignore():
    a = L.ATTN_WEIGHT_REGISTER["hip_flash_attn"]()
    q = torch.zeros(8, 1, 128, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="does not cover"):
        a.apply(q, q, q, cu_seqlens_q=torch.tensor([0, 8]), cu_seqlens_kv=torch.tensor([0, 6]))
    with pytest.raises(ValueError, match="max_seqlen"):
        a.apply(q, q, q, max_seqlen_q=8, max_seqlen_kv=6)
    with pytest.raises(NotImplementedError, match="one sequence"):
        a.apply(q, q, q, cu_seqlens_q=torch.tensor([0, 4, 8]))


def test_options_roundtrip_and_reject_unknown_names():
    assert hip.get_option("attn_kernel") in (16, 4, 8)
    old = hip.get_option("attn_round_sync")
    hip.set_option("attn_round_sync", 0)
    assert hip.get_option("attn_round_sync") == 0
    hip.set_option("attn_round_sync", old)
    with pytest.raises(hip.Lx2vError, match="unknown option"):
        hip.set_option("no_such_option", 1)
    with pytest.raises(hip.Lx2vError, match="attn_qb"):
        hip.set_option("attn_qb", 7)
    assert hip.get_option("attn_fp8_p") == 0          # exact exps unless asked otherwise
    with pytest.raises(hip.Lx2vError, match="attn_fp8_p"):
        hip.set_option("attn_fp8_p", 2)


def test_static_scratch_entry_point_rejects_oversize_without_touching_the_gpu():
    lib = hip.load()
    cap = hip.attn_fp8_static_bytes()
    assert cap == 32 << 20
    assert hip.attn_fp8_workspace_bytes(4680, 4680, 12) <= cap < hip.attn_fp8_workspace_bytes(32760, 32760, 12)
    buf = ctypes.create_string_buffer(4096 + 16)
    p = (ctypes.addressof(buf) + 15) & ~15
    assert lib.lx2v_attn_fwd_fp8_static(p, 1536, p, 1536, p, 1536, p, 1536, 32760, 32760, 12, 128, 0.1, None) == -2
    assert b"fixed array" in lib.lx2v_last_error()


def test_ulysses_chunk_count_follows_the_payload_limit():
    """A chunk's q|k|v payload per peer stays under the limit RCCL moves whole: more, narrower head
    chunks at sizes where two would not fit; multi-GPU shapes keep their default."""
    from lightx2v_amd import ulysses

    lim = ulysses.MAX_PEER_BYTES
    assert lim == 512 << 20
    # 14B at 720p on a group of one rank: 40 heads x 75 600 tokens = 2.3 GB -> five chunks of eight heads
    c = ulysses.chunks_for_payload(40, 2, 75600, 128)
    widths = ulysses.split_heads(40, c)
    assert c == 5 and widths == [8] * 5
    assert all(75600 * 3 * w * 128 * 2 <= lim for w in widths)
    # groups of 2 / 4 / 8 at the same size keep the default chunk count
    for world in (2, 4, 8):
        assert ulysses.chunks_for_payload(40 // world, ulysses.default_chunks(world), 75600 // world, 128) == ulysses.default_chunks(world)
    # never more chunks than heads
    assert ulysses.chunks_for_payload(2, 1, 10**7, 128) == 2
