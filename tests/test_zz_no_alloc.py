"""The fp8 attention entry point that allocates nothing (`lx2v_attn_fwd_fp8_static`: the quantised
q / K / V images live in a fixed per-GPU array inside the library) against the entry point that
takes a workspace, through the C ABI, on two calls in a row, the second with a smaller input (the
array then still holds the first call's images)."""
import ctypes

import pytest
import torch

from lightx2v_amd import hip

pytestmark = pytest.mark.gpu


def _qkv(cuda, s, heads, seed):
    g = torch.Generator(device=cuda).manual_seed(seed)
    return tuple(torch.randn(s, heads, 128, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16) for _ in range(3))


def _call_static(q, k, v, out):
    """Straight through ctypes: raw pointers, sizes, strides, the current stream."""
    lib = hip.load()
    sq, heads, d = q.shape
    rc = lib.lx2v_attn_fwd_fp8_static(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), out.data_ptr(), out.stride(0),
                                      sq, k.shape[0], heads, d, ctypes.c_float(d ** -0.5), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.lx2v_last_error()


def test_static_scratch_entry_point_matches_the_workspace_one(cuda):
    big = _qkv(cuda, 4680, 12, 1)      # 22.6 MB of images: most of the 32 MiB array
    small = _qkv(cuda, 1000, 5, 2)     # then a smaller call over the same array
    for q, k, v in (big, small):
        sq, heads, d = q.shape
        ref = hip.attn_fwd_fp8(q, k, v)            # existing entry point (workspace allocated by the wrapper)
        out = torch.full((sq, heads * d), float("nan"), dtype=torch.bfloat16, device=cuda)
        before = torch.cuda.memory_allocated()
        _call_static(q, k, v, out)
        assert torch.cuda.memory_allocated() == before
        torch.cuda.synchronize()
        assert torch.equal(out, ref)               # same kernels, same images: bit for bit
    # cross-attention shape (sq != skv) and a strided q|k|v buffer
    g = torch.Generator(device=cuda).manual_seed(3)
    qkv = torch.randn(700, 3 * 4 * 128, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16)
    q, k, v = (qkv[:, i * 512:(i + 1) * 512].unflatten(1, (4, 128)) for i in range(3))
    ref = hip.attn_fwd_fp8(q[:300], k, v)
    out = torch.empty((300, 512), dtype=torch.bfloat16, device=cuda)
    _call_static(q[:300], k, v, out)
    assert torch.equal(out, ref)


def test_static_scratch_entry_point_rejects_what_does_not_fit(cuda):
    q, k, v = _qkv(cuda, 64, 1, 4)
    lib = hip.load()
    out = torch.empty((64, 128), dtype=torch.bfloat16, device=cuda)
    rc = lib.lx2v_attn_fwd_fp8_static(q.data_ptr(), 128, k.data_ptr(), 128, v.data_ptr(), 128, out.data_ptr(), 128,
                                      200000, 200000, 40, 128, ctypes.c_float(0.1), None)   # sizes only: nothing is launched
    assert rc == -2 and b"fixed array" in lib.lx2v_last_error()
    with pytest.raises(hip.Lx2vError, match="expected shape"):
        hip.attn_fwd_fp8_static(q, k, v, out[:32])


def test_static_scratch_entry_point_under_graph_capture(cuda):
    """What the entry point is for: one warm-up call outside capture (it resolves the device's copy of
    the array), then the call is recorded into a graph and replayed twice; every replay reproduces
    the workspace entry point's result bit for bit.  The second shape is long enough per head
    (60 000 keys) for the round alignment, which a captured launch must leave off."""
    for s, heads, seed in ((1500, 4, 5), (60000, 1, 6)):
        q, k, v = _qkv(cuda, s, heads, seed)
        ref = hip.attn_fwd_fp8(q, k, v)
        out = torch.empty_like(ref)
        _call_static(q, k, v, out)                 # outside capture
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                _call_static(q, k, v, out)
        for _ in range(2):
            out.fill_(float("nan"))
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
        hip.attn_fwd_fp8(q, k, v)                  # and the eager path still runs afterwards (counters untouched by the replays)
        torch.cuda.synchronize()
