#!/usr/bin/env python3
"""Full-size check and timing of the sequence-parallel attention path on an RCCL group of one rank
against the plain attention call on the same q, k, v (14B at 720p: 75 600 tokens, 40 heads)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightx2v_amd as L  # noqa: E402
from lightx2v_amd import hip, ulysses  # noqa: E402
from lightx2v_amd.registry_factory import ATTN_WEIGHT_REGISTER  # noqa: E402

S, H, d = int(os.environ.get("S", 75600)), int(os.environ.get("H", 40)), 128
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.randn(S, H, d, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
plain = ATTN_WEIGHT_REGISTER["hip_flash_attn"]()
uly = ATTN_WEIGHT_REGISTER["hip_ulysses"]()
group = dist.new_group(ranks=[0])


def t(f, n=3):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = f()
    e1.record()
    torch.cuda.synchronize()
    return r, e0.elapsed_time(e1) / n


ref, t_ref = t(lambda: plain.apply(q, k, v, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=S, max_seqlen_kv=S))
got, t_got = t(lambda: uly.apply(q, k, v, img_qkv_len=S, cu_seqlens_qkv=[0, S], attention_module=plain, seq_p_group=group))
diff = (got.float() - ref.float()).abs()
print(f"plain {t_ref:.2f} ms   ulysses(one rank, RCCL) {t_got:.2f} ms   max|diff| {float(diff.max()):.3e}  mean|ref| {float(ref.float().abs().mean()):.3e}  "
      f"equal={bool(torch.equal(got, ref))}", flush=True)
per_head = diff.view(S, H, d).amax(dim=(0, 2))
print("per-head max|diff|:", [f"{float(x):.1e}" for x in per_head], flush=True)

# ---- stage by stage, one chunk of w heads ----
w = H // 2
for c in range(2):
    cols = slice(c * w * d, (c + 1) * w * d)
    q2, k2, v2 = (x.view(S, H * d) for x in (q, k, v))
    send = hip.ulysses_pack_qkv(q2[:, cols], k2[:, cols], v2[:, cols], 1)
    torch.cuda.synchronize()
    ok_pack = all(bool(torch.equal(send[0, :, i], x[:, cols])) for i, x in enumerate((q2, k2, v2)))
    recv = torch.full((S, 3, w * d), float("nan"), device=dev, dtype=torch.bfloat16)
    work = dist.all_to_all_single(recv.view(1, S, 3, w * d), send, group=group, async_op=True)
    work.wait()
    torch.cuda.synchronize()
    ok_a2a = bool(torch.equal(recv.view(-1), send.view(-1)))
    bad = (recv.view(-1) != send.view(-1)).nonzero()
    r = recv.view(S, 3, w, d)
    o = plain.apply(q=r[:, 0], k=r[:, 1], v=r[:, 2], cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=S, max_seqlen_kv=S)
    torch.cuda.synchronize()
    dd = float((o.float() - ref[:, cols].float()).abs().max())
    print(f"chunk {c}: pack ok={ok_pack}  all_to_all ok={ok_a2a} (first bad element {int(bad[0]) if bad.numel() else -1}, bad count {bad.numel()} of {send.numel()})  attention on recv: max|diff| {dd:.3e}", flush=True)
    o2 = plain.apply(q=send.view(S, 3, w, d)[:, 0], k=send.view(S, 3, w, d)[:, 1], v=send.view(S, 3, w, d)[:, 2], cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=S, max_seqlen_kv=S)
    print(f"         attention on the send buffer: max|diff| {float((o2.float() - ref[:, cols].float()).abs().max()):.3e}", flush=True)
dist.destroy_process_group()
