import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import hip
dev = torch.device("cuda:0")
S, H, T = 75600, 40, 512
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(S, H, 128, generator=g, device=dev).to(torch.bfloat16)
kv = torch.randn(T, 2 * H * 128, generator=g, device=dev).to(torch.bfloat16)
k, v = kv[:, :H * 128].unflatten(1, (H, 128)), kv[:, H * 128:].unflatten(1, (H, 128))
def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for rs in (1, 0, 1, 0):
    hip.set_option("attn_kernel", 16); hip.set_option("attn_qb", 0); hip.set_option("attn_round_sync", rs)
    ms = timed(lambda: hip.attn_fwd(q, k, v))
    print(f"round_sync {rs}: {ms:.3f} ms", flush=True)
hip.set_option("attn_round_sync", 1)
for kern in (16,):
    for qb in ((0, 5, 4) if kern == 16 else (0,)):
        hip.set_option("attn_kernel", kern); hip.set_option("attn_qb", qb)
        ms = timed(lambda: hip.attn_fwd(q, k, v))
        print(f"kernel {kern} qb {qb}: {ms:.3f} ms = {4*S*T*H*128/ms/1e9:.0f} TFLOP/s; q+o traffic {2*S*H*128*2/ms/1e6:.0f} GB/s", flush=True)
