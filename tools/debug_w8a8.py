"""Debugging aid: where the persistent W8A8 GEMM differs from the one-barrier kernel (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from lightx2v_amd import hip

dev = torch.device("cuda:0")
for int8 in (True, False):
    for (M, N, K) in [(256, 256, 128), (256, 256, 256), (256, 256, 512), (512, 512, 384)]:
        g = torch.Generator(device=dev).manual_seed(1)
        a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(torch.bfloat16)
        quant = hip.quant_int8_rowwise if int8 else hip.quant_fp8_rowwise
        gemm = hip.gemm_int8 if int8 else hip.gemm_fp8
        aq, a_s = quant(a)
        wq, w_s = quant(w)
        hip.set_option("gemm_kernel", 1)
        ref = gemm(aq, a_s, wq, w_s, None).float()
        hip.set_option("gemm_kernel", 0)
        got = gemm(aq, a_s, wq, w_s, None).float()
        torch.cuda.synchronize()
        bad = (got != ref)
        print(f"int8={int8} M={M} N={N} K={K}: mismatches {int(bad.sum())} of {M * N}, max diff {float((got - ref).abs().max()):.4f}")
        if bad.any():
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print("   bad rows:", rows[:40].tolist(), "... count", rows.numel())
            print("   bad cols:", cols[:40].tolist(), "... count", cols.numel())
            r, c = int(rows[0]), int(cols[0])
            print("   sample got/ref:", got[r, c:c + 8].tolist(), ref[r, c:c + 8].tolist())
            exact = (aq.float() @ wq.float().t()) * a_s[:, None] * w_s[None, :]
            print("   exact:", exact[r, c:c + 8].tolist())
