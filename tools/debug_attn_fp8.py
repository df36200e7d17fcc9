"""GPU box: decode the fp8 attention workspace and compare each piece with the CPU emulation."""
import math, sys, torch
sys.path.insert(0, ".")
from lightx2v_amd import hip
from oracle import attn_fp8_ref as A8, wan_block_ref as R

def cosf(a, b): return R.cosine(a.float().cpu(), b.float().cpu())
dev = torch.device("cuda:0")
sq = skv = 64; H = 1
g = torch.Generator().manual_seed(1)
mode = sys.argv[1] if len(sys.argv) > 1 else "rand"
q = torch.randn(sq, H, 128, generator=g); k = torch.randn(skv, H, 128, generator=g); v = torch.randn(skv, H, 128, generator=g)
if mode == "flatscale":  # every block has amax exactly 1 -> all scales equal
    for t in (q, k, v):
        t.clamp_(-1, 1)
    q.view(sq, H, 4, 32)[..., 0] = 1.0 / (128 ** -0.5 * math.log2(math.e)); k.view(skv, H, 4, 32)[..., 0] = 1.0; v[::1, :, :] = v.clamp(-1, 1); v[0::4] = v[0::4].sign()
q, k, v = (t.to(torch.bfloat16) for t in (q, k, v))
need = hip.attn_fp8_workspace_bytes(sq, skv, H)
ws = torch.zeros(need, dtype=torch.uint8, device=dev)
out = hip.attn_fwd_fp8(q.to(dev), k.to(dev), v.to(dev), workspace=ws).float().cpu()
w = ws.cpu()
q8 = w[: sq * H * 128].view(torch.float8_e4m3fn).view(sq, H, 128)
qs = w[sq * H * 128: sq * H * 132].view(sq, H, 4).int()
c = 128 ** -0.5 * math.log2(math.e)
q_ref, qe_ref = A8.mx_quant(q.float() * c)
print("Q bytes equal:", torch.equal(q8.view(torch.uint8), q_ref.view(torch.uint8)), " Q scales equal:", torch.equal(qs, qe_ref.int()))
off = ((sq * H * 132 + 255) // 256) * 256
blob = w[off: off + 17152]
kd = torch.empty(64, 128, dtype=torch.uint8)
for key in range(64):
    for ch in range(8):
        p = key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)
        kd[key, ch * 16:(ch + 1) * 16] = blob[p:p + 16]
ks = blob[8192:8704].view(64, 2, 4)[:, :, :2].permute(0, 2, 1).reshape(64, 4).int()  # [key][h][ks] -> block 2 ks + h
k_ref, ke_ref = A8.mx_quant(k.float())
print("K bytes equal:", torch.equal(kd, k_ref.view(torch.uint8)[:, 0]), " K scales equal:", torch.equal(ks, ke_ref[:, 0].int()))
vd = torch.empty(128, 64, dtype=torch.uint8)
for d in range(128):
    for ch in range(4):
        p = 8704 + d * 64 + ((ch ^ ((d >> 2) & 3)) << 4)
        vd[d, ch * 16:(ch + 1) * 16] = blob[p:p + 16]
vs = blob[16896:17152].view(2, 32, 4).int()  # [h][d&31][d>>5]
slots = torch.tensor([32 * (j >> 4) + 8 * ((j & 15) >> 2) + 4 * h + (j & 3) for h in range(2) for j in range(32)])
vq_ref, ve_ref = A8.mx_quant(v.float()[:, 0], dim=0)  # [key][d], e: [2][d]
print("V bytes equal:", torch.equal(vd, vq_ref.view(torch.uint8)[slots].t().contiguous()), " V scales equal:",
      torch.equal(vs.permute(0, 2, 1).reshape(2, 128), ve_ref.int()))
ref = R.sdpa(q.float(), k.float(), v.float()); emu = A8.attention_fp8(q, k, v)
print("out vs fp32", cosf(out, ref), " out vs emu", cosf(out, emu), " emu vs fp32", cosf(emu, ref))
