// Flash-style attention forward for gfx950, head_dim 128, bf16 in / bf16 out, non-causal.
//
// Workgroup = 8 waves = 256 query rows of one head (32 rows per wave); K/V tiles of 64 keys are
// double-buffered in LDS (2 x (16 KiB K + 16 KiB V)), staged through registers: the global
// loads of tile t+2 are issued while tile t is being consumed and are written to LDS one tile
// later, so HBM/L2 latency hides under a full tile of MFMA work.
//
// Per wave and tile:
//   S^T = K Q^T      v_mfma_f32_32x32x16_bf16, A = K rows (ds_read_b128 from an XOR-swizzled
//                    [64][128] image), B = Q^T fragments held in registers for the whole kernel.
//                    The result has the query row on the LANE and 32 of the 64 keys in the lane's
//                    registers, so the row maximum / row sum are in-lane reductions plus one
//                    half-wave exchange, and the per-row rescale is a per-lane multiply.
//   online softmax   fp32: p = exp2(s*c - m*c), c = scale*log2(e); the rescale of O is skipped
//                    (wave-uniform branch) on tiles where no row maximum grew.
//   O^T += V^T P^T   the S^T accumulator registers, converted pairwise to bf16, ARE the B operand
//                    (no LDS round trip, no shuffles); A = V^T fragments gathered with
//                    ds_read_b64_tr_b16 from a [8 keys][32 cols] sub-tiled V image in the k order
//                    the accumulator layout dictates (key = 16s + 8(j>>2) + 4h + (j&3)).
// Optional fused 3-axis RoPE: q and k are rotated (interleaved pairs, fp32) as they are loaded.
#include <stdlib.h>

#include "attn_common.h"

using namespace lx2v_attn;

namespace {

// SHORT_KV only gives the cross-attention launches (a few hundred keys) their own kernel symbol,
// so per-kernel profiles separate them from the long self-attention launches.
//
// VARIANT 0: plain order per tile: S^T, stage, softmax, PV, barrier.
// VARIANT 1: same work with the LDS reads software-pipelined: the K fragments of the first
// k-steps are requested right after the tile barrier, which sits BEFORE the last four PV MFMAs of
// the previous tile, and the remaining K fragments are requested behind the MFMAs that free their
// registers, so no S^T MFMA waits a full LDS round trip.
template <bool ROPE, bool SHORT_KV, int VARIANT>
__global__ __launch_bounds__(NTHREADS) void attn_fwd_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) char lds[4 * KV_TILE_BYTES];  // [buf][K|V]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;

  const int qb = blockIdx.x % p.nqb;
  const int head = blockIdx.x / p.nqb;
  const int q0 = qb * QROWS + wid * 32;

  // ---- Q^T fragments: lane holds c * Q[q0 + r32][16 ks + 8 h .. +7] for ks = 0..7 ----
  bf16x8 qf[8];
  u32x4 qraw[8];
  const int qrow_c = min(q0 + r32, p.sq - 1);
  {
    const uint16_t* qp = p.q + (int64_t)qrow_c * p.ldq + head * HD + h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qraw[ks] = *reinterpret_cast<const u32x4*>(qp + ks * 16);
  }

  // ---- K/V staging: thread owns chunks c = tid + 512 i (i = 0,1): row = c >> 4, ch = c & 15 ----
  const int srow0 = tid >> 4, sch = tid & 15;  // second chunk: row + 32
  const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(p.k + head * HD, (uint32_t)((int64_t)p.skv * p.ldk * 2));
  const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(p.v + head * HD, (uint32_t)((int64_t)p.skv * p.ldv * 2));
  const __amdgpu_buffer_rsrc_t t_rs = make_rsrc(ROPE ? p.rope : nullptr, ROPE ? (uint32_t)((int64_t)p.skv * HD * 4) : 0u);
  int k_vo[2], v_vo[2], t_vo[2];  // per-lane byte offsets inside a tile; the tile offset is scalar
  int k_wr[2], v_wr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = srow0 + 32 * i;
    k_vo[i] = (int)(row * p.ldk * 2) + sch * 16;
    v_vo[i] = (int)(row * p.ldv * 2) + sch * 16;
    t_vo[i] = row * (HD * 4) + sch * 16;
    k_wr[i] = k_lds_off(row, sch);
    v_wr[i] = KV_TILE_BYTES + v_lds_off(row, sch);
  }
  const int k_tile_bytes = (int)(KVB * p.ldk * 2), v_tile_bytes = (int)(KVB * p.ldv * 2);
  u32x4 kreg[2], vreg[2];
  f32x4 rcos[2], rsin[2];  // only live when ROPE
  // Loads only: nothing here may consume the loaded registers, or the wait for them would land
  // at issue time instead of one tile later.
  auto load_tile = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rs, k_vo[i], t * k_tile_bytes, 0));
      vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rs, v_vo[i], t * v_tile_bytes, 0));
      if (ROPE) {
        rcos[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(t_rs, t_vo[i], t * (KVB * HD * 4), 0));
        rsin[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(t_rs, t_vo[i] + 256, t * (KVB * HD * 4), 0));
      }
    }
  };
  // Rotate (ROPE) and write to LDS buffer BUF.
  auto write_tile = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    char* base = lds + BUF * 2 * KV_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 kv = kreg[i];
      if (ROPE) {
        float f[8];
        unpack8(kv, f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = f[2 * j], b = f[2 * j + 1];
          f[2 * j] = a * rcos[i][j] - b * rsin[i][j];
          f[2 * j + 1] = a * rsin[i][j] + b * rcos[i][j];
        }
        kv = pack8(f);
      }
      *reinterpret_cast<u32x4*>(base + k_wr[i]) = kv;
      *reinterpret_cast<u32x4*>(base + v_wr[i]) = vreg[i];
    }
  };

  // ---- fragment read offsets (per-lane VGPRs; buffer / key-block / d-block offsets are immediates) ----
  // K (A operand of S^T): row = kb*32 + r32, chunk = 2 ks + h, swizzled by row & 15 == r32 & 15
  int k_rd[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_rd[ks] = r32 * 256 + (((2 * ks + h) ^ (r32 & 15)) << 4);
  // V^T (A operand of O^T): per 16-lane group g2 = (lane>>4)&1, lane-in-group li = lane & 15 -> q4 = li>>2, p4 = li&3
  //   block rows r0 = 32 kb + 16 s + 8 u + 4 h (u = 0,1 -> elements 0..3 / 4..7), cols 32 db + 16 g2 .. +15
  //   this lane supplies row r0 + q4, chunk c0 + (p4 >> 1) with c0 = 4 db + 2 g2, byte +8 (p4 & 1)
  const int g2 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  int v_rd[4][2];  // [kstep = 2 kb + s][u]; db adds 512 * db (chunk >> 2 == db since c0 = 4 db + ...)
#pragma unroll
  for (int ks4 = 0; ks4 < 4; ++ks4)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 16 * ks4 + 8 * u + 4 * h + q4;
      const int chl = 2 * g2 + (p4 >> 1);  // chunk & 3
      v_rd[ks4][u] = KV_TILE_BYTES + 2048 * (row >> 3) + 64 * (row & 7) + 16 * (chl ^ ((row >> 2) & 3)) + 8 * (p4 & 1);
    }

  f32x16 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
  // Running maximum m of this lane's query row (log2 domain, scores already scaled through Q).
  // negm holds -m in all 16 registers: it is the C operand of the first S^T MFMA of every tile, so
  // the accumulators come out as s - m and exp2 needs no per-element subtract.
  f32x16 negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  float l_run = 0.f;

  const int nt = (p.skv + KVB - 1) / KVB;
  const bool ragged = (p.skv & (KVB - 1)) != 0;

  // ---- S^T - m = K (cQ)^T - m for the tile in LDS buffer BUF ----
  auto qk = [&](auto bufc, f32x16 (&s)[2]) {
    constexpr int BUF = decltype(bufc)::value;
    const char* kb_base = lds + BUF * 2 * KV_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(kb_base + kb * 8192 + k_rd[ks]);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], ks == 0 ? negm : s[kb], 0, 0, 0);
      }
    }
  };
  constexpr int KPF = 2;  // VARIANT 1: k-steps of K fragments kept in flight ahead of their MFMAs
  bf16x8 ka[2 * KPF];     // [2 (ks % KPF) + kb]
  auto k_prefetch = [&](auto bufc) __attribute__((always_inline)) {  // k-steps 0..KPF-1 of the tile in buffer BUF
    constexpr int BUF = decltype(bufc)::value;
    const char* kb_base = lds + BUF * 2 * KV_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < KPF; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) ka[2 * ks + kb] = *reinterpret_cast<const bf16x8*>(kb_base + kb * 8192 + k_rd[ks]);
  };
  auto qk_pipe = [&](auto bufc, f32x16 (&s)[2]) {
    constexpr int BUF = decltype(bufc)::value;
    const char* kb_base = lds + BUF * 2 * KV_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[2 * (ks % KPF) + kb], qf[ks], ks == 0 ? negm : s[kb], 0, 0, 0);
        if (ks + KPF < 8) ka[2 * (ks % KPF) + kb] = *reinterpret_cast<const bf16x8*>(kb_base + kb * 8192 + k_rd[ks + KPF]);
      }
    }
  };

  // ---- stage tile t+1 into the buffer tile t does not use, issue the loads of tile t+2 ----
  auto stage = [&](auto bufc, int t) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    if (t + 1 < nt) {
      write_tile(IntC<BUF ^ 1>{});
      if (t + 2 < nt) load_tile(t + 2);
    }
  };

  // ---- online softmax: s holds (score - m); leaves P = exp2(s) as bf16 B-operand fragments ----
  auto softmax = [&](int t, f32x16 (&s)[2], bf16x8 (&pf)[4]) {
    if (ragged && t == nt - 1) {  // keys past skv (last tile only)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = t * KVB + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= p.skv) s[kb][i] = -1e30f;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
    mx = half_swap_max(mx);
    // Rare path: some row's maximum grew by more than the threshold (always taken on tile 0, where
    // it sets m to the tile's true row maximum).  Everything at the old scale -- O, l and this
    // tile's not yet exponentiated scores -- moves to the new scale exactly once.
    if (t == 0 || __any(mx > kRescaleThr)) {
      const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kb][i] -= delta;
      const float nm = negm[0] - delta;
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = nm;
    }
    float lsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float e = __builtin_amdgcn_exp2f(s[kb][8 * sidx + j]);
          lsum += e;
          f[j] = (__bf16)e;
        }
        pf[kb * 2 + sidx] = f;
      }
    l_run += lsum;
  };

  // ---- O^T += V^T P^T for k-steps [KS0, KS1) against the tile in LDS buffer BUF ----
  auto pv = [&](auto bufc, auto ks0c, auto ks1c, const bf16x8 (&pf)[4]) {
    constexpr int BUF = decltype(bufc)::value;
    const char* kb_base = lds + BUF * 2 * KV_TILE_BYTES;
#pragma unroll
    for (int ks4 = decltype(ks0c)::value; ks4 < decltype(ks1c)::value; ++ks4) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x4 lo = tr_read(kb_base + v_rd[ks4][0] + 512 * db);
        const bf16x4 hi = tr_read(kb_base + v_rd[ks4][1] + 512 * db);
        const bf16x8 a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[ks4], o[db], 0, 0, 0);
      }
    }
  };

  // VARIANT 0 tile body
  auto tile_plain = [&](auto bufc, int t, f32x16 (&s)[2]) {
    bf16x8 pf[4];
    qk(bufc, s);
    stage(bufc, t);
    softmax(t, s, pf);
    pv(bufc, IntC<0>{}, IntC<4>{}, pf);
    __syncthreads();
  };
  // VARIANT 1 tile body: the barrier sits before the last four PV MFMAs; right after it the first
  // K fragments of tile t+1 are requested.
  auto tile_pipe = [&](auto bufc, int t, f32x16 (&s)[2]) {
    constexpr int BUF = decltype(bufc)::value;
    const char* kb_base = lds + BUF * 2 * KV_TILE_BYTES;
    bf16x8 pf[4];
    qk_pipe(bufc, s);
    stage(bufc, t);
    softmax(t, s, pf);
    pv(bufc, IntC<0>{}, IntC<3>{}, pf);
    bf16x8 va[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const bf16x4 lo = tr_read(kb_base + v_rd[3][0] + 512 * db);
      const bf16x4 hi = tr_read(kb_base + v_rd[3][1] + 512 * db);
      va[db] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    __syncthreads();  // every LDS read of tile t is back, tile t+1 is fully written
    if (t + 1 < nt) k_prefetch(IntC<BUF ^ 1>{});
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[db], pf[3], o[db], 0, 0, 0);
  };

  // ---- prologue: K0/V0 -> LDS, K1/V1 in flight, Q fragments scaled and final before the loop ----
  load_tile(0);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    float f[8];
    unpack8(qraw[ks], f);
    if (ROPE) {
      const float* cs = p.rope + (int64_t)qrow_c * HD + 8 * ks + 4 * h;
      const f32x4 co = *reinterpret_cast<const f32x4*>(cs), si = *reinterpret_cast<const f32x4*>(cs + 64);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = f[2 * i], b = f[2 * i + 1];
        f[2 * i] = a * co[i] - b * si[i];
        f[2 * i + 1] = a * si[i] + b * co[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] *= p.c;
    u32x4 v = pack8(f);
    asm volatile("" : "+v"(v));  // the wait for the Q loads lands here, not inside the tile loop
    qf[ks] = __builtin_bit_cast(bf16x8, v);
  }
  write_tile(IntC<0>{});
  if (nt > 1) load_tile(1);
  __syncthreads();

  f32x16 s[2];
  {
    int t = 0;
    if (VARIANT == 1) {
      k_prefetch(IntC<0>{});
      for (; t + 1 < nt; t += 2) {
        tile_pipe(IntC<0>{}, t, s);
        tile_pipe(IntC<1>{}, t + 1, s);
      }
      if (t < nt) tile_pipe(IntC<0>{}, t, s);
    } else {
      for (; t + 1 < nt; t += 2) {
        tile_plain(IntC<0>{}, t, s);
        tile_plain(IntC<1>{}, t + 1, s);
      }
      if (t < nt) tile_plain(IntC<0>{}, t, s);
    }
  }

  // ---- normalise and store: o[db][i] = O[q0 + r32][32 db + 8 (i>>2) + 4 h + (i&3)] ----
  const float inv = 1.0f / half_swap_sum(l_run);
  const int qrow = q0 + r32;
  if (qrow < p.sq) {
    uint16_t* op = p.out + (int64_t)qrow * p.ldo + head * HD + 4 * h;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 pk = {pack2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv), pack2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + 32 * db + 8 * g) = pk;
      }
  }
}

}  // namespace

int lx2v_attn_launch_w4(const lx2v_attn::AttnParams& p, hipStream_t st);   // attn_fwd_w4.hip
int lx2v_attn_launch_x16(const lx2v_attn::AttnParams& p, hipStream_t st);  // attn_fwd_x16.hip

int64_t lx2v_attn_x16_workspace_bytes(int64_t sq, int64_t skv, int heads);  // attn_fwd_x16.hip

extern "C" int64_t lx2v_attn_bf16_workspace_bytes(int64_t sq, int64_t skv, int heads) {
  if (sq <= 0 || skv <= 0 || heads <= 0 || sq >= (1LL << 31) || skv >= (1LL << 31) || lx2v_options().attn_kernel != 16) return 0;
  return lx2v_attn_x16_workspace_bytes(sq, skv, heads);
}

extern "C" int lx2v_attn_fwd_bf16(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                                  const uint16_t* v, int64_t ldv, uint16_t* out, int64_t ldo,
                                  int64_t sq, int64_t skv, int heads, int head_dim, float scale,
                                  const float* rope_cos_sin, void* stream) {
  return lx2v_attn_fwd_bf16_ws(q, ldq, k, ldk, v, ldv, out, ldo, sq, skv, heads, head_dim, scale, rope_cos_sin, nullptr, 0, stream);
}

extern "C" int lx2v_attn_fwd_bf16_ws(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                                     const uint16_t* v, int64_t ldv, uint16_t* out, int64_t ldo,
                                     int64_t sq, int64_t skv, int heads, int head_dim, float scale,
                                     const float* rope_cos_sin, void* workspace, int64_t workspace_bytes, void* stream) {
  LX2V_REQUIRE(sq == 0 || (q && k && v && out), LX2V_EINVAL, "attn_fwd: null tensor");
  LX2V_REQUIRE(head_dim == HD, LX2V_EUNSUP, "attn_fwd: head_dim=%d, only 128 is built", head_dim);
  LX2V_REQUIRE(heads > 0 && sq >= 0 && skv >= 0, LX2V_EINVAL, "attn_fwd: bad shape sq=%lld skv=%lld heads=%d", (long long)sq, (long long)skv, heads);
  LX2V_REQUIRE(sq < (1LL << 31) && skv < (1LL << 31), LX2V_EUNSUP, "attn_fwd: sequence lengths must fit in 31 bits");
  LX2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, LX2V_EINVAL, "attn_fwd: token strides must be multiples of 8 elements");
  LX2V_REQUIRE(ldq >= (int64_t)heads * HD && ldk >= (int64_t)heads * HD && ldv >= (int64_t)heads * HD && ldo >= (int64_t)heads * HD, LX2V_EINVAL,
               "attn_fwd: token stride smaller than heads*head_dim");
  LX2V_REQUIRE(lx2v_aligned16(q) && lx2v_aligned16(k) && lx2v_aligned16(v) && lx2v_aligned16(out), LX2V_EINVAL, "attn_fwd: tensors must be 16-byte aligned");
  if (rope_cos_sin != nullptr) {
    LX2V_REQUIRE(sq == skv, LX2V_EINVAL, "attn_fwd: fused RoPE needs sq == skv (self-attention)");
    LX2V_REQUIRE(lx2v_aligned16(rope_cos_sin), LX2V_EINVAL, "attn_fwd: rope table must be 16-byte aligned");
  }
  if (sq == 0) return LX2V_OK;
  LX2V_REQUIRE(skv > 0, LX2V_EINVAL, "attn_fwd: skv must be > 0 when sq > 0");
  AttnParams p;
  p.q = q; p.k = k; p.v = v; p.out = out; p.rope = rope_cos_sin;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.sq = (int)sq; p.skv = (int)skv; p.heads = heads;
  p.nqb = (int)((sq + QROWS - 1) / QROWS);
  p.c = scale * 1.4426950408889634f;
  p.workspace = workspace;
  p.workspace_bytes = workspace == nullptr ? 0 : workspace_bytes;
  const int64_t nwg = (int64_t)p.nqb * heads;
  LX2V_REQUIRE(nwg < (1LL << 31), LX2V_EUNSUP, "attn_fwd: grid too large");
  const dim3 grid((unsigned)nwg), block(NTHREADS);
  LX2V_REQUIRE((skv + 8 * KVB) * ldk * 2 < (1LL << 32) && (skv + 8 * KVB) * ldv * 2 < (1LL << 32), LX2V_EUNSUP,
               "attn_fwd: K/V extent (skv * token stride) must stay below 4 GiB for 32-bit buffer offsets");
  hipStream_t st = (hipStream_t)stream;
  // Kernel choice: a one-wave-per-SIMD kernel unless RoPE is fused or LX2V_ATTN_KERNEL=w8 asks
  // for the 8-wave one.  Of the two, the 16x16x32 kernel is the default (it needs more cycles than
  // the 32x32x16 one but the chip holds a higher clock under it); LX2V_ATTN_KERNEL=w4 selects the
  // 32x32x16 kernel.
  const int kernel = lx2v_options().attn_kernel;
  if (rope_cos_sin == nullptr && kernel != 8) {
    const int rc = kernel == 16 ? lx2v_attn_launch_x16(p, st) : lx2v_attn_launch_w4(p, st);
    if (rc != LX2V_OK) return rc;
    LX2V_CHECK_LAUNCH("attn_fwd(one wave per SIMD)");
    return LX2V_OK;
  }
  // Tuning switch attn_variant (8-wave kernel): 0 = plain order, 1 (default) = pipelined K-fragment reads.
  const int variant = lx2v_options().attn_variant;
#define LX2V_ATTN_LAUNCH(R, SK)                                                                  \
  do {                                                                                           \
    if (variant == 0) hipLaunchKernelGGL((attn_fwd_kernel<R, SK, 0>), grid, block, 0, st, p);       \
    else hipLaunchKernelGGL((attn_fwd_kernel<R, SK, 1>), grid, block, 0, st, p);                    \
  } while (0)
  if (rope_cos_sin != nullptr) hipLaunchKernelGGL((attn_fwd_kernel<true, false, 0>), grid, block, 0, st, p);  // variant 1 would spill with the rotation registers
  else if (skv <= 2048) LX2V_ATTN_LAUNCH(false, true);
  else LX2V_ATTN_LAUNCH(false, false);
#undef LX2V_ATTN_LAUNCH
  LX2V_CHECK_LAUNCH("attn_fwd");
  return LX2V_OK;
}
