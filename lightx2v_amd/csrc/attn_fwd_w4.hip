// Attention forward, one wave per SIMD: 4 waves x 64 query rows, half-tile software pipeline.
//
// Workgroup = 4 waves = 256 query rows of one head.  Each wave owns two 32-row query blocks and,
// alone on its SIMD, the whole 512-entry register file: O (128 registers), the prescaled Q^T
// fragments (64) live in the accumulator half; the scores, the running-maximum
// accumulator init (an MFMA's C and D share one file), probabilities and staging registers in the
// architectural half.  hipcc cannot be told
// which half an MFMA operand goes to, so every MFMA here is an asm statement with the register
// class of each operand spelled out; everything else is ordinary HIP.
//
// The unit of work is a HALF tile of 32 keys.  Half tile k goes through three stages,
//     QK(k): 8 MFMA steps  S(k) = K(k) Q^T - m      (both query blocks per step)
//     SM(k): exp2 / row sum / row max / bf16 convert of S(k)   (VALU only)
//     PV(k): 8 MFMA steps  O += V(k)^T P(k)
// and the loop runs them skewed: period k issues QK(k+1) and PV(k-1) interleaved step by step
// (one K fragment read or two V^T fragment reads per step, each feeding two MFMAs) while the VALU
// works through SM(k) in the gaps, one chunk of two scores per step.  With no second wave on the
// SIMD to fill bubbles this balance is what keeps the MFMA pipe busy: every period has the same
// 32 MFMAs and the same VALU load.
//
// Scores are exponentiated against the running maximum as it stood when their accumulation
// started (it enters through the C operand of the first MFMA), so no exp waits for a maximum.
// The maximum of S(k) is tracked on the side and checked once at the end of period k, before any
// P(k) is used; the rare fix-up re-bases O, l, m, S(k) (redoing P(k)) and the already complete
// S(k+1).  Half tile 0 always takes the fix-up so that m starts at a true row maximum.
//
// LDS: a ring of four 16 KiB tiles for K and one for V (same images as the 8-wave kernel), filled
// by LDS-DMA (buffer_load ... lds: no staging registers, no LDS write instructions; the bank
// swizzle of both images is applied to each lane's SOURCE address because the DMA writes
// lane-linear).  K(t) is read in periods 2t-1 and 2t, V(t) in 2t+1 and 2t+2; one barrier per tile
// (at step 12 of every even period 2t) retires K(t) and V(t-1), after which period 2t+1 issues the
// DMA of K(t+4) and V(t+3) into their slots.  Three such groups stay in flight across each
// barrier (counted vmcnt), i.e. every tile is requested three tiles before it is needed.
// Rows past the last key come back as zeros from the buffer bounds check.
#include "attn_common.h"

using namespace lx2v_attn;

namespace {

#ifndef W4_ABL
#define W4_ABL 0  // timing ablations (results are wrong with any bit set): 1 softmax VALU, 2 fragment reads, 4 staging, 8 barrier, 16 maximum check
#endif
constexpr int W4_THREADS = 256;
constexpr int RING = 4;                        // tiles per LDS ring
constexpr int K_RING = 0;                      // LDS byte offset of K slot 0
constexpr int V_RING = RING * KV_TILE_BYTES;   // same for V
constexpr int DMA_PER_GROUP = 8;               // DMA instructions a wave issues per tile (4 K + 4 V)


// D (VGPR) = A (VGPR: K rows) x B (AGPR: Q^T) + C (VGPR: -m broadcast; C and D share one file)
LX2V_DEVICE void mfma_qk_first(f32x16& s, const u32x4& kf, const u32x4& qf, const f32x16& negm) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s) : "v"(kf), "a"(qf), "v"(negm));
}
// D (VGPR) += A (VGPR) x B (AGPR)
LX2V_DEVICE void mfma_qk(f32x16& s, const u32x4& kf, const u32x4& qf) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(kf), "a"(qf));
}
// D (AGPR) += A (VGPR: V^T) x B (VGPR: P)
LX2V_DEVICE void mfma_pv(f32x16& o, const u32x4& vf, const u32x4& pf) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(vf), "v"(pf));
}

__global__ __launch_bounds__(W4_THREADS) void attn_fwd_w4_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * RING * KV_TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;

  const int qb = blockIdx.x % p.nqb;
  const int head = blockIdx.x / p.nqb;
  const int q0 = qb * QROWS + wid * 64;  // block b covers rows q0 + 32 b .. +31

  // ---- Q^T fragments, raw: lane holds Q[row][16 ks + 8 h .. +7] ----
  u32x4 qraw[2][8];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int qrow_c = min(q0 + 32 * b + r32, p.sq - 1);
    const uint16_t* qp = p.q + (int64_t)qrow_c * p.ldq + head * HD + h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qraw[b][ks] = *reinterpret_cast<const u32x4*>(qp + ks * 16);
  }

  // ---- K/V staging by LDS-DMA: one wave instruction fills 1 KiB of a tile image, lane L the 16
  // bytes at 16 L.  Wave w owns pieces j = w + 4 i (i < 4) of each tile.
  //   K image: piece j = rows 4j..4j+3, lane = (row & 3, chunk position); position c holds chunk c ^ (row & 15)
  //   V image: piece j = rows 8 (j>>1) .. +7, d-blocks 2 (j&1), 2 (j&1) + 1; lane = (d-block & 1, row & 7, position)
  // In both the i-dependence is "16 rows further", which goes into the scalar offset.
  // The DMA is an asm statement: hipcc would drain vmcnt to 0 before every later transposed LDS read
  // if it knew an LDS write was in flight; completion is counted by hand at the tile barrier.
  const uint32_t wid_u = __builtin_amdgcn_readfirstlane(wid);
  auto descriptor = [&](const uint16_t* base, int64_t bytes) __attribute__((always_inline)) -> u32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));  // stride 0: raw buffer
    d[2] = __builtin_amdgcn_readfirstlane((uint32_t)bytes);      // loads at or past it return 0
    d[3] = 0x00020000u;
    return d;
  };
  const u32x4 k_rs = descriptor(p.k + head * HD, (int64_t)p.skv * p.ldk * 2);
  const u32x4 v_rs = descriptor(p.v + head * HD, (int64_t)p.skv * p.ldv * 2);
  const int krow = 4 * wid + (lane >> 4);
  const int k_vo = (int)(krow * p.ldk * 2) + (((lane & 15) ^ (krow & 15)) << 4);
  const int vrow = 8 * (wid >> 1) + ((lane >> 2) & 7);
  const int v_vo = (int)(vrow * p.ldv * 2) + ((4 * (2 * (wid & 1) + (lane >> 5)) + ((lane & 3) ^ ((vrow >> 2) & 3))) << 4);
  const uint32_t k_tile_bytes = (uint32_t)(KVB * p.ldk * 2), v_tile_bytes = (uint32_t)(KVB * p.ldv * 2);
  const uint32_t k_q_bytes = k_tile_bytes / 4, v_q_bytes = v_tile_bytes / 4;  // 16 rows
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds);
  // lane L of the wave: 16 bytes from base + voff + soff to LDS byte dst + 16 L
  auto dma16 = [&](const u32x4& rs, int voff, uint32_t soff, uint32_t dst) __attribute__((always_inline)) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst), "s"(soff) : "memory");
  };
  auto dma_k = [&](int t, auto slotc, auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    dma16(k_rs, k_vo, (uint32_t)t * k_tile_bytes + i * k_q_bytes, lds_base + K_RING + decltype(slotc)::value * KV_TILE_BYTES + 4096 * i + 1024 * wid_u);
  };
  auto dma_v = [&](int t, auto slotc, auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    dma16(v_rs, v_vo, (uint32_t)t * v_tile_bytes + i * v_q_bytes, lds_base + V_RING + decltype(slotc)::value * KV_TILE_BYTES + 4096 * i + 1024 * wid_u);
  };

  // ---- fragment read offsets inside one tile image (derivation: the 8-wave kernel) ----
  int k_rd[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_rd[ks] = r32 * 256 + (((2 * ks + h) ^ (r32 & 15)) << 4);
  const int g2 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  int v_rd[4][2];
#pragma unroll
  for (int ks4 = 0; ks4 < 4; ++ks4)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 16 * ks4 + 8 * u + 4 * h + q4;
      const int chl = 2 * g2 + (p4 >> 1);
      v_rd[ks4][u] = V_RING + 2048 * (row >> 3) + 64 * (row & 7) + 16 * (chl ^ ((row >> 2) & 3)) + 8 * (p4 & 1);
      asm volatile("" : "+v"(v_rd[ks4][u]));  // ring base stays in the register: ds offsets are 16-bit
    }
  // The fragment step s of a period needs.  Period parity PAR and the ring slot BUF = t % 4 of the
  // period's tile t fix every buffer: even steps read the K rows of half tile k+1 (k = 2t+PAR),
  // odd steps the V^T fragments of half tile k-1; m = s/2 is the k-step (K) or
  // (k-step within the half, d-block) pair (V).
  auto frag_read = [&](auto parc, auto bufc, auto sc) __attribute__((always_inline)) -> u32x4 {
    constexpr int PAR = decltype(parc)::value, BUF = decltype(bufc)::value, s = decltype(sc)::value, m = s >> 1;
    if constexpr ((s & 1) == 0) {
      constexpr int off = K_RING + (PAR == 0 ? BUF : (BUF + 1) % RING) * KV_TILE_BYTES + (PAR == 0 ? 8192 : 0);
      return *reinterpret_cast<const u32x4*>(lds + off + k_rd[m]);
    } else {
      constexpr int off = (PAR == 0 ? (BUF + RING - 1) % RING : BUF) * KV_TILE_BYTES + 512 * (m & 3);  // + V_RING inside v_rd
      constexpr int ks4 = (PAR == 0 ? 2 : 0) + (m >> 2);
      const bf16x4 lo = tr_read(lds + off + v_rd[ks4][0]);
      const bf16x4 hi = tr_read(lds + off + v_rd[ks4][1]);
      return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };

  f32x16 o[2][4];   // accumulator file
  f32x16 negm[2];   // -m of the lane's row in all 16 registers
  u32x4 qf[2][8];   // c * Q^T fragments (accumulator file)
  float l_run[2];
  f32x16 s0[2], s1[2];          // scores of the even / odd half tiles
  u32x4 pfa[2][2], pfb[2][2];   // [block][k-step]: P of the even / odd half tiles
  u32x4 ring[4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    l_run[b] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[b][i] = 0.f;
    asm volatile("" : "+v"(negm[b]));
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int i = 0; i < 16; ++i) o[b][db][i] = 0.f;
      asm volatile("" : "+a"(o[b][db]));
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) pfa[b][f] = pfb[b][f] = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) s1[b][i] = -1e30f;  // "half tile -1": every key masked
  }

  const int nt = (p.skv + KVB - 1) / KVB;

  auto mask_half = [&](f32x16& sv, int key0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (key0 + (i & 3) + 8 * (i >> 2) + 4 * h >= p.skv) sv[i] = -1e30f;
  };

  // Rare path, entered before any P of half tile k is used: move O, l, m, S(k) and the complete
  // S(k+1) of block b to the new maximum and redo P(k).  MFMAs on these registers may have been
  // issued a few cycles ago and others follow right after, hence the explicit wait states.
  auto fixup = [&](auto bc, f32x16& sc, f32x16& sn, u32x4 (&pc)[2], float delta, float& lsa, float& lsb) __attribute__((always_inline)) {
    constexpr int b = decltype(bc)::value;
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(o[b][0]), "+a"(o[b][1]), "+a"(o[b][2]), "+a"(o[b][3]), "+v"(sc), "+v"(sn));
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    l_run[b] *= alpha;
    static_for<0, 4>([&](auto dbc) __attribute__((always_inline)) {  // one tuple at a time through the architectural file
      constexpr int db = decltype(dbc)::value;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[b][db][i] *= alpha;
      asm volatile("" : "+a"(o[b][db]));
    });
    const float nm = negm[b][0] - delta;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[b][i] = nm;
    float ls = 0.f;
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sc[i] -= delta;
      sn[i] -= delta;
      e[i] = __builtin_amdgcn_exp2f(sc[i]);
      ls += e[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) pc[i >> 2][i & 3] = pack2(e[2 * i], e[2 * i + 1]);
    lsa = ls;
    lsb = 0.f;
    asm volatile("s_nop 1" : "+a"(o[b][0]), "+a"(o[b][1]), "+a"(o[b][2]), "+a"(o[b][3]), "+v"(negm[b]), "+v"(pc[0]), "+v"(pc[1]), "+v"(sn));
  };

  // One period (k = 2t + PAR) of the pipeline.  Periods outside the real half tiles (k < 0 before
  // the first tile, k >= 2 nt at the end) run the same code on zero tiles with every key masked.
  auto period = [&](auto parc, auto bufc, int t) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value, BUF = decltype(bufc)::value;
    f32x16 (&Sc)[2] = PAR == 0 ? s0 : s1;        // S(k): exponentiated in this period
    f32x16 (&Sn)[2] = PAR == 0 ? s1 : s0;        // S(k+1): accumulated in this period
    u32x4 (&Pc)[2][2] = PAR == 0 ? pfa : pfb;    // P(k): produced
    u32x4 (&Pp)[2][2] = PAR == 0 ? pfb : pfa;    // P(k-1): consumed
    float mx[2] = {-3e38f, -3e38f}, lsa[2] = {0.f, 0.f}, lsb[2] = {0.f, 0.f};
    float ea = 0.f;
    const int key0 = t * KVB + 32 * PAR;
    if (key0 + 32 > p.skv) {  // wave-uniform: only the last real half tile and the zero tiles
      mask_half(Sc[0], key0);
      mask_half(Sc[1], key0);
    }
    static_for<0, 16>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, m = s >> 1;
      constexpr int bsm = s & 1, c = s >> 1;  // softmax chunk of this step: scores 2c, 2c+1 of block bsm
      const u32x4 f = ring[s & 3];
      // ---- gap 1: MFMA of block 0, first half of the chunk ----
      if constexpr ((s & 1) == 0) {
        if constexpr (m == 0) mfma_qk_first(Sn[0], f, qf[0][0], negm[0]);
        else mfma_qk(Sn[0], f, qf[0][m]);
      } else {
        mfma_pv(o[0][m & 3], f, Pp[0][m >> 2]);
      }
      if constexpr (!(W4_ABL & 1)) {
        const float a = Sc[bsm][2 * c], b2 = Sc[bsm][2 * c + 1];
        mx[bsm] = fmaxf(mx[bsm], fmaxf(a, b2));
        ea = __builtin_amdgcn_exp2f(a);
        lsa[bsm] += ea;
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- gap 2: MFMA of block 1, fragment refill, second half of the chunk, staging ----
      if constexpr ((s & 1) == 0) {
        if constexpr (m == 0) mfma_qk_first(Sn[1], f, qf[1][0], negm[1]);
        else mfma_qk(Sn[1], f, qf[1][m]);
      } else {
        mfma_pv(o[1][m & 3], f, Pp[1][m >> 2]);
      }
      if constexpr (W4_ABL & 2) asm volatile("" : "+v"(ring[s & 3]));
      else if constexpr (s + 4 < 16) ring[s & 3] = frag_read(parc, bufc, IntC<s + 4>{});
      else if constexpr (PAR == 0) ring[s & 3] = frag_read(IntC<1>{}, bufc, IntC<s - 12>{});
      else ring[s & 3] = frag_read(IntC<0>{}, IntC<(BUF + 1) % RING>{}, IntC<s - 12>{});
      if constexpr (!(W4_ABL & 1)) {
        const float eb = __builtin_amdgcn_exp2f(Sc[bsm][2 * c + 1]);
        lsb[bsm] += eb;
        Pc[bsm][c >> 2][c & 3] = pack2(ea, eb);
      }
      if constexpr (PAR == 1 && s < 8 && !(W4_ABL & 4)) {  // the DMA group of this tile: K(t+4), V(t+3)
        if constexpr (s < 4) dma_k(t + 4, bufc, IntC<s>{});
        else dma_v(t + 3, IntC<(BUF + RING - 1) % RING>{}, IntC<s - 4>{});
      }
      if constexpr (PAR == 0 && s == 11 && !(W4_ABL & 8)) {
        // K(t+1) and V(t) (the group issued three tiles ago) have landed, this wave's reads of
        // K(t) and V(t-1) are back; after the barrier that holds for every wave.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0070 | ((2 * DMA_PER_GROUP) & 15) | (((2 * DMA_PER_GROUP) >> 4) << 14));  // vmcnt(16) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- maximum check of half tile k (the exps above must not sink below the branch).  Each
    // lane holds the maximum over its 16 of the row's 32 keys; "any lane above the threshold" needs
    // no exchange, the row maximum itself is formed inside the rare path.
    asm volatile("" : "+v"(Pc[0][0]), "+v"(Pc[0][1]), "+v"(lsa[0]), "+v"(lsb[0]));
    asm volatile("" : "+v"(Pc[1][0]), "+v"(Pc[1][1]), "+v"(lsa[1]), "+v"(lsb[1]));
    const bool first = PAR == 0 && t == 0;  // half tile 0: m becomes its true row maximum
    if (!(W4_ABL & 16) && __builtin_expect(first || __any(fmaxf(mx[0], mx[1]) > kRescaleThr), 0)) {
      static_for<0, 2>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        const float mxc = half_swap_max(mx[b]);
        fixup(bc, Sc[b], Sn[b], Pc[b], first ? mxc : fmaxf(mxc, 0.f), lsa[b], lsb[b]);
      });
    }
    l_run[0] += lsa[0] + lsb[0];
    l_run[1] += lsa[1] + lsb[1];
  };

  // ---- prologue: Q fragments scaled and parked; DMA groups of tiles -4, -3, -2 (K0; K1, V0; K2,
  // V1); V slot 3 zeroed by hand (it stands for V(-1): the two periods before half tile 0 multiply
  // it by P = 0) ----
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float f[8];
      unpack8(qraw[b][ks], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] *= p.c;
      qf[b][ks] = pack8(f);
      asm volatile("" : "+a"(qf[b][ks]));
    }
  static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4*>(lds + V_RING + 3 * KV_TILE_BYTES + 4096 * decltype(ic)::value + tid * 16) = u32x4{0, 0, 0, 0};
    dma_k(0, IntC<0>{}, ic);
  });
  static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { dma_k(1, IntC<1>{}, ic); });
  static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { dma_v(0, IntC<0>{}, ic); });
  static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { dma_k(2, IntC<2>{}, ic); });
  static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { dma_v(1, IntC<1>{}, ic); });
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0070 | ((2 * DMA_PER_GROUP) & 15) | (((2 * DMA_PER_GROUP) >> 4) << 14));  // K0 landed (16 younger DMAs may fly), zeros written
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // fragments of steps 0..3 of the first period run, (t = -1, odd), whose tile has ring slot 3
  static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { ring[decltype(ic)::value] = frag_read(IntC<1>{}, IntC<3>{}, ic); });

  // Periods run in the fixed rotation (t-1, odd) (t, even) (t, odd) (t+1, even) ... over four
  // tiles per trip, so that every LDS offset is static; the last period needed is (nt, even), which
  // holds PV of the last half tile.
  for (int t = 0;; t += 4) {
    period(IntC<1>{}, IntC<3>{}, t - 1);
    period(IntC<0>{}, IntC<0>{}, t);
    if (t == nt) break;
    period(IntC<1>{}, IntC<0>{}, t);
    period(IntC<0>{}, IntC<1>{}, t + 1);
    if (t + 1 == nt) break;
    period(IntC<1>{}, IntC<1>{}, t + 1);
    period(IntC<0>{}, IntC<2>{}, t + 2);
    if (t + 2 == nt) break;
    period(IntC<1>{}, IntC<2>{}, t + 2);
    period(IntC<0>{}, IntC<3>{}, t + 3);
    if (t + 3 == nt) break;
  }
  // no DMA may still be writing this workgroup's LDS when it is handed to the next one
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

  // ---- normalise and store: o[b][db][i] = O[q0 + 32 b + r32][32 db + 8 (i>>2) + 4 h + (i&3)] ----
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(o[b][0]), "+a"(o[b][1]), "+a"(o[b][2]), "+a"(o[b][3]));
    const float inv = 1.0f / half_swap_sum(l_run[b]);
    const int qrow = q0 + 32 * b + r32;
    if (qrow < p.sq) {
      uint16_t* op = p.out + (int64_t)qrow * p.ldo + head * HD + 4 * h;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 pk = {pack2(o[b][db][4 * g] * inv, o[b][db][4 * g + 1] * inv), pack2(o[b][db][4 * g + 2] * inv, o[b][db][4 * g + 3] * inv)};
          *reinterpret_cast<u32x2*>(op + 32 * db + 8 * g) = pk;
        }
    }
  }
}

}  // namespace

// Launch for p already validated by lx2v_attn_fwd_bf16 (no RoPE; K/V extent + 8 tiles < 4 GiB).
int lx2v_attn_launch_w4(const lx2v_attn::AttnParams& p, hipStream_t st) {
  const dim3 grid((unsigned)((int64_t)p.nqb * p.heads));
  hipLaunchKernelGGL(attn_fwd_w4_kernel, grid, dim3(W4_THREADS), 0, st, p);
  return LX2V_OK;
}
