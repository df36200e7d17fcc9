// Attention forward, one wave per SIMD, on v_mfma_f32_16x16x32_bf16.
//
// Same structure as attn_fwd_w4.hip (4 waves x 64 or 80 query rows, half-tile software pipeline, K and
// V rings filled by LDS-DMA, one barrier per tile, running maximum as accumulator init, overflow
// check instead of a tracked maximum), re-tiled for the 16x16x32 instruction: back to back it
// moves half the accumulator traffic of 32x32x16 per cycle, and under the chip's power limit the
// same softmax filler stream beside it sustains about 13 % more FLOP/s.
//
// Per wave: NQB = 4 or 5 (1 in the tail launch) blocks of 16 query rows (qb).  A half tile of 32 keys is two key blocks (kb).
//   QK(k): S[kb][qb] = K(kb) Q(qb)^T - m   2 kb x 4 k-steps of 32 channels x NQB = 8 NQB MFMAs
//   SM(k): exp2 / row sum / bf16 convert of the 8 NQB scores a lane holds
//   PV(k): O[qb][db] += V(db)^T P(qb)      8 blocks of 16 channels (db) x NQB = 8 NQB MFMAs, the
//          whole half tile being one contraction step
// A step of the pipeline is one LDS fragment (a K fragment on even steps, a V^T fragment on odd
// ones) and the NQB MFMAs (one per qb) that use it; a period has 16 steps = 16 NQB MFMAs.  An
// MFMA of this shape holds vector issue for 8 of its 16 cycles, so each gap takes 8 cycles of
// other work; the softmax of 4 NQB score pairs fills the 16 NQB gaps exactly, four gaps a pair:
// exp | exp | add + convert | add.  With NQB = 5 (448 registers) a workgroup covers 320 query
// rows, so K/V are streamed from L2 a fifth less often and every fragment read, DMA piece and
// wait is shared by five MFMAs instead of four.
//
// Operand maps (lane: c = lane & 15, g = lane >> 4).  A key block is NOT 16 consecutive keys: row
// c of block kb is key 16 (c >> 3) + 8 ((c >> 2) & 1) + 4 kb + (c & 3) of the half tile, so that the
// accumulators of lane group g are keys 8 g + 4 kb + i (i < 4), and the lane's eight probabilities
// (kb = 0, 1) are keys 8 g .. 8 g + 7 in order: converted pairwise they ARE the B operand of PV
// with the contraction in natural key order, and the V^T fragment is two transposed reads of
// four consecutive keys whose two blocks per 32-lane half lie 8 rows apart (conflict-free on the
// 8-row x 32-column sub-tiled V image).  The K image is swizzled by the MFMA row index c instead
// of the key index, which keeps its ds_read_b128 conflict-free under that row permutation.
#include <algorithm>

#include "attn_w4_common.h"

using namespace lx2v_attn;

namespace {

#ifndef X16_ABL
#define X16_ABL 0  // timing ablations (results are wrong with any bit set): 1 softmax VALU, 2 fragment reads, 4 staging, 8 barrier, 16 overflow check, 32 K fragment reads, 64 V fragment reads, 128 exps (moves instead), 256 softmax VALU other than exps, 512 K staging, 1024 V staging
#endif
#ifndef X16_LAG
#define X16_LAG 0  // test builds only: s_sleep argument for one wave per tile, directly behind the tile barrier
#endif
constexpr int X16_THREADS = 256;
constexpr int RING = 4;                        // tiles per LDS ring
constexpr int K_RING = 0;                      // LDS byte offset of K slot 0
constexpr int V_RING = RING * KV_TILE_BYTES;   // same for V
constexpr float kRescaleSum = 64.0f;           // 2^kRescaleThr
constexpr int DMA_PER_GROUP = 8;               // DMA instructions a wave issues per tile (4 K + 4 V)

// D (VGPR) = A (VGPR: K rows) x B (AGPR: Q^T) + C (VGPR: -m broadcast)
LX2V_DEVICE void mfma_qk_first(f32x4& s, const u32x4& kf, const u32x4& qf, const f32x4& negm) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(s) : "v"(kf), "a"(qf), "v"(negm));
}
// D (VGPR) += A (VGPR) x B (AGPR)
LX2V_DEVICE void mfma_qk(f32x4& s, const u32x4& kf, const u32x4& qf) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(kf), "a"(qf));
}
// D (AGPR) += A (VGPR: V^T) x B (VGPR: P)
LX2V_DEVICE void mfma_pv(f32x4& o, const u32x4& vf, const u32x4& pf) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(vf), "v"(pf));
}

// K image swizzle key of tile row `row`: the MFMA row index that reads it.
LX2V_DEVICE int k_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2) | (((row >> 4) & 1) << 3); }

// SHORT_KV only names a second symbol for short key sequences (cross-attention), so that kernel
// profiles keep self- and cross-attention apart.
template <bool SHORT_KV, int NQB>
__global__ __launch_bounds__(X16_THREADS) void attn_fwd_x16_kernel(const AttnParams p, const RoundSync rs) {
  __shared__ __attribute__((aligned(16))) char lds[2 * RING * KV_TILE_BYTES];
  round_enter(rs);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;

  // (key split: part-major copies of the (head, query block) grid, each over its own key range)
  int bid = blockIdx.x, part = 0, key_base = 0, skv = p.skv;
  if (p.nparts > 1) {
    const int per = p.nqb * p.heads;
    part = bid / per;
    bid -= part * per;
    key_base = part * p.part_keys;
    skv = min(p.part_keys, p.skv - key_base);
  }
  const int qb_wg = bid % p.nqb;
  const int head = bid / p.nqb;
  const int q0 = qb_wg * (64 * NQB) + wid * (16 * NQB);  // block qb covers rows q0 + 16 qb .. +15

  // ---- Q^T fragments, raw: lane holds Q[q0 + 16 qb + c][32 ks + 8 g .. +7] ----
  u32x4 qraw[NQB][4];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int qrow_c = min(q0 + 16 * qb + c16, p.sq - 1);
    const uint16_t* qp = p.q + (int64_t)qrow_c * p.ldq + head * HD + g * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qraw[qb][ks] = *reinterpret_cast<const u32x4*>(qp + ks * 32);
  }

  // ---- K/V staging by LDS-DMA: one wave instruction fills 1 KiB of a tile image, lane L the 16
  // bytes at 16 L.  Wave w owns pieces j = w + 4 i (i < 4) of each tile.
  //   K image: piece j = rows 4j..4j+3, lane = (row & 3, chunk position); position c holds chunk c ^ k_swz(row)
  //   V image: piece j = rows 8 (j>>1) .. +7, d-blocks 2 (j&1), 2 (j&1) + 1; lane = (d-block & 1, row & 7, position)
  // In both the i-dependence is "16 rows further", which goes into the scalar offset (and, for K,
  // into bit 3 of the swizzle key).
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
  const u32x4 k_rs = descriptor(p.k + (int64_t)key_base * p.ldk + head * HD, (int64_t)skv * p.ldk * 2);
  const u32x4 v_rs = descriptor(p.v + (int64_t)key_base * p.ldv + head * HD, (int64_t)skv * p.ldv * 2);
  const int krow = 4 * wid + (lane >> 4);
  const int vrow = 8 * (wid >> 1) + ((lane >> 2) & 7);
  int k_vo[4], v_vo[4];  // piece i: 16 rows further
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    k_vo[i] = (int)((krow + 16 * i) * p.ldk * 2) + (((lane & 15) ^ k_swz(krow + 16 * i)) << 4);
    v_vo[i] = (int)((vrow + 16 * i) * p.ldv * 2) + ((4 * (2 * (wid & 1) + (lane >> 5)) + ((lane & 3) ^ ((vrow >> 2) & 3))) << 4);
  }
  const uint32_t k_tile_bytes = (uint32_t)(KVB * p.ldk * 2), v_tile_bytes = (uint32_t)(KVB * p.ldv * 2);
  const uint32_t dma_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds) + 1024 * wid_u;
  // lane L of the wave: 16 bytes from base + voff + soff to LDS byte dma_base + DST + 16 L.  M0 is
  // set by one add and not restored (nothing else in this kernel uses it).  The DMA is an asm
  // statement: hipcc would drain vmcnt to 0 before every later LDS read if it knew an LDS write
  // was in flight; completion is counted by hand at the tile barrier.
  // Two statements so that the loop can put them into different MFMA gaps: M0 needs one wait
  // state between its write and the DMA that reads it, which the instructions in between supply.
  auto dma_m0 = [&](auto dstc) __attribute__((always_inline)) {
    const uint32_t base = dma_base;
    asm volatile("s_add_u32 m0, %0, %1" :: "s"(base), "i"(decltype(dstc)::value) : "memory", "scc");
  };
  auto dma_go = [&](const u32x4& rsrc, int voff, uint32_t soff) __attribute__((always_inline)) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };
  auto dma16 = [&](const u32x4& rsrc, int voff, uint32_t soff, auto dstc) __attribute__((always_inline)) {
    dma_m0(dstc);
    asm volatile("s_nop 0");
    dma_go(rsrc, voff, soff);
  };
  auto dma_k = [&](int t, auto slotc, auto ic, uint32_t soff = 0) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    dma16(k_rs, k_vo[i], (uint32_t)t * k_tile_bytes + soff, IntC<K_RING + decltype(slotc)::value * KV_TILE_BYTES + 4096 * i>{});
  };
  auto dma_v = [&](int t, auto slotc, auto ic, uint32_t soff = 0) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    dma16(v_rs, v_vo[i], (uint32_t)t * v_tile_bytes + soff, IntC<V_RING + decltype(slotc)::value * KV_TILE_BYTES + 4096 * i>{});
  };

  // ---- fragment read offsets inside one tile image ----
  // K (A operand of S^T), block kb of half tile hf: lane reads row 32 hf + 4 kb + krd_row, chunk
  // 4 ks + g, stored at position (4 ks + g) ^ c.  hf and kb are immediates (8192 hf + 1024 kb).
  const int krd_row = (c16 & 3) + 8 * ((c16 >> 2) & 1) + 16 * (c16 >> 3);
  int k_rd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_rd[ks] = krd_row * 256 + (((4 * ks + g) ^ c16) << 4);
  // V^T (A operand of O^T), channel block db of half tile hf: per 16-lane group, lane 4 q4 + p4
  // supplies row 32 hf + 8 g + 4 u + q4 (u = 0, 1: elements 0..3 / 4..7), channels 16 db + 4 p4 ..
  // In the sub-tiled image that is 8192 hf + 2048 g + 512 (db >> 1) + 256 u + 64 q4 + 8 (p4 & 1)
  //   + 16 (((p4 >> 1) ^ u) | (((db ^ g) & 1) << 1)): four lane bases [u][db & 1], the rest immediates.
  const int q4 = c16 >> 2, p4 = c16 & 3;
  int v_rd[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int dp = 0; dp < 2; ++dp) {
      v_rd[u][dp] = V_RING + 2048 * g + 256 * u + 64 * q4 + 8 * (p4 & 1) + 16 * (((p4 >> 1) ^ u) | (((dp ^ g) & 1) << 1));
      asm volatile("" : "+v"(v_rd[u][dp]));  // ring base stays in the register: ds offsets are 16-bit
    }
  // Fragment of step s of a period.  Period parity PAR and the ring slot BUF = t % 4 of the
  // period's tile t fix every buffer: even steps read K of half tile k+1 (k = 2t+PAR), m = s/2 =
  // 4 kb + ks; odd steps the V^T fragment db = s/2 of half tile k-1.
  auto frag_read_k = [&](auto parc, auto bufc, auto sc) __attribute__((always_inline)) -> u32x4 {
    constexpr int PAR = decltype(parc)::value, BUF = decltype(bufc)::value, m = decltype(sc)::value >> 1;
    constexpr int off = K_RING + (PAR == 0 ? BUF : (BUF + 1) % RING) * KV_TILE_BYTES + (PAR == 0 ? 8192 : 0) + 1024 * (m >> 2);
    return *reinterpret_cast<const u32x4*>(lds + off + k_rd[m & 3]);
  };
  auto frag_read_v = [&](auto parc, auto bufc, auto sc, auto uc) __attribute__((always_inline)) -> bf16x4 {
    constexpr int PAR = decltype(parc)::value, BUF = decltype(bufc)::value, db = decltype(sc)::value >> 1;
    constexpr int off = (PAR == 0 ? (BUF + RING - 1) % RING : BUF) * KV_TILE_BYTES + (PAR == 0 ? 8192 : 0) + 512 * (db >> 1);  // + V_RING inside v_rd
    return tr_read(lds + off + v_rd[decltype(uc)::value][db & 1]);
  };
  auto frag_read = [&](auto parc, auto bufc, auto sc) __attribute__((always_inline)) -> u32x4 {
    if constexpr ((decltype(sc)::value & 1) == 0) {
      return frag_read_k(parc, bufc, sc);
    } else {
      const bf16x4 lo = frag_read_v(parc, bufc, sc, IntC<0>{}), hi = frag_read_v(parc, bufc, sc, IntC<1>{});
      return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };

  f32x4 o[NQB][8];    // accumulator file: o[qb][db][i] = O[q0 + 16 qb + c][16 db + 4 g + i]
  f32x4 negm[NQB];    // -m of the lane's row of block qb in all 4 registers
  u32x4 qf[NQB][4];   // c * Q^T fragments (accumulator file)
  float l_run[NQB];
  f32x4 s0[2][NQB], s1[2][NQB];   // [kb][qb]: scores of the even / odd half tiles
  u32x4 pfa[NQB], pfb[NQB];       // [qb]: P of the even / odd half tiles
  u32x4 ring[4];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    l_run[qb] = 0.f;
    negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(negm[qb]));
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      o[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+a"(o[qb][db]));
    }
    pfa[qb] = pfb[qb] = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) s1[kb][qb] = f32x4{-1e30f, -1e30f, -1e30f, -1e30f};  // "half tile -1": every key masked
  }

  const int nt = (skv + KVB - 1) / KVB;

  // S[kb][qb][i] of the half tile starting at key0 is key key0 + 8 g + 4 kb + i
  auto mask_half = [&](f32x4 (&sv)[2][NQB], int key0) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (key0 + 8 * g + 4 * kb + i >= skv) {
#pragma unroll
          for (int qb = 0; qb < NQB; ++qb) sv[kb][qb][i] = -1e30f;
        }
  };

  // Rare path, entered before any P of half tile k is used: move O, l, m, S(k) and the complete
  // S(k+1) of block qb to the row maximum of S(k) (never downwards, except for half tile 0, whose
  // maximum becomes m whatever its sign) and redo P(k).  MFMAs on these registers may have been
  // issued a few cycles ago and others follow right after, hence the explicit wait states.
  auto fixup = [&](auto qbc, f32x4 (&sc)[2][NQB], f32x4 (&sn)[2][NQB], u32x4& pc, bool first, float& ls) __attribute__((always_inline)) {
    constexpr int qb = decltype(qbc)::value;
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(o[qb][0]), "+a"(o[qb][1]), "+a"(o[qb][2]), "+a"(o[qb][3]), "+a"(o[qb][4]), "+a"(o[qb][5]), "+a"(o[qb][6]), "+a"(o[qb][7]),
                 "+v"(sc[0][qb]), "+v"(sc[1][qb]), "+v"(sn[0][qb]), "+v"(sn[1][qb]));
    float mxl = sc[0][qb][0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mxl = fmaxf(mxl, sc[i >> 2][qb][i & 3]);
    mxl = fmaxf(mxl, __shfl_xor(mxl, 16, 64));
    const float mxc = fmaxf(mxl, __shfl_xor(mxl, 32, 64));
    const float delta = first ? mxc : fmaxf(mxc, 0.f);
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    l_run[qb] *= alpha;
    static_for<0, 8>([&](auto dbc) __attribute__((always_inline)) {  // one tuple at a time through the architectural file
      constexpr int db = decltype(dbc)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[qb][db][i] *= alpha;
      asm volatile("" : "+a"(o[qb][db]));
    });
    const float nm = negm[qb][0] - delta;
#pragma unroll
    for (int i = 0; i < 4; ++i) negm[qb][i] = nm;
    float e[8];
    ls = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[i >> 2][qb][i & 3] -= delta;
      sn[i >> 2][qb][i & 3] -= delta;
      e[i] = __builtin_amdgcn_exp2f(sc[i >> 2][qb][i & 3]);
      ls += e[i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) pc[j] = pack2(e[2 * j], e[2 * j + 1]);
    asm volatile("s_nop 1" : "+a"(o[qb][0]), "+a"(o[qb][1]), "+a"(o[qb][2]), "+a"(o[qb][3]), "+a"(o[qb][4]), "+a"(o[qb][5]), "+a"(o[qb][6]), "+a"(o[qb][7]),
                 "+v"(negm[qb]), "+v"(pc), "+v"(sn[0][qb]), "+v"(sn[1][qb]));
  };

  // One period (k = 2t + PAR) of the pipeline.  Periods outside the real half tiles (k < 0 before
  // the first tile, k >= 2 nt at the end) run the same code on zero tiles with every key masked.
  auto period = [&](auto parc, auto bufc, int t) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value, BUF = decltype(bufc)::value;
    f32x4 (&Sc)[2][NQB] = PAR == 0 ? s0 : s1;     // S(k): exponentiated in this period
    f32x4 (&Sn)[2][NQB] = PAR == 0 ? s1 : s0;     // S(k+1): accumulated in this period
    u32x4 (&Pc)[NQB] = PAR == 0 ? pfa : pfb;      // P(k): produced
    u32x4 (&Pp)[NQB] = PAR == 0 ? pfb : pfa;      // P(k-1): consumed
    constexpr int NPAIR = 4 * NQB;  // softmax pairs per period; a period has 16 NQB = 4 NPAIR gaps
    float ls[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) ls[qb] = 0.f;
    float ea[NPAIR], eb[NPAIR];  // exps of pair p = (qb = p >> 2, j = p & 3): scores [j >> 1][qb][2 (j & 1)], + 1
    const int key0 = t * KVB + 32 * PAR;
    if (__builtin_expect(key0 + 32 > skv, 0)) mask_half(Sc, key0);  // wave-uniform: only the last real half tile and the zero tiles (kept out of the fall-through path)
    // DMA of this period: odd periods bring K(t+4) into the slot K(t) just left, even periods
    // V(t+2) into the slot V(t-2) left at the previous barrier; four pieces each, on even steps:
    // 0, 4, 8, 10 of the even periods (before their barrier, so that the count there is the same
    // on every trip), 4, 8, 12, 14 of the odd ones (well after the barrier that released the slot).
    const uint32_t dma_soff = __builtin_amdgcn_readfirstlane(PAR == 1 ? (uint32_t)(t + 4) * k_tile_bytes : (uint32_t)(t + 2) * v_tile_bytes);
    bf16x4 vlo, vhi;
    static_for<0, 16>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, m = s >> 1;
      constexpr int dma_i = PAR == 1 ? (s == 4 ? 0 : s == 8 ? 1 : s == 12 ? 2 : s == 14 ? 3 : -1) : (s == 0 ? 0 : s == 4 ? 1 : s == 8 ? 2 : s == 10 ? 3 : -1);
      constexpr bool DMA = dma_i >= 0 && !(X16_ABL & 4) && !(X16_ABL & (PAR == 1 ? 512 : 1024));
      const u32x4 f = ring[s & 3];
      // One gap per query block: the MFMA, then what is not VALU work (it issues in the MFMA's
      // shadow), then the gap's share of the softmax.  Gap G = NQB s + qb of the period carries
      // phase G & 3 of pair G >> 2:  0 first exp | 1 second exp | 2 row-sum add and conversion of
      // the previous pair | 3 second add of the previous pair (this order measured 66.5 cycles per
      // four gaps in isolation, exp | add + convert | exp | add 70.3: tools/probe_valu_beside_mfma.hip).
      // Consumers run a pair behind their exps, so that no instruction waits on a transcendental
      // just issued.
      static_for<0, NQB>([&](auto qbc) __attribute__((always_inline)) {
        constexpr int qb = decltype(qbc)::value, G = NQB * s + qb, pr = G >> 2, ph = G & 3;
        constexpr int sq = pr >> 2, sj = pr & 3, pq = (pr - 1) >> 2, pj = (pr - 1) & 3;
        if constexpr ((s & 1) == 0) {
          if constexpr ((m & 3) == 0) mfma_qk_first(Sn[m >> 2][qb], f, qf[qb][0], negm[qb]);
          else mfma_qk(Sn[m >> 2][qb], f, qf[qb][m & 3]);
        } else {
          mfma_pv(o[qb][m], f, Pp[qb]);
        }
        // Where the step's non-VALU instructions go: beside this MFMA shape an LDS read is free
        // in a gap with spare issue time and costs 3 cycles in an "add" gap (phase 3), 7 in an
        // "add + convert" or second "exp" gap, 10 in a first "exp" gap (tools/probe_valu_beside_mfma.hip).
        // So the read that replaces this step's fragment goes to the step's phase-3 gap (G3), as
        // does the DMA piece of an even step (measured 0.2 % better there than in GX); the first
        // V^T read of an odd step goes to a phase-1 or -2 gap behind gap 0 (GX); the M0 update of
        // the DMA stays in gap 0.
        constexpr int p0 = (NQB * s) & 3;                      // phase of the step's gap 0
        constexpr int G3 = NQB == 1 ? 0 : (3 - p0 + 4) & 3;     // first gap with phase 3 (NQB >= 4: always < NQB)
        constexpr int GX = NQB == 1 ? 0 : (((1 - p0 + 4) & 3) != 0 ? ((1 - p0 + 4) & 3) : ((2 - p0 + 4) & 3));  // phase 1, or phase 2 when phase 1 is gap 0
        static_assert(NQB == 1 || (G3 < NQB && GX < NQB && GX != G3 && GX != 0), "gap placement");
        if constexpr (qb == 0 && DMA) {
          if constexpr (PAR == 1) dma_m0(IntC<K_RING + BUF * KV_TILE_BYTES + 4096 * dma_i>{});
          else dma_m0(IntC<V_RING + ((BUF + 2) % RING) * KV_TILE_BYTES + 4096 * dma_i>{});
          if constexpr (NQB == 1) asm volatile("s_nop 0");  // M0 write -> use, same gap
        }
        // The next step's fragment is claimed in the step's last phase-2 gap, so that the compiler's
        // wait for it is issued in an MFMA's shadow instead of in front of the next step (a wait
        // costs nothing extra there; next to the read in the phase-3 gap it would cost 4 cycles).
        constexpr int GW = ((2 - p0 + 4) & 3) + (((2 - p0 + 4) & 3) + 4 < NQB ? 4 : 0);
        if constexpr (NQB > 1 && qb == GW) asm volatile("" :: "v"(ring[(s + 1) & 3]));
        if constexpr (qb == GX) {
          if constexpr ((s & 1) == 1 && !(X16_ABL & (2 | 64))) {  // first of the two V^T reads of the fragment four steps on
            if constexpr (s + 4 < 16) vlo = frag_read_v(parc, bufc, IntC<s + 4>{}, IntC<0>{});
            else if constexpr (PAR == 0) vlo = frag_read_v(IntC<1>{}, bufc, IntC<s - 12>{}, IntC<0>{});
            else vlo = frag_read_v(IntC<0>{}, IntC<(BUF + 1) % RING>{}, IntC<s - 12>{}, IntC<0>{});
          }
          if constexpr (DMA && NQB == 1) {
            if constexpr (PAR == 1) dma_go(k_rs, k_vo[dma_i], dma_soff);
            else dma_go(v_rs, v_vo[dma_i], dma_soff);
          }
        }
        if constexpr (NQB > 1 && qb == G3 && DMA) {  // DMA steps are even steps: the gap's other guest is the K read
          if constexpr (PAR == 1) dma_go(k_rs, k_vo[dma_i], dma_soff);
          else dma_go(v_rs, v_vo[dma_i], dma_soff);
        }
        if constexpr (qb == G3) {
          // The fragment of step s + 4 is requested (it lands in fresh registers: this step's later
          // MFMAs keep f).
          if constexpr (NQB == 1) asm volatile("" :: "v"(ring[(s + 1) & 3]));
          if constexpr ((X16_ABL & 2) || ((X16_ABL & 32) && (s & 1) == 0) || ((X16_ABL & 64) && (s & 1) == 1)) {
            asm volatile("" : "+v"(ring[s & 3]));
          } else if constexpr ((s & 1) == 0) {
            if constexpr (s + 4 < 16) ring[s & 3] = frag_read_k(parc, bufc, IntC<s + 4>{});
            else if constexpr (PAR == 0) ring[s & 3] = frag_read_k(IntC<1>{}, bufc, IntC<s - 12>{});
            else ring[s & 3] = frag_read_k(IntC<0>{}, IntC<(BUF + 1) % RING>{}, IntC<s - 12>{});
          } else {
            if constexpr (s + 4 < 16) vhi = frag_read_v(parc, bufc, IntC<s + 4>{}, IntC<1>{});
            else if constexpr (PAR == 0) vhi = frag_read_v(IntC<1>{}, bufc, IntC<s - 12>{}, IntC<1>{});
            else vhi = frag_read_v(IntC<0>{}, IntC<(BUF + 1) % RING>{}, IntC<s - 12>{}, IntC<1>{});
          }
        }
        if constexpr (qb == NQB - 1 && (s & 1) == 1 && !(X16_ABL & (2 | 64)))
          ring[s & 3] = __builtin_bit_cast(u32x4, __builtin_shufflevector(vlo, vhi, 0, 1, 2, 3, 4, 5, 6, 7));
        if constexpr (!(X16_ABL & 1)) {
          if constexpr (ph == 0) {
            if constexpr (X16_ABL & 128) ea[pr] = Sc[sj >> 1][sq][2 * (sj & 1)];
            else ea[pr] = exp2_op(Sc[sj >> 1][sq][2 * (sj & 1)]);
          } else if constexpr (ph == 1) {
            if constexpr (X16_ABL & 128) eb[pr] = Sc[sj >> 1][sq][2 * (sj & 1) + 1];
            else eb[pr] = exp2_op(Sc[sj >> 1][sq][2 * (sj & 1) + 1]);
            if constexpr (X16_ABL & 256) asm volatile("" :: "v"(ea[pr]), "v"(eb[pr]));
          } else if constexpr (pr > 0 && !(X16_ABL & 256)) {
            if constexpr (ph == 2) {
              add_op(ls[pq], ea[pr - 1]);
              Pc[pq][pj] = cvt_pk_op(ea[pr - 1], eb[pr - 1]);
            } else {
              add_op(ls[pq], eb[pr - 1]);
            }
          }
        }
        // Tile barrier: K(t+1) and V(t) (issued three tiles ago) have landed; after the barrier that
        // holds for every wave.  Slot reuse is safe in program order, not by timing: every read of
        // K(t) and V(t-1) -- the slots that DMA instructions issued after this barrier overwrite --
        // precedes the barrier in each wave's instruction stream (the last ones are the fragment
        // requests of steps 10 and 11), so when any wave issues such a DMA, all four waves' reads
        // of the slot are already in the CU's LDS queue, which serves its requests in order; the
        // DMA's data arrives there nine steps and a memory round trip later.  They are not waited
        // for here (their consumers wait, within four steps).  X16_LAG holds one wave back behind
        // the barrier to check the first half of that argument (tests/test_attn_fullsize_gpu.py).
        if constexpr (qb == NQB - 1 && PAR == 0 && s == 11 && !(X16_ABL & 8)) {
          asm volatile("" ::: "memory");
          __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * DMA_PER_GROUP) & 15) | (((2 * DMA_PER_GROUP) >> 4) << 14));  // vmcnt(16)
          __builtin_amdgcn_s_barrier();
#if X16_LAG
          if (wid == (t & 3)) __builtin_amdgcn_s_sleep(X16_LAG);  // 64 X16_LAG cycles behind the others, a different wave every tile
#endif
          asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    // ---- overflow check of half tile k, before any P(k) is used.  Every exp is positive, so a
    // lane whose 8 exps of a row sum to at most 2^kRescaleThr holds no P above that bound: no row
    // maximum is tracked in the loop, the rare path forms it. ----
    if constexpr (!(X16_ABL & (1 | 256))) {  // consumers of the last pair's exps
      ls[NQB - 1] += ea[NPAIR - 1] + eb[NPAIR - 1];
      Pc[NQB - 1][3] = pack2(ea[NPAIR - 1], eb[NPAIR - 1]);
    }
    float lmax = 0.f;   // (the sums are positive: the first one needs no max with it)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      asm volatile("" : "+v"(Pc[qb]), "+v"(ls[qb]));
      lmax = qb == 0 ? ls[0] : fmaxf(lmax, ls[qb]);
    }
    const bool first = PAR == 0 && t == 0;  // half tile 0: m becomes its true row maximum
    // (half tile 0 takes the rare path through the threshold -- a scalar select made long before -- so that the period ends in one
    // compare and one branch: with `first || over` the compiler tests `first` in a branch of its own, with `first | over` in
    // five scalar instructions behind the compare)
    const float thr = __builtin_bit_cast(float, first ? (int)0xbf800000 : __builtin_bit_cast(int, kRescaleSum));  // (selected as an integer: a scalar select, no v_cndmask)
    if (!(X16_ABL & 16) && __builtin_expect(__any(lmax > thr), 0)) {
      static_for<0, NQB>([&](auto qbc) __attribute__((always_inline)) { fixup(qbc, Sc, Sn, Pc[decltype(qbc)::value], first, ls[decltype(qbc)::value]); });
    }
    // (as single adds: left to itself the compiler pairs them into v_pk_add_f32, which costs a lone wave about 18 issue
    // cycles where two v_add_f32 cost 8, and computes them on both sides of the branch above)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) add_op(l_run[qb], ls[qb]);
  };

  // ---- prologue: Q fragments scaled and parked; DMA groups of tiles -4, -3, -2 (K0; K1, V0; K2,
  // V1); V slot 3 zeroed by hand (it stands for V(-1): the two periods before half tile 0 multiply
  // it by P = 0) ----
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float f[8];
      unpack8(qraw[qb][ks], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] *= p.c;
      qf[qb][ks] = pack8(f);
      asm volatile("" : "+a"(qf[qb][ks]));
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

  round_leave(rs);

  // ---- normalise and store ----
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(o[qb][0]), "+a"(o[qb][1]), "+a"(o[qb][2]), "+a"(o[qb][3]), "+a"(o[qb][4]), "+a"(o[qb][5]), "+a"(o[qb][6]), "+a"(o[qb][7]));
    float l = l_run[qb];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const int qrow = q0 + 16 * qb + c16;
    if (p.nparts > 1) {
      // partial result of this key range: O as accumulated (relative to m), and (m, l) of the row
      if (qrow < p.sq) {
        const int64_t prow = ((int64_t)part * p.heads + head) * p.sq + qrow;
        float* po = p.part_o + prow * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < 8; ++db) *reinterpret_cast<f32x4*>(po + 16 * db) = o[qb][db];
        if (g == 0) *reinterpret_cast<f32x2*>(p.part_ml + prow * 2) = f32x2{-negm[qb][0], l};
      }
      continue;
    }
    const float inv = 1.0f / l;
    if (qrow < p.sq) {
      uint16_t* op = p.out + (int64_t)qrow * p.ldo + head * HD + 4 * g;
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        u32x2 pk = {pack2(o[qb][db][0] * inv, o[qb][db][1] * inv), pack2(o[qb][db][2] * inv, o[qb][db][3] * inv)};
        *reinterpret_cast<u32x2*>(op + 16 * db) = pk;
      }
    }
  }
}

}  // namespace

// Merge of a key split: out[row, head] = sum_j 2^(m_j - M) O_j / sum_j 2^(m_j - M) l_j with M = max_j m_j (scores
// are in log2 units).  One thread per (row, head, 4 channels).
__global__ __launch_bounds__(256) void attn_merge_parts_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, uint16_t* __restrict__ out,
                                                               int64_t ldo, int sq, int heads, int nparts) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)sq * heads * 32;
  if (idx >= total) return;
  const int c4 = (int)(idx & 31);
  const int64_t rh = idx >> 5;           // head-major, row-minor like the partial buffers
  const int head = (int)(rh / sq);
  const int row = (int)(rh - (int64_t)head * sq);
  const int64_t stride = (int64_t)heads * sq;  // rows per part
  float mx = -3.0e38f;
  for (int j = 0; j < nparts; ++j) mx = fmaxf(mx, part_ml[(j * stride + rh) * 2]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float lsum = 0.f;
  for (int j = 0; j < nparts; ++j) {
    const f32x2 ml = *reinterpret_cast<const f32x2*>(part_ml + (j * stride + rh) * 2);
    const float w = __builtin_amdgcn_exp2f(ml[0] - mx);
    const f32x4 oj = *reinterpret_cast<const f32x4*>(part_o + (j * stride + rh) * HD + 4 * c4);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += w * oj[i];
    lsum += w * ml[1];
  }
  const float inv = 1.0f / lsum;
  const u32x2 pk = {pack2(acc[0] * inv, acc[1] * inv), pack2(acc[2] * inv, acc[3] * inv)};
  *reinterpret_cast<u32x2*>(out + (int64_t)row * ldo + head * HD + 4 * c4) = pk;
}

// Launch for p already validated by lx2v_attn_fwd_bf16 (no RoPE; K/V extent + 8 tiles < 4 GiB).
// Query rows per workgroup = 4 waves x 16 NQB.  Five blocks per wave stream K/V a fifth less often
// and measure about 4 % faster per row, but workgroups run in rounds of one per CU and a round
// costs in proportion to NQB, so the choice goes by rounds x NQB (14B/720p: 46 x 4 + tail against
// 37 x 5 x 0.96 + tail; 1.3B/480p: 6 x 4 against 5 x 5 x 0.96).  A call becomes TWO launches when its
// last round would be poorly filled:
//  * few workgroups left, all of the last head: the main launch stops at the last full round and
//    the remaining query rows of that head go to a launch with one block per wave (64 rows per
//    workgroup: more, much shorter workgroups instead of a nearly empty full-length round;
//    14B/720p: 8 workgroups of 320 rows become 37 of 64, 77.4 -> 76.1 ms per call);
//  * otherwise, when it pays by the same cost model: the query ROWS are split -- rows [0, row0) of
//    every head in full rounds of the main block count, the rest of every head with four blocks or
//    one block per wave.  This is the case of launches with few heads (a head chunk of a
//    sequence-parallel rank: 5 heads x 75 600 rows = 4.63 rounds of 320-row workgroups become 4
//    full rounds + one round of 256-row workgroups).
// Options attn_qb = 4|5 fix NQB and attn_tail = 0 keeps everything in one launch (A/B switches).
namespace {

// rows [row0, row1) of every head (or of the last head only) with `m` blocks per wave
int launch_rows(const lx2v_attn::AttnParams& p, int64_t row0, int64_t row1, int m, bool last_head_only, hipStream_t st, const char* what) {
  lx2v_attn::AttnParams pp = p;
  const int64_t h0 = last_head_only ? p.heads - 1 : 0;
  pp.q = p.q + row0 * p.ldq + h0 * HD;
  pp.out = p.out + row0 * p.ldo + h0 * HD;
  pp.k = p.k + h0 * HD;
  pp.v = p.v + h0 * HD;
  pp.heads = last_head_only ? 1 : p.heads;
  pp.sq = (int)(row1 - row0);
  pp.nqb = (int)((pp.sq + 64 * m - 1) / (64 * m));
  const int64_t nwg = (int64_t)pp.nqb * pp.heads;
  if (nwg == 0) return LX2V_OK;
  switch (m) {
    case 5: return launch_with_rounds(attn_fwd_x16_kernel<true, 5>, attn_fwd_x16_kernel<false, 5>, X16_THREADS, pp, nwg, st, what);
    case 4: return launch_with_rounds(attn_fwd_x16_kernel<true, 4>, attn_fwd_x16_kernel<false, 4>, X16_THREADS, pp, nwg, st, what);
    default: return launch_with_rounds(attn_fwd_x16_kernel<true, 1>, attn_fwd_x16_kernel<false, 1>, X16_THREADS, pp, nwg, st, what);
  }
}

}  // namespace

namespace {

struct Plan {
  int m, t;            // blocks per wave of the main launch / of the second launch (0: none)
  int64_t row0;        // first row of the second launch
  bool last_head_only; // the second launch covers the last head only
  int split;           // key ranges of the second launch (1: none)
  double cost;
};

int device_cus() {
  static int cus_cached[64];
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
    if (cus_cached[dev] == 0) {
      int v = 0;
      cus_cached[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    cus = cus_cached[dev];
  }
  return cus;
}

// Bytes of partial results a key split of `rows` query rows of every head into `split` ranges needs.
int64_t split_bytes(int64_t rows, int heads, int split) { return rows * heads * split * (int64_t)(HD + 2) * 4; }

// The launch plan of a call.  Workgroups run in rounds of one per CU and a round costs in proportion to the
// blocks per wave, so the choice goes by rounds x cost per round (units of "64 rows per wave": m blocks cost m,
// 0.96 m for m = 5, about 1.3 for m = 1, whose workgroups spend relatively more on their prologue; a second
// launch adds 0.15).  `max_split_bytes`: what a key split may use for its partial results (0: none).
Plan make_plan(const lx2v_attn::AttnParams& p, int cus, int64_t max_split_bytes) {
  auto blocks = [&](int m) { return (int64_t)((p.sq + 64 * m - 1) / (64 * m)); };
  auto rounds = [&](int64_t wgs) { return (double)((wgs + cus - 1) / cus); };
  auto per_round = [](int m) { return m == 5 ? 4.8 : m == 4 ? 4.0 : 1.3; };
  // the workgroups of a last round that is worth a launch of their own: few (one round of 64-row
  // workgroups takes their rows) and all of the last head
  auto tail_of = [&](int m) -> int64_t {
    const int64_t wgs = blocks(m) * p.heads, rest = wgs % cus;
    return (wgs > cus && rest > 0 && rest <= blocks(m) && rest * m <= cus) ? rest : 0;
  };
  const bool tail_on = lx2v_options().attn_tail != 0;
  const int64_t ntiles = (p.skv + KVB - 1) / KVB;
  auto plan_for = [&](int m) {
    const int64_t nb = blocks(m), wgs = nb * p.heads;
    Plan best{m, 0, p.sq, false, 1, rounds(wgs) * per_round(m)};
    if (!tail_on || wgs <= cus) return best;
    if (const int64_t rest = tail_of(m); rest > 0) {
      const int64_t row0 = (nb - rest) * 64 * m;
      const Plan c{m, 1, row0, true, 1, (double)(wgs / cus) * per_round(m) + 1.3};
      if (c.cost < best.cost) best = c;
      // ... or, given a workspace, the same rows as whole workgroups over 1 / sp of the keys each, as many ranges as
      // fill one round (14B/720p: 8 workgroups x 32 ranges of 37 tiles instead of 37 workgroups of all 1182 tiles)
      int sp = (int)std::min<int64_t>(std::min<int64_t>(32, cus / rest), ntiles / 8);
      if (sp >= 2) sp = (int)((ntiles + (ntiles + sp - 1) / sp - 1) / ((ntiles + sp - 1) / sp));  // as many ranges as are not empty
      if (sp >= 2 && split_bytes(p.sq - row0, 1, sp) <= max_split_bytes) {
        const Plan k{m, m, row0, true, sp, (double)(wgs / cus) * per_round(m) + per_round(m) / sp + 0.08 + 0.1 + 0.15};
        if (k.cost < best.cost) best = k;
      }
      return best;
    }
    // rows split: `full` full rounds of the main block count, the other rows of every head behind them
    const int64_t full = wgs / cus, nb_main = full * cus / p.heads;
    if (nb_main > 0 && nb_main < nb) {
      const int64_t row0 = nb_main * 64 * m, rest_rows = p.sq - row0;
      const double main_cost = rounds(nb_main * p.heads) * per_round(m);
      for (int t : {4, 1}) {
        if (t == m) continue;
        const int64_t twgs = (rest_rows + 64 * t - 1) / (64 * t) * p.heads;
        const Plan c{m, t, row0, false, 1, main_cost + rounds(twgs) * per_round(t) + 0.15};
        if (c.cost < 0.99 * best.cost) best = c;
      }
      // ... and the same rows with their KEYS split into `sp` ranges: workgroups of 1 / sp the length fill the last
      // round(s) more evenly than whole ones can (a launch of few heads is a few rounds long, and its last round
      // is a large share of it).  Each range costs a prologue and a partial-result store of its own (0.08), the
      // merge pass 0.1.
      for (int t : {5, 4}) {
        const int64_t twgs = (rest_rows + 64 * t - 1) / (64 * t) * p.heads;
        for (int sp = 2; sp <= 8 && sp * 8 <= ntiles; ++sp) {
          if (split_bytes(rest_rows, p.heads, sp) > max_split_bytes) break;
          const Plan c{m, t, row0, false, sp, main_cost + rounds(twgs * sp) * (per_round(t) / sp + 0.08) + 0.1 + 0.15};
          if (c.cost < 0.985 * best.cost) best = c;
        }
      }
    }
    return best;
  };
  Plan plan = plan_for(5);
  if (const Plan p4 = plan_for(4); p4.cost < plan.cost) plan = p4;
  if (lx2v_options().attn_qb != 0) plan = plan_for(lx2v_options().attn_qb);
  return plan;
}

}  // namespace

// Upper bound of the workspace a call of this shape can use (the plan is made without a limit).
int64_t lx2v_attn_x16_workspace_bytes(int64_t sq, int64_t skv, int heads) {
  lx2v_attn::AttnParams p{};
  p.sq = (int)sq; p.skv = (int)skv; p.heads = heads;
  const Plan plan = make_plan(p, device_cus(), INT64_MAX);
  return plan.split > 1 ? (split_bytes(p.sq - plan.row0, plan.last_head_only ? 1 : heads, plan.split) + 255) / 256 * 256 : 0;
}

namespace {

// Rows [row0, sq) of every head (or of the last head only), keys in `split` ranges of whole tiles: partial results
// into the workspace, then the merge.
int launch_key_split(const lx2v_attn::AttnParams& p, int64_t row0, int t, int split, bool last_head_only, hipStream_t st) {
  lx2v_attn::AttnParams pp = p;
  const int64_t h0 = last_head_only ? p.heads - 1 : 0;
  const int64_t rest_rows = p.sq - row0;
  const int64_t ntiles = (p.skv + KVB - 1) / KVB;
  pp.q = p.q + row0 * p.ldq + h0 * HD;
  pp.k = p.k + h0 * HD;
  pp.v = p.v + h0 * HD;
  pp.heads = last_head_only ? 1 : p.heads;
  pp.sq = (int)rest_rows;
  pp.nqb = (int)((rest_rows + 64 * t - 1) / (64 * t));
  pp.nparts = split;
  pp.part_keys = (int)((ntiles + split - 1) / split) * KVB;
  pp.part_o = reinterpret_cast<float*>(p.workspace);
  pp.part_ml = pp.part_o + rest_rows * pp.heads * split * HD;
  const int64_t nwg = (int64_t)pp.nqb * pp.heads * split;
  const int rc = t == 5 ? launch_with_rounds(attn_fwd_x16_kernel<true, 5>, attn_fwd_x16_kernel<false, 5>, X16_THREADS, pp, nwg, st, "attn_fwd(x16 key split)")
                        : launch_with_rounds(attn_fwd_x16_kernel<true, 4>, attn_fwd_x16_kernel<false, 4>, X16_THREADS, pp, nwg, st, "attn_fwd(x16 key split)");
  if (rc != LX2V_OK) return rc;
  const int64_t threads = rest_rows * pp.heads * 32;
  hipLaunchKernelGGL(attn_merge_parts_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, pp.part_o, pp.part_ml, p.out + row0 * p.ldo + h0 * HD, p.ldo,
                     (int)rest_rows, pp.heads, split);
  return LX2V_OK;
}

}  // namespace

int lx2v_attn_launch_x16(const lx2v_attn::AttnParams& p, hipStream_t st) {
  const int cus = device_cus();
  const bool ws_ok = p.workspace != nullptr && (reinterpret_cast<uintptr_t>(p.workspace) & 15u) == 0;
  const Plan plan = make_plan(p, cus, ws_ok ? p.workspace_bytes : 0);
  auto blocks = [&](int m) { return (int64_t)((p.sq + 64 * m - 1) / (64 * m)); };
  if (plan.t == 0) return launch_rows(p, 0, p.sq, plan.m, false, st, "attn_fwd(x16)");
  if (plan.last_head_only) {
    // every head but the last in full, the last head up to row0; then the last head's remaining rows
    lx2v_attn::AttnParams pp = p;
    pp.nqb = (int)blocks(plan.m);
    const int64_t wgs = (int64_t)pp.nqb * p.heads;
    const int64_t nwg = wgs - wgs % cus;
    const int rc = plan.m == 5 ? launch_with_rounds(attn_fwd_x16_kernel<true, 5>, attn_fwd_x16_kernel<false, 5>, X16_THREADS, pp, nwg, st, "attn_fwd(x16)")
                               : launch_with_rounds(attn_fwd_x16_kernel<true, 4>, attn_fwd_x16_kernel<false, 4>, X16_THREADS, pp, nwg, st, "attn_fwd(x16)");
    if (rc != LX2V_OK) return rc;
    if (plan.split > 1) return launch_key_split(p, plan.row0, plan.t, plan.split, true, st);
    return launch_rows(p, plan.row0, p.sq, 1, true, st, "attn_fwd(x16 tail)");
  }
  const int rc = launch_rows(p, 0, plan.row0, plan.m, false, st, "attn_fwd(x16)");
  if (rc != LX2V_OK) return rc;
  if (plan.split == 1) return launch_rows(p, plan.row0, p.sq, plan.t, false, st, "attn_fwd(x16 rows tail)");
  return launch_key_split(p, plan.row0, plan.t, plan.split, false, st);
}
