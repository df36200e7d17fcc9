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
#include "attn_w4_common.h"

using namespace lx2v_attn;

namespace {

#ifndef W4_ABL
#define W4_ABL 0  // timing ablations (results are wrong with any bit set): 1 softmax VALU, 2 fragment reads, 4 staging, 8 barrier, 16 maximum check, 32 K fragment reads, 64 V fragment reads, 128 exps, 256 softmax VALU other than exps
#endif
constexpr int W4_THREADS = 256;
constexpr int RING = 4;                        // tiles per LDS ring
constexpr int K_RING = 0;                      // LDS byte offset of K slot 0
constexpr int V_RING = RING * KV_TILE_BYTES;   // same for V
constexpr float kRescaleSum = 64.0f;             // 2^kRescaleThr
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

// SHORT_KV only names a second symbol for short key sequences (cross-attention), so that kernel
// profiles keep self- and cross-attention apart.
template <bool SHORT_KV>
__global__ __launch_bounds__(W4_THREADS) void attn_fwd_w4_kernel(const AttnParams p, const RoundSync rs) {
  __shared__ __attribute__((aligned(16))) char lds[2 * RING * KV_TILE_BYTES];
  round_enter(rs);

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
  const int vrow = 8 * (wid >> 1) + ((lane >> 2) & 7);
  int k_vo[4], v_vo[4];  // piece i: 16 rows further
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    k_vo[i] = (int)((krow + 16 * i) * p.ldk * 2) + (((lane & 15) ^ (krow & 15)) << 4);
    v_vo[i] = (int)((vrow + 16 * i) * p.ldv * 2) + ((4 * (2 * (wid & 1) + (lane >> 5)) + ((lane & 3) ^ ((vrow >> 2) & 3))) << 4);
  }
  const uint32_t k_tile_bytes = (uint32_t)(KVB * p.ldk * 2), v_tile_bytes = (uint32_t)(KVB * p.ldv * 2);
  const uint32_t dma_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds) + 1024 * wid_u;
  // lane L of the wave: 16 bytes from base + voff + soff to LDS byte dma_base + DST + 16 L.  Every
  // instruction of a lone wave costs issue time, scalar ones included: M0 is set by one add and
  // not restored (nothing else in this kernel uses it).
  auto dma16 = [&](const u32x4& rs, int voff, uint32_t soff, auto dstc) __attribute__((always_inline)) {
    const uint32_t base = dma_base;  // (a capture used only as an asm operand is not captured)
    asm volatile("s_add_u32 m0, %2, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 :: "v"(voff), "s"(rs), "s"(base), "s"(soff), "i"(decltype(dstc)::value) : "memory", "scc");
  };
  auto dma_k = [&](int t, auto slotc, auto ic, uint32_t soff = 0) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    dma16(k_rs, k_vo[i], (uint32_t)t * k_tile_bytes + soff, IntC<K_RING + decltype(slotc)::value * KV_TILE_BYTES + 4096 * i>{});
  };
  auto dma_v = [&](int t, auto slotc, auto ic, uint32_t soff = 0) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    dma16(v_rs, v_vo[i], (uint32_t)t * v_tile_bytes + soff, IntC<V_RING + decltype(slotc)::value * KV_TILE_BYTES + 4096 * i>{});
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
  auto frag_read_k = [&](auto parc, auto bufc, auto sc) __attribute__((always_inline)) -> u32x4 {
    constexpr int PAR = decltype(parc)::value, BUF = decltype(bufc)::value, m = decltype(sc)::value >> 1;
    constexpr int off = K_RING + (PAR == 0 ? BUF : (BUF + 1) % RING) * KV_TILE_BYTES + (PAR == 0 ? 8192 : 0);
    return *reinterpret_cast<const u32x4*>(lds + off + k_rd[m]);
  };
  auto frag_read_v = [&](auto parc, auto bufc, auto sc, auto uc) __attribute__((always_inline)) -> bf16x4 {
    constexpr int PAR = decltype(parc)::value, BUF = decltype(bufc)::value, m = decltype(sc)::value >> 1;
    constexpr int off = (PAR == 0 ? (BUF + RING - 1) % RING : BUF) * KV_TILE_BYTES + 512 * (m & 3);  // + V_RING inside v_rd
    constexpr int ks4 = (PAR == 0 ? 2 : 0) + (m >> 2);
    return tr_read(lds + off + v_rd[ks4][decltype(uc)::value]);
  };
  auto frag_read = [&](auto parc, auto bufc, auto sc) __attribute__((always_inline)) -> u32x4 {
    if constexpr ((decltype(sc)::value & 1) == 0) {
      return frag_read_k(parc, bufc, sc);
    } else {
      const bf16x4 lo = frag_read_v(parc, bufc, sc, IntC<0>{}), hi = frag_read_v(parc, bufc, sc, IntC<1>{});
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
  // S(k+1) of block b to the row maximum of S(k) (never downwards, except for half tile 0, whose
  // maximum becomes m whatever its sign) and redo P(k).  MFMAs on these registers may have been
  // issued a few cycles ago and others follow right after, hence the explicit wait states.
  auto fixup = [&](auto bc, f32x16& sc, f32x16& sn, u32x4 (&pc)[2], bool first, float& ls) __attribute__((always_inline)) {
    constexpr int b = decltype(bc)::value;
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(o[b][0]), "+a"(o[b][1]), "+a"(o[b][2]), "+a"(o[b][3]), "+v"(sc), "+v"(sn));
    float mxl = sc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mxl = fmaxf(mxl, sc[i]);
    const float mxc = half_swap_max(mxl);
    const float delta = first ? mxc : fmaxf(mxc, 0.f);
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
    float e[16];
    ls = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sc[i] -= delta;
      sn[i] -= delta;
      e[i] = __builtin_amdgcn_exp2f(sc[i]);
      ls += e[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) pc[i >> 2][i & 3] = pack2(e[2 * i], e[2 * i + 1]);
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
    float lsa[2] = {0.f, 0.f}, lsb[2] = {0.f, 0.f};
    float ea[16], eb[16];  // exps of step s (block s & 1, chunk s >> 1)
    const int key0 = t * KVB + 32 * PAR;
    if (key0 + 32 > p.skv) {  // wave-uniform: only the last real half tile and the zero tiles
      mask_half(Sc[0], key0);
      mask_half(Sc[1], key0);
    }
    // DMA of this period: odd periods bring K(t+4) into the slot K(t) just left, even periods
    // V(t+2) into the slot V(t-2) left at the previous barrier; four pieces each, on even steps
    // (one fragment read instruction instead of two): steps 0, 4, 8, 10 of the even periods (before
    // their barrier, so that the count there is the same on every trip), steps 4, 8, 12, 14 of the
    // odd ones (well after the barrier that released the slot).
    const uint32_t dma_soff = __builtin_amdgcn_readfirstlane(PAR == 1 ? (uint32_t)(t + 4) * k_tile_bytes : (uint32_t)(t + 2) * v_tile_bytes);
    bf16x4 vlo;
    static_for<0, 16>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, m = s >> 1;
      constexpr int bsm = s & 1, c = s >> 1;  // softmax chunk of this step: scores 2c, 2c+1 of block bsm
      constexpr bool NO_READ = (W4_ABL & 2) || ((W4_ABL & 32) && (s & 1) == 0) || ((W4_ABL & 64) && (s & 1) == 1);
      constexpr int dma_i = PAR == 1 ? (s == 4 ? 0 : s == 8 ? 1 : s == 12 ? 2 : s == 14 ? 3 : -1) : (s == 0 ? 0 : s == 4 ? 1 : s == 8 ? 2 : s == 10 ? 3 : -1);
      const u32x4 f = ring[s & 3];
      // ---- gap 1: MFMA of block 0 ----
      if constexpr ((s & 1) == 0) {
        if constexpr (m == 0) mfma_qk_first(Sn[0], f, qf[0][0], negm[0]);
        else mfma_qk(Sn[0], f, qf[0][m]);
      } else {
        mfma_pv(o[0][m & 3], f, Pp[0][m >> 2]);
      }
      // The exps of this step's chunk are only issued here; their consumers (row sum, bf16 pack) run
      // one step later, so that no instruction waits on a transcendental just issued: with a
      // single wave on the SIMD nothing else would fill that wait.
      if constexpr (!(W4_ABL & 1)) {
        if constexpr (W4_ABL & 128) ea[s] = Sc[bsm][2 * c];
        else ea[s] = exp2_op(Sc[bsm][2 * c]);
        if constexpr (W4_ABL & 256) asm volatile("" :: "v"(ea[s]));
        else if constexpr (s > 0) {
          constexpr int pc = (s - 1) >> 1;
          add_op(lsa[bsm ^ 1], ea[s - 1]);
          Pc[bsm ^ 1][pc >> 2][pc & 3] = cvt_pk_op(ea[s - 1], eb[s - 1]);
        }
      }
      if constexpr ((s & 1) == 1 && !NO_READ) {  // first of the two V^T reads of the fragment four steps on
        if constexpr (s + 4 < 16) vlo = frag_read_v(parc, bufc, IntC<s + 4>{}, IntC<0>{});
        else if constexpr (PAR == 0) vlo = frag_read_v(IntC<1>{}, bufc, IntC<s - 12>{}, IntC<0>{});
        else vlo = frag_read_v(IntC<0>{}, IntC<(BUF + 1) % RING>{}, IntC<s - 12>{}, IntC<0>{});
      }
      if constexpr (dma_i >= 0 && !(W4_ABL & 4)) {
        if constexpr (PAR == 1) dma_k(0, bufc, IntC<dma_i>{}, dma_soff);
        else dma_v(0, IntC<(BUF + 2) % RING>{}, IntC<dma_i>{}, dma_soff);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- gap 2: MFMA of block 1 ----
      if constexpr ((s & 1) == 0) {
        if constexpr (m == 0) mfma_qk_first(Sn[1], f, qf[1][0], negm[1]);
        else mfma_qk(Sn[1], f, qf[1][m]);
      } else {
        mfma_pv(o[1][m & 3], f, Pp[1][m >> 2]);
      }
      if constexpr (NO_READ) {
        asm volatile("" : "+v"(ring[s & 3]));
      } else if constexpr ((s & 1) == 0) {
        if constexpr (s + 4 < 16) ring[s & 3] = frag_read_k(parc, bufc, IntC<s + 4>{});
        else if constexpr (PAR == 0) ring[s & 3] = frag_read_k(IntC<1>{}, bufc, IntC<s - 12>{});
        else ring[s & 3] = frag_read_k(IntC<0>{}, IntC<(BUF + 1) % RING>{}, IntC<s - 12>{});
      } else {
        bf16x4 vhi;
        if constexpr (s + 4 < 16) vhi = frag_read_v(parc, bufc, IntC<s + 4>{}, IntC<1>{});
        else if constexpr (PAR == 0) vhi = frag_read_v(IntC<1>{}, bufc, IntC<s - 12>{}, IntC<1>{});
        else vhi = frag_read_v(IntC<0>{}, IntC<(BUF + 1) % RING>{}, IntC<s - 12>{}, IntC<1>{});
        ring[s & 3] = __builtin_bit_cast(u32x4, __builtin_shufflevector(vlo, vhi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
      if constexpr (!(W4_ABL & 1)) {
        if constexpr (W4_ABL & 128) eb[s] = Sc[bsm][2 * c + 1];
        else eb[s] = exp2_op(Sc[bsm][2 * c + 1]);
        if constexpr (W4_ABL & 256) asm volatile("" :: "v"(eb[s]));
        else if constexpr (s > 0) add_op(lsb[bsm ^ 1], eb[s - 1]);
      }
      if constexpr (PAR == 0 && s == 11 && !(W4_ABL & 8)) {
        // K(t+1) and V(t) (issued three tiles ago) have landed; after the barrier that holds for
        // every wave.  This wave's last reads of K(t) and V(t-1) were issued before this point but
        // are not waited for: an issued LDS read completes within ~10^2 cycles whatever its wave
        // does next, while the first DMA into those slots is issued 8 steps (>= 512 cycles) after the
        // barrier and lands an L2 latency later still.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * DMA_PER_GROUP) & 15) | (((2 * DMA_PER_GROUP) >> 4) << 14));  // vmcnt(16)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- overflow check of half tile k, before any P(k) is used.  Every exp is positive, so a
    // lane whose 16 exps sum to at most 2^kRescaleThr holds no P above that bound: no row maximum
    // is tracked in the loop, the rare path forms it.  (The exps above must not sink below the branch.)
    float t0 = lsa[0] + lsb[0], t1 = lsa[1] + lsb[1];
    if constexpr (!(W4_ABL & (1 | 256))) {  // consumers of the last step's exps
      t1 += ea[15] + eb[15];
      Pc[1][1][3] = pack2(ea[15], eb[15]);
    }
    asm volatile("" : "+v"(Pc[0][0]), "+v"(Pc[0][1]), "+v"(t0));
    asm volatile("" : "+v"(Pc[1][0]), "+v"(Pc[1][1]), "+v"(t1));
    const bool first = PAR == 0 && t == 0;  // half tile 0: m becomes its true row maximum
    if (!(W4_ABL & 16) && __builtin_expect(first || __any(fmaxf(t0, t1) > kRescaleSum), 0)) {
      fixup(IntC<0>{}, Sc[0], Sn[0], Pc[0], first, t0);
      fixup(IntC<1>{}, Sc[1], Sn[1], Pc[1], first, t1);
    }
    l_run[0] += t0;
    l_run[1] += t1;
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

  round_leave(rs);

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
  return launch_with_rounds(attn_fwd_w4_kernel<true>, attn_fwd_w4_kernel<false>, W4_THREADS, p, (int64_t)p.nqb * p.heads, st, "attn_fwd(w4)");
}
