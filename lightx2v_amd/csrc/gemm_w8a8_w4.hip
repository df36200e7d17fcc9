// W8A8 Linear (fp8 e4m3 or int8 operands, fp32 / int32 accumulation, bf16 out) on MFMA, persistent,
// one wave per SIMD:  C[M,N] = epilogue((A8[M,K] @ W8[N,K]^T) * a_scale[m] * w_scale[n] + bias).
//
// The design of the bf16 kernel in gemm_bf16_w4.hip on one-byte operands.  Workgroup = 4 waves
// (2 x 2), one per SIMD, each owning a 128 x 128 corner of a 256 x 256 output tile (64 accumulator
// tuples, 60 of them in the accumulator half of the register file); one workgroup per CU walks its
// share of the output tiles, so the stream of K-tiles never drains between tiles.  A K-tile is 128
// bytes of K per row -- the same 32 KiB per operand as the bf16 kernel's 64 elements -- and ONE
// v_mfma_f32_16x16x128_f8f6f4 per accumulator tuple covers its whole depth (int8: two
// v_mfma_i32_16x16x64_i8 on the first and second 16 bytes of the lane's 32): half as many LDS-DMA
// pieces, fragment reads, waits and barriers per FLOP as the bf16 kernel pays.
//
// LDS: a ring of five 32 KiB slots; half-tile h = 2 j + o (o = 0: A of K-tile j, o = 1: W of K-tile
// j) goes through slot h % 5.  Registers: the 8 A fragments and the 8 W fragments of one K-tile
// (8 registers each, 128 in all, as many as the bf16 kernel's two K-steps).  Per K-tile j:
//   first half   W fragments 0-3 (the MFMA's A operand: output columns) against all 8 A fragments,
//                W-major.  Reads W[2..7] of K-tile j -- every read of W(j) precedes X.  The 8 DMA
//                pieces of A(j + 2) go to the slot W(j - 1) left at the previous barrier.
//   X            s_waitcnt vmcnt(8) lgkmcnt(0); s_barrier: K-tile j + 1 is complete in LDS.
//   second half  W fragments 4-7, A-major: A fragment mt meets its last MFMA early, and A[mt] of
//                K-tile j + 1 is read into its registers right behind that MFMA, so the A reads are
//                spread over the half; W[0], W[1] of K-tile j + 1 go to the registers the first half
//                left.  The 8 DMA pieces of W(j + 2) go to the slot of A(j), whose reads all precede X.
// One instruction class besides the MFMA per MFMA shadow (M0 write, DMA instruction, one fragment's
// two reads).
// The slot images and the staging are the bf16 kernel's (128-byte rows, chunk ^= (row >> 1) & 7,
// raw buffer descriptors: rows past M / N arrive as zeros).
//
// Epilogue as in the bf16 kernel (eight passes of 16 rows through the idle W(j) slot as fp32 / int32,
// eight consecutive columns per lane, 256-byte row segments out), with the two scales applied first:
// v = acc * a_scale[m] * w_scale[n] + bias, the arithmetic of the 8-wave kernels in gemm_fp8.hip, so
// results are bit-identical to theirs.
#include <algorithm>

#include "gemm_fp8_common.h"

namespace {

#ifndef W8_ABL
#define W8_ABL 0  // timing ablations (wrong results): 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no wait + barrier per K-tile, 8 no epilogue, 16 no MFMAs
#endif
constexpr int W4_THREADS = 256;
constexpr int EPI_PARTIAL = 3;  // kernel instance that leaves the raw accumulators of a K range (internal)
constexpr int NSLOT = 5;
constexpr int W4_LDS = NSLOT * TILE_BYTES;  // all 160 KiB of the CU
constexpr int W4_GROUP_M = 4;               // row panels per column sweep (the bf16 kernel's measured choice)

template <int N>
struct IntC {
  static constexpr int value = N;
};
template <int B, int E, class F>
LX2V_DEVICE void static_for(F&& f) {
  if constexpr (B < E) {
    f(IntC<B>{});
    static_for<B + 1, E>(f);
  }
}

LX2V_DEVICE void tile_origin_w4(const GemmParams& p, int vbid, int& m0, int& n0) {
  const int nwg = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(vbid, nwg);
  const int group_sz = W4_GROUP_M * p.tiles_n;
  const int g = t / group_sz;
  const int first_m = g * W4_GROUP_M;
  const int gm = min(p.tiles_m - first_m, W4_GROUP_M);
  const int in_g = t - g * group_sz;
  m0 = (first_m + in_g % gm) * BM;
  n0 = (in_g / gm) * BN;
}

constexpr bool acc_in_agpr(int nt, int mt) { return !(nt == 7 && mt >= 6); }  // 62 of the 64 tuples (the compiler keeps a few accumulator registers to itself)

// acc (+)= W fragment (MFMA A operand: output columns) x A fragment (B operand: rows).
// FIRST: the K-tile that opens an output tile starts its accumulators from 0.
// int8: HALF selects the first / second 16 bytes of both fragments (the two calls of a tuple are
// eight MFMAs apart: the second reads the first one's result).
// A fragment = the lane's 32 K bytes as two 16-byte halves (two LDS reads, issued in two MFMA shadows).
struct Frag8 {
  i32x4 lo, hi;
};
template <bool INT8, bool IN_AGPR, bool FIRST, int HALF>
LX2V_DEVICE void mfma_acc(f32x4& acc, const Frag8& wfr, const Frag8& afr) {
  if constexpr (W8_ABL & 16) {
    asm volatile("" ::"v"(wfr.lo), "v"(wfr.hi), "v"(afr.lo), "v"(afr.hi));
  } else if constexpr (!INT8) {
    const i32x8 wf = __builtin_shufflevector(wfr.lo, wfr.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8 af = __builtin_shufflevector(afr.lo, afr.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    if constexpr (FIRST) {
      if constexpr (IN_AGPR) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0" : "=a"(acc) : "v"(wf), "v"(af));
      else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0" : "=&v"(acc) : "v"(wf), "v"(af));
    } else {
      if constexpr (IN_AGPR) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc) : "v"(wf), "v"(af));
      else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(acc) : "v"(wf), "v"(af));
    }
  } else {
    const i32x4 w4 = HALF == 0 ? wfr.lo : wfr.hi;
    const i32x4 a4 = HALF == 0 ? afr.lo : afr.hi;
    if constexpr (FIRST && HALF == 0) {
      if constexpr (IN_AGPR) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=a"(acc) : "v"(w4), "v"(a4));
      else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=&v"(acc) : "v"(w4), "v"(a4));
    } else {
      if constexpr (IN_AGPR) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(w4), "v"(a4));
      else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(w4), "v"(a4));
    }
  }
}
template <bool IN_AGPR>
LX2V_DEVICE void acc_to_lds(int lds_addr, const f32x4& acc) {
  if constexpr (IN_AGPR) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "a"(acc) : "memory");
  else asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(acc) : "memory");
}

// ... or of a global store at `sbase` (wave-uniform) + `voff` bytes.
template <bool IN_AGPR>
LX2V_DEVICE void acc_to_global(int voff, const f32x4& acc, uint64_t sbase) {
  if constexpr (IN_AGPR) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "a"(acc), "s"(sbase) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(acc), "s"(sbase) : "memory");
}

LX2V_DEVICE u32x4 raw_descriptor(const void* base, int64_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  u32x4 d;
  d[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));  // stride 0: raw buffer
  d[2] = __builtin_amdgcn_readfirstlane((uint32_t)(bytes < 0 ? 0 : (bytes > 0xffffffffLL ? 0xffffffffLL : bytes)));  // loads at or past it return 0
  d[3] = 0x00020000u;
  return d;
}

template <int EPI, bool INT8>
__global__ __launch_bounds__(W4_THREADS) void gemm_w8a8_w4_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char lds[W4_LDS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;

  // EPI == EPI_PARTIAL: one workgroup per (tail tile, K range); everything else is the same stream of one item.
  constexpr bool PARTIAL = EPI == EPI_PARTIAL;
  const int ntiles = p.ntiles_run;
  const int grid = gridDim.x;
  int nk = p.K / BK;
  int part_vbid = 0;    // virtual block id of the partial item's tile
  int64_t k_first = 0;  // first K byte of the item
  if constexpr (PARTIAL) {
    const int item = blockIdx.x, part = item % p.tail_split;
    part_vbid = p.tail_first + item / p.tail_split;
    k_first = (int64_t)part * p.tail_nk * BK;
    nk = min(p.tail_nk, nk - part * p.tail_nk);
  }

  // ---- staging: DMA instruction q (0..7) of wave w fills rows 64 w + 8 q .. + 7 of the tile image
  // (1 KiB of LDS, lane L the 16 bytes at 16 L: row 64 w + 8 q + (L >> 3), position L & 7 holding
  // source chunk (L & 7) ^ ((row >> 1) & 7)); four pieces share one M0 through the instruction's
  // 12-bit offset field, which is added to the LDS address and to the memory address alike.
  auto piece_voff = [&](int q, int64_t ld) __attribute__((always_inline)) -> int {
    const int row = 64 * wid + 8 * q + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    return (int)(row * ld) + chunk * 16 - (q & 3) * 1024;  // the offset field adds (q & 3) KiB back
  };
  int a_pv[8], w_pv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    a_pv[q] = piece_voff(q, p.lda);
    w_pv[q] = piece_voff(q, p.ldw);
    asm volatile("" : "+v"(a_pv[q]), "+v"(w_pv[q]));
  }
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds);
  const uint32_t dma_base = lds_base + 8192 * (uint32_t)wid;

  // DMA cursor: the K-tile staged next, two K-tiles ahead of the one computed.
  int d_ord = 0;  // ordinal of the cursor's output tile in this workgroup's list
  int d_kt = 0;
  u32x4 a_rs, w_rs;
  // Descriptors of the cursor's tile.  Past the workgroup's last tile they are empty (0 bytes): the
  // DMA instructions of the stream's last two K-tiles fetch nothing and write zeros into a slot
  // nobody reads, so the loop body needs no "last tiles" variants.
  auto d_panel = [&]() __attribute__((always_inline)) {
    const int vbid = PARTIAL ? part_vbid : d_ord * grid + (int)blockIdx.x;
    const bool live = PARTIAL ? d_ord == 0 : vbid < ntiles;
    int m0 = 0, n0 = 0;
    if (live) tile_origin_w4(p, vbid, m0, n0);
    a_rs = raw_descriptor(p.A + k_first + (int64_t)m0 * p.lda, live ? ((int64_t)p.M - m0) * p.lda - k_first : 0);
    w_rs = raw_descriptor(p.W + k_first + (int64_t)n0 * p.ldw, live ? ((int64_t)p.N - n0) * p.ldw - k_first : 0);
  };
  d_panel();
  auto dma_m0 = [&](uint32_t slot_base, auto dstc) __attribute__((always_inline)) {
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(slot_base), "i"(decltype(dstc)::value) : "memory", "scc");
  };
  auto dma_go = [&](const u32x4& rsrc, int voff, uint32_t soff, auto ofsc) __attribute__((always_inline)) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff), "s"(rsrc), "s"(soff), "n"(decltype(ofsc)::value) : "memory");
  };
  uint32_t d_soff = 0;  // byte offset of the cursor's K-tile inside a row (same for A and W)
  auto d_begin = [&]() __attribute__((always_inline)) { d_soff = (uint32_t)d_kt * BK; };
  auto dma_piece_m0 = [&](uint32_t slot_base, auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    if constexpr ((q & 3) == 0) dma_m0(slot_base, IntC<1024 * q>{});
  };
  auto dma_piece_go = [&](auto halfc, auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    if constexpr (decltype(halfc)::value == 0) dma_go(a_rs, a_pv[q], d_soff, IntC<1024 * (q & 3)>{});
    else dma_go(w_rs, w_pv[q], d_soff, IntC<1024 * (q & 3)>{});
  };
  auto dma_piece = [&](uint32_t slot_base, auto halfc, auto qc) __attribute__((always_inline)) {  // prologue only
    dma_piece_m0(slot_base, qc);
    asm volatile("s_nop 0");
    dma_piece_go(halfc, qc);
  };
  auto d_advance = [&]() __attribute__((always_inline)) {  // after the W half of a K-tile
    ++d_kt;
    if (d_kt == nk) {
      d_kt = 0;
      ++d_ord;
      d_panel();
    }
    // (the descriptors are merged here from two paths; pinned to scalar registers, or the compiler may move the
    // merge to vector registers, which the DMA statements cannot take)
    asm volatile("" : "+s"(a_rs), "+s"(w_rs));
  };

  // ---- fragment read offsets inside a slot image: the lane's 32 K bytes are the two adjacent
  // 16-byte chunks 2 fq, 2 fq + 1 of row (corner row + fr); mt / nt add 2048 each ----
  int rd_a[2], rd_w[2];
  {
    const int arow = wm * 128 + fr, wrow = wn * 128 + fr;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      rd_a[c] = arow * 128 + (((2 * fq + c) ^ ((arow >> 1) & 7)) << 4);
      rd_w[c] = wrow * 128 + (((2 * fq + c) ^ ((wrow >> 1) & 7)) << 4);
    }
  }
  auto read_half = [&](const char* base, const int (&rd)[2], int i, int c) __attribute__((always_inline)) -> i32x4 {
    return *reinterpret_cast<const i32x4*>(base + rd[c] + i * 2048);
  };

  f32x4 acc[8][8];  // acc[nt][mt][j] = C'[m = wm*128 + mt*16 + fr][n = wn*128 + nt*16 + fq*4 + j] (int8: the int32 bit patterns)
  Frag8 fa[8];      // A fragments of the current K-tile (second half: replaced one by one by the next K-tile's)
  Frag8 fw[8];      // W fragments of the current K-tile (second half: fw[0], fw[1] already the next K-tile's)

  // One K-tile.  sa..sfree: byte offsets of the slots A(j), W(j), A(j+1), W(j+1) and the free one.
  // Extras ride in the shadow of the MFMA they follow: u = MFMA position inside the half (int8: of the
  // first MFMA of the pair).
  auto ktile = [&](auto FIRSTc, int sa, int sw, int sa1, int sw1, int sfree, bool relaxed) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(FIRSTc)::value != 0;
    const char* lw = lds + sw;
    const char* la1 = lds + sa1;
    const char* lw1 = lds + sw1;
    const uint32_t dfree = dma_base + (uint32_t)sfree, dsa = dma_base + (uint32_t)sa;
    constexpr bool DMA = !(W8_ABL & 1), RD = !(W8_ABL & 2);
    // DMA piece q of a half: M0 (pieces 0 and 4) in shadow 4 q, the instruction in shadow 4 q + 1
    auto dma_extra = [&](uint32_t dslot, auto halfc, auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if constexpr (DMA && (u & 3) == 0) dma_piece_m0(dslot, IntC<u / 4>{});
      if constexpr (DMA && (u & 3) == 1) dma_piece_go(halfc, IntC<u / 4>{});
    };
    // ---- first half: W-major; the two reads of W[2 + i] of this K-tile in shadows 4 i + 2, 4 i + 3; DMA of A(j + 2) into the free slot ----
    d_begin();
    static_for<0, 4>([&](auto ntc) __attribute__((always_inline)) {
      constexpr int nt = decltype(ntc)::value;
      static_for<0, 8>([&](auto mc) __attribute__((always_inline)) {
        constexpr int mt = decltype(mc)::value, u = 8 * nt + mt;
        mfma_acc<INT8, acc_in_agpr(nt, mt), FIRST, 0>(acc[nt][mt], fw[nt], fa[mt]);
        dma_extra(dfree, IntC<0>{}, IntC<u>{});
        if constexpr (RD && (u & 3) == 2 && u / 4 < 6) fw[2 + u / 4].lo = read_half(lw, rd_w, 2 + u / 4, 0);
        if constexpr (RD && (u & 3) == 3 && u / 4 < 6) fw[2 + u / 4].hi = read_half(lw, rd_w, 2 + u / 4, 1);
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (INT8) {
        static_for<0, 8>([&](auto mc) __attribute__((always_inline)) {
          constexpr int mt = decltype(mc)::value;
          mfma_acc<INT8, acc_in_agpr(nt, mt), FIRST, 1>(acc[nt][mt], fw[nt], fa[mt]);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    });
    // ---- X: this wave's DMA except the 8 pieces just issued has landed, its reads of W(j) are done;
    // behind the barrier that holds for every wave (K-tile j + 1 complete, W(j) and A(j) released).
    // `relaxed`: the previous K-tile ended in an epilogue, whose 32 stores (issued for every row and column of the
    // tile; those outside C are dropped by the range check of their descriptor) are younger than the data waited
    // for: 40 may stay outstanding.
    if constexpr (!(W8_ABL & 4)) {
      if (relaxed) asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // ---- second half: A-major over W[4..7]; DMA of W(j + 2) into the slot of A(j).  A[mt] of K-tile j + 1 goes out
    // behind the last MFMA of A[mt]: first half in shadow 4 mt + 3, second half in shadow 4 mt + 6 (one LDS read
    // fills an MFMA shadow; two in one shadow hold the next MFMA back); W[0], W[1] of K-tile j + 1 in shadows 4, 8, 12, 20 ----
    auto second_extra = [&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      dma_extra(dsa, IntC<1>{}, uc);
      if constexpr (RD) {
        if constexpr ((u & 3) == 3) fa[u / 4].lo = read_half(la1, rd_a, u / 4, 0);
        if constexpr ((u & 3) == 2 && u >= 6) fa[u / 4 - 1].hi = read_half(la1, rd_a, u / 4 - 1, 1);
        if constexpr (u == 4) fw[0].lo = read_half(lw1, rd_w, 0, 0);
        if constexpr (u == 8) fw[0].hi = read_half(lw1, rd_w, 0, 1);
        if constexpr (u == 12) fw[1].lo = read_half(lw1, rd_w, 1, 0);
        if constexpr (u == 20) fw[1].hi = read_half(lw1, rd_w, 1, 1);
      }
    };
    static_for<0, 8>([&](auto mc) __attribute__((always_inline)) {
      constexpr int mt = decltype(mc)::value;
      if constexpr (!INT8) {
        static_for<4, 8>([&](auto ntc) __attribute__((always_inline)) {
          constexpr int nt = decltype(ntc)::value;
          mfma_acc<INT8, acc_in_agpr(nt, mt), FIRST, 0>(acc[nt][mt], fw[nt], fa[mt]);
          second_extra(IntC<4 * mt + (nt - 4)>{});
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        // int8: the pair of MFMAs of a tuple four MFMAs apart; the extras ride behind the second pass, whose last
        // MFMA is the last use of A[mt] (extras of the odd positions behind the first pass: one per shadow)
        static_for<4, 8>([&](auto ntc) __attribute__((always_inline)) {
          constexpr int nt = decltype(ntc)::value;
          mfma_acc<INT8, acc_in_agpr(nt, mt), FIRST, 0>(acc[nt][mt], fw[nt], fa[mt]);
          if constexpr (nt == 4 || nt == 5) second_extra(IntC<4 * mt + (nt - 4)>{});   // M0 / DMA / W' reads: nothing of A[mt]
          __builtin_amdgcn_sched_barrier(0);
        });
        static_for<4, 8>([&](auto ntc) __attribute__((always_inline)) {
          constexpr int nt = decltype(ntc)::value;
          mfma_acc<INT8, acc_in_agpr(nt, mt), FIRST, 1>(acc[nt][mt], fw[nt], fa[mt]);
          if constexpr (nt == 6 || nt == 7) second_extra(IntC<4 * mt + (nt - 4)>{});   // position 4 mt + 2: hi of A[mt - 1]; 4 mt + 3: lo of A[mt], behind its last use
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    });
    if constexpr (RD) fa[7].hi = read_half(la1, rd_a, 7, 1);
    d_advance();
  };

  // ---- epilogue of the output tile at (m0, n0): eight passes, one per 16-row block mt ----
  auto epilogue = [&](int m0, int n0, int epi_base) __attribute__((always_inline)) {
    int fr = lane & 15, fq = lane >> 4;
    asm volatile("" : "+v"(fr), "+v"(fq));  // (opaque: nothing of this is hoisted in front of the K-loop)
    const int stg = (int)(size_t)(__attribute__((address_space(3))) char*)lds + epi_base + wid * 8192;
    int waddr[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) waddr[nt] = stg + fr * 512 + (((nt * 4 + fq) ^ fr) << 4);
    const int gn = n0 + wn * 128 + fr * 8;  // this lane's eight output columns
    const bool col_ok = gn < p.N;           // N % 8 == 0: all eight or none
    // Stores go through a raw buffer descriptor of the tile's row panel: rows past M lie beyond its range and lanes
    // whose columns lie past N get an offset beyond it, and the hardware drops such stores -- no branch around a
    // store, so every pass issues the same memory operations and the waits the compiler places between them
    // (memory operations retire in order) are exact counts instead of worst cases over skipped stores.
    const __amdgpu_buffer_rsrc_t c_rs = [&]() __attribute__((always_inline)) {
      const uint64_t a = reinterpret_cast<uint64_t>(p.C + (int64_t)m0 * p.ldc);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));  // (the builtin returns int)
      void* pb = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
      const int rows = min(p.M - m0, 256);  // the panel's rows inside C (the launcher keeps 256 rows of C below 2 GiB)
      return __builtin_amdgcn_make_buffer_rsrc(pb, 0, __builtin_amdgcn_readfirstlane(rows * p.ldc * 2), 0x00020000);
    }();
    const int c_col = gn * 2;  // byte offset of the lane's columns inside a row
    constexpr int C_NOWHERE = (int)0x80000000u;  // an offset at or past the end of any such range
    float bv[8], gv[8], swv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bv[j] = 0.f;
      gv[j] = 1.f;
      swv[j] = 0.f;
    }
    if (col_ok) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.w_scale + gn), s1 = *reinterpret_cast<const f32x4*>(p.w_scale + gn + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { swv[j] = s0[j]; swv[4 + j] = s1[j]; }
      if (p.bias != nullptr) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + gn), b1 = *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { bv[j] = b0[j]; bv[4 + j] = b1[j]; }
      }
      if (EPI == LX2V_EPI_RESIDUAL && p.gate != nullptr) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gate + gn), g1 = *reinterpret_cast<const f32x4*>(p.gate + gn + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { gv[j] = g0[j]; gv[4 + j] = g1[j]; }
      }
    }
    // the last MFMAs were issued a few cycles ago and the LDS writes below read their results
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(bv[j]), "+v"(gv[j]), "+v"(swv[j]));  // their loads are waited for here, once
    // The 128 row scales of this wave's corner: two per lane (rows lane and 64 + lane), fetched per pass
    // with a lane permute -- no memory operation per pass, so no pass waits behind the stores before it.
    float sa_lane[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) sa_lane[h] = p.a_scale[min(m0 + wm * 128 + 64 * h + lane, p.M - 1)];
#pragma unroll
    for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(sa_lane[h]));
    // Residual rows are loaded TWO passes ahead (clamped addresses, no branch: the only conditional operation of a
    // pass is the store).  Memory operations retire in order, so a pass that waits for its residual rows also waits
    // for every store issued before their loads; one pass ahead that was the previous pass's stores, a full write
    // latency per pass.  The ring of three passes (four 16-byte rows each) lives in W fragment variables that are
    // dead between K-tiles (fw[2..7]); what they hold when the epilogue ends is complete, consumed data.
    auto load_rows = [&](auto mtc) __attribute__((always_inline)) {
      constexpr int mt = decltype(mtc)::value;
      if constexpr (EPI == LX2V_EPI_RESIDUAL && mt < 8) {
        i32x4 r[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int gm = min(m0 + wm * 128 + mt * 16 + it * 4 + fq, p.M - 1);
          r[it] = *reinterpret_cast<const i32x4*>(p.resid + (int64_t)gm * p.ldr + (col_ok ? gn : 0));
        }
        Frag8 &d0 = fw[2 + 2 * (mt % 3)], &d1 = fw[3 + 2 * (mt % 3)];
        d0.lo = r[0];
        d0.hi = r[1];
        d1.lo = r[2];
        d1.hi = r[3];
      }
    };
    load_rows(IntC<0>{});
    load_rows(IntC<1>{});
    static_for<0, 8>([&](auto mtc) __attribute__((always_inline)) {
      constexpr int mt = decltype(mtc)::value;
      load_rows(IntC<mt + 2>{});
      static_for<0, 8>([&](auto ntc) __attribute__((always_inline)) {
        constexpr int nt = decltype(ntc)::value;
        acc_to_lds<acc_in_agpr(nt, mt)>(waddr[nt], acc[nt][mt]);
      });
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 4 + fq;
        const int gm = m0 + wm * 128 + mt * 16 + r;
        const char* row = lds + epi_base + wid * 8192 + r * 512;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(row + (((2 * fr) ^ r) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(row + (((2 * fr + 1) ^ r) << 4));
        float v[8];
        const i32x4 lo_i = __builtin_bit_cast(i32x4, lo), hi_i = __builtin_bit_cast(i32x4, hi);  // int8: the exact int32 sums
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = INT8 ? (float)lo_i[j] : lo[j];
          v[4 + j] = INT8 ? (float)hi_i[j] : hi[j];
        }
        const float sa = __shfl(sa_lane[mt >> 2], (mt & 3) * 16 + r, 64);  // row mt * 16 + r of the corner
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = dequant_bias(v[j], sa, swv[j], bv[j]);
          if (EPI == LX2V_EPI_GELU_TANH) v[j] = gelu_tanh(v[j]);
          if (EPI == LX2V_EPI_RESIDUAL) v[j] *= gv[j];
        }
        if (EPI == LX2V_EPI_RESIDUAL) {
          float y[8];
          const Frag8& rr = fw[2 + 2 * (mt % 3) + (it >> 1)];
          const i32x4 rh = (it & 1) ? rr.hi : rr.lo;
          unpack8(__builtin_bit_cast(u32x4, rh), y);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (float)(__bf16)v[j] + y[j];  // the product is rounded to bf16 before the add, as in the 8-wave kernels
        }
        const u32x4 out = pack8(v);
        __builtin_amdgcn_raw_buffer_store_b128(out, c_rs, col_ok ? (int)((int64_t)(gm - m0) * p.ldc * 2) + c_col : C_NOWHERE, 0, 0);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- a partial item's result: the raw accumulators, tuple by tuple in fragment order (1 KiB per wave and tuple) ----
  auto store_partial = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMAs were issued a few cycles ago
    const float* base = p.partials + ((int64_t)blockIdx.x * 4 + wid) * (64 * 64 * 4);
    const uint64_t bu = reinterpret_cast<uint64_t>(base);
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)bu), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(bu >> 32));
    const uint64_t sbase = ((uint64_t)bhi << 32) | blo;
    static_for<0, 8>([&](auto ntc) __attribute__((always_inline)) {
      static_for<0, 8>([&](auto mtc) __attribute__((always_inline)) {
        constexpr int nt = decltype(ntc)::value, mt = decltype(mtc)::value;
        acc_to_global<acc_in_agpr(nt, mt)>(lane * 16 + (nt * 8 + mt) * 1024, acc[nt][mt], sbase);
      });
    });
  };

  // ---- prologue: half-tiles 0..3 (K-tiles 0 and 1 of the stream), then A[0..7], W[0], W[1] of K-tile 0 ----
  const int my_tiles = PARTIAL ? 1 : (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / grid + 1 : 0;
  const int64_t total = (int64_t)my_tiles * nk;  // K-tiles in this workgroup's stream
  if (total == 0) return;
  static_for<0, 2>([&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    d_begin();
    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) { dma_piece(dma_base + (2 * t) * TILE_BYTES, IntC<0>{}, qc); });
    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) { dma_piece(dma_base + (2 * t + 1) * TILE_BYTES, IntC<1>{}, qc); });
    d_advance();
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[i].lo = read_half(lds, rd_a, i, 0);
    fa[i].hi = read_half(lds, rd_a, i, 1);
    fw[i].lo = fw[i].hi = i32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    fw[i].lo = read_half(lds + TILE_BYTES, rd_w, i, 0);
    fw[i].hi = read_half(lds + TILE_BYTES, rd_w, i, 1);
  }

  // ---- the stream ----
  int ord = 0, kt = 0;   // output tile ordinal and K-tile of stream position j
  bool relaxed = false;  // the previous K-tile ended in an epilogue (32 stores per wave)
  int r5 = 0;            // (2 j) % 5: slot of A(j); W(j) r5 + 1, A(j+1) r5 + 2, W(j+1) r5 + 3, free r5 + 4 (all mod 5)
  auto slot = [&](int k) __attribute__((always_inline)) -> int {
    const int s5 = r5 + k;
    return (s5 >= NSLOT ? s5 - NSLOT : s5) * TILE_BYTES;
  };
  for (int64_t j = 0; j < total; ++j) {
    const int sa = slot(0), sw = slot(1), sa1 = slot(2), sw1 = slot(3), sfree = slot(4);
    if (kt == 0) ktile(IntC<1>{}, sa, sw, sa1, sw1, sfree, relaxed);
    else ktile(IntC<0>{}, sa, sw, sa1, sw1, sfree, false);
    relaxed = false;
    if (++kt == nk) {
      if constexpr (PARTIAL) {
        store_partial();
        break;
      }
      int m0, n0;
      tile_origin_w4(p, ord * grid + (int)blockIdx.x, m0, n0);
      relaxed = !(W8_ABL & 8);
      if constexpr (!(W8_ABL & 8)) {
        epilogue(m0, n0, sw);  // W(j)'s slot is idle until A(j + 3) is staged into it, in the next first half ...
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... by every wave into every wave's share of the slot: all must have left the epilogue
      }
      // The fragment registers that are dead here (fw[2..7]) stay occupied across the epilogue: handed to its
      // loads and stores, they would be overwritten by the next K-tile's LDS reads while those memory operations
      // still count as pending, and the compiler would protect them with a vmcnt wait inside the K-loop.
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(fa[i].lo), "+v"(fa[i].hi), "+v"(fw[i].lo), "+v"(fw[i].hi));
      kt = 0;
      ++ord;
    }
    r5 += 2;
    if (r5 >= NSLOT) r5 -= NSLOT;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may still be writing this workgroup's LDS when it ends
}

// Tail tiles of a launch whose last round was split along K: one thread adds the parts of 8 consecutive outputs of a
// row in a fixed order (int8: exact int32 sums, so the result equals the unsplit kernel's bit for bit; fp8: fp32 sums
// in another order) and applies scales and epilogue with the persistent kernel's arithmetic.
template <int EPI, bool INT8>
__global__ __launch_bounds__(256) void gemm_w8a8_tail_fixup_kernel(const GemmParams p) {
  const int t = blockIdx.x >> 5;  // tail tile
  const int e = ((blockIdx.x & 31) << 8) + threadIdx.x;
  const int r = e & 255, c = (e >> 8) * 8;  // consecutive threads: consecutive rows (consecutive lanes of a fragment)
  int m0, n0;
  tile_origin_w4(p, p.tail_first + t, m0, n0);
  const int gm = m0 + r, gn = n0 + c;
  if (gm >= p.M || gn >= p.N) return;
  const int wave = (r >> 7) * 2 + (c >> 7), mt = (r & 127) >> 4, fr = r & 15, nt = (c & 127) >> 4, fq0 = (c & 15) >> 2;
  const float* src = p.partials + ((((int64_t)t * p.tail_split * 4 + wave) * 64 + nt * 8 + mt) * 64 + fq0 * 16 + fr) * 4;
  const int64_t part_stride = (int64_t)4 * 64 * 64 * 4;  // floats per item
  float v[8];
  if constexpr (INT8) {
    int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  #pragma unroll 4
  for (int part = 0; part < p.tail_split; ++part) {
      const i32x4 lo = *reinterpret_cast<const i32x4*>(src + part * part_stride);
      const i32x4 hi = *reinterpret_cast<const i32x4*>(src + part * part_stride + 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[j] += lo[j]; acc[4 + j] += hi[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)acc[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  #pragma unroll 4
  for (int part = 0; part < p.tail_split; ++part) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(src + part * part_stride);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(src + part * part_stride + 64);  // the next column group: 16 lanes on
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += lo[j]; v[4 + j] += hi[j]; }
    }
  }
  const float sa = p.a_scale[gm];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float bv = p.bias != nullptr ? p.bias[gn + j] : 0.f;
    const float gv = (EPI == LX2V_EPI_RESIDUAL && p.gate != nullptr) ? p.gate[gn + j] : 1.f;
    v[j] = dequant_bias(v[j], sa, p.w_scale[gn + j], bv);
    if (EPI == LX2V_EPI_GELU_TANH) v[j] = gelu_tanh(v[j]);
    if (EPI == LX2V_EPI_RESIDUAL) v[j] *= gv;
  }
  if (EPI == LX2V_EPI_RESIDUAL) {
    float y[8];
    unpack8(*reinterpret_cast<const u32x4*>(p.resid + (int64_t)gm * p.ldr + gn), y);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)(__bf16)v[j] + y[j];
  }
  *reinterpret_cast<u32x4*>(p.C + (int64_t)gm * p.ldc + gn) = pack8(v);
}

struct TailPlan {
  int rem, split, nk_part;
  int64_t bytes;
};

// Is the last round of one tile per CU worth splitting along K?  Only when at most a quarter of the CUs would
// have a tile in it and K is long enough for two more launches to pay.
TailPlan tail_plan(const GemmParams& p, int cus) {
  TailPlan t{0, 1, 0, 0};
  const int64_t ntiles = (int64_t)p.tiles_m * p.tiles_n;
  const int nk = p.K / BK;
  if (ntiles <= cus || nk < 64) return t;  // (K = 5120: a round is 68 us long and the split breaks even, measured)
  const int rem = (int)(ntiles % cus);
  if (rem == 0 || rem * 4 > cus) return t;
  int split = std::min(cus / rem, 16);
  const int per = (nk + split - 1) / split;
  split = (nk + per - 1) / per;  // as many parts as are not empty
  t = TailPlan{rem, split, per, (int64_t)rem * split * 4 * 64 * 64 * 4 * (int64_t)sizeof(float)};
  return t;
}

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

template <bool INT8>
void launch_all(GemmParams& p, int epilogue, const TailPlan& tp, int cus, hipStream_t st) {
  const dim3 grid((unsigned)(p.ntiles_run < cus ? p.ntiles_run : cus)), block(W4_THREADS);
  switch (epilogue) {
    case LX2V_EPI_NONE: hipLaunchKernelGGL((gemm_w8a8_w4_kernel<LX2V_EPI_NONE, INT8>), grid, block, 0, st, p); break;
    case LX2V_EPI_GELU_TANH: hipLaunchKernelGGL((gemm_w8a8_w4_kernel<LX2V_EPI_GELU_TANH, INT8>), grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL((gemm_w8a8_w4_kernel<LX2V_EPI_RESIDUAL, INT8>), grid, block, 0, st, p); break;
  }
  if (tp.rem == 0) return;
  hipLaunchKernelGGL((gemm_w8a8_w4_kernel<EPI_PARTIAL, INT8>), dim3((unsigned)(tp.rem * tp.split)), block, 0, st, p);
  const dim3 fgrid((unsigned)(tp.rem * 32));
  switch (epilogue) {
    case LX2V_EPI_NONE: hipLaunchKernelGGL((gemm_w8a8_tail_fixup_kernel<LX2V_EPI_NONE, INT8>), fgrid, dim3(256), 0, st, p); break;
    case LX2V_EPI_GELU_TANH: hipLaunchKernelGGL((gemm_w8a8_tail_fixup_kernel<LX2V_EPI_GELU_TANH, INT8>), fgrid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((gemm_w8a8_tail_fixup_kernel<LX2V_EPI_RESIDUAL, INT8>), fgrid, dim3(256), 0, st, p); break;
  }
}

}  // namespace

int64_t lx2v_gemm_w8a8_w4_workspace_bytes(const lx2v_w8a8::GemmParams& p) { return tail_plan(p, device_cus()).bytes; }

// One workgroup per CU (the 160 KiB of LDS allow no second one), never more than there are tiles.  With a
// workspace, a sparsely filled last round becomes a K split of its tiles (see GemmParams).
int lx2v_gemm_w8a8_launch_w4(lx2v_w8a8::GemmParams& p, bool int8, int epilogue, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  const int cus = device_cus();
  const int64_t ntiles = (int64_t)p.tiles_m * p.tiles_n;
  TailPlan tp = tail_plan(p, cus);
  if (tp.bytes == 0 || workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 15u) != 0 || workspace_bytes < tp.bytes) tp = TailPlan{0, 1, 0, 0};
  p.ntiles_run = (int)(ntiles - tp.rem);
  p.tail_first = p.ntiles_run;
  p.tail_split = tp.split;
  p.tail_nk = tp.nk_part;
  p.partials = reinterpret_cast<float*>(workspace);
  if (int8) launch_all<true>(p, epilogue, tp, cus, st);
  else launch_all<false>(p, epilogue, tp, cus, st);
  return LX2V_OK;
}
