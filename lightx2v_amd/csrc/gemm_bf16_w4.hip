// bf16 Linear on MFMA, persistent, one wave per SIMD:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T + bias).
//
// Workgroup = 4 waves (2 x 2), one per SIMD, each owning a 128 x 128 corner of a 256 x 256 output
// tile: 64 accumulator tuples (60 of them in the accumulator half of the 512-entry register file),
// so a K-step of 32 is 64 independent v_mfma_f32_16x16x32_bf16 against 16 fragment reads -- a
// third fewer LDS bytes per MFMA than the 128 x 64 corners of the 8-wave kernels.  One workgroup
// per CU; the grid is one workgroup per CU and every workgroup walks its share of the output
// tiles, so the stream of K-tiles never drains between tiles: the loads of a tile's first K-tiles
// fly during the previous tile's epilogue.
//
// LDS: a ring of five 32 KiB slots, one operand's K-tile each (all 160 KiB of the CU).  The stream
// of half-tiles h = 2 j + o (o = 0: A of K-tile j, o = 1: W of K-tile j) goes through slot h % 5.
// Per K-tile j, two K-steps of 64 MFMAs each:
//   step 0:  MFMAs on the fragments of (j, 0), read a K-step earlier; the 16 fragment reads of
//            (j, 1); the 8 LDS-DMA instructions of half-tile 2j + 4 (A of K-tile j + 2) into the
//            slot W(j - 1) left at the previous barrier.
//   X:       s_waitcnt vmcnt(8) lgkmcnt(0); s_barrier -- all of this wave's DMA but the 8 just issued
//            has landed (so K-tile j + 1 is complete) and its reads of K-tile j are done; behind
//            the barrier that holds for every wave.
//   step 1:  MFMAs on (j, 1); fragment reads of (j + 1, 0); the 8 DMA instructions of half-tile
//            2j + 5 (W of K-tile j + 2) into the slot of A(j).
// So the DMA is issued at a constant rate (one instruction per 8 MFMAs per wave), a half-tile has
// one to one and a half K-tiles of MFMAs to land, there is one barrier per K-tile, no LDS latency
// in front of an MFMA and nothing but MFMAs on the vector pipe inside the loop.  During step 1 of
// K-tile j the slot of W(j) is idle: when j closes an output tile, the epilogue stages through it.
// The slot images are the 128-byte-row images of the 8-wave kernels (chunk ^= (row >> 1) & 7 on
// the DMA's source address and on the read).
//
// Staging is `buffer_load_dwordx4 ... lds` against raw buffer descriptors of the tile's row panels:
// rows past M (or N) are out of the descriptor's range and arrive as zeros.  The DMA and the MFMAs
// are asm statements (MFMA operands in the accumulator half need explicit register classes; an
// LDS-DMA the compiler knows about would make it drain vmcnt before LDS reads), so their order in
// the source is their order in the wave.
//
// Epilogue per wave, eight passes of 16 rows: the accumulator tuples go to LDS as fp32 straight
// from the accumulator file, come back as eight consecutive columns per lane, get bias / GELU /
// gate in fp32 and leave as whole 256-byte row segments of bf16 (residual read the same way, one
// pass ahead).
#ifndef LX2V_GROUP_M
#define LX2V_GROUP_M 4  // measured: 4 row panels per column sweep beat 8 and 16 for this kernel (3.15 vs 3.30 vs 3.66 ms at [75600, 5120] x [5120, 5120])
#endif
#include <algorithm>

#include "gemm_bf16_common.h"

namespace {

#ifndef W4_PROBE
#define W4_PROBE 0  // timing probes of the DMA instruction's issue cost (wrong results)
#endif
#ifndef W4_DMA_GROUPS
#define W4_DMA_GROUPS 2  // the 8 DMA pieces of a half-step go to MFMA groups 0, D, 2 D, ... (1: the first half of the step, 2: spread over all of it)
#endif
#ifndef W4_ABL
#define W4_ABL 0  // timing ablations (wrong results; 1024: with 8 only, half as many MFMAs of the 32x32x16 shape): 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no wait + barrier per K-tile, 8 no epilogue, 16 no MFMAs, 32 wait for the DMA only (no barrier), 64 barrier only (no waits), 128 no relaxed wait behind an epilogue, 256 every workgroup stages tile (0, 0) (all L2 hits), 512 the epilogue computes but does not store
#endif
constexpr int W4_THREADS = 256;
constexpr int EPI_PARTIAL = 3;                // kernel instance that leaves raw fp32 accumulators of a K range (internal)
constexpr int NSLOT = 5;                      // ring of operand K-tiles
constexpr int W4_LDS = NSLOT * TILE_BYTES;    // all 160 KiB of the CU

struct Frag {
  u32x4 a[8];  // activations, rows wm*128 + mt*16 + (lane & 15), 8 bf16 of chunk 4 ks + (lane >> 4)
  u32x4 w[8];  // weights, rows wn*128 + nt*16 + (lane & 15)
};

// acc += W fragment (MFMA A operand: output columns) x A fragment (B operand: rows).  60 of the 64
// accumulator tuples live in the accumulator half of the register file, the last four (nt = 7,
// mt >= 4) in the architectural half: the compiler keeps a few accumulator registers to itself,
// and asked for all 256 it spills tuples inside the loop.
// FIRST: the K-step that opens an output tile starts its accumulators from 0.
template <bool IN_AGPR, bool FIRST>
LX2V_DEVICE void mfma_acc(f32x4& acc, const u32x4& wf, const u32x4& af) {
  if constexpr (W4_ABL & 16) {
    asm volatile("" ::"v"(wf), "v"(af));
  } else if constexpr (FIRST) {
    if constexpr (IN_AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(wf), "v"(af));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(wf), "v"(af));
  } else {
    if constexpr (IN_AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(wf), "v"(af));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wf), "v"(af));
  }
}
constexpr bool acc_in_agpr(int nt, int mt) { return !(nt == 7 && mt >= 4); }
// An accumulator tuple leaves its file as the data operand of an LDS write (DS instructions take
// accumulator registers directly), so every use of those tuples is an asm statement with the
// accumulator register class and the register allocator has no reason to re-home them (given
// ordinary reads it copies all of them out in front of the epilogue and pays for the registers
// with spills inside the K-loop).
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
typedef __attribute__((ext_vector_type(16))) float f32x16;
// (timing build only: named accumulator registers, which the compiler is told are clobbered -- with tied 16-register
// operands it copies the tuples around in front of every MFMA)
template <int B>
LX2V_DEVICE void mfma_big(const u32x4& wf, const u32x4& af) {
  if constexpr (B == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" ::"v"(wf), "v"(af) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  if constexpr (B == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(wf), "v"(af) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
  if constexpr (B == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(wf), "v"(af) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
  if constexpr (B == 3) asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" ::"v"(wf), "v"(af) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
  if constexpr (B == 4) asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" ::"v"(wf), "v"(af) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
  if constexpr (B == 5) asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" ::"v"(wf), "v"(af) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
  if constexpr (B == 6) asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" ::"v"(wf), "v"(af) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
  if constexpr (B == 7) asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" ::"v"(wf), "v"(af) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
}
template <int NT, int MT>
LX2V_DEVICE void pin_acc(f32x4& acc) {
  if constexpr (acc_in_agpr(NT, MT)) asm volatile("" : "+a"(acc));
  else asm volatile("" : "+v"(acc));
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

template <int EPI>
__global__ __launch_bounds__(W4_THREADS) void gemm_bf16_w4_kernel(const GemmParams p) {
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
  int part_vbid = 0;      // virtual block id of the partial item's tile
  int64_t k_first = 0;    // first K column of the item
  if constexpr (PARTIAL) {
    const int item = blockIdx.x, part = item % p.tail_split;
    part_vbid = p.tail_first + item / p.tail_split;
    k_first = (int64_t)part * p.tail_nk * BK;
    nk = min(p.tail_nk, nk - part * p.tail_nk);
  }

  // ---- staging: DMA instruction q (0..7) of wave w fills rows 64 w + 8 q .. + 7 of the tile image
  // (1 KiB of LDS, lane L the 16 bytes at 16 L: row 64 w + 8 q + (L >> 3), position L & 7 holding
  // source chunk (L & 7) ^ ((row >> 1) & 7)).  A wave's pieces are 1 KiB apart, so the instruction's
  // 12-bit offset field (added to the LDS address and to the memory address alike) reaches four of
  // them from one M0: two M0 writes per half-tile instead of eight.
  auto piece_voff = [&](int q, int64_t ld) __attribute__((always_inline)) -> int {
    const int row = 64 * wid + 8 * q + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    return (int)(row * ld * 2) + chunk * 16 - (q & 3) * 1024;  // the offset field adds (q & 3) KiB back
  };
  // Per-piece byte offsets of this lane (piece q: 32 q rows further down), kept in registers so that
  // a piece costs one scalar instruction (M0) and the DMA instruction, nothing else.
  int a_pv[8], w_pv[8];  // a_pv is set with the A descriptor (row stride of the region)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    w_pv[q] = piece_voff(q, p.ldw);
    asm volatile("" : "+v"(w_pv[q]));
  }
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds);
  const uint32_t dma_base = lds_base + 8192 * (uint32_t)wid;

  // DMA cursor: the K-tile staged next, two K-tiles ahead of the one computed.
  int d_ord = 0;                       // ordinal of the cursor's output tile in this workgroup's list
  int d_kt = 0;
  ACursor d_cur(p);
  int d_region = 0;
  u32x4 a_rs, w_rs;
  // Descriptors of the cursor's tile (and A region).  Past the workgroup's last tile they are
  // empty (0 bytes): the DMA instructions of the stream's last two K-tiles then fetch nothing and
  // write zeros into a stage nobody reads, so the loop body needs no "last tiles" variants.
  auto d_panel = [&]() __attribute__((always_inline)) {
    const int vbid = PARTIAL ? part_vbid : d_ord * grid + (int)blockIdx.x;
    const bool live = PARTIAL ? d_ord == 0 : vbid < ntiles;
    int m0 = 0, n0 = 0;
    if (live && !(W4_ABL & 256)) tile_origin(p, vbid, m0, n0);
    const int64_t ld = d_region ? p.lda2 : p.lda;
    a_rs = raw_descriptor(p.A + k_first + (int64_t)m0 * ld, live ? p.a_bytes - ((int64_t)m0 * ld + k_first) * 2 : 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a_pv[q] = piece_voff(q, ld);
      asm volatile("" : "+v"(a_pv[q]));
    }
    w_rs = raw_descriptor(p.W + k_first + (int64_t)n0 * p.ldw, live ? p.w_bytes - ((int64_t)n0 * p.ldw + k_first) * 2 : 0);
  };
  d_panel();

  // One DMA instruction: lane L's 16 bytes from rsrc + voff + soff + OFS to LDS byte m0 + OFS + 16 L.
  auto dma_m0 = [&](uint32_t slot_base, auto dstc) __attribute__((always_inline)) {  // M0 = slot base + constant
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(slot_base), "i"(decltype(dstc)::value) : "memory", "scc");
  };
  auto dma_go = [&](const u32x4& rsrc, int voff, uint32_t soff, auto ofsc) __attribute__((always_inline)) {
#if W4_PROBE == 1  // M0 only
    asm volatile("" ::"v"(voff), "s"(rsrc), "s"(soff) : "memory");
#else
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff), "s"(rsrc), "s"(soff), "n"(decltype(ofsc)::value) : "memory");
#endif
  };
  // The cursor's half-tile in 8 pieces; piece q fills rows 64 wid + 8 q .. + 7 of the slot at LDS
  // byte `slot_base` (which includes this wave's 8 KiB share).  Pieces 0 and 4 set M0.
  uint32_t d_asoff = 0, d_wsoff = 0;  // scalar byte offsets of the cursor's K-tile
  auto d_begin = [&]() __attribute__((always_inline)) {
    d_asoff = (uint32_t)(d_cur.koff * 2);
    d_wsoff = (uint32_t)d_kt * (BK * 2);
  };
  auto dma_piece_m0 = [&](uint32_t slot_base, auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    if constexpr ((q & 3) == 0) dma_m0(slot_base, IntC<1024 * q>{});
  };
  auto dma_piece_go = [&](auto halfc, auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    if constexpr (decltype(halfc)::value == 0) dma_go(a_rs, a_pv[q], d_asoff, IntC<1024 * (q & 3)>{});
    else dma_go(w_rs, w_pv[q], d_wsoff, IntC<1024 * (q & 3)>{});
  };
  auto dma_piece_a = [&](uint32_t slot_base, auto qc) __attribute__((always_inline)) {
    dma_piece_m0(slot_base, qc);
    asm volatile("s_nop 0");
    dma_piece_go(IntC<0>{}, qc);
  };
  auto dma_piece_w = [&](uint32_t slot_base, auto qc) __attribute__((always_inline)) {
    dma_piece_m0(slot_base, qc);
    asm volatile("s_nop 0");
    dma_piece_go(IntC<1>{}, qc);
  };
  auto d_advance = [&]() __attribute__((always_inline)) {  // after the W half of a K-tile
    ++d_kt;
    if (d_kt == nk) {
      d_kt = 0;
      ++d_ord;
      d_cur = ACursor(p);
      d_region = 0;
      d_panel();
    } else if (d_cur.advance(p)) {
      d_region = 1;
      d_panel();
    }
  };

  // ---- fragment read offsets inside a slot image, per K-step; mt / nt add 2048 each ----
  int rd_a[2], rd_w[2];
  {
    const int arow = wm * 128 + fr, wrow = wn * 128 + fr;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      rd_a[ks] = arow * 128 + (((ks * 4 + fq) ^ ((arow >> 1) & 7)) << 4);
      rd_w[ks] = wrow * 128 + (((ks * 4 + fq) ^ ((wrow >> 1) & 7)) << 4);
    }
  }
  // (slot bases are scalars that change every K-tile: one add per operand and K-step)
  auto read_frag = [&](const char* base, int i) __attribute__((always_inline)) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(base + i * 2048);
  };

  f32x4 acc[8][8];  // acc[nt][mt][j] = C'[m = wm*128 + mt*16 + fr][n = wn*128 + nt*16 + fq*4 + j]
  // (no initialisation: the first K-step of every output tile writes them from 0)
  Frag f0, f1;  // fragments of K-step 0 / 1 of the current K-tile

  // One K-step: 16 groups of 4 MFMAs on `cur` (W fragment nt = g >> 1, A fragments 4 (g & 1) .. + 3).
  // The first 8 groups carry the 16 fragment reads of the next K-step into `nxt` (A fragments first:
  // the next step needs all of them in its first two groups), from the slot images at `na` / `nw`;
  // every second group issues one piece of the cursor's half-tile into the slot at `dslot`.
  // (Past the end of the stream the reads return stale bytes that no MFMA uses.)
  auto kstep = [&](Frag& cur, Frag& nxt, const char* na, const char* nw, uint32_t dslot, auto HALFc, auto FIRSTc) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(FIRSTc)::value != 0;
    static_for<0, 16>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value, nt = g >> 1, mt0 = 4 * (g & 1);
      constexpr bool DMA = g % W4_DMA_GROUPS == 0 && g / W4_DMA_GROUPS < 8 && !(W4_ABL & 1);
      constexpr int q = g / W4_DMA_GROUPS;
      // One instruction class per MFMA shadow: M0, the DMA instruction, then one fragment read each
      // (a wave alone on its SIMD issues about one instruction every four cycles, so an MFMA covers three).
      if constexpr (W4_ABL & 1024) mfma_big<((nt >> 1) * 4 + (mt0 >> 1)) % 8>(cur.w[nt], cur.a[mt0]);
      else mfma_acc<acc_in_agpr(nt, mt0), FIRST>(acc[nt][mt0], cur.w[nt], cur.a[mt0]);
      if constexpr (DMA) dma_piece_m0(dslot, IntC<q>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(W4_ABL & 1024)) mfma_acc<acc_in_agpr(nt, mt0 + 1), FIRST>(acc[nt][mt0 + 1], cur.w[nt], cur.a[mt0 + 1]);
      if constexpr (DMA) dma_piece_go(HALFc, IntC<q>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (W4_ABL & 1024) mfma_big<((nt >> 1) * 4 + ((mt0 + 2) >> 1)) % 8>(cur.w[nt], cur.a[mt0 + 2]);
      else mfma_acc<acc_in_agpr(nt, mt0 + 2), FIRST>(acc[nt][mt0 + 2], cur.w[nt], cur.a[mt0 + 2]);
      if constexpr (W4_ABL & 2) {
      } else if constexpr (g < 4) {
        nxt.a[2 * g] = read_frag(na, 2 * g);
      } else if constexpr (g < 8) {
        nxt.w[2 * g - 8] = read_frag(nw, 2 * g - 8);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(W4_ABL & 1024)) mfma_acc<acc_in_agpr(nt, mt0 + 3), FIRST>(acc[nt][mt0 + 3], cur.w[nt], cur.a[mt0 + 3]);
      if constexpr (W4_ABL & 2) {
      } else if constexpr (g < 4) {
        nxt.a[2 * g + 1] = read_frag(na, 2 * g + 1);
      } else if constexpr (g < 8) {
        nxt.w[2 * g - 7] = read_frag(nw, 2 * g - 7);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // AFTER_EPI: the barrier of the first K-tile behind an epilogue.  The epilogue's 32 stores are
  // younger than the half-tile that has to have landed and older than the 8 DMA instructions just
  // issued; memory operations retire in order, so 40 may stay outstanding.  (Every epilogue issues
  // all 32: the stores of rows and columns outside C are dropped by the range check of their
  // buffer descriptor, not branched around.)
  auto sync_x = [&](auto AFTER_EPIc) __attribute__((always_inline)) {
    if constexpr (W4_ABL & 4) return;
    if constexpr (!(W4_ABL & 64)) {
      if constexpr (decltype(AFTER_EPIc)::value != 0 && !(W4_ABL & 128)) asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    }
    if constexpr (!(W4_ABL & 32)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- epilogue of the output tile at (m0, n0) ----
  // Eight passes, one per 16-row block mt: the wave's eight tuples of that block go to its 8 KiB of
  // staging as fp32 (16 rows x 512 B; 16-byte chunk c of row r at position c ^ r), come back as
  // eight consecutive columns per lane, and leave as whole 256-byte row segments of bf16.
  auto epilogue = [&](int m0, int n0, int epi_base) __attribute__((always_inline)) {
    // Lane coordinates behind an opaque statement: everything below that depends only on the lane
    // would otherwise be computed once in front of the K-loop and sit in registers the loop needs
    // for its fragments.
    int fr = lane & 15, fq = lane >> 4;
    asm volatile("" : "+v"(fr), "+v"(fq));
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
    float bv[8], gv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bv[j] = 0.f;
      gv[j] = 1.f;
    }
    if (col_ok) {
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
    // An unconditional use of the bias / gate registers: their loads are then waited for here, once.
    // Their only other uses sit in the conditional store blocks below, and a load that is still
    // pending on the path around such a block makes the compiler wait for ALL memory operations
    // (vmcnt(0)) in front of every one of them, i.e. for the previous store's completion.
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(bv[j]), "+v"(gv[j]));
    // Residual rows are loaded THREE passes ahead (clamped addresses, no branch: the only conditional operation
    // of a pass is the store).  Memory operations retire in order, so a pass that waits for its residual rows also
    // waits for every store issued before their loads; one pass ahead that was the previous pass's stores, a full
    // write latency per pass.  The ring of four passes (16 registers each) lives in the fragment variables of
    // K-step 1, which are dead between K-tiles: what they hold when the epilogue ends is complete, consumed data.
    auto resbuf = [&](auto bc, auto itc) __attribute__((always_inline)) -> u32x4& {
      constexpr int b = decltype(bc)::value, it = decltype(itc)::value;
      if constexpr (b < 2) return f1.a[(b & 1) * 4 + it];
      else return f1.w[(b & 1) * 4 + it];
    };
    auto load_resid = [&](auto mtc) __attribute__((always_inline)) {
      constexpr int mt = decltype(mtc)::value;
      if constexpr (EPI == LX2V_EPI_RESIDUAL && mt < 8) {
        static_for<0, 4>([&](auto itc) __attribute__((always_inline)) {
          constexpr int it = decltype(itc)::value;
          const int gm = min(m0 + wm * 128 + mt * 16 + it * 4 + fq, p.M - 1);
          resbuf(IntC<mt & 3>{}, itc) = *reinterpret_cast<const u32x4*>(p.resid + (int64_t)gm * p.ldr + (col_ok ? gn : 0));
        });
      }
    };
    load_resid(IntC<0>{});
    load_resid(IntC<1>{});
    load_resid(IntC<2>{});
    // Software pipeline over the passes: the tuples of pass mt + 1 are written to the staging rows right behind the
    // reads of pass mt and in front of its arithmetic (one wave's LDS instructions execute in order, so those
    // writes cannot overtake the reads), i.e. the LDS round trip of a pass runs under the previous pass's
    // conversions and stores.
    auto write_pass = [&](auto mtc) __attribute__((always_inline)) {
      constexpr int mt = decltype(mtc)::value;
      if constexpr (mt < 8) {
        static_for<0, 8>([&](auto ntc) __attribute__((always_inline)) {
          constexpr int nt = decltype(ntc)::value;
          acc_to_lds<acc_in_agpr(nt, mt)>(waddr[nt], acc[nt][mt]);
        });
      }
    };
    write_pass(IntC<0>{});
    static_for<0, 8>([&](auto mtc) __attribute__((always_inline)) {
      constexpr int mt = decltype(mtc)::value;
      load_resid(IntC<mt + 3>{});
      f32x4 lo[4], hi[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 4 + fq;
        const char* row = lds + epi_base + wid * 8192 + r * 512;
        lo[it] = *reinterpret_cast<const f32x4*>(row + (((2 * fr) ^ r) << 4));
        hi[it] = *reinterpret_cast<const f32x4*>(row + (((2 * fr + 1) ^ r) << 4));
      }
      asm volatile("" ::: "memory");
      write_pass(IntC<mt + 1>{});
      static_for<0, 4>([&](auto itc) __attribute__((always_inline)) {
        constexpr int it = decltype(itc)::value;
        const int r = it * 4 + fq;
        const int gm = m0 + wm * 128 + mt * 16 + r;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo[it][j]; v[4 + j] = hi[it][j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] += bv[j];
          if (EPI == LX2V_EPI_GELU_TANH) v[j] = gelu_tanh(v[j]);
          if (EPI == LX2V_EPI_RESIDUAL) v[j] *= gv[j];
        }
        if (EPI == LX2V_EPI_RESIDUAL) {
          float y[8];
          unpack8(resbuf(IntC<mt & 3>{}, itc), y);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (float)(__bf16)v[j] + y[j];  // the product is rounded to bf16 before the add, as in the 8-wave kernels
        }
        const u32x4 out = pack8(v);
        if constexpr (W4_ABL & 512) asm volatile("" ::"v"(out));
        else __builtin_amdgcn_raw_buffer_store_b128(out, c_rs, col_ok ? (int)((int64_t)(gm - m0) * p.ldc * 2) + c_col : C_NOWHERE, 0, 0);
      });
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
        const int voff = lane * 16 + (nt * 8 + mt) * 1024;
        acc_to_global<acc_in_agpr(nt, mt)>(voff, acc[nt][mt], sbase);
      });
    });
  };

  // ---- prologue: half-tiles 0..3 (K-tiles 0 and 1 of the stream), then the fragments of (0, 0) ----
  const int my_tiles = PARTIAL ? 1 : (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / grid + 1 : 0;
  const int64_t total = (int64_t)my_tiles * nk;  // K-tiles in this workgroup's stream
  if (total == 0) return;
  static_for<0, 2>([&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    d_begin();
    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) { dma_piece_a(dma_base + (2 * t) * TILE_BYTES, qc); });
    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) { dma_piece_w(dma_base + (2 * t + 1) * TILE_BYTES, qc); });
    d_advance();
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f0.a[i] = read_frag(lds + rd_a[0], i);
    f0.w[i] = read_frag(lds + TILE_BYTES + rd_w[0], i);
  }

  // ---- the stream ----
  int ord = 0, kt = 0;  // output tile ordinal and K-tile of stream position j
  bool relaxed = false; // the previous K-tile ended in an epilogue (32 stores per wave)
  int r5 = 0;           // (2 j) % 5: slot of A(j); W(j) r5 + 1, A(j+1) r5 + 2, W(j+1) r5 + 3, free r5 + 4 (all mod 5)
  auto slot = [&](int k) __attribute__((always_inline)) -> int {  // byte offset of slot (r5 + k) % 5
    const int s5 = r5 + k;
    return (s5 >= NSLOT ? s5 - NSLOT : s5) * TILE_BYTES;
  };
  for (int64_t j = 0; j < total; ++j) {
    const int sa = slot(0), sw = slot(1), sa1 = slot(2), sw1 = slot(3), sfree = slot(4);
    // step 0: reads of (j, 1); DMA of A(j + 2) into the free slot
    d_begin();
    if (kt == 0) {
      kstep(f0, f1, lds + sa + rd_a[1], lds + sw + rd_w[1], dma_base + (uint32_t)sfree, IntC<0>{}, IntC<1>{});
      if (relaxed) sync_x(IntC<1>{});
      else sync_x(IntC<0>{});
    } else {
      kstep(f0, f1, lds + sa + rd_a[1], lds + sw + rd_w[1], dma_base + (uint32_t)sfree, IntC<0>{}, IntC<0>{});
      sync_x(IntC<0>{});
    }
    // step 1: reads of (j + 1, 0); DMA of W(j + 2) into the slot A(j) has left
    kstep(f1, f0, lds + sa1 + rd_a[0], lds + sw1 + rd_w[0], dma_base + (uint32_t)sa, IntC<1>{}, IntC<0>{});
    d_advance();
    relaxed = false;
    if (++kt == nk) {
      if constexpr (PARTIAL) {
        store_partial();
        break;
      }
      int m0, n0;
      tile_origin(p, ord * grid + (int)blockIdx.x, m0, n0);
      relaxed = !(W4_ABL & (8 | 512));
      if constexpr (!(W4_ABL & 8)) {
        epilogue(m0, n0, sw);  // W(j)'s slot is idle until A(j + 3) is staged into it, in the next step 0 ...
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... by every wave into every wave's share of the slot: all must have left the epilogue
      }
      // Keep the (dead) fragment registers of K-step 1 occupied across the epilogue: handed to its
      // loads and stores, they would be overwritten by the next K-step's LDS reads while those
      // memory operations are still counted as pending, and the compiler would protect them with a
      // vmcnt wait inside the K-loop -- which also waits for the LDS-DMA in flight.
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(f1.a[i]), "+v"(f1.w[i]));
      kt = 0;
      ++ord;
    }
    r5 += 2;
    if (r5 >= NSLOT) r5 -= NSLOT;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may still be writing this workgroup's LDS when it ends
}

// Tail tiles of a launch whose last round was split along K: one thread sums the parts of 8 consecutive outputs
// of a row in a fixed order (part 0 first) and applies the epilogue of the persistent kernel, same arithmetic.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_tail_fixup_kernel(const GemmParams p) {
  const int t = blockIdx.x >> 5;  // tail tile
  const int e = ((blockIdx.x & 31) << 8) + threadIdx.x;
  const int r = e & 255, c = (e >> 8) * 8;  // consecutive threads: consecutive rows (consecutive lanes of a fragment)
  int m0, n0;
  tile_origin(p, p.tail_first + t, m0, n0);
  const int gm = m0 + r, gn = n0 + c;
  if (gm >= p.M || gn >= p.N) return;
  const int wave = (r >> 7) * 2 + (c >> 7), mt = (r & 127) >> 4, fr = r & 15, nt = (c & 127) >> 4, fq0 = (c & 15) >> 2;
  const float* src = p.partials + ((((int64_t)t * p.tail_split * 4 + wave) * 64 + nt * 8 + mt) * 64 + fq0 * 16 + fr) * 4;
  const int64_t part_stride = (int64_t)4 * 64 * 64 * 4;  // floats per item
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int part = 0; part < p.tail_split; ++part) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(src + part * part_stride);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(src + part * part_stride + 64);  // the next column group: 16 lanes on
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] += lo[j]; v[4 + j] += hi[j]; }
  }
  float bv[8], gv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bv[j] = p.bias != nullptr ? p.bias[gn + j] : 0.f;
    gv[j] = (EPI == LX2V_EPI_RESIDUAL && p.gate != nullptr) ? p.gate[gn + j] : 1.f;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[j] += bv[j];
    if (EPI == LX2V_EPI_GELU_TANH) v[j] = gelu_tanh(v[j]);
    if (EPI == LX2V_EPI_RESIDUAL) v[j] *= gv[j];
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
// have a tile in it, K is long enough for two more launches to pay, and A is a plain matrix.
TailPlan tail_plan(const GemmParams& p, int cus) {
  TailPlan t{0, 1, 0, 0};
  const int64_t ntiles = (int64_t)p.tiles_m * p.tiles_n;
  const int nk = p.K / BK;
  if (ntiles <= cus || p.a_kseg != 0 || nk < 64) return t;
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

}  // namespace

// Workspace a call of this shape can use (0: its last round is not split).
int64_t lx2v_gemm_w4_workspace_bytes(const lx2v_gemm::GemmParams& p) { return tail_plan(p, device_cus()).bytes; }

// Launch for p already validated by lx2v_gemm_bf16.  One workgroup per CU (the 160 KiB of LDS allow
// no second one), never more than there are tiles.  With a workspace, a sparsely filled last round
// becomes a K split of its tiles (see GemmParams).
int lx2v_gemm_launch_w4(lx2v_gemm::GemmParams& p, int epilogue, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  const int cus = device_cus();
  const int64_t ntiles = (int64_t)p.tiles_m * p.tiles_n;
  TailPlan tp = tail_plan(p, cus);
  if (tp.bytes == 0 || workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 15u) != 0 || workspace_bytes < tp.bytes) tp = TailPlan{0, 1, 0, 0};
  p.ntiles_run = (int)(ntiles - tp.rem);
  p.tail_first = p.ntiles_run;
  p.tail_split = tp.split;
  p.tail_nk = tp.nk_part;
  p.partials = reinterpret_cast<float*>(workspace);
  const dim3 grid((unsigned)(p.ntiles_run < cus ? p.ntiles_run : cus)), block(W4_THREADS);
  switch (epilogue) {
    case LX2V_EPI_NONE: hipLaunchKernelGGL(gemm_bf16_w4_kernel<LX2V_EPI_NONE>, grid, block, 0, st, p); break;
    case LX2V_EPI_GELU_TANH: hipLaunchKernelGGL(gemm_bf16_w4_kernel<LX2V_EPI_GELU_TANH>, grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL(gemm_bf16_w4_kernel<LX2V_EPI_RESIDUAL>, grid, block, 0, st, p); break;
  }
  if (tp.rem > 0) {
    hipLaunchKernelGGL(gemm_bf16_w4_kernel<EPI_PARTIAL>, dim3((unsigned)(tp.rem * tp.split)), block, 0, st, p);
    const dim3 fgrid((unsigned)(tp.rem * 32));
    switch (epilogue) {
      case LX2V_EPI_NONE: hipLaunchKernelGGL(gemm_tail_fixup_kernel<LX2V_EPI_NONE>, fgrid, dim3(256), 0, st, p); break;
      case LX2V_EPI_GELU_TANH: hipLaunchKernelGGL(gemm_tail_fixup_kernel<LX2V_EPI_GELU_TANH>, fgrid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL(gemm_tail_fixup_kernel<LX2V_EPI_RESIDUAL>, fgrid, dim3(256), 0, st, p); break;
    }
  }
  return LX2V_OK;
}
