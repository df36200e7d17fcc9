// bf16 Linear on MFMA for gfx950: C[M,N] = epilogue(A[M,K] @ W[N,K]^T + bias).
//
// Both operands are K-contiguous (the checkpoint's [out,in] weight layout is exactly the
// "B transposed" form MFMA wants), so A and W tiles are staged the same way.
//
// Tile: 256 x 256 x 64 per 512-thread workgroup (8 waves as 2(M) x 4(N), 128 x 64 per wave),
// one workgroup per CU (128 KiB LDS: 2 stages x (A 32 KiB + W 32 KiB)).
// Staging: LDS-DMA (global_load_lds, 16 B per lane).  The LDS image of a tile is [256 rows][8
// chunks of 16 B]; the DMA writes lane-linear, so the bank-conflict swizzle
// (chunk ^= (row >> 1) & 7) is applied to the per-lane SOURCE address and again on the read.
// With it the 16 lanes of every ds_read_b128 lane group hit 16 distinct 16-B slots.
// MFMA: v_mfma_f32_16x16x32_bf16 with the W fragment as the A operand, so each lane ends up
// with 4 consecutive output columns of one row (8 contiguous bytes after packing).
// Epilogue: bias / GELU / gate in fp32 on the accumulators, staged per wave through LDS as
// bf16, then whole 128-B row segments are written (and the residual read) with 16-B accesses.
#include <stdlib.h>

#include <type_traits>

#include "gemm_bf16_common.h"

namespace {

constexpr int NTHREADS = 512;

// Epilogue shared by both schedules: fp32 math on the accumulators -> per-wave LDS image (bf16) ->
// whole 128-B row segments.  acc[nt][mt][j] = C'[m = wm*128 + mt*16 + fr][n = wn*64 + nt*16 + fq*4 + j].
// Every wave of the workgroup must have finished reading the operand tiles in `lds`.
template <int EPI>
LX2V_DEVICE void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[4][8], char* lds, int m0, int n0, int wid, int wm, int wn, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  char* stg = lds + wid * 16384;  // 128 rows x 128 B per wave
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int ncol = n0 + wn * 64 + nt * 16 + fq * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f};
    if (ncol < p.N) {
      if (p.bias != nullptr) bv = *reinterpret_cast<const f32x4*>(p.bias + ncol);
      if (EPI == LX2V_EPI_RESIDUAL && p.gate != nullptr) gv = *reinterpret_cast<const f32x4*>(p.gate + ncol);
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[nt][mt][j] + bv[j];
        if (EPI == LX2V_EPI_GELU_TANH) v[j] = gelu_tanh(v[j]);
        if (EPI == LX2V_EPI_RESIDUAL) v[j] *= gv[j];
      }
      const int r = mt * 16 + fr;                 // row inside the wave's 128-row image
      const int slot = nt * 4 + fq;               // 8-byte slot inside the 128-B row
      const int off = r * 128 + ((((slot >> 1) ^ (r & 7)) << 4) | ((slot & 1) << 3));
      u32x2 pk = {pack2(v[0], v[1]), pack2(v[2], v[3])};
      *reinterpret_cast<u32x2*>(stg + off) = pk;
    }
  }
  // Each wave reads back only its own image: a wave-level wait on the LDS writes is enough.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int c8 = lane & 7;  // 16-B chunk inside the 128-B row segment
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int r = it * 8 + (lane >> 3);
    const int gm_ = m0 + wm * 128 + r;
    const int gn = n0 + wn * 64 + c8 * 8;
    u32x4 val = *reinterpret_cast<const u32x4*>(stg + r * 128 + ((c8 ^ (r & 7)) << 4));
    if (gm_ < p.M && gn < p.N) {
      if (EPI == LX2V_EPI_RESIDUAL) {
        float a[8], b[8];
        unpack8(val, a);
        unpack8(*reinterpret_cast<const u32x4*>(p.resid + (int64_t)gm_ * p.ldr + gn), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        val = pack8(a);
      }
      *reinterpret_cast<u32x4*>(p.C + (int64_t)gm_ * p.ldc + gn) = val;
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];  // [stage][A|W][32 KiB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;

  // ---- tile coordinates (XCD-aware, grouped along M) ----
  const int nwg = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int group_sz = GROUP_M * p.tiles_n;
  const int g = t / group_sz;
  const int first_m = g * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_g = t - g * group_sz;
  const int tile_m = first_m + in_g % gm;
  const int tile_n = in_g / gm;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread staging sources: 4 DMA instructions per operand per K-tile ----
  // LDS byte offset o = i*8192 + tid*16 -> row = o >> 7, physical chunk = (o >> 4) & 7.
  const uint16_t* a_src[4];
  const uint16_t* w_src[4];
  int a_row[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = i * 8192 + tid * 16;
    const int row = o >> 7;
    const int chunk = ((o >> 4) & 7) ^ ((row >> 1) & 7);
    const int ar = min(m0 + row, p.M - 1);
    const int wr = min(n0 + row, p.N - 1);
    a_src[i] = p.A + (int64_t)ar * p.lda + chunk * 8;
    w_src[i] = p.W + (int64_t)wr * p.ldw + chunk * 8;
    a_row[i] = ar;
  }
  ACursor a_cur(p);  // the K-tile staged next

  auto stage = [&](int s, int kt) {
    char* la = lds + s * 2 * TILE_BYTES + wid * 1024;
    char* lw = la + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + a_cur.koff, la + i * 8192);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w_src[i] + (int64_t)kt * BK, lw + i * 8192);
    if (a_cur.advance(p)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a_src[i] += (int64_t)a_row[i] * (p.lda2 - p.lda);
    }
  };

  // ---- fragment read offsets ----
  // A operand of the MFMA = W fragment: row n = wn*64 + nt*16 + (lane & 15), chunk = ks*4 + (lane >> 4)
  // B operand of the MFMA = A fragment: row m = wm*128 + mt*16 + (lane & 15)
  const int fr = lane & 15, fq = lane >> 4;
  int a_rd[2], w_rd[2];  // byte offsets for ks = 0,1 at mt/nt = 0 (rows step by 16 -> +2048 B, swizzle unchanged)
  {
    const int arow = wm * 128 + fr, wrow = wn * 64 + fr;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_rd[ks] = arow * 128 + (((ks * 4 + fq) ^ ((arow >> 1) & 7)) << 4);
      w_rd[ks] = wrow * 128 + (((ks * 4 + fq) ^ ((wrow >> 1) & 7)) << 4);
    }
  }

  f32x4 acc[4][8];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk && !(GEMM_ABL & 2)) stage(cur ^ 1, kt + 1);
    const char* la = lds + cur * 2 * TILE_BYTES;
    const char* lw = la + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[8], wf[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(lw + w_rd[ks] + nt * 2048);
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) af[mt] = *reinterpret_cast<const bf16x8*>(la + a_rd[ks] + mt * 2048);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
    }
    if (!(GEMM_ABL & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(GEMM_ABL & 4)) __syncthreads();
  }

  gemm_epilogue<EPI>(p, acc, lds, m0, n0, wid, wm, wn, lane);
}


// ---------------------------------------------------------------------------------------------
// Ping-pong variant (the default for K >= 128).  Same tile, fragments, swizzle and epilogue; what
// changes is the schedule.  The 8 waves form two groups (wave rows 0 and 1; on every SIMD one wave of
// each).  A K-tile is computed in four phases, one 64 x 32 quadrant of the wave's 128 x 64 output
// each: [LDS fragment reads + 2 LDS-DMA instructions] barrier [16 MFMAs] barrier.  Group 1 runs one
// barrier behind group 0, so on each SIMD one wave's reads and DMA issue overlap the other wave's
// MFMAs.  Staging units are the row sets one phase reads (16 KiB each):
//     Am0 = A rows {0-63, 128-191}   read in phase 1        Bn0 = W rows {64w + 0-31}   read in phase 1 (kept for phase 4)
//     Am1 = A rows {64-127, 192-255} read in phase 3        Bn1 = W rows {64w + 32-63}  read in phase 2
// and travel in the fixed stream order Am0(t) Bn0(t) Bn1(t) Am1(t) Am0(t+1) ..., one unit per phase:
//     phase 1 of tile u issues Bn1(u+1), phase 2 Am1(u+1), phase 3 Am0(u+2), phase 4 Bn0(u+2)
// -- each into a region whose last read lies at least two phases back -- and every phase waits with
// vmcnt(8): all but the newest four units have landed, which is exactly what the NEXT phase reads.
// A unit is read one phase after the wait + barrier that retires it (LDS-DMA data is ordered for
// another wave's ds_read only by the issuing wave's vmcnt followed by a barrier).
constexpr int A_BUF = BM * BK * 2;           // 32 KiB per operand per buffer
constexpr int W_BASE = 2 * A_BUF;            // LDS: [A buf0][A buf1][W buf0][W buf1]
enum { TILE_FULL = 0, TILE_PENULT = 1, TILE_LAST = 2 };

template <int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_pp_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;

  const int nwg = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int group_sz = GROUP_M * p.tiles_n;
  const int g = t / group_sz;
  const int first_m = g * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_g = t - g * group_sz;
  const int tile_m = first_m + in_g % gm;
  const int tile_n = in_g / gm;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- staging: unit U in {0: Am0, 1: Bn0, 2: Bn1, 3: Am1}, two DMA instructions (i) per unit ----
  // unit row r = i*64 + wid*8 + (lane >> 3), physical chunk = lane & 7; 8 consecutive unit rows are 8
  // consecutive tile rows, so each instruction fills 1 KiB of contiguous LDS (lane-linear).
  // Sources are a wave-uniform base (tile origin + K offset, scalar registers) plus a per-lane
  // 32-bit byte offset inside the tile's row panel.
  uint32_t soff[4][2];
  uint32_t soff2[2][2];  // soff of the A units (0: Am0, 1: Am1) under region 2's row stride
  int dst[4][2];  // LDS byte offset of the wave's first lane, inside buffer 0
  const char* a_base[2] = {reinterpret_cast<const char*>(p.A + (int64_t)m0 * p.lda), nullptr};  // row panel origin for Am0 / Am1
  a_base[1] = a_base[0];
  const char* w_base = reinterpret_cast<const char*>(p.W + (int64_t)n0 * p.ldw);
#pragma unroll
  for (int U = 0; U < 4; ++U)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r0 = i * 64 + wid * 8, r = r0 + (lane >> 3);
      const bool is_a = (U == 0 || U == 3);
      const int row = is_a ? (r >> 6) * 128 + (U == 3 ? 64 : 0) + (r & 63) : (r >> 5) * 64 + (U == 2 ? 32 : 0) + (r & 31);
      const int row0 = is_a ? (r0 >> 6) * 128 + (U == 3 ? 64 : 0) + (r0 & 63) : (r0 >> 5) * 64 + (U == 2 ? 32 : 0) + (r0 & 31);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const int rr = is_a ? min(row, p.M - 1 - m0) : min(row, p.N - 1 - n0);  // rows past the edge re-read the last one
      soff[U][i] = (uint32_t)((int64_t)rr * (is_a ? p.lda : p.ldw) * 2 + chunk * 16);
      if (is_a) soff2[U == 3][i] = (uint32_t)((int64_t)rr * p.lda2 * 2 + chunk * 16);
      dst[U][i] = (is_a ? 0 : W_BASE) + row0 * 128;
    }
  auto enter_region2 = [&](int U) {  // U = 0 (Am0) or 3 (Am1): the unit's cursor now gives offsets of region 2
    a_base[U == 3] = reinterpret_cast<const char*>(p.A + (int64_t)m0 * p.lda2);
#pragma unroll
    for (int i = 0; i < 2; ++i) soff[U][i] = soff2[U == 3][i];
  };
  ACursor a_issue(p);  // the tile whose Am0 is issued next
  ACursor a_am1(p);    // the tile whose Am1 is issued next

  auto issue = [&](int U, int buf, int64_t koff) {
    if (GEMM_ABL & 2) return;
    const char* base = (U == 0 ? a_base[0] : U == 3 ? a_base[1] : w_base) + koff * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(base + soff[U][i], lds + dst[U][i] + buf * A_BUF);
  };

  const int fr = lane & 15, fq = lane >> 4;
  int a_rd[2], w_rd[2];
  {
    const int arow = wm * 128 + fr, wrow = wn * 64 + fr;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_rd[ks] = arow * 128 + (((ks * 4 + fq) ^ ((arow >> 1) & 7)) << 4);
      w_rd[ks] = W_BASE + wrow * 128 + (((ks * 4 + fq) ^ ((wrow >> 1) & 7)) << 4);
    }
  }

  f32x4 acc[4][8];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;  // >= 2 (checked by the launcher)

  // ---- prologue: Am0(0) Bn0(0) Bn1(0) Am1(0) Am0(1) Bn0(1) ----
  issue(0, 0, a_issue.koff);
  issue(1, 0, 0);
  issue(2, 0, 0);
  issue(3, 0, a_am1.koff);
  if (a_issue.advance(p)) enter_region2(0);
  if (a_am1.advance(p)) enter_region2(3);
  issue(0, 1, a_issue.koff);
  issue(1, 1, BK);
  if (a_issue.advance(p)) enter_region2(0);  // now at tile 2
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind

  bf16x8 af[4][2], b0[2][2], b1[2][2];

  // One K-tile = four phases.  KIND (the same for every wave): FULL issues all four units, PENULT
  // (tile nk-2) only the two for tile nk-1, LAST none; the waits shrink with what is left in flight.
  auto tile = [&](auto buf_c, int u) {
    constexpr int BUF = decltype(buf_c)::value;
    const int KIND = u + 2 < nk ? TILE_FULL : (u + 1 < nk ? TILE_PENULT : TILE_LAST);  // wave-uniform
    constexpr int RD = BUF * A_BUF;  // immediate offset of this tile's buffer for the fragment reads
    const char* base = lds;
    // ---- phase 1: quadrant (m0, n0) ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) b0[nt][ks] = *reinterpret_cast<const bf16x8*>(base + w_rd[ks] + RD + nt * 2048);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt][ks] = *reinterpret_cast<const bf16x8*>(base + a_rd[ks] + RD + mt * 2048);
    if (KIND != TILE_LAST) issue(2, BUF ^ 1, (int64_t)(u + 1) * BK);
    if (KIND == TILE_LAST) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[nt][ks], af[mt][ks], acc[nt][mt], 0, 0, 0);
    if (!(GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: quadrant (m0, n1) ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) b1[nt][ks] = *reinterpret_cast<const bf16x8*>(base + w_rd[ks] + RD + (nt + 2) * 2048);
    if (KIND != TILE_LAST) issue(3, BUF ^ 1, a_am1.koff);
    if (KIND == TILE_LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt + 2][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[nt][ks], af[mt][ks], acc[nt + 2][mt], 0, 0, 0);
    if (!(GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (a_am1.advance(p)) enter_region2(3);
    // ---- phase 3: quadrant (m1, n1) ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt][ks] = *reinterpret_cast<const bf16x8*>(base + a_rd[ks] + RD + (mt + 4) * 2048);
    if (KIND == TILE_FULL) issue(0, BUF, a_issue.koff);
    if (KIND == TILE_FULL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (KIND == TILE_PENULT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt + 2][mt + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[nt][ks], af[mt][ks], acc[nt + 2][mt + 4], 0, 0, 0);
    if (!(GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: quadrant (m1, n0), operands already in registers ----
    if (KIND == TILE_FULL) issue(1, BUF, (int64_t)(u + 2) * BK);
    if (KIND == TILE_FULL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (KIND == TILE_PENULT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!(GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[nt][ks], af[mt][ks], acc[nt][mt + 4], 0, 0, 0);
    if (!(GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (KIND == TILE_FULL && a_issue.advance(p)) enter_region2(0);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  for (int u = 0; u < nk; u += 2) {
    tile(B0{}, u);
    if (u + 1 < nk) tile(B1{}, u + 1);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // group 0 gives back the barrier group 1 spent at the start

  if (GEMM_ABL & 32) {  // timing ablation: one value per lane instead of the epilogue
    float sacc = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) sacc += acc[nt][mt][0] + acc[nt][mt][1] + acc[nt][mt][2] + acc[nt][mt][3];
    if (sacc == 123.f) p.C[tid] = 1;
    return;
  }
  gemm_epilogue<EPI>(p, acc, lds, m0, n0, wid, wm, wn, lane);
}

}  // namespace

int lx2v_gemm_launch_w4(lx2v_gemm::GemmParams& p, int epilogue, void* workspace, int64_t workspace_bytes, hipStream_t st);  // gemm_bf16_w4.hip
int64_t lx2v_gemm_w4_workspace_bytes(const lx2v_gemm::GemmParams& p);

extern "C" int lx2v_gemm_bf16(const uint16_t* A, int64_t lda, const lx2v_kseg* kseg,
                              const uint16_t* W, int64_t ldw, const float* bias,
                              uint16_t* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                              const uint16_t* resid, int64_t ldr, const float* gate, void* stream) {
  return lx2v_gemm_bf16_ws(A, lda, kseg, W, ldw, bias, C, ldc, M, N, K, epilogue, resid, ldr, gate, nullptr, 0, stream);
}

extern "C" int64_t lx2v_gemm_bf16_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || M >= (1LL << 31) || N >= (1LL << 31) || K >= (1LL << 31) || K % BK != 0) return 0;
  GemmParams p{};
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_m = (int)((M + BM - 1) / BM);
  p.tiles_n = (int)((N + BN - 1) / BN);
  if ((int64_t)p.tiles_m * p.tiles_n >= (1LL << 31) || lx2v_options().gemm_kernel != 0) return 0;
  return lx2v_gemm_w4_workspace_bytes(p);
}

extern "C" int lx2v_gemm_bf16_ws(const uint16_t* A, int64_t lda, const lx2v_kseg* kseg,
                                 const uint16_t* W, int64_t ldw, const float* bias,
                                 uint16_t* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                                 const uint16_t* resid, int64_t ldr, const float* gate,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  LX2V_REQUIRE(M == 0 || (A && W && C), LX2V_EINVAL, "gemm: null tensor");
  LX2V_REQUIRE(M >= 0 && N > 0 && K > 0, LX2V_EINVAL, "gemm: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  LX2V_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), LX2V_EUNSUP, "gemm: dimensions must fit in 31 bits");
  LX2V_REQUIRE(K % BK == 0, LX2V_EUNSUP, "gemm: K=%lld must be a multiple of %d", (long long)K, BK);
  LX2V_REQUIRE(N % 8 == 0, LX2V_EUNSUP, "gemm: N=%lld must be a multiple of 8", (long long)N);
  LX2V_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0, LX2V_EINVAL, "gemm: row strides must be multiples of 8 elements");
  LX2V_REQUIRE(ldw >= K && ldc >= N, LX2V_EINVAL, "gemm: ldw/ldc smaller than the row length");
  LX2V_REQUIRE(lx2v_aligned16(A) && lx2v_aligned16(W) && lx2v_aligned16(C), LX2V_EINVAL, "gemm: tensors must be 16-byte aligned");
  LX2V_REQUIRE(bias == nullptr || lx2v_aligned16(bias), LX2V_EINVAL, "gemm: bias must be 16-byte aligned");
  lx2v_kseg ks = {0, 0, 0, 0, 0, 0, 0};
  if (kseg != nullptr) ks = *kseg;
  if (ks.seg != 0) {
    const int64_t k1 = ks.split > 0 ? ks.split : K;  // columns of region 1
    LX2V_REQUIRE(ks.split >= 0 && ks.split < K, LX2V_EINVAL, "gemm: kseg.split=%lld outside [0, K)", (long long)ks.split);
    LX2V_REQUIRE(ks.seg > 0 && ks.seg % BK == 0 && k1 % ks.seg == 0, LX2V_EINVAL, "gemm: kseg.seg=%lld must be a multiple of %d dividing its region of %lld columns", (long long)ks.seg, BK, (long long)k1);
    LX2V_REQUIRE(ks.stride % 8 == 0 && lda >= ks.seg, LX2V_EINVAL, "gemm: segmented A needs lda >= kseg.seg and an 8-element aligned segment stride");
    if (ks.split > 0) {
      LX2V_REQUIRE(ks.seg2 > 0 && ks.seg2 % BK == 0 && (K - ks.split) % ks.seg2 == 0, LX2V_EINVAL, "gemm: kseg.seg2=%lld must be a multiple of %d dividing the %lld columns after the split", (long long)ks.seg2, BK, (long long)(K - ks.split));
      LX2V_REQUIRE(ks.stride2 % 8 == 0 && ks.base2 % 8 == 0 && ks.lda2 % 8 == 0 && ks.lda2 >= ks.seg2, LX2V_EINVAL, "gemm: region 2 needs lda2 >= kseg.seg2 and 8-element aligned lda2, stride and base");
    }
  } else {
    LX2V_REQUIRE(lda >= K, LX2V_EINVAL, "gemm: lda smaller than K");
  }
  LX2V_REQUIRE(epilogue == LX2V_EPI_NONE || epilogue == LX2V_EPI_GELU_TANH || epilogue == LX2V_EPI_RESIDUAL, LX2V_EINVAL, "gemm: unknown epilogue %d", epilogue);
  if (epilogue == LX2V_EPI_RESIDUAL) {
    LX2V_REQUIRE(resid != nullptr && ldr % 8 == 0 && ldr >= N && lx2v_aligned16(resid), LX2V_EINVAL, "gemm: residual epilogue needs an aligned resid with ldr >= N");
    LX2V_REQUIRE(gate == nullptr || lx2v_aligned16(gate), LX2V_EINVAL, "gemm: gate must be 16-byte aligned");
  }
  if (M == 0) return LX2V_OK;

  GemmParams p;
  p.A = A; p.W = W; p.bias = bias; p.C = C; p.resid = resid; p.gate = gate;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.a_kseg = ks.seg; p.a_kseg_stride = ks.stride;
  p.a_split = ks.seg != 0 ? ks.split : 0; p.a_kseg2 = ks.seg2; p.a_kseg2_stride = ks.stride2; p.a_base2 = ks.base2; p.lda2 = ks.split > 0 ? ks.lda2 : lda;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_m = (int)((M + BM - 1) / BM);
  p.tiles_n = (int)((N + BN - 1) / BN);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  LX2V_REQUIRE(nwg < (1LL << 31), LX2V_EUNSUP, "gemm: too many tiles");
  // extents, for the bounds of the buffer descriptors the persistent kernel stages through
  int64_t a_end = (M - 1) * lda + K;
  if (ks.seg != 0) {
    const int64_t k1 = ks.split > 0 ? ks.split : K;
    a_end = (k1 / ks.seg - 1) * ks.stride + (M - 1) * lda + ks.seg;
    if (ks.split > 0) {
      const int64_t e2 = ks.base2 + ((K - ks.split) / ks.seg2 - 1) * ks.stride2 + (M - 1) * ks.lda2 + ks.seg2;
      a_end = e2 > a_end ? e2 : a_end;
    }
  }
  p.a_bytes = a_end * 2;
  p.w_bytes = ((N - 1) * ldw + K) * 2;
  const dim3 grid((unsigned)nwg), block(NTHREADS);
  hipStream_t st = (hipStream_t)stream;
  // Kernel choice (option gemm_kernel): 0 = the persistent one-wave-per-SIMD kernel, 1 = 8 waves with
  // one barrier per K-tile, 2 = 8 waves in the ping-pong schedule.  All three give identical results.
  const int which = lx2v_options().gemm_kernel;
  if (which == 0 && p.a_bytes < (1LL << 32) && p.w_bytes < (1LL << 32) && ldc < (1LL << 22)) {  // (its stores address one 256-row panel of C with 31 bits)
    const int rc = lx2v_gemm_launch_w4(p, epilogue, workspace, workspace_bytes, st);
    if (rc != LX2V_OK) return rc;
    LX2V_CHECK_LAUNCH("gemm(persistent)");
    return LX2V_OK;
  }
  const bool simple = which == 1;
  if (simple || K < 2 * BK) {
    switch (epilogue) {
      case LX2V_EPI_NONE: hipLaunchKernelGGL(gemm_bf16_kernel<LX2V_EPI_NONE>, grid, block, 0, st, p); break;
      case LX2V_EPI_GELU_TANH: hipLaunchKernelGGL(gemm_bf16_kernel<LX2V_EPI_GELU_TANH>, grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL(gemm_bf16_kernel<LX2V_EPI_RESIDUAL>, grid, block, 0, st, p); break;
    }
  } else {
    switch (epilogue) {
      case LX2V_EPI_NONE: hipLaunchKernelGGL(gemm_bf16_pp_kernel<LX2V_EPI_NONE>, grid, block, 0, st, p); break;
      case LX2V_EPI_GELU_TANH: hipLaunchKernelGGL(gemm_bf16_pp_kernel<LX2V_EPI_GELU_TANH>, grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL(gemm_bf16_pp_kernel<LX2V_EPI_RESIDUAL>, grid, block, 0, st, p); break;
    }
  }
  LX2V_CHECK_LAUNCH("gemm");
  return LX2V_OK;
}
