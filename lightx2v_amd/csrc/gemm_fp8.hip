// fp8 (OCP e4m3) W8A8 Linear on the CDNA4 block-scaled MFMA for gfx950:
//   C[M,N] = epilogue((A8[M,K] @ W8[N,K]^T) * a_scale[m] * w_scale[n] + bias)
// with a per-token (row of A) and a per-output-channel (row of W) fp32 scale, bf16 output.
//
// Same structure as the bf16 GEMM (256 x 256 tile, 8 waves as 2 x 4, LDS-DMA staging, the same
// 128-byte-row LDS image and swizzle); a K-tile is 128 fp8 values = the same 128 bytes per row, and
// one v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales) covers its whole depth: half as many MFMAs
// per byte staged, each at twice the bf16 rate.  A lane's operand is 32 consecutive K bytes
// (k-block lane/16), i.e. two adjacent 16-byte chunks of the swizzled row.
//
// The one-barrier kernel also serves int8 W8A8 (lx2v_gemm_int8): same tiles, same LDS image, the
// lane's 32 K bytes go through two v_mfma_i32_16x16x64_i8 (first and second 16 bytes; both operands
// use the same split, so every K index meets its partner), accumulation is exact in int32 and the
// accumulators are converted to fp32 in front of the shared epilogue.
#include "gemm_fp8_common.h"

namespace {

constexpr int NTHREADS = 512;

// Epilogue shared by both schedules: scales, bias and the optional GELU / gate in fp32 on the
// accumulators -> per-wave LDS image (bf16) -> whole 128-B row segments.
// acc[nt][mt][j] = C'[m = wm*128 + mt*16 + fr][n = wn*64 + nt*16 + fq*4 + j].  Every wave of the
// workgroup must have finished reading the operand tiles in `lds`.
template <int EPI>
LX2V_DEVICE void gemm_fp8_epilogue(const GemmParams& p, f32x4 (&acc)[4][8], char* lds, int m0, int n0, int wid, int wm, int wn, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  char* stg = lds + wid * 16384;  // 128 rows x 128 B per wave
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int ncol = n0 + wn * 64 + nt * 16 + fq * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f}, sw = {0.f, 0.f, 0.f, 0.f};
    if (ncol < p.N) {
      sw = *reinterpret_cast<const f32x4*>(p.w_scale + ncol);
      if (p.bias != nullptr) bv = *reinterpret_cast<const f32x4*>(p.bias + ncol);
      if (EPI == LX2V_EPI_RESIDUAL && p.gate != nullptr) gv = *reinterpret_cast<const f32x4*>(p.gate + ncol);
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      float v[4];
      const float sa = p.a_scale[min(m0 + wm * 128 + mt * 16 + fr, p.M - 1)];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = dequant_bias(acc[nt][mt][j], sa, sw[j], bv[j]);
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

template <int EPI, bool INT8 = false>
__global__ __launch_bounds__(NTHREADS) void gemm_fp8_kernel(const GemmParams p) {
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
  const uint8_t* a_src[4];
  const uint8_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = i * 8192 + tid * 16;
    const int row = o >> 7;
    const int chunk = ((o >> 4) & 7) ^ ((row >> 1) & 7);
    const int ar = min(m0 + row, p.M - 1);
    const int wr = min(n0 + row, p.N - 1);
    a_src[i] = p.A + (int64_t)ar * p.lda + chunk * 16;
    w_src[i] = p.W + (int64_t)wr * p.ldw + chunk * 16;
  }
  auto stage = [&](int s, int kt) {
    char* la = lds + s * 2 * TILE_BYTES + wid * 1024;
    char* lw = la + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + (int64_t)kt * BK, la + i * 8192);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w_src[i] + (int64_t)kt * BK, lw + i * 8192);
  };

  // ---- fragment read offsets ----
  // A operand of the MFMA = W fragment: row n = wn*64 + nt*16 + (lane & 15), K bytes 32*(lane >> 4) .. +31
  // B operand of the MFMA = A fragment: row m = wm*128 + mt*16 + (lane & 15)
  const int fr = lane & 15, fq = lane >> 4;
  int a_rd[2], w_rd[2];  // byte offsets of the two 16-byte chunks at mt/nt = 0 (rows step by 16 -> +2048 B, swizzle unchanged)
  {
    const int arow = wm * 128 + fr, wrow = wn * 64 + fr;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      a_rd[c] = arow * 128 + (((2 * fq + c) ^ ((arow >> 1) & 7)) << 4);
      w_rd[c] = wrow * 128 + (((2 * fq + c) ^ ((wrow >> 1) & 7)) << 4);
    }
  }
  auto frag = [&](const char* base, const int (&rd)[2], int t) __attribute__((always_inline)) -> i32x8 {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(base + rd[0] + t * 2048);
    const u32x4 hi = *reinterpret_cast<const u32x4*>(base + rd[1] + t * 2048);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };

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
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* la = lds + cur * 2 * TILE_BYTES;
    const char* lw = la + TILE_BYTES;
    i32x8 wf[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[nt] = frag(lw, w_rd, nt);
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {  // two halves of the wave's 8 row blocks: 64 fragment registers live at a time
      i32x8 af[4];
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) af[m4] = frag(la, a_rd, mh * 4 + m4);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
          if constexpr (INT8) {
            // int32 accumulators in the same registers (bit pattern kept in the f32x4 slots until the loop ends)
            i32x4 c = __builtin_bit_cast(i32x4, acc[nt][mh * 4 + m4]);
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(i32x4{wf[nt][0], wf[nt][1], wf[nt][2], wf[nt][3]}, i32x4{af[m4][0], af[m4][1], af[m4][2], af[m4][3]}, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(i32x4{wf[nt][4], wf[nt][5], wf[nt][6], wf[nt][7]}, i32x4{af[m4][4], af[m4][5], af[m4][6], af[m4][7]}, c, 0, 0, 0);
            acc[nt][mh * 4 + m4] = __builtin_bit_cast(f32x4, c);
          } else {
            acc[nt][mh * 4 + m4] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], af[m4], acc[nt][mh * 4 + m4], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if constexpr (INT8) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const i32x4 c = __builtin_bit_cast(i32x4, acc[nt][mt]);
        acc[nt][mt] = f32x4{(float)c[0], (float)c[1], (float)c[2], (float)c[3]};
      }
  }

  gemm_fp8_epilogue<EPI>(p, acc, lds, m0, n0, wid, wm, wn, lane);
}


// ---------------------------------------------------------------------------------------------
// Ping-pong variant (opt-in, see the launcher): the schedule of the bf16 GEMM's ping-pong kernel on
// the fp8 operands.  Two wave groups one barrier apart; a K-tile (128 bytes of K) is four quadrant
// phases of 8 MFMAs; staging units are the row sets one phase reads (Am0, Bn0, Bn1, Am1, 16 KiB each),
// streamed in that fixed order with four units in flight behind a counted vmcnt(8):
//     phase 1 of tile u issues Bn1(u+1), phase 2 Am1(u+1), phase 3 Am0(u+2), phase 4 Bn0(u+2),
// each unit read one phase after the wait + barrier that retires it.
constexpr int A_BUF = BM * BK;       // 32 KiB per operand per buffer
constexpr int W_BASE = 2 * A_BUF;    // LDS: [A buf0][A buf1][W buf0][W buf1]
enum { TILE_FULL = 0, TILE_PENULT = 1, TILE_LAST = 2 };

template <int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_fp8_pp_kernel(const GemmParams p) {
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
  // unit row r = i*64 + wid*8 + (lane >> 3), physical chunk = lane & 7; sources are a wave-uniform
  // base (tile origin + K offset) plus a per-lane 32-bit byte offset inside the tile's row panel.
  uint32_t soff[4][2];
  int dst[4][2];
  const char* a_base = reinterpret_cast<const char*>(p.A + (int64_t)m0 * p.lda);
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
      soff[U][i] = (uint32_t)((int64_t)rr * (is_a ? p.lda : p.ldw) + chunk * 16);
      dst[U][i] = (is_a ? 0 : W_BASE) + row0 * 128;
    }
  auto issue = [&](int U, int buf, int tile_idx) {
    const char* base = ((U == 0 || U == 3) ? a_base : w_base) + (int64_t)tile_idx * BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(base + soff[U][i], lds + dst[U][i] + buf * A_BUF);
  };

  // fragment = the lane's 32 consecutive K bytes = two adjacent 16-byte chunks of the swizzled row
  const int fr = lane & 15, fq = lane >> 4;
  int a_rd[2], w_rd[2];
  {
    const int arow = wm * 128 + fr, wrow = wn * 64 + fr;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      a_rd[c] = arow * 128 + (((2 * fq + c) ^ ((arow >> 1) & 7)) << 4);
      w_rd[c] = W_BASE + wrow * 128 + (((2 * fq + c) ^ ((wrow >> 1) & 7)) << 4);
    }
  }

  f32x4 acc[4][8];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;  // >= 2 (checked by the launcher)

  // ---- prologue: Am0(0) Bn0(0) Bn1(0) Am1(0) Am0(1) Bn0(1) ----
  issue(0, 0, 0);
  issue(1, 0, 0);
  issue(2, 0, 0);
  issue(3, 0, 0);
  issue(0, 1, 1);
  issue(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind

  i32x8 af[4], b0[2], b1[2];
  auto frag = [&](const int (&rd)[2], int off) __attribute__((always_inline)) -> i32x8 {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(lds + rd[0] + off);
    const u32x4 hi = *reinterpret_cast<const u32x4*>(lds + rd[1] + off);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
#define LX2V_FP8_MFMA(W_, A_, C_) C_ = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(W_, A_, C_, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)

  auto tile = [&](auto buf_c, int u) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr int RD = BUF * A_BUF;
    const int KIND = u + 2 < nk ? TILE_FULL : (u + 1 < nk ? TILE_PENULT : TILE_LAST);  // wave-uniform
    // ---- phase 1: quadrant (m0, n0) ----
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b0[nt] = frag(w_rd, RD + nt * 2048);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) af[mt] = frag(a_rd, RD + mt * 2048);
    if (KIND != TILE_LAST) issue(2, BUF ^ 1, u + 1);
    if (KIND == TILE_LAST) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) LX2V_FP8_MFMA(b0[nt], af[mt], acc[nt][mt]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: quadrant (m0, n1) ----
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b1[nt] = frag(w_rd, RD + (nt + 2) * 2048);
    if (KIND != TILE_LAST) issue(3, BUF ^ 1, u + 1);
    if (KIND == TILE_LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) LX2V_FP8_MFMA(b1[nt], af[mt], acc[nt + 2][mt]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: quadrant (m1, n1) ----
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) af[mt] = frag(a_rd, RD + (mt + 4) * 2048);
    if (KIND == TILE_FULL) issue(0, BUF, u + 2);
    if (KIND == TILE_FULL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (KIND == TILE_PENULT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) LX2V_FP8_MFMA(b1[nt], af[mt], acc[nt + 2][mt + 4]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: quadrant (m1, n0), operands already in registers ----
    if (KIND == TILE_FULL) issue(1, BUF, u + 2);
    if (KIND == TILE_FULL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (KIND == TILE_PENULT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) LX2V_FP8_MFMA(b0[nt], af[mt], acc[nt][mt + 4]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
#undef LX2V_FP8_MFMA
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  for (int u = 0; u < nk; u += 2) {
    tile(B0{}, u);
    if (u + 1 < nk) tile(B1{}, u + 1);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // group 0 gives back the barrier group 1 spent at the start

  gemm_fp8_epilogue<EPI>(p, acc, lds, m0, n0, wid, wm, wn, lane);
}

}  // namespace

static int gemm_w8a8(bool int8, const uint8_t* A, int64_t lda, const float* a_scale, const uint8_t* W, int64_t ldw, const float* w_scale,
                     const float* bias, uint16_t* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                     const uint16_t* resid, int64_t ldr, const float* gate, void* workspace, int64_t workspace_bytes, void* stream) {
  LX2V_REQUIRE(M == 0 || (A && W && C && a_scale && w_scale), LX2V_EINVAL, "gemm_fp8: null tensor");
  LX2V_REQUIRE(M >= 0 && N > 0 && K > 0, LX2V_EINVAL, "gemm_fp8: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  LX2V_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), LX2V_EUNSUP, "gemm_fp8: dimensions must fit in 31 bits");
  LX2V_REQUIRE(K % BK == 0, LX2V_EUNSUP, "gemm_fp8: K=%lld must be a multiple of %d", (long long)K, BK);
  LX2V_REQUIRE(N % 8 == 0, LX2V_EUNSUP, "gemm_fp8: N=%lld must be a multiple of 8", (long long)N);
  LX2V_REQUIRE(lda % 16 == 0 && ldw % 16 == 0 && ldc % 8 == 0, LX2V_EINVAL, "gemm_fp8: fp8 row strides must be multiples of 16, ldc of 8");
  LX2V_REQUIRE(lda >= K && ldw >= K && ldc >= N, LX2V_EINVAL, "gemm_fp8: row stride smaller than the row length");
  LX2V_REQUIRE(lx2v_aligned16(A) && lx2v_aligned16(W) && lx2v_aligned16(C) && lx2v_aligned16(w_scale), LX2V_EINVAL, "gemm_fp8: tensors must be 16-byte aligned");
  LX2V_REQUIRE(bias == nullptr || lx2v_aligned16(bias), LX2V_EINVAL, "gemm_fp8: bias must be 16-byte aligned");
  LX2V_REQUIRE(epilogue == LX2V_EPI_NONE || epilogue == LX2V_EPI_GELU_TANH || epilogue == LX2V_EPI_RESIDUAL, LX2V_EINVAL, "gemm_fp8: unknown epilogue %d", epilogue);
  if (epilogue == LX2V_EPI_RESIDUAL) {
    LX2V_REQUIRE(resid != nullptr && ldr % 8 == 0 && ldr >= N && lx2v_aligned16(resid), LX2V_EINVAL, "gemm_fp8: residual epilogue needs an aligned resid with ldr >= N");
    LX2V_REQUIRE(gate == nullptr || lx2v_aligned16(gate), LX2V_EINVAL, "gemm_fp8: gate must be 16-byte aligned");
  }
  if (M == 0) return LX2V_OK;

  GemmParams p;
  p.A = A; p.W = W; p.a_scale = a_scale; p.w_scale = w_scale; p.bias = bias; p.C = C; p.resid = resid; p.gate = gate;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_m = (int)((M + BM - 1) / BM);
  p.tiles_n = (int)((N + BN - 1) / BN);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  LX2V_REQUIRE(nwg < (1LL << 31), LX2V_EUNSUP, "gemm_fp8: too many tiles");
  const dim3 grid((unsigned)nwg), block(NTHREADS);
  hipStream_t st = (hipStream_t)stream;
  if (int8) LX2V_REQUIRE(K <= 131072, LX2V_EUNSUP, "gemm_int8: K=%lld: the int32 accumulators hold 127 * 127 * K up to K = 131072", (long long)K);
  // Default (gemm_kernel = 0): the persistent one-wave-per-SIMD kernel.  Its staging addresses rows through
  // 32-bit buffer offsets, so operands whose extent reaches 4 GiB stay with the 8-wave kernels.
  if (lx2v_options().gemm_kernel == 0 && M * lda < (1LL << 32) && N * ldw < (1LL << 32) && ldc < (1LL << 22)) {  // (its stores address one 256-row panel of C with 31 bits)
    const int rc = lx2v_gemm_w8a8_launch_w4(p, int8, epilogue, workspace, workspace_bytes, st);
    if (rc != LX2V_OK) return rc;
    LX2V_CHECK_LAUNCH(int8 ? "gemm_int8(w4)" : "gemm_fp8(w4)");
    return LX2V_OK;
  }
  // The 8-wave kernels (gemm_kernel = 1: one barrier per K-tile, 2.0-2.24 PFLOP/s at 14B shapes; 2: the ping-pong
  // schedule, 1.87-2.11, fp8 only): A/B switches, and the path of operands beyond 4 GiB.
  if (int8) {
    switch (epilogue) {
      case LX2V_EPI_NONE: hipLaunchKernelGGL((gemm_fp8_kernel<LX2V_EPI_NONE, true>), grid, block, 0, st, p); break;
      case LX2V_EPI_GELU_TANH: hipLaunchKernelGGL((gemm_fp8_kernel<LX2V_EPI_GELU_TANH, true>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((gemm_fp8_kernel<LX2V_EPI_RESIDUAL, true>), grid, block, 0, st, p); break;
    }
    LX2V_CHECK_LAUNCH("gemm_int8");
    return LX2V_OK;
  }
  const bool simple = lx2v_options().gemm_kernel != 2;
  if (simple || K < 2 * BK) {
    switch (epilogue) {
      case LX2V_EPI_NONE: hipLaunchKernelGGL(gemm_fp8_kernel<LX2V_EPI_NONE>, grid, block, 0, st, p); break;
      case LX2V_EPI_GELU_TANH: hipLaunchKernelGGL(gemm_fp8_kernel<LX2V_EPI_GELU_TANH>, grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL(gemm_fp8_kernel<LX2V_EPI_RESIDUAL>, grid, block, 0, st, p); break;
    }
  } else {
    switch (epilogue) {
      case LX2V_EPI_NONE: hipLaunchKernelGGL(gemm_fp8_pp_kernel<LX2V_EPI_NONE>, grid, block, 0, st, p); break;
      case LX2V_EPI_GELU_TANH: hipLaunchKernelGGL(gemm_fp8_pp_kernel<LX2V_EPI_GELU_TANH>, grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL(gemm_fp8_pp_kernel<LX2V_EPI_RESIDUAL>, grid, block, 0, st, p); break;
    }
  }
  LX2V_CHECK_LAUNCH("gemm_fp8");
  return LX2V_OK;
}

extern "C" int lx2v_gemm_fp8(const uint8_t* A, int64_t lda, const float* a_scale, const uint8_t* W, int64_t ldw, const float* w_scale,
                             const float* bias, uint16_t* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                             const uint16_t* resid, int64_t ldr, const float* gate, void* stream) {
  return gemm_w8a8(false, A, lda, a_scale, W, ldw, w_scale, bias, C, ldc, M, N, K, epilogue, resid, ldr, gate, nullptr, 0, stream);
}

extern "C" int lx2v_gemm_int8(const int8_t* A, int64_t lda, const float* a_scale, const int8_t* W, int64_t ldw, const float* w_scale,
                              const float* bias, uint16_t* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                              const uint16_t* resid, int64_t ldr, const float* gate, void* stream) {
  return gemm_w8a8(true, reinterpret_cast<const uint8_t*>(A), lda, a_scale, reinterpret_cast<const uint8_t*>(W), ldw, w_scale, bias, C, ldc, M, N, K, epilogue,
                   resid, ldr, gate, nullptr, 0, stream);
}

extern "C" int64_t lx2v_gemm_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || M >= (1LL << 31) || N >= (1LL << 31) || K >= (1LL << 31) || K % BK != 0) return 0;
  GemmParams p{};
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_m = (int)((M + BM - 1) / BM);
  p.tiles_n = (int)((N + BN - 1) / BN);
  if ((int64_t)p.tiles_m * p.tiles_n >= (1LL << 31) || lx2v_options().gemm_kernel != 0) return 0;
  return lx2v_gemm_w8a8_w4_workspace_bytes(p);
}

extern "C" int lx2v_gemm_fp8_ws(const uint8_t* A, int64_t lda, const float* a_scale, const uint8_t* W, int64_t ldw, const float* w_scale,
                                const float* bias, uint16_t* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                                const uint16_t* resid, int64_t ldr, const float* gate, void* workspace, int64_t workspace_bytes, void* stream) {
  return gemm_w8a8(false, A, lda, a_scale, W, ldw, w_scale, bias, C, ldc, M, N, K, epilogue, resid, ldr, gate, workspace, workspace_bytes, stream);
}

extern "C" int lx2v_gemm_int8_ws(const int8_t* A, int64_t lda, const float* a_scale, const int8_t* W, int64_t ldw, const float* w_scale,
                                 const float* bias, uint16_t* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                                 const uint16_t* resid, int64_t ldr, const float* gate, void* workspace, int64_t workspace_bytes, void* stream) {
  return gemm_w8a8(true, reinterpret_cast<const uint8_t*>(A), lda, a_scale, reinterpret_cast<const uint8_t*>(W), ldw, w_scale, bias, C, ldc, M, N, K, epilogue,
                   resid, ldr, gate, workspace, workspace_bytes, stream);
}
