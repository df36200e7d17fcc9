// Shared by the W8A8 GEMM kernels (gemm_fp8.hip: 8-wave schedules; gemm_w8a8_w4.hip: the persistent
// one-wave-per-SIMD kernel): tile constants, parameters and the tile rasterisation.
#pragma once

#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace lx2v_w8a8 {

struct GemmParams {
  const uint8_t* A;
  const uint8_t* W;
  const float* a_scale;
  const float* w_scale;
  const float* bias;
  uint16_t* C;
  const uint16_t* resid;
  const float* gate;
  int64_t lda, ldw, ldc, ldr;
  int M, N, K;
  int tiles_m, tiles_n;
  // Persistent kernel: the tiles it runs (virtual block ids below ntiles_run); the tiles of a sparsely filled last
  // round [tail_first, tiles_m * tiles_n) go to a second launch that splits their K range into tail_split parts of
  // tail_nk K-tiles (raw accumulators into `partials` in fragment order: fp32, or the exact int32 sums of int8) and
  // a third that adds the parts in a fixed order and applies scales and epilogue.
  int ntiles_run;
  int tail_first, tail_split, tail_nk;
  float* partials;
};

}  // namespace lx2v_w8a8

// The persistent kernel's launcher (gemm_w8a8_w4.hip), for p already validated by gemm_w8a8(), and the workspace a
// call of that shape can use (0: its last round is not split).
int lx2v_gemm_w8a8_launch_w4(lx2v_w8a8::GemmParams& p, bool int8, int epilogue, void* workspace, int64_t workspace_bytes, hipStream_t st);
int64_t lx2v_gemm_w8a8_w4_workspace_bytes(const lx2v_w8a8::GemmParams& p);

namespace {

using lx2v_w8a8::GemmParams;

constexpr int BM = 256, BN = 256, BK = 128;  // BK in fp8 / int8 elements = bytes
constexpr int TILE_BYTES = BM * BK;          // 32 KiB, same for A and W
constexpr int GROUP_M = 8;                   // tile rasterisation: 8 row-panels share one column sweep

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

LX2V_DEVICE void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)l, 16, 0, 0);
}

// Bijective remap so that blocks sharing an XCD (same blockIdx % 8) get a contiguous run of tiles.
// Dequantise one sum and add the bias: (v * sa) * sw + b, the last step fused, in exactly this association.  Every
// W8A8 kernel goes through here so that the schedules agree bit for bit; under -ffast-math the plain expression
// is re-associated differently from one kernel (and one build) to the next.
LX2V_DEVICE float dequant_bias(float v, float sa, float sw, float b) {
#pragma clang fp reassociate(off) contract(off)
  const float t = v * sa;
  return __builtin_fmaf(t, sw, b);
}

LX2V_DEVICE int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Virtual block id (blockIdx, or round * grid + blockIdx of the persistent kernel) -> tile origin:
// XCD-aware, grouped along M (GROUP_M row panels share one sweep over the columns).
LX2V_DEVICE void tile_origin(const GemmParams& p, int vbid, int& m0, int& n0) {
  const int nwg = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(vbid, nwg);
  const int group_sz = GROUP_M * p.tiles_n;
  const int g = t / group_sz;
  const int first_m = g * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_g = t - g * group_sz;
  m0 = (first_m + in_g % gm) * BM;
  n0 = (in_g / gm) * BN;
}

}  // namespace
