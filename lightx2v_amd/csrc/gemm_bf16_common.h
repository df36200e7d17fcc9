// Shared by the bf16 GEMM kernels (gemm_bf16.hip: 8-wave schedules; gemm_bf16_w4.hip: the
// persistent one-wave-per-SIMD kernel): tile constants, parameters, the segmented-A cursor and the
// tile rasterisation.
#pragma once

#include "common.h"

namespace lx2v_gemm {

struct GemmParams {
  const uint16_t* A;
  const uint16_t* W;
  const float* bias;
  uint16_t* C;
  const uint16_t* resid;
  const float* gate;
  int64_t lda, ldw, ldc, ldr;
  int64_t a_kseg, a_kseg_stride;  // a_kseg == 0: plain matrix
  int64_t a_split, a_kseg2, a_kseg2_stride, a_base2, lda2;  // a_split > 0: from K column a_split on, segments of a_kseg2 columns starting at A + a_base2, row stride lda2
  int64_t a_bytes, w_bytes;  // extent of A / W from their first element to the end of the last row read
  int M, N, K;
  int tiles_m, tiles_n;
  // Persistent kernel: the tiles it runs (virtual block ids below ntiles_run).  When the last round of one tile per
  // CU would be sparsely filled, its tiles [tail_first, tiles_m * tiles_n) are left to a second launch that splits
  // their K range into tail_split parts of tail_nk K-tiles (one workgroup per part, raw fp32 accumulators into
  // `partials` in fragment order) and a third that sums the parts in a fixed order and applies the epilogue.
  int ntiles_run;
  int tail_first, tail_split, tail_nk;
  float* partials;
};

}  // namespace lx2v_gemm

namespace {

using lx2v_gemm::GemmParams;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 32 KiB, same for A and W
#ifndef LX2V_GROUP_M
#define LX2V_GROUP_M 8
#endif
constexpr int GROUP_M = LX2V_GROUP_M;    // tile rasterisation: 8 row-panels share one column sweep
#ifndef GEMM_ABL
#define GEMM_ABL 0  // timing ablations (wrong results): 1 no load wait, 2 no staging, 4 no barrier, 8 no priority raise (ping-pong), 16 no fragment reads (ping-pong), 32 no epilogue (ping-pong)
#endif


template <int N>
struct IntC {
  static constexpr int value = N;
};

// f(IntC<B>{}), ... f(IntC<E-1>{}): an unrolled loop whose index is a constant expression in every step.
template <int B, int E, class F>
LX2V_DEVICE void static_for(F&& f) {
  if constexpr (B < E) {
    f(IntC<B>{});
    static_for<B + 1, E>(f);
  }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

LX2V_DEVICE void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)l, 16, 0, 0);
}

// Position of a K-tile inside A when A's columns are stored as segments (see lx2v_kseg): `koff` is
// the element offset of the tile from the row start; advance() moves one K-tile on.  All fields are
// wave-uniform.
struct ACursor {
  int64_t koff = 0, within = 0, kpos = 0, seg, stride;
  LX2V_DEVICE explicit ACursor(const GemmParams& p) : seg(p.a_kseg), stride(p.a_kseg_stride) {}
  // Returns true when the cursor has just entered region 2 (the caller then switches its per-lane
  // row offsets from lda to lda2).
  LX2V_DEVICE bool advance(const GemmParams& p) {
    koff += BK;
    if (p.a_kseg != 0) {
      within += BK;
      kpos += BK;
      if (kpos == p.a_split) {  // never true for a_split == 0 (kpos >= BK)
        koff = p.a_base2;
        within = 0;
        seg = p.a_kseg2;
        stride = p.a_kseg2_stride;
        return true;
      }
      if (within == seg) {
        within = 0;
        koff += stride - seg;
      }
    }
    return false;
  }
};

// Bijective remap so that blocks sharing an XCD (same blockIdx % 8) get a contiguous run of tiles.
LX2V_DEVICE int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Virtual block id (blockIdx, or round * grid + blockIdx of a persistent kernel) -> tile origin:
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
