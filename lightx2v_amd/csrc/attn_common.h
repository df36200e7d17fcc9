// Shared pieces of the attention kernels (parameters, LDS image layouts, small device helpers).
#pragma once

#include "common.h"

namespace lx2v_attn {

constexpr int HD = 128;       // head dim
constexpr int QROWS = 256;    // query rows per workgroup
constexpr int KVB = 64;       // keys per tile
constexpr int NTHREADS = 512;  // 8-wave kernels
constexpr int KV_TILE_BYTES = KVB * HD * 2;  // 16 KiB

struct AttnParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* out;
  const float* rope;  // [tokens][128] cos(64) || sin(64), or null
  int64_t ldq, ldk, ldv, ldo;
  int sq, skv, heads, nqb;
  float c;  // scale * log2(e)
  // Key split (16x16x32 kernel only).  nparts > 1: workgroup ids run part-major over `nparts` copies of the
  // (head, query block) grid, part j attends keys [j * part_keys, (j + 1) * part_keys) and leaves, instead of
  // normalised bf16 rows, its unnormalised O (fp32 [nparts][heads][sq][128] at part_o) and per row the pair
  // (m, l) (fp32 [nparts][heads][sq][2] at part_ml) for lx2v's merge kernel.
  float* part_o = nullptr;
  float* part_ml = nullptr;
  int nparts = 1, part_keys = 0;
  void* workspace = nullptr;       // launcher-side only: where a plan with a key split may put its partial results
  int64_t workspace_bytes = 0;
};

LX2V_DEVICE int k_lds_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
LX2V_DEVICE int v_lds_off(int row, int ch) {
  return 2048 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

LX2V_DEVICE bf16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
}

// rotate the 4 interleaved pairs of one 8-element chunk: pairs pair0..pair0+3 of table row `cs`
LX2V_DEVICE u32x4 rope_chunk(u32x4 v, const float* __restrict__ cs, int pair0) {
  float f[8];
  unpack8(v, f);
  const f32x4 co = *reinterpret_cast<const f32x4*>(cs + pair0);
  const f32x4 si = *reinterpret_cast<const f32x4*>(cs + 64 + pair0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = f[2 * i], b = f[2 * i + 1];
    f[2 * i] = a * co[i] - b * si[i];
    f[2 * i + 1] = a * si[i] + b * co[i];
  }
  return pack8(f);
}

LX2V_DEVICE float half_swap_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
LX2V_DEVICE float half_swap_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Wave-uniform buffer descriptor over [base, base + bytes): loads past the end return 0, which is
// how key rows >= skv of the last tile become zero rows without any per-lane clamping.
LX2V_DEVICE __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  void* pb = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(pb, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int N>
struct IntC {
  static constexpr int value = N;
};

// f(IntC<B>{}), f(IntC<B+1>{}), ... f(IntC<E-1>{}): an unrolled loop whose index is a constant
// expression in every step (no reliance on the loop unroller).
template <int B, int E, class F>
LX2V_DEVICE void static_for(F&& f) {
  if constexpr (B < E) {
    f(IntC<B>{});
    static_for<B + 1, E>(f);
  }
}

// Rescale threshold in log2 units: O and l are rescaled only on tiles where some row's running
// maximum would grow by more than this, so P stays <= 2^kRescaleThr (bf16 keeps its relative
// precision at any magnitude; O and l accumulate in fp32).
constexpr float kRescaleThr = 6.0f;

}  // namespace lx2v_attn
