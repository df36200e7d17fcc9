// Pieces shared by the two fp8 attention kernels (attn_fp8.hip: quantisation pre-pass, 8-wave kernel
// and entry points; attn_fp8_w4.hip: one-wave-per-SIMD kernel): the per-tile K/V image and the
// kernel parameters.
#pragma once

#include "attn_common.h"

namespace lx2v_attn_fp8 {

typedef int i32x8 __attribute__((ext_vector_type(8)));

// One 64-key tile of one head, as the pre-pass writes it and the kernels stage it: K rows (128 B each,
// 16-byte chunks swizzled), K scales, V^T rows (64 B per channel, chunks swizzled), V scales.
constexpr int K_DATA = 0, K_SCALE = 8192, V_DATA = 8704, V_SCALE = 16896, TILE_BLOB = 17152;
constexpr int K_PART = V_DATA;              // bytes [0, 8704): K rows + K scales
constexpr int V_PART = TILE_BLOB - V_DATA;  // bytes [8704, 17152): V^T rows + V scales
constexpr int UNIT_SCALE = 127;             // E8M0 byte of 2^0

struct Fp8AttnParams {
  const uint8_t* q8;       // [sq][heads][128]
  const uint8_t* qs;       // [sq][heads][4]
  const uint8_t* kv;       // [heads][tiles][TILE_BLOB]
  uint16_t* out;
  int64_t ldo;
  int sq, skv, heads, nqb, tiles;
};

}  // namespace lx2v_attn_fp8

// One wave per SIMD, 4 waves x 64 query rows (attn_fp8_w4.hip); p.nqb = workgroups per head.
int lx2v_attn_fp8_launch_w4(const lx2v_attn_fp8::Fp8AttnParams& p, hipStream_t st);
