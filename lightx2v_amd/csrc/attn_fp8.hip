// fp8 attention forward for gfx950 (CDNA4 block-scaled fp8 MFMA), head_dim 128, bf16 in / bf16 out,
// non-causal.  Both contractions run on v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 MFMA rate):
//
//   S^T = K (cQ)^T   Q and K are quantised to e4m3 in blocks of 32 channels with a power-of-two
//                    (E8M0) scale per (token, head, block) -- the MX layout the instruction
//                    applies by itself, so the accumulators are the true scaled scores and the
//                    online softmax is the bf16 kernel's: the running maximum enters as the
//                    accumulator init, p = exp2(s - m), the rescale of O is skipped while no row
//                    maximum grows by more than 2^6.
//   O^T += V^T P^T   P (<= 2^6) is converted to e4m3 in registers: the S^T accumulator layout (query
//                    row on the lane, 16 keys of each 32-key block in the lane's registers) IS the
//                    B operand once the contraction slots are numbered to match, slot j of lane
//                    half h = key 32 (j >> 4) + 8 ((j & 15) >> 2) + 4 h + (j & 3).  V is quantised
//                    with one E8M0 scale per (channel, 32 consecutive keys) and stored transposed in
//                    exactly that slot order, so its fragment is two 16-byte LDS reads.
//
// Operand / scale layout of the instruction (measured, tools/probe_mfma_scale.hip): a lane (row r,
// half h) supplies 32 bytes; bytes 0-15 are elements 16 h .. 16 h + 15 of scale block 0, bytes 16-31
// the same elements of scale block 1 (a block = 32 consecutive k).  The scale of block b is read
// from lane (r, half b), byte `opsel` of the scale register.
//
// A quantisation pre-pass writes, per head and 64-key tile, one 16.75 KiB blob that is already the
// LDS image (K rows swizzled for conflict-free ds_read_b128, K scales, V^T rows swizzled, V scales;
// every scale sits in the byte an immediate `opsel` picks, so no lane extracts one); staging a tile
// is a linear copy, done by LDS-DMA.  Workgroup = 8 waves x 32 query rows; the blobs are
// double-buffered in LDS, tile t+1 is in flight while tile t is consumed.
#include <mutex>

#include "attn_fp8_common.h"

using namespace lx2v_attn;
using namespace lx2v_attn_fp8;

namespace {

constexpr int PAIR_BLOB = 2 * TILE_BLOB;  // staging unit: two consecutive tiles (one barrier per unit)

// Smallest power of two 2^e with amax / 2^e <= 448 (the e4m3 maximum), from the bits of amax:
// 448 = 1.75 * 2^8, so e = exponent(amax) - 8, one more when the mantissa exceeds 1.75.  Returns the
// biased E8M0 byte, clamped to [1, 254]; amax = 0 gives 1 (every value quantises to 0).
LX2V_DEVICE int e8m0_for(float amax) {
  const uint32_t b = __float_as_uint(amax);
  int e = (int)(b >> 23) - 8 + ((b & 0x7fffffu) > 0x600000u ? 1 : 0);
  return min(max(e, 1), 254);
}
LX2V_DEVICE float inv_scale(int e8) { return __uint_as_float((uint32_t)(254 - e8) << 23); }  // 2^-(e8-127)

LX2V_DEVICE u32x2 quant8(const float (&f)[8], float inv) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
  return u32x2{(uint32_t)lo, (uint32_t)hi};
}

// Q (IS_K = false): x * premul -> e4m3 [token][head][128] + E8M0 scales [token][head][4].
// K (IS_K = true): the K part of the tile blobs, rows past `tokens` zero-filled up to tiles * 64.
// One thread = 8 channels; the 4 threads of a 32-channel block share the block maximum.
template <bool IS_K>
__global__ __launch_bounds__(256) void mxq_rows_kernel(const uint16_t* __restrict__ x, int64_t ldx, int tokens, int heads, float premul,
                                                       uint8_t* __restrict__ out, uint8_t* __restrict__ scales, int tiles) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c16 = (int)(gid & 15);
  const int64_t th = gid >> 4;
  const int head = (int)(th % heads);
  const int64_t token = th / heads;
  const int64_t limit = IS_K ? (int64_t)tiles * KVB : tokens;
  if (token >= limit) return;  // whole 4-lane groups leave together (16 threads per (token, head))
  float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (token < tokens) {
    unpack8(*reinterpret_cast<const u32x4*>(x + token * ldx + head * HD + c16 * 8), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] *= premul;
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(f[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  const int e8 = e8m0_for(amax);
  const u32x2 q = quant8(f, inv_scale(e8));
  if (IS_K) {
    const int64_t tile = token >> 6;
    const int key = (int)(token & 63);
    uint8_t* blob = out + ((int64_t)head * tiles + tile) * TILE_BLOB;
    *reinterpret_cast<u32x2*>(blob + K_DATA + key * 128 + (((c16 >> 1) ^ ((key >> 1) & 7)) << 4) + (c16 & 1) * 8) = q;
    if ((c16 & 3) == 0) {  // block b = c16 >> 2 = 2 ks + h is byte ks of the dword lane half h reads
      const int b = c16 >> 2;
      blob[K_SCALE + (key * 2 + (b & 1)) * 4 + (b >> 1)] = (uint8_t)e8;
    }
  } else {
    const int64_t row = token * heads + head;
    *reinterpret_cast<u32x2*>(out + row * HD + c16 * 8) = q;
    if ((c16 & 3) == 0) scales[row * 4 + (c16 >> 2)] = (uint8_t)e8;
  }
}

// V part of the tile blobs: thread = two adjacent channels of one (tile, head); reads their 64 keys as dwords (each
// key row is a coalesced 256-byte read over the block's 64 lanes), quantises the two groups of 32 consecutive keys per
// channel and writes the two 64-byte V^T rows in slot order.
__global__ __launch_bounds__(64) void mxq_v_kernel(const uint16_t* __restrict__ v, int64_t ldv, int tokens, int heads, uint8_t* __restrict__ out, int tiles) {
  const int tile = blockIdx.x, head = blockIdx.y, d0 = 2 * threadIdx.x;
  float val[2][64];
  const bool dword_ok = (ldv & 1) == 0 && (reinterpret_cast<uintptr_t>(v) & 3) == 0;   // every row's channel pair is a dword
#pragma unroll
  for (int key = 0; key < 64; ++key) {
    const int64_t tok = (int64_t)tile * KVB + key;
    uint32_t pair = 0;
    if (tok < tokens) {
      const uint16_t* src = v + tok * ldv + head * HD + d0;
      pair = dword_ok ? *reinterpret_cast<const uint32_t*>(src) : ((uint32_t)src[0] | ((uint32_t)src[1] << 16));
    }
    val[0][key] = __uint_as_float(pair << 16);
    val[1][key] = __uint_as_float(pair & 0xffff0000u);
  }
  uint8_t* blob = out + ((int64_t)head * tiles + tile) * TILE_BLOB;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    const int d = d0 + ch;
    int e8[2];
    float inv[2];
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {  // scale groups: keys 0-31 and 32-63 of the tile
      float amax = 0.f;
#pragma unroll
      for (int key = 0; key < 32; ++key) amax = fmaxf(amax, fabsf(val[ch][32 * grp + key]));
      e8[grp] = e8m0_for(amax);
      inv[grp] = inv_scale(e8[grp]);
      blob[V_SCALE + (grp * 32 + (d & 31)) * 4 + (d >> 5)] = (uint8_t)e8[grp];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 2; ++c) {  // 16-slot chunk c of half h = keys 32 c + 8 (j >> 2) + 4 h + (j & 3), j = 0..15
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          a[i] = val[ch][32 * c + 8 * (i >> 2) + 4 * h + (i & 3)];
          b[i] = val[ch][32 * c + 8 * ((i + 8) >> 2) + 4 * h + (i & 3)];
        }
        const u32x2 lo = quant8(a, inv[c]), hi = quant8(b, inv[c]);
        const int chunk = 2 * h + c;
        *reinterpret_cast<u32x4*>(blob + V_DATA + d * 64 + ((chunk ^ ((d >> 2) & 3)) << 4)) = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
  }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
LX2V_DEVICE void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)l, 16, 0, 0); }

LX2V_DEVICE i32x8 frag32(const char* p0, const char* p1) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(p0), b = *reinterpret_cast<const u32x4*>(p1);
  return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
}

template <bool SHORT_KV>
__global__ __launch_bounds__(NTHREADS) void attn_fwd_fp8_kernel(const Fp8AttnParams p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * PAIR_BLOB];  // two staging units of two tiles each

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;

  const int qb = blockIdx.x % p.nqb;
  const int head = blockIdx.x / p.nqb;
  const int q0 = qb * QROWS + wid * 32;
  const int qrow_c = min(q0 + r32, p.sq - 1);

  // ---- Q fragments (B operand of S^T): per 64-channel step ks the lane holds channels 16 h .. +15 of
  // both 32-channel blocks (bytes 0-15: block 2 ks, bytes 16-31: block 2 ks + 1) and the scale of block 2 ks + h ----
  i32x8 qf[2];
  int qsc;
  {
    const uint8_t* qp = p.q8 + ((int64_t)qrow_c * p.heads + head) * HD + 16 * h;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(qp + 64 * ks), b = *reinterpret_cast<const u32x4*>(qp + 64 * ks + 32);
      qf[ks] = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    }
    const uint32_t sc = *reinterpret_cast<const uint32_t*>(p.qs + ((int64_t)qrow_c * p.heads + head) * 4);
    qsc = (int)(((sc >> (8 * h)) & 0xffu) | (((sc >> (8 * (2 + h))) & 0xffu) << 8));  // byte ks = scale of block 2 ks + h
  }

  // ---- staging: a pair of tile blobs (34 304 B, the tile count is even) is copied linearly by
  // LDS-DMA, 16 B per lane and instruction: four full instructions and one of 96 lanes ----
  const char* kv_base = reinterpret_cast<const char*>(p.kv) + (int64_t)head * p.tiles * TILE_BLOB;
  const int wave = __builtin_amdgcn_readfirstlane(wid);
  auto issue_pair = [&](auto bufc, int t) __attribute__((always_inline)) {
    char* base = lds + decltype(bufc)::value * PAIR_BLOB;
    const char* src = kv_base + (int64_t)t * TILE_BLOB;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(src + (uint32_t)((tid + 512 * i) * 16), base + 8192 * i + wave * 1024);
    if (tid < 96) glds16(src + (uint32_t)((tid + 2048) * 16), base + 32768 + wave * 1024);
  };

  // ---- fragment read offsets ----
  // K (A operand of S^T): key = 32 kb + r32, chunks 4 ks + h and 4 ks + 2 + h (swizzled by (key >> 1) & 7)
  int k_rd[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i) k_rd[ks][i] = K_DATA + r32 * 128 + (((4 * ks + 2 * i + h) ^ ((r32 >> 1) & 7)) << 4);
  const int ks_rd = K_SCALE + (r32 * 2 + h) * 4;
  // V^T (A operand of O^T): d = 32 db + r32, slots 32 h .. +31 = chunks 2 h, 2 h + 1 (swizzled by (d >> 2) & 3);
  // bytes 0-15 are keys of the tile's first 32 (scale group 0), bytes 16-31 of its last 32 (group 1);
  // this lane supplies the scale of group h
  int v_rd[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) v_rd[i] = V_DATA + r32 * 64 + (((2 * h + i) ^ ((r32 >> 2) & 3)) << 4);
  const int vs_rd = V_SCALE + (h * 32 + r32) * 4;

  f32x16 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
  f32x16 negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  float l_run = 0.f;

  const int nt = p.tiles;

  // Stages of one tile (SUB = 0, 1 of the staging unit in LDS buffer BUF).  Fragment reads are issued
  // in whole batches well ahead of the MFMAs that consume them (the scheduler fences keep them
  // there): K fragments before the previous tile's O^T MFMAs, V fragments before the exponentials.
  f32x16 s[2];
  i32x8 kf[2][2], vf[4], pf;
  int ksc[2], vsc;
  auto read_k = [&](auto bufc, auto subc) __attribute__((always_inline)) {
    const char* base = lds + decltype(bufc)::value * PAIR_BLOB + decltype(subc)::value * TILE_BLOB;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      ksc[kb] = *reinterpret_cast<const int*>(base + ks_rd + kb * 256);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kf[kb][ks] = frag32(base + k_rd[ks][0] + kb * 4096, base + k_rd[ks][1] + kb * 4096);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto read_v = [&](auto bufc, auto subc) __attribute__((always_inline)) {
    const char* base = lds + decltype(bufc)::value * PAIR_BLOB + decltype(subc)::value * TILE_BLOB;
    vsc = *reinterpret_cast<const int*>(base + vs_rd);
#pragma unroll
    for (int db = 0; db < 4; ++db) vf[db] = frag32(base + v_rd[0] + db * 2048, base + v_rd[1] + db * 2048);
    __builtin_amdgcn_sched_barrier(0);
  };
  // S^T - m: the two key blocks alternate, so no MFMA waits on the one just issued
  auto qk = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 2>([&](auto ksc_) {
      constexpr int ks = decltype(ksc_)::value;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        s[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf[kb][ks], qf[ks], ks == 0 ? negm : s[kb], 0, 0, ks, ksc[kb], ks, qsc);
    });
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto pv = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 4>([&](auto dbc) {
      constexpr int db = decltype(dbc)::value;
      o[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf[db], pf, o[db], 0, 0, db, vsc, 0, UNIT_SCALE);
    });
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto softmax = [&](int t) __attribute__((always_inline)) {
    // ---- online softmax ----
    if ((t + 1) * KVB > p.skv) {  // the last real tile when ragged, and the all-padding tile of an odd count
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = t * KVB + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= p.skv) s[kb][i] = -1e30f;
        }
    }
    // P = exp2(s) in e4m3, slot j of this lane half = s[j >> 4][j & 15].  No maximum is taken on
    // the common path: every exp2 is positive, so "this lane's sum <= 2^6" bounds each of its P (e4m3
    // holds up to 448).  When some lane exceeds it (or on tile 0, where m is still 0) the row maximum
    // is formed, everything at the old scale -- O, l and this tile's scores -- moves to the new one
    // exactly once, and the exponentials are redone.
    float lsum;
    auto exps = [&]() __attribute__((always_inline)) {
      f32x2 acc2 = {0.f, 0.f};  // packed adds: two lanes of the sum per instruction
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(s[kb][4 * w + j]);
          int pk = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], 0, false);
          pk = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], pk, true);
          pf[kb * 4 + w] = pk;
          acc2 += f32x2{e[0], e[1]};
          acc2 += f32x2{e[2], e[3]};
        }
      lsum = acc2[0] + acc2[1];
    };
    if (t != 0) exps();
    if (t == 0 || __any(!(lsum <= 64.f))) {
      float mx = s[0][0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
      mx = half_swap_max(mx);
      const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kb][i] -= delta;
      const float nm = negm[0] - delta;
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = nm;
      exps();
    }
    l_run += lsum;
    __builtin_amdgcn_sched_barrier(0);
  };
  // One staging unit = two tiles = one barrier.  (Running the two wave groups half a period apart --
  // strictly alternating MFMA and exponential phases, or carrying one group's second tile over the
  // barrier -- was measured 5-20 % slower than letting the hardware interleave the waves.)
  auto pair = [&](auto bufc, int t) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    if (t + 2 < nt) issue_pair(IntC<BUF ^ 1>{}, t + 2);  // that buffer's last reader finished before the barrier just passed
    read_k(bufc, IntC<0>{});
    qk();
    read_v(bufc, IntC<0>{});
    softmax(t);
    read_k(bufc, IntC<1>{});
    pv();
    qk();
    read_v(bufc, IntC<1>{});
    softmax(t + 1);
    pv();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next unit has landed
    __syncthreads();
  };
  issue_pair(IntC<0>{}, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    int t = 0;
    for (; t + 2 < nt; t += 4) {
      pair(IntC<0>{}, t);
      pair(IntC<1>{}, t + 2);
    }
    if (t < nt) pair(IntC<0>{}, t);
  }

  // ---- normalise and store: o[db][i] = O[q0 + r32][32 db + 8 (i>>2) + 4 h + (i&3)] ----
  const float inv = 1.0f / half_swap_sum(l_run);
  const int qrow = q0 + r32;
  if (qrow < p.sq) {
    uint16_t* op = p.out + (int64_t)qrow * p.ldo + head * HD + 4 * h;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 pk = {pack2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv), pack2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + 32 * db + 8 * g) = pk;
      }
  }
}

int64_t tiles_of(int64_t skv) { return 2 * ((skv + 2 * KVB - 1) / (2 * KVB)); }  // even: tiles are staged in pairs

}  // namespace

extern "C" int64_t lx2v_attn_fp8_workspace_bytes(int64_t sq, int64_t skv, int heads) {
  if (sq < 0 || skv < 0 || heads <= 0) return -1;
  const int64_t q = ((sq * heads * (HD + 4) + 255) / 256) * 256;
  return q + (int64_t)heads * tiles_of(skv) * TILE_BLOB;
}

// Pre-pass + attention over `workspace` (device memory of at least lx2v_attn_fp8_workspace_bytes
// bytes, every byte the kernels read is written by the pre-pass of the same call).
static int attn_fwd_fp8_on(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                           uint16_t* out, int64_t ldo, int64_t sq, int64_t skv, int heads, int head_dim, float scale,
                           void* workspace, int64_t workspace_bytes, void* stream) {
  LX2V_REQUIRE(sq == 0 || (q && k && v && out && workspace), LX2V_EINVAL, "attn_fwd_fp8: null tensor");
  LX2V_REQUIRE(head_dim == HD, LX2V_EUNSUP, "attn_fwd_fp8: head_dim=%d, only 128 is built", head_dim);
  LX2V_REQUIRE(heads > 0 && sq >= 0 && skv >= 0, LX2V_EINVAL, "attn_fwd_fp8: bad shape sq=%lld skv=%lld heads=%d", (long long)sq, (long long)skv, heads);
  LX2V_REQUIRE(sq < (1LL << 31) && skv < (1LL << 31), LX2V_EUNSUP, "attn_fwd_fp8: sequence lengths must fit in 31 bits");
  if (sq == 0) return LX2V_OK;
  LX2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, LX2V_EINVAL, "attn_fwd_fp8: token strides must be multiples of 8 elements");
  LX2V_REQUIRE(ldq >= (int64_t)heads * HD && ldk >= (int64_t)heads * HD && ldv >= (int64_t)heads * HD && ldo >= (int64_t)heads * HD, LX2V_EINVAL,
               "attn_fwd_fp8: token stride smaller than heads*head_dim");
  LX2V_REQUIRE(lx2v_aligned16(q) && lx2v_aligned16(k) && lx2v_aligned16(v) && lx2v_aligned16(out) && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, LX2V_EINVAL,
               "attn_fwd_fp8: tensors must be 16-byte aligned, the workspace 256-byte aligned");
  LX2V_REQUIRE(workspace_bytes >= lx2v_attn_fp8_workspace_bytes(sq, skv, heads), LX2V_EINVAL, "attn_fwd_fp8: workspace of %lld bytes, need %lld", (long long)workspace_bytes,
               (long long)lx2v_attn_fp8_workspace_bytes(sq, skv, heads));
  LX2V_REQUIRE(skv > 0, LX2V_EINVAL, "attn_fwd_fp8: skv must be > 0 when sq > 0");
  const int64_t tiles = tiles_of(skv);
  LX2V_REQUIRE(tiles * TILE_BLOB < (1LL << 32), LX2V_EUNSUP, "attn_fwd_fp8: one head's K/V image must stay below 4 GiB");
  hipStream_t st = (hipStream_t)stream;
  uint8_t* ws = static_cast<uint8_t*>(workspace);
  uint8_t* q8 = ws;
  uint8_t* qs = ws + sq * heads * HD;
  uint8_t* kv = ws + ((sq * heads * (HD + 4) + 255) / 256) * 256;

  const int64_t qthreads = sq * heads * 16, kthreads = tiles * KVB * heads * 16;
  LX2V_REQUIRE((qthreads + 255) / 256 < (1LL << 31) && (kthreads + 255) / 256 < (1LL << 31), LX2V_EUNSUP, "attn_fwd_fp8: grid too large");
  hipLaunchKernelGGL(mxq_rows_kernel<false>, dim3((unsigned)((qthreads + 255) / 256)), dim3(256), 0, st, q, ldq, (int)sq, heads, scale * 1.4426950408889634f, q8, qs, 0);
  hipLaunchKernelGGL(mxq_rows_kernel<true>, dim3((unsigned)((kthreads + 255) / 256)), dim3(256), 0, st, k, ldk, (int)skv, heads, 1.0f, kv, nullptr, (int)tiles);
  hipLaunchKernelGGL(mxq_v_kernel, dim3((unsigned)tiles, (unsigned)heads), dim3(64), 0, st, v, ldv, (int)skv, heads, kv, (int)tiles);

  Fp8AttnParams p;
  p.q8 = q8; p.qs = qs; p.kv = kv; p.out = out; p.ldo = ldo;
  p.sq = (int)sq; p.skv = (int)skv; p.heads = heads; p.tiles = (int)tiles;
  p.nqb = (int)((sq + QROWS - 1) / QROWS);
  const int64_t nwg = (int64_t)p.nqb * heads;
  LX2V_REQUIRE(nwg < (1LL << 31), LX2V_EUNSUP, "attn_fwd_fp8: grid too large");
  if (lx2v_options().attn_fp8_kernel != 8) return lx2v_attn_fp8_launch_w4(p, st);
  if (skv <= 2048) hipLaunchKernelGGL(attn_fwd_fp8_kernel<true>, dim3((unsigned)nwg), dim3(NTHREADS), 0, st, p);
  else hipLaunchKernelGGL(attn_fwd_fp8_kernel<false>, dim3((unsigned)nwg), dim3(NTHREADS), 0, st, p);
  LX2V_CHECK_LAUNCH("attn_fwd_fp8");
  return LX2V_OK;
}

extern "C" int lx2v_attn_fwd_fp8(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                                 uint16_t* out, int64_t ldo, int64_t sq, int64_t skv, int heads, int head_dim, float scale,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  return attn_fwd_fp8_on(q, ldq, k, ldk, v, ldv, out, ldo, sq, skv, heads, head_dim, scale, workspace, workspace_bytes, stream);
}

// ---- the variant that allocates nothing: the quantised q / K / V images go to a fixed array ----
// Each GPU the library runs on holds its own copy of `scratch` (at its own address) until the
// process ends.  In host code the array's name is only a stand-in for that memory: its address on
// the current device comes from hipGetSymbolAddress, once per device, and everything below uses
// that pointer.
static __device__ __align__(256) unsigned char scratch[LX2V_ATTN_FP8_STATIC_BYTES];

static int static_scratch_of_current_device(unsigned char** ptr_out) {
  static std::mutex mu;
  static unsigned char* ptr_of[64];  // per device
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    lx2v_set_error("attn_fwd_fp8_static: hipGetDevice failed or device index %d out of range", dev);
    return LX2V_EHIP;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (ptr_of[dev] == nullptr) {
    unsigned char* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || ptr == nullptr) {
      lx2v_set_error("attn_fwd_fp8_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return LX2V_EHIP;
    }
    ptr_of[dev] = ptr;
  }
  *ptr_out = ptr_of[dev];
  return LX2V_OK;
}

extern "C" int64_t lx2v_attn_fp8_static_bytes(void) { return LX2V_ATTN_FP8_STATIC_BYTES; }

extern "C" int lx2v_attn_fwd_fp8_static(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                                        uint16_t* out, int64_t ldo, int64_t sq, int64_t skv, int heads, int head_dim, float scale,
                                        void* stream) {
  const int64_t need = lx2v_attn_fp8_workspace_bytes(sq, skv, heads);
  LX2V_REQUIRE(need >= 0, LX2V_EINVAL, "attn_fwd_fp8_static: bad shape sq=%lld skv=%lld heads=%d", (long long)sq, (long long)skv, heads);
  LX2V_REQUIRE(need <= (int64_t)LX2V_ATTN_FP8_STATIC_BYTES, LX2V_EUNSUP,
               "attn_fwd_fp8_static: sq=%lld skv=%lld heads=%d needs %lld bytes of scratch, the fixed array holds %lld; use lx2v_attn_fwd_fp8 with a workspace",
               (long long)sq, (long long)skv, heads, (long long)need, (long long)LX2V_ATTN_FP8_STATIC_BYTES);
  unsigned char* ptr = nullptr;
  const int rc = static_scratch_of_current_device(&ptr);
  if (rc != LX2V_OK) return rc;
  return attn_fwd_fp8_on(q, ldq, k, ldk, v, ldv, out, ldo, sq, skv, heads, head_dim, scale, ptr, LX2V_ATTN_FP8_STATIC_BYTES, stream);
}
