// Bandwidth-bound kernels of the Wan block: LayerNorm(+affine)(+AdaLN modulate), whole-row
// RMSNorm (+ per-head interleaved RoPE), gate-residual, GELU(tanh), Ulysses q/k/v pack.
//
// Row kernels give one row to one 64-lane wave: every lane owns 16-byte chunks (8 bf16) at
// chunk index lane + 64*c, so a wave-instruction moves 1 KiB of contiguous row, the row lives in
// registers between the statistics pass and the output pass (one HBM read, one HBM write), and
// the reductions are pure cross-lane shuffles (no LDS, no barrier).
#include "common.h"

namespace {

constexpr int kRowsPerBlock = 4;  // 4 waves of 64 lanes
#ifndef LN_ABL
#define LN_ABL 0  // timing ablation (wrong results): 1 the modulation vectors are not read
#endif
// Activation rows are read once and written once by these kernels: non-temporal accesses keep them
// from displacing what the cache holds (measured at [75600, 5120]: LN+modulate 5.31 -> 5.63 TB/s,
// GELU 4.98 -> 5.42 TB/s; LN_STREAMING=0 restores ordinary accesses for A/B).
#ifndef LN_STREAMING
#define LN_STREAMING 1
#endif
LX2V_DEVICE u32x4 ld_stream(const u32x4* p) {
#if LN_STREAMING
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
LX2V_DEVICE void st_stream(u32x4* p, const u32x4 v) {
#if LN_STREAMING
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

template <int NC>
struct RowRegs {
  float f[NC][8];
};

template <int NC>
LX2V_DEVICE void load_row(const uint16_t* __restrict__ p, int nchunks, int lane, RowRegs<NC>& r) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + 64 * c;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ch < nchunks) v = ld_stream(reinterpret_cast<const u32x4*>(p + (int64_t)ch * 8));
    unpack8(v, r.f[c]);
  }
}

// the same row from fp32 storage (the residual stream kept in fp32): chunk = 8 floats = two 16-byte loads
template <int NC>
LX2V_DEVICE void load_row_f32(const float* __restrict__ p, int nchunks, int lane, RowRegs<NC>& r) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + 64 * c;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (ch < nchunks) {
      lo = *reinterpret_cast<const f32x4*>(p + (int64_t)ch * 8);
      hi = *reinterpret_cast<const f32x4*>(p + (int64_t)ch * 8 + 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r.f[c][i] = lo[i];
      r.f[c][4 + i] = hi[i];
    }
  }
}

// out = (LN(x) * w + b) * (1 + scale) + shift
// Q8: the bf16-rounded result is not written; the row is quantised to OCP e4m3 with one scale per
// row (qscale[row] = max|row| / 448, out8 = row / qscale) while it is still in registers -- the
// bytes and scales are those the row-wise quantisation kernel produces from the bf16 output.  This
// file is built with fast-math, so the two quotients are formed in double precision and rounded
// once: that is the correctly rounded single-precision quotient.
// XF32: x holds fp32 rows (row stride ldx in elements) instead of bf16.
template <int NC, bool Q8, bool XF32 = false>
__global__ __launch_bounds__(256) void ln_modulate_kernel(
    const void* __restrict__ x, int64_t ldx, const float* __restrict__ ln_w,
    const float* __restrict__ ln_b, const float* __restrict__ scale, const float* __restrict__ shift,
    const uint16_t* __restrict__ scale_rows, const uint16_t* __restrict__ shift_rows, int64_t ld_rows,
    uint16_t* __restrict__ out, int64_t ldo, int64_t rows, int dim, float eps, uint8_t* __restrict__ out8, float* __restrict__ qscale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunks = dim >> 3;
  RowRegs<NC> r;
  if constexpr (XF32) load_row_f32<NC>(static_cast<const float*>(x) + row * ldx, nchunks, lane, r);
  else load_row<NC>(static_cast<const uint16_t*>(x) + row * ldx, nchunks, lane, r);

  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r.f[c][i];
  const float mean = wave_sum(s) / (float)dim;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool live = (lane + 64 * c) < nchunks;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float d = live ? r.f[c][i] - mean : 0.f;
      r.f[c][i] = d;
      ss += d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)dim + eps);

  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + 64 * c;
    if (ch >= nchunks) continue;
    const int col = ch * 8;
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = r.f[c][i] * rstd;
    if (ln_w != nullptr) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(ln_w + col), w1 = *reinterpret_cast<const f32x4*>(ln_w + col + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln_b + col), b1 = *reinterpret_cast<const f32x4*>(ln_b + col + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y[i] = y[i] * w0[i] + b0[i];
        y[i + 4] = y[i + 4] * w1[i] + b1[i];
      }
    }
    if (scale != nullptr) {
#if LN_ABL & 1  // timing ablation (wrong results): the modulation vectors are not read
      f32x4 s0 = {0.1f, 0.1f, 0.1f, 0.1f}, s1 = s0, h0 = s0, h1 = s0;
#else
      f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + col), s1 = *reinterpret_cast<const f32x4*>(scale + col + 4);
      f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + col), h1 = *reinterpret_cast<const f32x4*>(shift + col + 4);
#endif
      if (scale_rows != nullptr) {  // per-token part of the modulation (bf16 rows), added to the vectors
        float sr[8], hr[8];
        unpack8(*reinterpret_cast<const u32x4*>(scale_rows + row * ld_rows + col), sr);
        unpack8(*reinterpret_cast<const u32x4*>(shift_rows + row * ld_rows + col), hr);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s0[i] += sr[i];
          s1[i] += sr[i + 4];
          h0[i] += hr[i];
          h1[i] += hr[i + 4];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y[i] = y[i] * (1.0f + s0[i]) + h0[i];
        y[i + 4] = y[i + 4] * (1.0f + s1[i]) + h1[i];
      }
    }
    if constexpr (Q8) {
      unpack8(pack8(y), r.f[c]);  // the bf16 values the unfused path would have stored
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(r.f[c][i]));
    } else {
      st_stream(reinterpret_cast<u32x4*>(out + row * ldo + col), pack8(y));
    }
  }
  if constexpr (Q8) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    // (the widened operands pass through empty asm statements: a double quotient of two widened
    // floats would otherwise be narrowed back to a -- here approximate -- single-precision division)
    double amax_d = (double)fmaxf(amax, 1e-12f);
    asm volatile("" : "+v"(amax_d));
    const float sc = (float)(amax_d / 448.0);
    double sc_d = (double)sc;
    asm volatile("" : "+v"(sc_d));
    const float inv = (float)(1.0 / sc_d);
    if (lane == 0) qscale[row] = sc;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + 64 * c;
      if (ch >= nchunks) continue;
      const float* f = r.f[c];
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
      int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
      *reinterpret_cast<u32x2*>(out8 + row * ldo + (int64_t)ch * 8) = u32x2{(uint32_t)lo, (uint32_t)hi};
    }
  }
}

// y = x * rsqrt(mean(x^2)+eps) * w, then optional per-head interleaved-pair rotation.
// do_norm: 0 = no normalisation, 1 = statistics over the whole row with a [dim] weight,
// 2 = statistics per head (head_dim a power of two, 8..512) with a [head_dim] weight shared by
// every head: a head's head_dim/8 chunks sit in adjacent lanes of one register slot, so its sum is a
// butterfly over that lane group.
template <int NC>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, const float* __restrict__ w, int do_norm,
    const float* __restrict__ cos_sin, int head_dim, uint16_t* __restrict__ out, int64_t ldo,
    int64_t rows, int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunks = dim >> 3;
  RowRegs<NC> r;
  load_row<NC>(x + row * ldx, nchunks, lane, r);
  float rstd[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) rstd[c] = 1.0f;
  if (do_norm == 1) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += r.f[c][i] * r.f[c][i];
    const float rs = rsqrtf(wave_sum(ss) / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) rstd[c] = rs;
  } else if (do_norm == 2) {
    const int lanes_per_head = head_dim >> 3;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += r.f[c][i] * r.f[c][i];
      for (int off = lanes_per_head >> 1; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
      rstd[c] = rsqrtf(ss / (float)head_dim + eps);
    }
  }
  const int half = head_dim >> 1;
  const float* cs_row = cos_sin != nullptr ? cos_sin + row * head_dim : nullptr;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + 64 * c;
    if (ch >= nchunks) continue;
    const int col = ch * 8;
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = r.f[c][i] * rstd[c];
    if (w != nullptr) {
      const int wcol = do_norm == 2 ? col % head_dim : col;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + wcol), w1 = *reinterpret_cast<const f32x4*>(w + wcol + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y[i] *= w0[i];
        y[i + 4] *= w1[i];
      }
    }
    if (cs_row != nullptr) {
      const int pair0 = (col % head_dim) >> 1;  // 4 pairs per chunk
      const f32x4 cs = *reinterpret_cast<const f32x4*>(cs_row + pair0);
      const f32x4 sn = *reinterpret_cast<const f32x4*>(cs_row + half + pair0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = y[2 * i], b = y[2 * i + 1];
        y[2 * i] = a * cs[i] - b * sn[i];
        y[2 * i + 1] = a * sn[i] + b * cs[i];
      }
    }
    st_stream(reinterpret_cast<u32x4*>(out + row * ldo + col), pack8(y));
  }
}

__global__ __launch_bounds__(256) void gate_residual_kernel(uint16_t* __restrict__ x, int64_t ldx,
                                                            const uint16_t* __restrict__ y, int64_t ldy,
                                                            const float* __restrict__ gate, const uint16_t* __restrict__ gate_rows, int64_t ld_rows,
                                                            int64_t rows, int dim) {
  const int nchunks = dim >> 3;
  const int64_t total = rows * nchunks;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nchunks;
    const int col = (int)(i - row * nchunks) * 8;
    float a[8], b[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + row * ldx + col), a);
    unpack8(*reinterpret_cast<const u32x4*>(y + row * ldy + col), b);
    if (gate != nullptr) {
      f32x4 g0 = *reinterpret_cast<const f32x4*>(gate + col), g1 = *reinterpret_cast<const f32x4*>(gate + col + 4);
      if (gate_rows != nullptr) {  // per-token part of the gate (bf16 rows)
        float gr[8];
        unpack8(*reinterpret_cast<const u32x4*>(gate_rows + row * ld_rows + col), gr);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g0[j] += gr[j];
          g1[j] += gr[j + 4];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] += b[j] * g0[j];
        a[j + 4] += b[j + 4] * g1[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += b[j];
    }
    *reinterpret_cast<u32x4*>(x + row * ldx + col) = pack8(a);
  }
}

__global__ __launch_bounds__(256) void gelu_tanh_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int64_t nchunks) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * blockDim.x) {
    float a[8];
    unpack8(ld_stream(reinterpret_cast<const u32x4*>(x + i * 8)), a);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = gelu_tanh(a[j]);
    st_stream(reinterpret_cast<u32x4*>(out + i * 8), pack8(a));
  }
}

// exact GELU: x Phi(x) = 0.5 x (1 + erf(x / sqrt 2))
__global__ __launch_bounds__(256) void gelu_erf_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int64_t nchunks) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * blockDim.x) {
    float a[8];
    unpack8(ld_stream(reinterpret_cast<const u32x4*>(x + i * 8)), a);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.5f * a[j] * (1.0f + erff(a[j] * 0.70710678118654752f));
    st_stream(reinterpret_cast<u32x4*>(out + i * 8), pack8(a));
  }
}

// out[dst][row][which][c] = src_which[row][dst*seg + c], seg = dim/world
__global__ __launch_bounds__(256) void ulysses_pack_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                           const uint16_t* __restrict__ k, int64_t ldk,
                                                           const uint16_t* __restrict__ v, int64_t ldv,
                                                           uint16_t* __restrict__ out, int64_t rows, int dim, int world) {
  const int nchunks = dim >> 3;
  const int seg_chunks = nchunks / world;
  const int64_t total = rows * nchunks * 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nchunks);
    const int64_t t = i / nchunks;
    const int which = (int)(t % 3);
    const int64_t row = t / 3;
    const uint16_t* src = which == 0 ? q + row * ldq : (which == 1 ? k + row * ldk : v + row * ldv);
    const u32x4 val = *reinterpret_cast<const u32x4*>(src + ch * 8);
    const int dst = ch / seg_chunks;
    const int c = ch - dst * seg_chunks;
    const int64_t o = (((int64_t)dst * rows + row) * 3 + which) * (int64_t)seg_chunks + c;
    *reinterpret_cast<u32x4*>(out + o * 8) = val;
  }
}

int pick_nc(int dim) {
  const int need = ((dim >> 3) + 63) / 64;
  const int opts[] = {1, 2, 3, 4, 6, 8, 10, 12, 16};
  for (int o : opts)
    if (o >= need) return o;
  return -1;
}

int ew_grid(int64_t work_items) {
  int64_t blocks = (work_items + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

#define DISPATCH_NC(nc, KERNEL, ...)                                                      \
  switch (nc) {                                                                           \
    case 1: hipLaunchKernelGGL(KERNEL<1>, grid, block, 0, st, __VA_ARGS__); break;        \
    case 2: hipLaunchKernelGGL(KERNEL<2>, grid, block, 0, st, __VA_ARGS__); break;        \
    case 3: hipLaunchKernelGGL(KERNEL<3>, grid, block, 0, st, __VA_ARGS__); break;        \
    case 4: hipLaunchKernelGGL(KERNEL<4>, grid, block, 0, st, __VA_ARGS__); break;        \
    case 6: hipLaunchKernelGGL(KERNEL<6>, grid, block, 0, st, __VA_ARGS__); break;        \
    case 8: hipLaunchKernelGGL(KERNEL<8>, grid, block, 0, st, __VA_ARGS__); break;        \
    case 10: hipLaunchKernelGGL(KERNEL<10>, grid, block, 0, st, __VA_ARGS__); break;      \
    case 12: hipLaunchKernelGGL(KERNEL<12>, grid, block, 0, st, __VA_ARGS__); break;      \
    default: hipLaunchKernelGGL(KERNEL<16>, grid, block, 0, st, __VA_ARGS__); break;      \
  }

namespace {
template <int NC> constexpr auto ln_modulate_bf16_kernel = ln_modulate_kernel<NC, false>;
template <int NC> constexpr auto ln_modulate_q8_kernel = ln_modulate_kernel<NC, true>;
template <int NC> constexpr auto ln_modulate_f32in_kernel = ln_modulate_kernel<NC, false, true>;
}  // namespace

// out8 == NULL: bf16 rows to `out`; otherwise e4m3 rows to `out8` (row stride ldo, in bytes) and their scales to `qscale`.
static int ln_modulate_launch(const void* x, int64_t ldx, const float* ln_w, const float* ln_b, const float* scale, const float* shift,
                              const uint16_t* scale_rows, const uint16_t* shift_rows, int64_t ld_rows, uint16_t* out, int64_t ldo,
                              int64_t rows, int dim, float eps, void* stream, uint8_t* out8 = nullptr, float* qscale = nullptr, bool x_f32 = false) {
  LX2V_REQUIRE(rows == 0 || (x && (out || (out8 && qscale))), LX2V_EINVAL, "ln_modulate: null tensor");
  LX2V_REQUIRE(rows >= 0 && dim > 0, LX2V_EINVAL, "ln_modulate: bad shape rows=%lld dim=%d", (long long)rows, dim);
  LX2V_REQUIRE(dim % 8 == 0 && dim <= 8192, LX2V_EUNSUP, "ln_modulate: dim=%d must be a multiple of 8 and <= 8192", dim);
  LX2V_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && ldx >= dim && ldo >= dim, LX2V_EINVAL, "ln_modulate: row strides must be multiples of 8 and >= dim");
  LX2V_REQUIRE(lx2v_aligned16(x) && lx2v_aligned16(out) && (reinterpret_cast<uintptr_t>(out8) & 7u) == 0, LX2V_EINVAL, "ln_modulate: tensors must be 16-byte aligned (e4m3 rows: 8-byte)");
  LX2V_REQUIRE((ln_w == nullptr) == (ln_b == nullptr), LX2V_EINVAL, "ln_modulate: ln_w and ln_b must both be given or both NULL");
  LX2V_REQUIRE((scale == nullptr) == (shift == nullptr), LX2V_EINVAL, "ln_modulate: scale and shift must both be given or both NULL");
  LX2V_REQUIRE((scale_rows == nullptr) == (shift_rows == nullptr), LX2V_EINVAL, "ln_modulate: scale_rows and shift_rows must both be given or both NULL");
  LX2V_REQUIRE(scale_rows == nullptr || (scale != nullptr && ld_rows % 8 == 0 && ld_rows >= dim && lx2v_aligned16(scale_rows) && lx2v_aligned16(shift_rows)), LX2V_EINVAL,
               "ln_modulate: per-token rows need the scale/shift vectors, 16-byte alignment and a row stride that is a multiple of 8 and >= dim");
  if (rows == 0) return LX2V_OK;
  const dim3 grid((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (x_f32) {
    LX2V_REQUIRE(out8 == nullptr, LX2V_EUNSUP, "ln_modulate: fp32 rows are not combined with the e4m3 output");
    DISPATCH_NC(pick_nc(dim), ln_modulate_f32in_kernel, x, ldx, ln_w, ln_b, scale, shift, scale_rows, shift_rows, ld_rows, out, ldo, rows, dim, eps, out8, qscale);
  } else if (out8 == nullptr) {
    DISPATCH_NC(pick_nc(dim), ln_modulate_bf16_kernel, x, ldx, ln_w, ln_b, scale, shift, scale_rows, shift_rows, ld_rows, out, ldo, rows, dim, eps, out8, qscale);
  } else {
    DISPATCH_NC(pick_nc(dim), ln_modulate_q8_kernel, x, ldx, ln_w, ln_b, scale, shift, scale_rows, shift_rows, ld_rows, out, ldo, rows, dim, eps, out8, qscale);
  }
  LX2V_CHECK_LAUNCH("ln_modulate");
  return LX2V_OK;
}

extern "C" int lx2v_ln_modulate_bf16(const uint16_t* x, int64_t ldx, const float* ln_w, const float* ln_b,
                                     const float* scale, const float* shift, uint16_t* out, int64_t ldo,
                                     int64_t rows, int dim, float eps, void* stream) {
  return ln_modulate_launch(x, ldx, ln_w, ln_b, scale, shift, nullptr, nullptr, 0, out, ldo, rows, dim, eps, stream);
}

extern "C" int lx2v_ln_modulate_rows_bf16(const uint16_t* x, int64_t ldx, const float* ln_w, const float* ln_b,
                                          const float* scale, const float* shift, const uint16_t* scale_rows,
                                          const uint16_t* shift_rows, int64_t ld_rows, uint16_t* out, int64_t ldo,
                                          int64_t rows, int dim, float eps, void* stream) {
  LX2V_REQUIRE(scale_rows != nullptr, LX2V_EINVAL, "ln_modulate_rows: per-token rows are NULL");
  return ln_modulate_launch(x, ldx, ln_w, ln_b, scale, shift, scale_rows, shift_rows, ld_rows, out, ldo, rows, dim, eps, stream);
}

extern "C" int lx2v_ln_modulate_f32in(const float* x, int64_t ldx, const float* ln_w, const float* ln_b, const float* scale, const float* shift,
                                      const uint16_t* scale_rows, const uint16_t* shift_rows, int64_t ld_rows, uint16_t* out, int64_t ldo,
                                      int64_t rows, int dim, float eps, void* stream) {
  return ln_modulate_launch(x, ldx, ln_w, ln_b, scale, shift, scale_rows, shift_rows, ld_rows, out, ldo, rows, dim, eps, stream, nullptr, nullptr, true);
}

extern "C" int lx2v_ln_modulate_quant_fp8(const uint16_t* x, int64_t ldx, const float* ln_w, const float* ln_b,
                                          const float* scale, const float* shift, const uint16_t* scale_rows,
                                          const uint16_t* shift_rows, int64_t ld_rows, uint8_t* out, int64_t ldo, float* row_scale,
                                          int64_t rows, int dim, float eps, void* stream) {
  LX2V_REQUIRE(rows == 0 || (out && row_scale), LX2V_EINVAL, "ln_modulate_quant_fp8: null output");
  return ln_modulate_launch(x, ldx, ln_w, ln_b, scale, shift, scale_rows, shift_rows, ld_rows, nullptr, ldo, rows, dim, eps, stream, out, row_scale);
}

extern "C" int lx2v_rmsnorm_rope_bf16(const uint16_t* x, int64_t ldx, const float* w, int do_norm,
                                      const float* rope_cos_sin, int head_dim, uint16_t* out, int64_t ldo,
                                      int64_t rows, int dim, float eps, void* stream) {
  LX2V_REQUIRE(rows == 0 || (x && out), LX2V_EINVAL, "rmsnorm_rope: null tensor");
  LX2V_REQUIRE(rows >= 0 && dim > 0, LX2V_EINVAL, "rmsnorm_rope: bad shape rows=%lld dim=%d", (long long)rows, dim);
  LX2V_REQUIRE(dim % 8 == 0 && dim <= 8192, LX2V_EUNSUP, "rmsnorm_rope: dim=%d must be a multiple of 8 and <= 8192", dim);
  LX2V_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && ldx >= dim && ldo >= dim, LX2V_EINVAL, "rmsnorm_rope: row strides must be multiples of 8 and >= dim");
  LX2V_REQUIRE(lx2v_aligned16(x) && lx2v_aligned16(out), LX2V_EINVAL, "rmsnorm_rope: tensors must be 16-byte aligned");
  if (rope_cos_sin != nullptr) {
    LX2V_REQUIRE(head_dim > 0 && head_dim % 8 == 0 && dim % head_dim == 0, LX2V_EINVAL,
                 "rmsnorm_rope: head_dim=%d must be a multiple of 8 dividing dim=%d", head_dim, dim);
    LX2V_REQUIRE(lx2v_aligned16(rope_cos_sin), LX2V_EINVAL, "rmsnorm_rope: rope table must be 16-byte aligned");
  }
  LX2V_REQUIRE(do_norm >= 0 && do_norm <= 2, LX2V_EINVAL, "rmsnorm_rope: do_norm=%d (0 none, 1 whole row, 2 per head)", do_norm);
  if (do_norm == 2)
    LX2V_REQUIRE(head_dim >= 8 && head_dim <= 512 && (head_dim & (head_dim - 1)) == 0 && dim % head_dim == 0, LX2V_EUNSUP,
                 "rmsnorm_rope: per-head statistics need head_dim=%d to be a power of two in [8, 512] dividing dim=%d", head_dim, dim);
  if (rows == 0) return LX2V_OK;
  const dim3 grid((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)), block(256);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_NC(pick_nc(dim), rmsnorm_rope_kernel, x, ldx, w, do_norm, rope_cos_sin, head_dim, out, ldo, rows, dim, eps);
  LX2V_CHECK_LAUNCH("rmsnorm_rope");
  return LX2V_OK;
}

// x (fp32 rows) += y (bf16 rows) * gate: the residual stream kept in fp32.  Product and sum are
// two roundings, as x.float() + y.float() * gate evaluates.
__global__ __launch_bounds__(256) void gate_residual_f32_kernel(float* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ y, int64_t ldy,
                                                                const float* __restrict__ gate, const uint16_t* __restrict__ gate_rows, int64_t ld_rows,
                                                                int64_t rows, int dim) {
  const int nchunks = dim >> 3;
  const int64_t total = rows * nchunks;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nchunks;
    const int col = (int)(i - row * nchunks) * 8;
    float* xp = x + row * ldx + col;
    f32x4 a0 = *reinterpret_cast<const f32x4*>(xp), a1 = *reinterpret_cast<const f32x4*>(xp + 4);
    float b[8];
    unpack8(*reinterpret_cast<const u32x4*>(y + row * ldy + col), b);
    if (gate != nullptr) {
      f32x4 g0 = *reinterpret_cast<const f32x4*>(gate + col), g1 = *reinterpret_cast<const f32x4*>(gate + col + 4);
      if (gate_rows != nullptr) {
        float gr[8];
        unpack8(*reinterpret_cast<const u32x4*>(gate_rows + row * ld_rows + col), gr);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g0[j] += gr[j];
          g1[j] += gr[j + 4];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0[j] += __fmul_rn(b[j], g0[j]);
        a1[j] += __fmul_rn(b[j + 4], g1[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0[j] += b[j];
        a1[j] += b[j + 4];
      }
    }
    *reinterpret_cast<f32x4*>(xp) = a0;
    *reinterpret_cast<f32x4*>(xp + 4) = a1;
  }
}

extern "C" int lx2v_gate_residual_f32(float* x, int64_t ldx, const uint16_t* y, int64_t ldy, const float* gate, const uint16_t* gate_rows,
                                      int64_t ld_rows, int64_t rows, int dim, void* stream) {
  LX2V_REQUIRE(rows == 0 || (x && y), LX2V_EINVAL, "gate_residual_f32: null tensor");
  LX2V_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0, LX2V_EINVAL, "gate_residual_f32: dim=%d must be a positive multiple of 8", dim);
  LX2V_REQUIRE(ldx % 4 == 0 && ldy % 8 == 0 && ldx >= dim && ldy >= dim, LX2V_EINVAL, "gate_residual_f32: row strides must be multiples of 4 (x) / 8 (y) and >= dim");
  LX2V_REQUIRE(lx2v_aligned16(x) && lx2v_aligned16(y), LX2V_EINVAL, "gate_residual_f32: tensors must be 16-byte aligned");
  LX2V_REQUIRE(gate_rows == nullptr || (gate != nullptr && ld_rows % 8 == 0 && ld_rows >= dim && lx2v_aligned16(gate_rows)), LX2V_EINVAL,
               "gate_residual_f32: per-token rows need the gate vector, 16-byte alignment and a row stride that is a multiple of 8 and >= dim");
  if (rows == 0) return LX2V_OK;
  hipLaunchKernelGGL(gate_residual_f32_kernel, dim3(ew_grid(rows * (dim >> 3))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, gate, gate_rows, ld_rows, rows, dim);
  LX2V_CHECK_LAUNCH("gate_residual_f32");
  return LX2V_OK;
}

static int gate_residual_launch(uint16_t* x, int64_t ldx, const uint16_t* y, int64_t ldy, const float* gate, const uint16_t* gate_rows,
                                int64_t ld_rows, int64_t rows, int dim, void* stream) {
  LX2V_REQUIRE(rows == 0 || (x && y), LX2V_EINVAL, "gate_residual: null tensor");
  LX2V_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0, LX2V_EINVAL, "gate_residual: dim=%d must be a positive multiple of 8", dim);
  LX2V_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= dim && ldy >= dim, LX2V_EINVAL, "gate_residual: row strides must be multiples of 8 and >= dim");
  LX2V_REQUIRE(lx2v_aligned16(x) && lx2v_aligned16(y), LX2V_EINVAL, "gate_residual: tensors must be 16-byte aligned");
  LX2V_REQUIRE(gate_rows == nullptr || (gate != nullptr && ld_rows % 8 == 0 && ld_rows >= dim && lx2v_aligned16(gate_rows)), LX2V_EINVAL,
               "gate_residual: per-token rows need the gate vector, 16-byte alignment and a row stride that is a multiple of 8 and >= dim");
  if (rows == 0) return LX2V_OK;
  hipLaunchKernelGGL(gate_residual_kernel, dim3(ew_grid(rows * (dim >> 3))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, gate, gate_rows, ld_rows, rows, dim);
  LX2V_CHECK_LAUNCH("gate_residual");
  return LX2V_OK;
}

extern "C" int lx2v_gate_residual_bf16(uint16_t* x, int64_t ldx, const uint16_t* y, int64_t ldy,
                                       const float* gate, int64_t rows, int dim, void* stream) {
  return gate_residual_launch(x, ldx, y, ldy, gate, nullptr, 0, rows, dim, stream);
}

extern "C" int lx2v_gate_residual_rows_bf16(uint16_t* x, int64_t ldx, const uint16_t* y, int64_t ldy, const float* gate,
                                            const uint16_t* gate_rows, int64_t ld_rows, int64_t rows, int dim, void* stream) {
  LX2V_REQUIRE(gate_rows != nullptr, LX2V_EINVAL, "gate_residual_rows: per-token rows are NULL");
  return gate_residual_launch(x, ldx, y, ldy, gate, gate_rows, ld_rows, rows, dim, stream);
}

extern "C" int lx2v_gelu_tanh_bf16(const uint16_t* x, uint16_t* out, int64_t n, void* stream) {
  LX2V_REQUIRE(n == 0 || (x && out), LX2V_EINVAL, "gelu_tanh: null tensor");
  LX2V_REQUIRE(n >= 0 && n % 8 == 0, LX2V_EINVAL, "gelu_tanh: n=%lld must be a multiple of 8", (long long)n);
  LX2V_REQUIRE(lx2v_aligned16(x) && lx2v_aligned16(out), LX2V_EINVAL, "gelu_tanh: tensors must be 16-byte aligned");
  if (n == 0) return LX2V_OK;
  hipLaunchKernelGGL(gelu_tanh_kernel, dim3(ew_grid(n >> 3)), dim3(256), 0, (hipStream_t)stream, x, out, n >> 3);
  LX2V_CHECK_LAUNCH("gelu_tanh");
  return LX2V_OK;
}

extern "C" int lx2v_gelu_erf_bf16(const uint16_t* x, uint16_t* out, int64_t n, void* stream) {
  LX2V_REQUIRE(n == 0 || (x && out), LX2V_EINVAL, "gelu_erf: null tensor");
  LX2V_REQUIRE(n >= 0 && n % 8 == 0, LX2V_EINVAL, "gelu_erf: n=%lld must be a multiple of 8", (long long)n);
  LX2V_REQUIRE(lx2v_aligned16(x) && lx2v_aligned16(out), LX2V_EINVAL, "gelu_erf: tensors must be 16-byte aligned");
  if (n == 0) return LX2V_OK;
  hipLaunchKernelGGL(gelu_erf_kernel, dim3(ew_grid(n >> 3)), dim3(256), 0, (hipStream_t)stream, x, out, n >> 3);
  LX2V_CHECK_LAUNCH("gelu_erf");
  return LX2V_OK;
}

extern "C" int lx2v_ulysses_pack_qkv_bf16(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                                          const uint16_t* v, int64_t ldv, uint16_t* out, int64_t rows,
                                          int dim, int world, void* stream) {
  LX2V_REQUIRE(rows == 0 || (q && k && v && out), LX2V_EINVAL, "ulysses_pack: null tensor");
  LX2V_REQUIRE(world > 0 && dim > 0 && dim % (8 * world) == 0, LX2V_EINVAL, "ulysses_pack: dim=%d must be a multiple of 8*world (world=%d)", dim, world);
  LX2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, LX2V_EINVAL, "ulysses_pack: row strides must be multiples of 8");
  LX2V_REQUIRE(lx2v_aligned16(q) && lx2v_aligned16(k) && lx2v_aligned16(v) && lx2v_aligned16(out), LX2V_EINVAL, "ulysses_pack: tensors must be 16-byte aligned");
  if (rows == 0) return LX2V_OK;
  hipLaunchKernelGGL(ulysses_pack_kernel, dim3(ew_grid(rows * (dim >> 3) * 3)), dim3(256), 0, (hipStream_t)stream, q, ldq, k, ldk, v, ldv, out, rows, dim, world);
  LX2V_CHECK_LAUNCH("ulysses_pack");
  return LX2V_OK;
}
