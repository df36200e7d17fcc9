// Denoise-step glue kernels around the Wan DiT blocks (all bandwidth-bound and small next to the
// blocks): latent -> patch-row gather for the patch-embedding GEMM, head output -> latent scatter
// (unpatchify), sinusoidal timestep embedding, SiLU, and the fp32 tensor arithmetic of the
// classifier-free-guidance combine and of the UniPC scheduler update.
#include "common.h"

namespace {

inline unsigned ew_blocks(int64_t items) {
  const int64_t b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

// latents fp32 [C, F, H, W] -> rows bf16 [F*(H/2)*(W/2), C*4], column = (c, ph, pw): the flattening
// order of a conv3d weight [D, C, 1, 2, 2].  One thread = one (token, channel): reads two pairs of
// adjacent floats, writes 8 contiguous bytes.  With `round_src` the latents are replaced by their
// bf16-rounded values on the way (the reference keeps its latents in bf16 between steps).
__global__ __launch_bounds__(256) void patchify_kernel(float* __restrict__ lat, uint16_t* __restrict__ out, int64_t ldo, int C, int F, int H, int W, int round_src) {
  const int h2 = H >> 1, w2 = W >> 1;
  const int64_t n = (int64_t)F * h2 * w2 * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t tok = i / C;
    const int x = (int)(tok % w2);
    tok /= w2;
    const int y = (int)(tok % h2);
    const int f = (int)(tok / h2);
    float* src = lat + (((int64_t)c * F + f) * H + 2 * y) * W + 2 * x;
    const f32x2 r0 = *reinterpret_cast<const f32x2*>(src);
    const f32x2 r1 = *reinterpret_cast<const f32x2*>(src + W);
    const u32x2 pk = {pack2(r0[0], r0[1]), pack2(r1[0], r1[1])};
    *reinterpret_cast<u32x2*>(out + (i / C) * ldo + c * 4) = pk;
    if (round_src) {
      *reinterpret_cast<f32x2*>(src) = f32x2{__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u)};
      *reinterpret_cast<f32x2*>(src + W) = f32x2{__uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
    }
  }
}

// head output bf16 [S, 4*C] with column = (ph, pw, c)  ->  fp32 [C, F, H, W]
__global__ __launch_bounds__(256) void unpatchify_kernel(const uint16_t* __restrict__ y, int64_t ldy, float* __restrict__ out, int C, int F, int H, int W) {
  const int h2 = H >> 1, w2 = W >> 1;
  const int64_t n = (int64_t)C * F * H * w2;  // one thread = two horizontally adjacent outputs
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % w2);
    int64_t r = i / w2;
    const int yy = (int)(r % H);
    r /= H;
    const int f = (int)(r % F);
    const int c = (int)(r / F);
    const int64_t tok = ((int64_t)f * h2 + (yy >> 1)) * w2 + x;
    const uint16_t* src = y + tok * ldy + ((yy & 1) * 2) * C + c;
    const f32x2 v = {__uint_as_float((uint32_t)src[0] << 16), __uint_as_float((uint32_t)src[C] << 16)};
    *reinterpret_cast<f32x2*>(out + (((int64_t)c * F + f) * H + yy) * W + 2 * x) = v;
  }
}

// out[i] = cos(t * 10000^(-i/half)), out[half + i] = sin(...): the angle is formed in fp32 like the
// reference does, its cosine and sine are evaluated in double (angles reach 1000 rad).
__global__ void timestep_embedding_kernel(float t, int half, uint16_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  const float freq = (float)pow(10000.0, -(double)((float)i / (float)half));
  const float ang = t * freq;
  const bf16x2 cs = {(__bf16)(float)cos((double)ang), (__bf16)(float)sin((double)ang)};
  out[i] = __builtin_bit_cast(uint32_t, cs) & 0xffffu;
  out[half + i] = __builtin_bit_cast(uint32_t, cs) >> 16;
}

__global__ __launch_bounds__(256) void silu_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = __uint_as_float((uint32_t)x[i] << 16);
    const bf16x2 r = {(__bf16)(v / (1.0f + expf(-v))), (__bf16)0.f};
    out[i] = __builtin_bit_cast(uint32_t, r) & 0xffffu;
  }
}

// src fp32 [C, N] (channel planes) -> rows bf16 [N, ldo], column c; columns C..kpad-1 are zeroed.
// One thread = one (token, chunk of 8 channels): its loads are coalesced across the wave (adjacent
// threads = adjacent tokens of one plane), its store is 16 bytes of one row.
__global__ __launch_bounds__(256) void channels_to_rows_kernel(const float* __restrict__ src, uint16_t* __restrict__ out, int64_t ldo, int C, int64_t N, int kpad) {
  const int chunks = kpad >> 3;
  const int64_t total = N * chunks;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t tok = i % N;
    const int c0 = (int)(i / N) * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = c0 + j < C ? src[(int64_t)(c0 + j) * N + tok] : 0.f;
    *reinterpret_cast<u32x4*>(out + tok * ldo + c0) = pack8(f);
  }
}

// rows bf16 [N, C] (row stride ldy) -> out fp32 [C, N]; same thread mapping, the other way round.
__global__ __launch_bounds__(256) void rows_to_channels_kernel(const uint16_t* __restrict__ y, int64_t ldy, float* __restrict__ out, int C, int64_t N) {
  const int chunks = (C + 7) >> 3;
  const int64_t total = N * chunks;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t tok = i % N;
    const int c0 = (int)(i / N) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < C) out[(int64_t)(c0 + j) * N + tok] = __uint_as_float((uint32_t)y[tok * ldy + c0 + j] << 16);
  }
}

// out[d] = bf16( sum_l w[l] x[l, d] / sum_l w[l] ), fp32 accumulation in row order.  One thread per
// column (adjacent threads = adjacent columns: coalesced row reads); the few thousand rows of a
// text sequence are walked serially.
__global__ __launch_bounds__(256) void weighted_mean_rows_kernel(const uint16_t* __restrict__ x, int64_t ldx, const float* __restrict__ w, int rows, int dim, uint16_t* __restrict__ out) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= dim) return;
  float acc = 0.f, cnt = 0.f;
  for (int l = 0; l < rows; ++l) {
    const float wl = w[l];
    acc += __uint_as_float((uint32_t)x[(int64_t)l * ldx + d] << 16) * wl;
    cnt += wl;
  }
  out[d] = pack2(acc / cnt, 0.f) & 0xffffu;
}

// out[r, :] = bf16(x[r, :] + v[:]) (bf16 operands, one rounding), 8 columns per thread
__global__ __launch_bounds__(256) void add_rowvec_kernel(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ v, uint16_t* __restrict__ out, int64_t ldo, int64_t rows, int dim) {
  const int chunks = dim >> 3;
  const int64_t total = rows * chunks;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / chunks;
    const int c0 = (int)(i - r * chunks) * 8;
    float a[8], b[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + r * ldx + c0), a);
    unpack8(*reinterpret_cast<const u32x4*>(v + c0), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += b[j];
    *reinterpret_cast<u32x4*>(out + r * ldo + c0) = pack8(a);
  }
}

struct LinComb {
  const float* x[4];
  float c[4];
  int terms;
};

// out = sum_k c[k] * x[k] (fp32, left to right); out may alias any input
__global__ __launch_bounds__(256) void lincomb_kernel(float* out, const LinComb lc, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(lc.x[0] + 4 * i) * lc.c[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (k < lc.terms) acc += *reinterpret_cast<const f32x4*>(lc.x[k] + 4 * i) * lc.c[k];
    *reinterpret_cast<f32x4*>(out + 4 * i) = acc;
  }
}

// out = uncond + g * (cond - uncond), the reference's own association
__global__ __launch_bounds__(256) void cfg_combine_kernel(float* out, const float* cond, const float* uncond, float g, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(cond + 4 * i), u = *reinterpret_cast<const f32x4*>(uncond + 4 * i);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = u[j] + g * (c[j] - u[j]);  // three roundings, as torch evaluates it (this file is built without contraction)
    *reinterpret_cast<f32x4*>(out + 4 * i) = r;
  }
}

// Row-wise symmetric fp8 (OCP e4m3) quantisation: scale[r] = max|x[r,:]| / 448, out = x / scale.
// Built with IEEE division (this file is compiled without fast-math): the scale and its reciprocal
// are the correctly rounded quotients a host computes.  A row is handled by `lpr` adjacent lanes
// (64 = one row per wave; short rows such as one head of 128 values take 16 lanes, four rows per
// wave); two sweeps over the row (absmax, then convert), the second one served by L2 -- rows can be
// wider than the register-resident row kernels allow (FFN inner dim).
// A source row whose columns are stored as segments (lx2v_kseg; seg == 0: a plain row).  All segment
// widths are multiples of 8, so a 16-byte chunk never straddles two segments.
struct SegSrc {
  int64_t seg, stride, split, seg2, stride2, base2, lda2;
};
// Eight consecutive source elements as fp32: bf16 (one 16-byte chunk) or fp32 (two).
LX2V_DEVICE void load8(const uint16_t* p, float (&f)[8]) { unpack8(*reinterpret_cast<const u32x4*>(p), f); }
LX2V_DEVICE void load8(const float* p, float (&f)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[i] = a[i];
    f[4 + i] = b[i];
  }
}
template <class T>
LX2V_DEVICE const T* seg_chunk(const T* x, int64_t ldx, const SegSrc& g, int64_t row, int col) {
  if (g.seg == 0) return x + row * ldx + col;
  if (g.split == 0 || col < g.split) {
    const int64_t s = col / g.seg;
    return x + s * g.stride + row * ldx + (col - s * g.seg);
  }
  const int64_t c2 = col - g.split, t = c2 / g.seg2;
  return x + g.base2 + t * g.stride2 + row * g.lda2 + (c2 - t * g.seg2);
}

template <class T>
__global__ __launch_bounds__(256) void quant_fp8_rowwise_kernel(const T* __restrict__ x, int64_t ldx, const SegSrc src, uint8_t* __restrict__ out, int64_t ldo,
                                                                float* __restrict__ scale, int64_t rows, int dim, int lpr) {
  const int lane = threadIdx.x & 63;
  const int l = lane & (lpr - 1);
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / lpr) + lane / lpr;
  if (row >= rows) return;
  const int nchunks = dim >> 3;
  float amax = 0.f;
  for (int ch = l; ch < nchunks; ch += lpr) {
    float f[8];
    load8(seg_chunk(x, ldx, src, row, ch * 8), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(f[i]));
  }
  for (int off = lpr >> 1; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  const float sc = __fdiv_rn(fmaxf(amax, 1e-12f), 448.0f);
  const float inv = __frcp_rn(sc);
  if (l == 0) scale[row] = sc;
  uint8_t* orow = out + row * ldo;
  for (int ch = l; ch < nchunks; ch += lpr) {
    float f[8];
    load8(seg_chunk(x, ldx, src, row, ch * 8), f);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
    *reinterpret_cast<u32x2*>(orow + (int64_t)ch * 8) = u32x2{(uint32_t)lo, (uint32_t)hi};
  }
}

// Row-wise symmetric int8 quantisation, round half to even, saturating to [-128, 127].  Same row handling as the fp8 kernel above.
//   weight_mode 0 (activations; the dynamic per-token form of vLLM's scaled_int8_quant): scale[r] = max|x[r,:]| / 127,
//     out = rint(x * (127 / max|x[r,:]|)); a row of zeros gets scale 0 and zeros.
//   weight_mode 1 (weights; the reference's IntegerQuantizer(8, True, "per_channel"), lightx2v/utils/quant_utils.py:43-49,111-113):
//     scale[r] = max(max|x[r,:]|, 1e-5) / 127, out = rint(x / scale[r]).
template <class T>
__global__ __launch_bounds__(256) void quant_int8_rowwise_kernel(const T* __restrict__ x, int64_t ldx, const SegSrc src, int8_t* __restrict__ out, int64_t ldo,
                                                                 float* __restrict__ scale, int64_t rows, int dim, int lpr, int weight_mode) {
  const int lane = threadIdx.x & 63;
  const int l = lane & (lpr - 1);
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / lpr) + lane / lpr;
  if (row >= rows) return;
  const int nchunks = dim >> 3;
  float amax = 0.f;
  for (int ch = l; ch < nchunks; ch += lpr) {
    float f[8];
    load8(seg_chunk(x, ldx, src, row, ch * 8), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(f[i]));
  }
  for (int off = lpr >> 1; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  const float inv = amax > 0.f ? __fdiv_rn(127.0f, amax) : 0.f;
  const float sc = weight_mode ? __fdiv_rn(fmaxf(amax, 1e-5f), 127.0f) : __fdiv_rn(amax, 127.0f);
  if (l == 0) scale[row] = sc;
  int8_t* orow = out + row * ldo;
  for (int ch = l; ch < nchunks; ch += lpr) {
    float f[8];
    load8(seg_chunk(x, ldx, src, row, ch * 8), f);
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float r = fminf(fmaxf(rintf(weight_mode ? __fdiv_rn(f[i], sc) : __fmul_rn(f[i], inv)), -128.0f), 127.0f);
      w[i >> 2] |= ((uint32_t)(int)r & 0xffu) << (8 * (i & 3));
    }
    *reinterpret_cast<u32x2*>(orow + (int64_t)ch * 8) = u32x2{w[0], w[1]};
  }
}

// out[r, :] = bf16(e4m3(q[r, :]) * bf16(scale[r])): the value a bf16 multiply of the widened
// byte by the bf16-rounded scale gives (one rounding: the fp32 product of two bf16 values is exact).
__global__ __launch_bounds__(256) void dequant_fp8_rowwise_kernel(const uint8_t* __restrict__ q, int64_t ldq, const float* __restrict__ scale,
                                                                  uint16_t* __restrict__ out, int64_t ldo, int64_t rows, int dim) {
  const int nchunks = dim >> 3;
  const int64_t total = rows * nchunks;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nchunks;
    const int col = (int)(i - row * nchunks) * 8;
    const u32x2 v = *reinterpret_cast<const u32x2*>(q + row * ldq + col);
    const float sc = __uint_as_float(pack2(scale[row], 0.f) << 16);
    float f[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[h], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[h], true);
      f[4 * h] = a[0] * sc;
      f[4 * h + 1] = a[1] * sc;
      f[4 * h + 2] = b[0] * sc;
      f[4 * h + 3] = b[1] * sc;
    }
    *reinterpret_cast<u32x4*>(out + row * ldo + col) = pack8(f);
  }
}

}  // namespace

extern "C" int lx2v_patchify_bf16(float* latents, uint16_t* out, int64_t ldo, int channels, int frames, int height, int width, int round_latents, void* stream) {
  LX2V_REQUIRE(latents && out, LX2V_EINVAL, "patchify: null tensor");
  LX2V_REQUIRE(channels > 0 && frames > 0 && height > 0 && width > 0 && height % 2 == 0 && width % 2 == 0, LX2V_EINVAL,
               "patchify: bad latent shape [%d,%d,%d,%d] (height and width must be even)", channels, frames, height, width);
  LX2V_REQUIRE((reinterpret_cast<uintptr_t>(latents) & 7u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0, LX2V_EINVAL, "patchify: tensors must be 8-byte aligned");
  LX2V_REQUIRE(ldo >= 4 * (int64_t)channels && ldo % 4 == 0, LX2V_EINVAL, "patchify: row stride must be a multiple of 4 and >= 4*channels");
  const int64_t n = (int64_t)frames * (height / 2) * (width / 2) * channels;
  hipLaunchKernelGGL(patchify_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, latents, out, ldo, channels, frames, height, width, round_latents);
  LX2V_CHECK_LAUNCH("patchify");
  return LX2V_OK;
}

extern "C" int lx2v_unpatchify_f32(const uint16_t* y, int64_t ldy, float* out, int channels, int frames, int height, int width, void* stream) {
  LX2V_REQUIRE(y && out, LX2V_EINVAL, "unpatchify: null tensor");
  LX2V_REQUIRE(channels > 0 && frames > 0 && height > 0 && width > 0 && height % 2 == 0 && width % 2 == 0, LX2V_EINVAL,
               "unpatchify: bad latent shape [%d,%d,%d,%d] (height and width must be even)", channels, frames, height, width);
  LX2V_REQUIRE(ldy >= 4 * (int64_t)channels, LX2V_EINVAL, "unpatchify: row stride smaller than 4*channels");
  LX2V_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7u) == 0, LX2V_EINVAL, "unpatchify: output must be 8-byte aligned");
  const int64_t n = (int64_t)channels * frames * height * (width / 2);
  hipLaunchKernelGGL(unpatchify_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, y, ldy, out, channels, frames, height, width);
  LX2V_CHECK_LAUNCH("unpatchify");
  return LX2V_OK;
}

extern "C" int lx2v_channels_to_rows_bf16(const float* src, uint16_t* out, int64_t ldo, int channels, int64_t tokens, int kpad, void* stream) {
  LX2V_REQUIRE(src && out, LX2V_EINVAL, "channels_to_rows: null tensor");
  LX2V_REQUIRE(channels > 0 && tokens > 0 && kpad >= channels && kpad % 8 == 0 && ldo >= kpad && ldo % 8 == 0, LX2V_EINVAL,
               "channels_to_rows: channels=%d tokens=%lld kpad=%d ldo=%lld (kpad >= channels, kpad and ldo multiples of 8, ldo >= kpad)", channels,
               (long long)tokens, kpad, (long long)ldo);
  LX2V_REQUIRE(lx2v_aligned16(out), LX2V_EINVAL, "channels_to_rows: output must be 16-byte aligned");
  hipLaunchKernelGGL(channels_to_rows_kernel, dim3(ew_blocks(tokens * (kpad / 8))), dim3(256), 0, (hipStream_t)stream, src, out, ldo, channels, tokens, kpad);
  LX2V_CHECK_LAUNCH("channels_to_rows");
  return LX2V_OK;
}

extern "C" int lx2v_rows_to_channels_f32(const uint16_t* y, int64_t ldy, float* out, int channels, int64_t tokens, void* stream) {
  LX2V_REQUIRE(y && out, LX2V_EINVAL, "rows_to_channels: null tensor");
  LX2V_REQUIRE(channels > 0 && tokens > 0 && ldy >= channels, LX2V_EINVAL, "rows_to_channels: channels=%d tokens=%lld ldy=%lld", channels, (long long)tokens,
               (long long)ldy);
  hipLaunchKernelGGL(rows_to_channels_kernel, dim3(ew_blocks(tokens * ((channels + 7) / 8))), dim3(256), 0, (hipStream_t)stream, y, ldy, out, channels, tokens);
  LX2V_CHECK_LAUNCH("rows_to_channels");
  return LX2V_OK;
}

extern "C" int lx2v_weighted_mean_rows_bf16(const uint16_t* x, int64_t ldx, const float* weights, int rows, int dim, uint16_t* out, void* stream) {
  LX2V_REQUIRE(x && weights && out, LX2V_EINVAL, "weighted_mean_rows: null tensor");
  LX2V_REQUIRE(rows > 0 && dim > 0 && ldx >= dim, LX2V_EINVAL, "weighted_mean_rows: rows=%d dim=%d ldx=%lld", rows, dim, (long long)ldx);
  hipLaunchKernelGGL(weighted_mean_rows_kernel, dim3((dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, weights, rows, dim, out);
  LX2V_CHECK_LAUNCH("weighted_mean_rows");
  return LX2V_OK;
}

extern "C" int lx2v_add_rowvec_bf16(const uint16_t* x, int64_t ldx, const uint16_t* vec, uint16_t* out, int64_t ldo, int64_t rows, int dim, void* stream) {
  LX2V_REQUIRE(rows == 0 || (x && vec && out), LX2V_EINVAL, "add_rowvec: null tensor");
  LX2V_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0 && ldx >= dim && ldo >= dim && ldx % 8 == 0 && ldo % 8 == 0, LX2V_EINVAL,
               "add_rowvec: rows=%lld dim=%d ldx=%lld ldo=%lld (dim and strides multiples of 8)", (long long)rows, dim, (long long)ldx, (long long)ldo);
  LX2V_REQUIRE(lx2v_aligned16(x) && lx2v_aligned16(vec) && lx2v_aligned16(out), LX2V_EINVAL, "add_rowvec: tensors must be 16-byte aligned");
  if (rows == 0) return LX2V_OK;
  hipLaunchKernelGGL(add_rowvec_kernel, dim3(ew_blocks(rows * (dim / 8))), dim3(256), 0, (hipStream_t)stream, x, ldx, vec, out, ldo, rows, dim);
  LX2V_CHECK_LAUNCH("add_rowvec");
  return LX2V_OK;
}

extern "C" int lx2v_timestep_embedding_bf16(float t, int dim, uint16_t* out, void* stream) {
  LX2V_REQUIRE(out, LX2V_EINVAL, "timestep_embedding: null tensor");
  LX2V_REQUIRE(dim > 0 && dim % 2 == 0, LX2V_EINVAL, "timestep_embedding: dim=%d must be positive and even", dim);
  const int half = dim / 2;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((half + 63) / 64), dim3(64), 0, (hipStream_t)stream, t, half, out);
  LX2V_CHECK_LAUNCH("timestep_embedding");
  return LX2V_OK;
}

extern "C" int lx2v_silu_bf16(const uint16_t* x, uint16_t* out, int64_t n, void* stream) {
  LX2V_REQUIRE(n == 0 || (x && out), LX2V_EINVAL, "silu: null tensor");
  LX2V_REQUIRE(n >= 0, LX2V_EINVAL, "silu: n=%lld", (long long)n);
  if (n == 0) return LX2V_OK;
  hipLaunchKernelGGL(silu_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, out, n);
  LX2V_CHECK_LAUNCH("silu");
  return LX2V_OK;
}

extern "C" int lx2v_lincomb_f32(float* out, int64_t n, int terms, const float* const* x, const float* coef, void* stream) {
  LX2V_REQUIRE(out && x && coef, LX2V_EINVAL, "lincomb: null argument");
  LX2V_REQUIRE(terms >= 1 && terms <= 4, LX2V_EINVAL, "lincomb: terms=%d must be 1..4", terms);
  LX2V_REQUIRE(n >= 0 && n % 4 == 0, LX2V_EINVAL, "lincomb: n=%lld must be a multiple of 4", (long long)n);
  LX2V_REQUIRE(lx2v_aligned16(out), LX2V_EINVAL, "lincomb: tensors must be 16-byte aligned");
  LinComb lc;
  lc.terms = terms;
  for (int k = 0; k < 4; ++k) {
    lc.x[k] = k < terms ? x[k] : x[0];
    lc.c[k] = k < terms ? coef[k] : 0.f;
    LX2V_REQUIRE(lc.x[k] != nullptr && lx2v_aligned16(lc.x[k]), LX2V_EINVAL, "lincomb: input %d is null or not 16-byte aligned", k);
  }
  if (n == 0) return LX2V_OK;
  hipLaunchKernelGGL(lincomb_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, out, lc, n / 4);
  LX2V_CHECK_LAUNCH("lincomb");
  return LX2V_OK;
}

extern "C" int lx2v_cfg_combine_f32(float* out, const float* cond, const float* uncond, float guide_scale, int64_t n, void* stream) {
  LX2V_REQUIRE(n == 0 || (out && cond && uncond), LX2V_EINVAL, "cfg_combine: null tensor");
  LX2V_REQUIRE(n >= 0 && n % 4 == 0, LX2V_EINVAL, "cfg_combine: n=%lld must be a multiple of 4", (long long)n);
  LX2V_REQUIRE(lx2v_aligned16(out) && lx2v_aligned16(cond) && lx2v_aligned16(uncond), LX2V_EINVAL, "cfg_combine: tensors must be 16-byte aligned");
  if (n == 0) return LX2V_OK;
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, out, cond, uncond, guide_scale, n / 4);
  LX2V_CHECK_LAUNCH("cfg_combine");
  return LX2V_OK;
}

namespace {

// Validates a segmented source description against `dim` and returns its device form (all zero = plain).
int seg_src_from(const lx2v_kseg* kseg, int64_t ldx, int dim, const char* what, SegSrc& g) {
  g = SegSrc{0, 0, 0, 0, 0, 0, 0};
  if (kseg == nullptr) {
    LX2V_REQUIRE(ldx >= dim, LX2V_EINVAL, "%s: row stride smaller than the row", what);
    return LX2V_OK;
  }
  LX2V_REQUIRE(kseg->seg > 0 && kseg->seg % 8 == 0 && kseg->stride % 8 == 0 && ldx >= kseg->seg, LX2V_EINVAL, "%s: segments must be multiples of 8 columns, 8-aligned, inside a row", what);
  const int64_t k1 = kseg->split > 0 ? kseg->split : dim;
  LX2V_REQUIRE(k1 % kseg->seg == 0 && k1 <= dim, LX2V_EINVAL, "%s: region 1 (%lld columns) must hold whole segments of %lld", what, (long long)k1, (long long)kseg->seg);
  if (kseg->split > 0) {
    LX2V_REQUIRE(kseg->seg2 > 0 && kseg->seg2 % 8 == 0 && kseg->stride2 % 8 == 0 && kseg->base2 % 8 == 0 && kseg->lda2 % 8 == 0 && kseg->lda2 >= kseg->seg2 &&
                     (dim - kseg->split) % kseg->seg2 == 0,
                 LX2V_EINVAL, "%s: region 2 must hold whole, 8-aligned segments", what);
  }
  g = SegSrc{kseg->seg, kseg->stride, kseg->split, kseg->seg2, kseg->stride2, kseg->base2, kseg->lda2};
  return LX2V_OK;
}

template <class T>
int quant_rows(bool int8, const T* x, int64_t ldx, const lx2v_kseg* kseg, void* out, int64_t ldo, float* scale, int64_t rows, int dim, int weight_mode, void* stream) {
  const char* what = int8 ? "quant_int8" : "quant_fp8";
  LX2V_REQUIRE(rows == 0 || (x && out && scale), LX2V_EINVAL, "%s: null tensor", what);
  LX2V_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0, LX2V_EINVAL, "%s: dim=%d must be a positive multiple of 8", what, dim);
  LX2V_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && ldo >= dim, LX2V_EINVAL, "%s: row strides must be multiples of 8, the output's >= dim", what);
  LX2V_REQUIRE(lx2v_aligned16(x) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0, LX2V_EINVAL, "%s: x must be 16-byte, out 8-byte aligned", what);
  SegSrc g;
  if (const int rc = seg_src_from(kseg, ldx, dim, what, g); rc != LX2V_OK) return rc;
  if (rows == 0) return LX2V_OK;
  const int nchunks = dim >> 3;
  int lpr = 64;
  if (nchunks <= 32 && (nchunks & (nchunks - 1)) == 0) lpr = nchunks;  // short rows share a wave
  const int64_t rows_per_block = 4 * (64 / lpr);
  const dim3 grid((unsigned)((rows + rows_per_block - 1) / rows_per_block));
  if (int8) hipLaunchKernelGGL(quant_int8_rowwise_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, g, (int8_t*)out, ldo, scale, rows, dim, lpr, weight_mode != 0);
  else hipLaunchKernelGGL(quant_fp8_rowwise_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, g, (uint8_t*)out, ldo, scale, rows, dim, lpr);
  LX2V_CHECK_LAUNCH(what);
  return LX2V_OK;
}

}  // namespace

extern "C" int lx2v_quant_fp8_rowwise(const uint16_t* x, int64_t ldx, uint8_t* out, int64_t ldo, float* scale, int64_t rows, int dim, void* stream) {
  return quant_rows(false, x, ldx, nullptr, out, ldo, scale, rows, dim, 0, stream);
}

extern "C" int lx2v_quant_fp8_rowwise_kseg(const uint16_t* x, int64_t ldx, const lx2v_kseg* kseg, uint8_t* out, int64_t ldo, float* scale, int64_t rows, int dim, void* stream) {
  return quant_rows(false, x, ldx, kseg, out, ldo, scale, rows, dim, 0, stream);
}

extern "C" int lx2v_quant_int8_rowwise(const uint16_t* x, int64_t ldx, int8_t* out, int64_t ldo, float* scale, int64_t rows, int dim, int weight_mode, void* stream) {
  return quant_rows(true, x, ldx, nullptr, out, ldo, scale, rows, dim, weight_mode, stream);
}

extern "C" int lx2v_quant_fp8_rowwise_f32(const float* x, int64_t ldx, uint8_t* out, int64_t ldo, float* scale, int64_t rows, int dim, void* stream) {
  return quant_rows(false, x, ldx, nullptr, out, ldo, scale, rows, dim, 0, stream);
}

extern "C" int lx2v_quant_int8_rowwise_f32(const float* x, int64_t ldx, int8_t* out, int64_t ldo, float* scale, int64_t rows, int dim, int weight_mode, void* stream) {
  return quant_rows(true, x, ldx, nullptr, out, ldo, scale, rows, dim, weight_mode, stream);
}

extern "C" int lx2v_quant_int8_rowwise_kseg(const uint16_t* x, int64_t ldx, const lx2v_kseg* kseg, int8_t* out, int64_t ldo, float* scale, int64_t rows, int dim, void* stream) {
  return quant_rows(true, x, ldx, kseg, out, ldo, scale, rows, dim, 0, stream);
}

extern "C" int lx2v_dequant_fp8_rowwise(const uint8_t* q, int64_t ldq, const float* scale, uint16_t* out, int64_t ldo, int64_t rows, int dim, void* stream) {
  LX2V_REQUIRE(rows == 0 || (q && out && scale), LX2V_EINVAL, "dequant_fp8: null tensor");
  LX2V_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0, LX2V_EINVAL, "dequant_fp8: dim=%d must be a positive multiple of 8", dim);
  LX2V_REQUIRE(ldq % 8 == 0 && ldo % 8 == 0 && ldq >= dim && ldo >= dim, LX2V_EINVAL, "dequant_fp8: row strides must be multiples of 8 and >= dim");
  LX2V_REQUIRE(lx2v_aligned16(out) && (reinterpret_cast<uintptr_t>(q) & 7u) == 0, LX2V_EINVAL, "dequant_fp8: out must be 16-byte, q 8-byte aligned");
  if (rows == 0) return LX2V_OK;
  hipLaunchKernelGGL(dequant_fp8_rowwise_kernel, dim3(ew_blocks(rows * (dim >> 3))), dim3(256), 0, (hipStream_t)stream, q, ldq, scale, out, ldo, rows, dim);
  LX2V_CHECK_LAUNCH("dequant_fp8");
  return LX2V_OK;
}
