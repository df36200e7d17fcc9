// Issue cost of VALU instructions placed behind the block-scaled fp8 MFMAs by a lone wave per SIMD:
// cycles per MFMA with 0..N copies of an instruction mix in every gap.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe_valu_beside_mfma_fp8.hip -o build/probe_valu_fp8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); } }
template <int J, int N, int KIND>
__device__ __forceinline__ void ops(float (&e)[8], float (&x)[8], float (&acc)[4], int (&pk)[4]) {
  if constexpr (N > 0) {
    constexpr int n = N - 1;
    if constexpr (KIND == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(e[(J + n) & 7]) : "v"(x[(J + n) & 7]));
    if constexpr (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[(J + n) & 3]) : "v"(e[(J + n + 4) & 7]));
    if constexpr (KIND == 2) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(pk[(J + n) & 3]) : "v"(e[(J + n + 2) & 7]), "v"(e[(J + n + 5) & 7]));
    if constexpr (KIND == 3) {  // packed add of two exps into a two-lane row sum
      f32x2 pr = {e[(J + 2 * n + 4) & 7], e[(J + 2 * n + 5) & 7]};
      f32x2& a2 = *reinterpret_cast<f32x2*>(&acc[2 * ((J + n) & 1)]);
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2) : "v"(pr));
    }
    if constexpr (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[(J + n) & 3]) : "v"(e[(J + n + 4) & 7]), "v"(e[(J + n + 6) & 7]));
    // other routes to two e4m3 values: the scaled conversion from fp32; through a packed bf16 / f16 pair
    if constexpr (KIND == 5) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(pk[(J + n) & 3]) : "v"(e[(J + n + 2) & 7]), "v"(e[(J + n + 5) & 7]), "v"(x[7]));
    if constexpr (KIND == 6) {
      int t;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(e[(J + n + 2) & 7]), "v"(e[(J + n + 5) & 7]));
      asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(pk[(J + n) & 3]) : "v"(t), "v"(x[7]));
    }
    if constexpr (KIND == 7) {
      int t;
      asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(t) : "v"(e[(J + n + 2) & 7]), "v"(e[(J + n + 5) & 7]));
      asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(pk[(J + n) & 3]) : "v"(t), "v"(x[7]));
    }
    if constexpr (KIND == 8) asm volatile("v_exp_f16 %0, %1" : "=v"(e[(J + n) & 7]) : "v"(x[(J + n) & 7]));
    if constexpr (KIND == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(pk[(J + n) & 3]) : "v"(e[(J + n + 2) & 7]), "v"(e[(J + n + 5) & 7]));
    // exp-free P: e4m3 code = 8 (x + 7) as an unsigned byte; dependent and independent chains
    if constexpr (KIND == 10) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(pk[0]) : "v"(e[(J + n + 2) & 7]));
    if constexpr (KIND == 11) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(pk[(J + n) & 3]) : "v"(e[(J + n + 2) & 7]));
    if constexpr (KIND == 12) asm volatile("v_fmaak_f32 %0, %1, %2, 0x42600000" : "=v"(e[(J + n) & 7]) : "v"(x[7]), "v"(x[(J + n) & 7]));
    if constexpr (KIND == 13) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[(J + n) & 7]) : "v"(x[7]), "v"(x[(J + n) & 7]), "v"(x[6]));
    if constexpr (KIND == 14) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(pk[0]) : "v"(e[(J + n + 2) & 7]), "v"(e[(J + n + 5) & 7]));
    if constexpr (KIND == 15) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[0]) : "v"(e[(J + n + 4) & 7]));
    if constexpr (KIND == 16) {
      f32x2 xx = {x[(J + 2 * n) & 6], x[((J + 2 * n) & 6) + 1]}, k8 = {x[7], x[7]}, kb = {x[6], x[6]};
      f32x2& yy = *reinterpret_cast<f32x2*>(&e[(J + 2 * n) & 6]);
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(yy) : "v"(xx), "v"(k8), "v"(kb));
    }
    if constexpr (KIND == 17) asm volatile("v_or_b32 %0, %0, %1" : "+v"(pk[(J + n) & 3]) : "v"(pk[(J + n + 2) & 3]));
    ops<J, N - 1, KIND>(e, x, acc, pk);
  }
}
template <int KIND, int N>
__global__ __launch_bounds__(256) void k1(const int* in, float* out, int iters, long long* cyc) {  // N copies of one instruction kind per gap of the 32x32x64 MFMA
  const int lane = threadIdx.x & 63;
  i32x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = in[(i * 512 + j * 64 + lane) & 4095]; b[i][j] = in[(i * 512 + j * 64 + lane + 2048) & 4095]; }
  int sa = 127, sb = 127;
  asm volatile("" : "+v"(sa), "+v"(sb));
  float x[8], e[8], acc[4] = {0.f, 0.f, 0.f, 0.f};
  int pk[4] = {};
  for (int i = 0; i < 8; ++i) { x[i] = -0.01f * (lane + i); e[i] = 0.5f + i; }
  x[7] = 1.0f;
  f32x16 c32[4] = {};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    sfor<0, 32>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c32[j & 3]) : "v"(a[j & 3]), "v"(b[(j >> 2) & 3]), "v"(sa), "v"(sb));
      ops<j, N, KIND>(e, x, acc, pk);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  const long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  float r = acc[0] + acc[1] + acc[2] + acc[3];
  for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) r += c32[j][q];
  for (int i = 0; i < 8; ++i) r += e[i];
  for (int i = 0; i < 4; ++i) r += (float)pk[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
// NA < 0: -NA packed adds instead of adds; NA >= 100: NA - 100 v_max3 instead
template <int J, int NE, int NA, int NC>
__device__ __forceinline__ void filler(float (&e)[8], float (&x)[8], float (&acc)[4], int (&pk)[4]) {
  ops<J, NE, 0>(e, x, acc, pk);
  if constexpr (NA >= 100) ops<J, NA - 100, 4>(e, x, acc, pk);
  else if constexpr (NA >= 0) ops<J, NA, 1>(e, x, acc, pk);
  else ops<J, -NA, 3>(e, x, acc, pk);
  ops<J, NC, 2>(e, x, acc, pk);
}
// SHAPE 0: 32x32x64 (16 accumulator registers), 1: 16x16x128 (4).  Per gap: NE exps, NA adds, NC fp8 converts.
template <int SHAPE, int NE, int NA, int NC, int NR = 0>
__global__ __launch_bounds__(256) void k(const int* in, float* out, int iters, long long* cyc) {
  const int lane = threadIdx.x & 63;
  __shared__ i32x4 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i32x4{in[i & 4095], in[(i + 1) & 4095], in[(i + 2) & 4095], in[(i + 3) & 4095]};
  __syncthreads();
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) i32x4*)(lds + lane);
  i32x4 fr[4] = {};
  asm volatile("" : "+a"(fr[0]), "+a"(fr[1]), "+a"(fr[2]), "+a"(fr[3]));
  i32x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = in[(i * 512 + j * 64 + lane) & 4095]; b[i][j] = in[(i * 512 + j * 64 + lane + 2048) & 4095]; }
  int sa = 127, sb = 127;
  asm volatile("" : "+v"(sa), "+v"(sb));
  float x[8], e[8], acc[4] = {0.f, 0.f, 0.f, 0.f};
  int pk[4] = {};
  for (int i = 0; i < 8; ++i) { x[i] = -0.01f * (lane + i); e[i] = 0.f; }
  f32x16 c32[4] = {};
  f32x4 c16[8] = {};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    sfor<0, 32>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (SHAPE == 0) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c32[j & 3]) : "v"(a[j & 3]), "v"(b[(j >> 2) & 3]), "v"(sa), "v"(sb));
      else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c16[j & 7]) : "v"(a[j & 3]), "v"(b[(j >> 2) & 3]), "v"(sa), "v"(sb));
      if constexpr (NR >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(fr[j & 3]) : "v"(laddr), "i"(1024 * (j & 15)));
      if constexpr (NR >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(fr[(j + 2) & 3]) : "v"(laddr), "i"(1024 * ((j + 5) & 15)));
      if constexpr (NR >= 1) asm volatile("s_waitcnt lgkmcnt(2)");
      filler<j, NE, NA, NC>(e, x, acc, pk);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  const long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  float r = acc[0] + acc[1] + acc[2] + acc[3];
  for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) r += c32[j][q];
  for (int j = 0; j < 8; ++j) for (int q = 0; q < 4; ++q) r += c16[j][q];
  for (int i = 0; i < 8; ++i) r += e[i];
  for (int i = 0; i < 4; ++i) r += (float)pk[i];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(fr[0]), "+a"(fr[1]), "+a"(fr[2]), "+a"(fr[3]));
  for (int i = 0; i < 4; ++i) r += (float)fr[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
  std::vector<int> h(4096, 0x38383838);
  int* din; float* dout; long long* dc;
  if (hipMalloc(&din, h.size() * 4) || hipMalloc(&dout, 256 * 256 * 4) || hipMalloc(&dc, 8)) return 1;
  (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  auto run = [&](auto kern, const char* name) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, din, dout, 2000, dc); (void)hipDeviceSynchronize(); }
    long long c = 0; (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %.2f cycles per MFMA\n", name, (double)c / 2000 / 32);
  };
  run(k<0, 0, 0, 0>, "32x32x64 bare");
  run(k<0, 2, 0, 0>, "32x32x64 + 2 exp"); run(k<0, 4, 0, 0>, "32x32x64 + 4 exp"); run(k<0, 6, 0, 0>, "32x32x64 + 6 exp"); run(k<0, 8, 0, 0>, "32x32x64 + 8 exp");
  run(k<0, 0, 4, 0>, "32x32x64 + 4 add"); run(k<0, 0, 8, 0>, "32x32x64 + 8 add"); run(k<0, 0, 0, 2>, "32x32x64 + 2 cvt"); run(k<0, 0, 0, 4>, "32x32x64 + 4 cvt");
  run(k<0, 4, 4, 2>, "32x32x64 + 4 exp 4 add 2 cvt (one softmax share)");
  run(k<0, 4, 0, 2>, "32x32x64 + 4 exp 2 cvt (row sums elsewhere)");
  run(k<0, 3, 3, 2>, "32x32x64 + 3 exp 3 add 2 cvt");
  run(k<0, 4, -2, 2>, "32x32x64 + 4 exp 2 pk_add 2 cvt");
  run(k<0, 4, 102, 2>, "32x32x64 + 4 exp 2 max3 2 cvt");
  run(k<0, 0, -4, 0>, "32x32x64 + 4 pk_add");
  run(k<0, 0, 104, 0>, "32x32x64 + 4 max3");
  run(k<0, 0, 0, 0, 1>, "32x32x64 + 1 ds_read_b128 -> AGPR");
  run(k<0, 0, 0, 0, 2>, "32x32x64 + 2 ds_read_b128 -> AGPR");
  run(k<0, 4, 4, 2, 1>, "32x32x64 + softmax share + 1 read");
  run(k<0, 4, 4, 2, 2>, "32x32x64 + softmax share + 2 reads");
  run(k<0, 4, 0, 2, 1>, "32x32x64 + 4 exp 2 cvt + 1 read");
  run(k<0, 4, -2, 2, 1>, "32x32x64 + 4 exp 2 pk_add 2 cvt + 1 read");
  run(k1<2, 2>, "32x32x64 + 2 v_cvt_pk_fp8_f32");
  run(k1<5, 2>, "32x32x64 + 2 v_cvt_scalef32_pk_fp8_f32");
  run(k1<6, 2>, "32x32x64 + 2 (cvt_pk_bf16_f32 + cvt_scalef32_pk_fp8_bf16)");
  run(k1<7, 2>, "32x32x64 + 2 (cvt_pkrtz_f16_f32 + cvt_scalef32_pk_fp8_f16)");
  run(k1<9, 2>, "32x32x64 + 2 v_cvt_pk_bf16_f32");
  run(k1<9, 4>, "32x32x64 + 4 v_cvt_pk_bf16_f32");
  run(k1<0, 4>, "32x32x64 + 4 v_exp_f32");
  run(k1<8, 4>, "32x32x64 + 4 v_exp_f16");
  run(k1<1, 4>, "32x32x64 + 4 v_add_f32");
  run(k1<10, 4>, "32x32x64 + 4 v_cvt_pk_u8_f32, one chain");
  run(k1<11, 4>, "32x32x64 + 4 v_cvt_pk_u8_f32, four registers");
  run(k1<11, 8>, "32x32x64 + 8 v_cvt_pk_u8_f32, four registers");
  run(k1<12, 4>, "32x32x64 + 4 v_fmaak_f32");
  run(k1<12, 8>, "32x32x64 + 8 v_fmaak_f32");
  run(k1<13, 4>, "32x32x64 + 4 v_fma_f32");
  run(k1<13, 8>, "32x32x64 + 8 v_fma_f32");
  run(k1<16, 4>, "32x32x64 + 4 v_pk_fma_f32");
  run(k1<14, 4>, "32x32x64 + 4 v_cvt_pk_fp8_f32, one chain");
  run(k1<2, 4>, "32x32x64 + 4 v_cvt_pk_fp8_f32, four registers");
  run(k1<15, 8>, "32x32x64 + 8 v_add_f32, one chain");
  run(k1<1, 8>, "32x32x64 + 8 v_add_f32, four chains");
  run(k1<1, 16>, "32x32x64 + 16 v_add_f32, four chains");
  run(k1<17, 16>, "32x32x64 + 16 v_or_b32");
  run(k<1, 0, 0, 0>, "16x16x128 bare");
  run(k<1, 1, 0, 0>, "16x16x128 + 1 exp"); run(k<1, 2, 0, 0>, "16x16x128 + 2 exp"); run(k<1, 3, 0, 0>, "16x16x128 + 3 exp"); run(k<1, 4, 0, 0>, "16x16x128 + 4 exp");
  run(k<1, 2, 2, 1>, "16x16x128 + 2 exp 2 add 1 cvt (one softmax share)");
  run(k<1, 2, 0, 1>, "16x16x128 + 2 exp 1 cvt");
  return 0;
}
