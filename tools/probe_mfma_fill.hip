// The two bf16 MFMA shapes with an attention-like filler stream between the MFMAs of a lone wave
// per SIMD: per 1024 MFMA cycles 32 v_exp_f32, 32 v_add_f32, 16 v_cvt_pk_bf16_f32 and 24 LDS reads,
// placed one group per gap.  Prints the sustained rate, i.e. schedule and held clock together.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_fill.hip -o build/probe_mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float exp2_op(float x) { float r; asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ void add_op(float& acc, float x) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x)); }
__device__ __forceinline__ unsigned cvt_op(float lo, float hi) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

template <int SHAPE, int FILL>
__global__ __launch_bounds__(256) void mfma_loop(const u32x4* in, float* out, int iters, long long* cyc) {
  __shared__ u32x4 lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
  __syncthreads();
  u32x4 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(i * 64 + lane) & 4095];
    b[i] = in[(i * 64 + lane + 1024) & 4095];
  }
  float x[32], e[32], acc0 = 0.f, acc1 = 0.f;
  unsigned pk[16];
  for (int i = 0; i < 32; ++i) { x[i] = -0.01f * (lane + i); e[i] = 0.f; }
  for (int i = 0; i < 16; ++i) pk[i] = 0;
  float r = 0.f;
  f32x16 c32[4] = {};
  f32x4 c16[8] = {};
  const u32x4* lp = lds + lane;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    // 32 filler groups per trip: group j = {exp x[j]; add e[j-1]; cvt on odd j; read on j % 4 != 3}
    auto filler_a = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (FILL & 1) e[j] = exp2_op(x[j]);
    };
    auto filler_b = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (FILL & 2) add_op((j & 1) ? acc1 : acc0, e[(j + 31) & 31]);
      if constexpr ((FILL & 4) && (j & 1)) pk[j >> 1] = cvt_op(e[(j + 30) & 31], e[(j + 31) & 31]);
      if constexpr ((FILL & 8) && (j & 3) != 3) a[(j + 4) & 7] = lp[64 * (j & 31)];
    };
    if constexpr (SHAPE == 32) {
      sfor<0, 32>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c32[j & 3]) : "v"(a[j & 7]), "v"(b[(j >> 2) & 7]));
        filler_a(jc);
        filler_b(jc);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      sfor<0, 32>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c16[(2 * j) & 7]) : "v"(a[j & 7]), "v"(b[(j >> 2) & 7]));
        filler_a(jc);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c16[(2 * j + 1) & 7]) : "v"(a[j & 7]), "v"(b[(j >> 2) & 7]));
        filler_b(jc);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) r += c32[j][k];
  for (int j = 0; j < 8; ++j) for (int k = 0; k < 4; ++k) r += c16[j][k];
  for (int i = 0; i < 16; ++i) r += (float)pk[i];
  out[blockIdx.x * 256 + threadIdx.x] = r + acc0 + acc1;
}

int main() {
  std::vector<uint16_t> h(4096 * 8);
  srand(1);
  for (auto& v : h) v = (uint16_t)(((rand() & 1) << 15) | ((126 + (rand() % 3)) << 7) | (rand() & 127));
  u32x4* din;
  float* dout;
  const int nwg = 256 * 8;
  if (hipMalloc(&din, h.size() * 2) != hipSuccess || hipMalloc(&dout, nwg * 256 * 4) != hipSuccess) return 1;
  if (hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 10000;
  long long* dcyc;
  if (hipMalloc(&dcyc, 8) != hipSuccess) return 1;
  auto run = [&](auto kern, const char* name) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, din, dout, iters, dcyc);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      long long c = 0;
      (void)hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
      const double flops = (double)nwg * 4 * iters * 32.0 * 2 * 32 * 32 * 16;
      if (rep) printf("%-28s %.2f ms  %.1f TFLOP/s  %.1f cycles per 1024 MFMA cycles\n", name, ms, flops / ms * 1e-9, (double)c / iters);
    }
  };
  run(mfma_loop<32, 0>, "32x32x16 bare");
  run(mfma_loop<16, 0>, "16x16x32 bare");
  run(mfma_loop<32, 1>, "32x32x16 exp");
  run(mfma_loop<16, 1>, "16x16x32 exp");
  run(mfma_loop<32, 3>, "32x32x16 exp add");
  run(mfma_loop<16, 3>, "16x16x32 exp add");
  run(mfma_loop<32, 7>, "32x32x16 exp add cvt");
  run(mfma_loop<16, 7>, "16x16x32 exp add cvt");
  run(mfma_loop<32, 8>, "32x32x16 lds");
  run(mfma_loop<16, 8>, "16x16x32 lds");
  run(mfma_loop<32, 15>, "32x32x16 exp add cvt lds");
  run(mfma_loop<16, 15>, "16x16x32 exp add cvt lds");
  run(mfma_loop<16, 6>, "16x16x32 add cvt");
  return 0;
}
