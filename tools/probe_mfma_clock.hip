// Sustained rate of the two bf16 MFMA shapes on register operands holding random data, one wave
// per SIMD: tells whether the shape alone changes the clock the chip holds under its power limit.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_clock.hip -o build/probe_mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    uint4 x = in[(i * 64 + lane) & 4095], y = in[(i * 64 + lane + 1024) & 4095];
    a[i] = __builtin_bit_cast(bf16x8, x);
    b[i] = __builtin_bit_cast(bf16x8, y);
  }
  float r = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 c[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + j) & 7], c[j], 0, 0, 0);
    for (int j = 0; j < 4; ++j)
      for (int k = 0; k < 16; ++k) r += c[j][k];
  } else {
    f32x4 c[8] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(i + j) & 7], c[j], 0, 0, 0);
    for (int j = 0; j < 8; ++j)
      for (int k = 0; k < 4; ++k) r += c[j][k];
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  std::vector<uint16_t> h(4096 * 8);
  srand(1);
  for (auto& v : h) {  // bf16 of magnitude ~1 with random sign and mantissa
    v = (uint16_t)(((rand() & 1) << 15) | ((126 + (rand() % 3)) << 7) | (rand() & 127));
  }
  uint4* din;
  float* dout;
  const int nwg = 256 * 16;
  hipMalloc(&din, h.size() * 2);
  hipMalloc(&dout, nwg * 256 * 4);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int shape : {32, 16, 32, 16}) {
    const int iters = 20000;  // 32 (or 64) MFMAs of 32768 (16384) x ... flops per trip
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(nwg), dim3(256), 0, 0, din, dout, iters);
      else hipLaunchKernelGGL(mfma_loop<16>, dim3(nwg), dim3(256), 0, 0, din, dout, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)nwg * 4 * iters * (shape == 32 ? 32.0 * 2 * 32 * 32 * 16 : 64.0 * 2 * 16 * 16 * 32);
      const double cyc = (double)(nwg / 256) * iters * 1024;  // MFMA cycles per SIMD
      printf("shape %dx%d: %.2f ms  %.1f TFLOP/s  implied clock %.3f GHz\n", shape, shape, ms, flops / ms * 1e-9, cyc / ms * 1e-6);
    }
  }
  return 0;
}
