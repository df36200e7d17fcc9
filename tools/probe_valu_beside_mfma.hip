// Issue cost of VALU instructions placed behind v_mfma_f32_16x16x32_bf16 by a lone wave per SIMD:
// cycles per MFMA gap with 0..4 copies of one instruction in every gap (16 = hidden).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe_valu_beside_mfma.hip -o build/probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); } }
template <int OP>
__device__ __forceinline__ void valu_op(float& e, float x) {
  if constexpr (OP == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(x));
  if constexpr (OP == 1) asm volatile("v_exp_legacy_f32 %0, %1" : "=v"(e) : "v"(x));
  if constexpr (OP == 2) asm volatile("v_exp_f16 %0, %1" : "=v"(e) : "v"(x));
  if constexpr (OP == 3) asm volatile("v_add_f32 %0, %1, %1" : "=v"(e) : "v"(x));
  if constexpr (OP == 4) asm volatile("v_ldexp_f32 %0, %1, %1" : "=v"(e) : "v"(x));
}
template <int OP, int NOPS>
__global__ __launch_bounds__(256) void k(const u32x4* in, float* out, int iters, long long* cyc) {
  const int lane = threadIdx.x & 63;
  u32x4 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(i * 64 + lane) & 4095]; b[i] = in[(i * 64 + lane + 1024) & 4095]; }
  float x[8], e[8];
  for (int i = 0; i < 8; ++i) { x[i] = -0.01f * (lane + i); e[i] = 0.f; }
  f32x4 c16[8] = {};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    sfor<0, 64>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c16[j & 7]) : "v"(a[j & 7]), "v"(b[(j >> 3) & 7]));
      sfor<0, NOPS>([&](auto nc) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
        valu_op<OP>(e[(j + n) & 7], x[(j + n) & 7]);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  const long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  float r = 0.f;
  for (int j = 0; j < 8; ++j) for (int q = 0; q < 4; ++q) r += c16[j][q];
  for (int i = 0; i < 8; ++i) r += e[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
// gap pattern: eight masks (512 = v_dot2c_f32_bf16 of the pair converted in the previous gap with a ones pair, 1024 = the same on the pair converted four gaps ago, 2048 = the VOP3P form v_dot2_f32_bf16 of 512; 1 exp, 2 add, 4 convert, 8 = exp issued last, 16 = a second add, 32 = a ds_read_b128 directly behind the MFMA, 64 = a second one, 128 = s_waitcnt lgkmcnt(8), 256 = s_add_u32 m0), repeated over the 64 gaps
template <int M0, int M1, int M2, int M3, int M4 = M0, int M5 = M1, int M6 = M2, int M7 = M3>
__global__ __launch_bounds__(256) void kp(const u32x4* in, float* out, int iters, long long* cyc) {
  const int lane = threadIdx.x & 63;
  u32x4 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(i * 64 + lane) & 4095]; b[i] = in[(i * 64 + lane + 1024) & 4095]; }
  float x[8], e[8], acc[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned pk[8] = {};
  unsigned ones = 0x3f803f80u;
  asm volatile("" : "+v"(ones));
  for (int i = 0; i < 8; ++i) { x[i] = -0.01f * (lane + i); e[i] = 0.f; }
  f32x4 c16[8] = {};
  __shared__ u32x4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = in[i];
  __syncthreads();
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) u32x4*)(lds + lane);
  u32x4 sink[4] = {};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    sfor<0, 64>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int MS[8] = {M0, M1, M2, M3, M4, M5, M6, M7};
      constexpr int M = MS[j & 7];
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c16[j & 7]) : "v"(a[j & 7]), "v"(b[(j >> 3) & 7]));
      if constexpr (M & 32) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink[j & 3]) : "v"(laddr), "i"(1024 * (j & 7)));
      if constexpr (M & 128) asm volatile("s_waitcnt lgkmcnt(8)");
      if constexpr (M & 64) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink[(j + 1) & 3]) : "v"(laddr), "i"(1024 * ((j + 3) & 7)));
      if constexpr (M & 256) asm volatile("s_add_u32 m0, %0, 0x1000" :: "s"(iters) : "scc");
      if constexpr ((M & 1) && !(M & 8)) asm volatile("v_exp_f32 %0, %1" : "=v"(e[j & 7]) : "v"(x[j & 7]));
      if constexpr (M & 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[j & 3]) : "v"(e[(j + 4) & 7]));
      if constexpr (M & 16) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[(j + 1) & 3]) : "v"(e[(j + 6) & 7]));
      if constexpr (M & 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[j & 7]) : "v"(e[(j + 3) & 7]), "v"(e[(j + 5) & 7]));
      if constexpr (M & 512) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[j & 3]) : "v"(pk[(j + 7) & 7]), "v"(ones));
      if constexpr (M & 1024) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[j & 3]) : "v"(pk[(j + 4) & 7]), "v"(ones));
      if constexpr (M & 2048) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(pk[(j + 7) & 7]), "v"(ones));
      if constexpr ((M & 1) && (M & 8)) asm volatile("v_exp_f32 %0, %1" : "=v"(e[j & 7]) : "v"(x[j & 7]));
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  const long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sink[0]), "+v"(sink[1]), "+v"(sink[2]), "+v"(sink[3]));
  float r = acc[0] + acc[1] + acc[2] + acc[3] + (float)(sink[0][0] + sink[1][1] + sink[2][2] + sink[3][3]);
  for (int j = 0; j < 8; ++j) for (int q = 0; q < 4; ++q) r += c16[j][q];
  for (int i = 0; i < 8; ++i) r += e[i] + (float)pk[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
  std::vector<uint16_t> h(4096 * 8, 0x3f80);
  u32x4* din; float* dout; long long* dc;
  if (hipMalloc(&din, h.size() * 2) || hipMalloc(&dout, 256 * 256 * 4) || hipMalloc(&dc, 8)) return 1;
  (void)hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  auto run = [&](auto kern, const char* name) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, din, dout, 2000, dc); (void)hipDeviceSynchronize(); }
    long long c = 0; (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-32s %.2f cycles per MFMA gap\n", name, (double)c / 2000 / 64);
  };
  run(k<0, 0>, "bare");
  run(k<0, 1>, "1 v_exp_f32"); run(k<0, 2>, "2 v_exp_f32"); run(k<0, 3>, "3 v_exp_f32");
  run(k<1, 1>, "1 v_exp_legacy_f32"); run(k<1, 2>, "2 v_exp_legacy_f32"); run(k<1, 3>, "3 v_exp_legacy_f32");
  run(k<2, 1>, "1 v_exp_f16"); run(k<2, 2>, "2 v_exp_f16"); run(k<2, 3>, "3 v_exp_f16");
  run(k<3, 1>, "1 v_add_f32"); run(k<3, 2>, "2 v_add_f32"); run(k<3, 3>, "3 v_add_f32"); run(k<3, 4>, "4 v_add_f32");
  run(k<4, 2>, "2 v_ldexp_f32"); run(k<4, 3>, "3 v_ldexp_f32");
  auto runp = [&](auto kern, const char* name) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, din, dout, 2000, dc); (void)hipDeviceSynchronize(); }
    long long c = 0; (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-32s %.2f cycles per 4 gaps (64 = hidden)\n", name, (double)c / 2000 / 16);
  };
  runp(kp<1, 6, 1, 2>, "E | AC | E | A");
  runp(kp<3, 4, 3, 0>, "EA | C | EA | -");
  runp(kp<3, 5, 2, 0>, "EA | EC | A | -");
  runp(kp<11, 4, 11, 0>, "AE | C | AE | -");
  runp(kp<11, 13, 2, 0>, "AE | CE | A | -");
  runp(kp<1, 2, 1, 6>, "E | A | E | AC");
  runp(kp<1, 2, 5, 2>, "E | A | EC | A");
  runp(kp<13, 2, 1, 2>, "CE | A | E | A");
  runp(kp<1, 1, 6, 2>, "E | E | AC | A");
  runp(kp<3, 3, 4, 0>, "EA | EA | C | -");
  runp(kp<1, 1, 2, 6>, "E | E | A | AC");
  runp(kp<1, 1, 18, 4>, "E | E | AA | C");
  runp(kp<1, 1, 4, 18>, "E | E | C | AA");
  runp(kp<1, 1, 1, 1, 6, 2, 6, 2>, "E | E | E | E | AC | A | AC | A");
  runp(kp<1, 1, 1, 1, 6, 6, 2, 2>, "E | E | E | E | AC | AC | A | A");
  runp(kp<1, 1, 1, 1, 18, 18, 4, 4>, "E | E | E | E | AA | AA | C | C");
  runp(kp<1, 6, 1, 2, 1, 6, 1, 2>, "E | AC | E | A (x2)");
  runp(kp<1, 1, 0, 0>, "E | E | - | - (exps only)");
  runp(kp<0, 0, 6, 2>, "- | - | AC | A (consumers only)");
  runp(kp<0, 0, 2, 2>, "- | - | A | A");
  runp(kp<0, 0, 4, 0>, "- | - | C | -");
  runp(kp<33, 1, 6, 2>, "RE | E | AC | A");
  runp(kp<1, 33, 6, 2>, "E | RE | AC | A");
  runp(kp<1, 1, 38, 2>, "E | E | RAC | A");
  runp(kp<1, 1, 6, 34>, "E | E | AC | RA");
  runp(kp<33, 1, 6, 34>, "RE | E | AC | RA");
  runp(kp<1, 1, 6, 34, 1, 1, 6, 2>, "E | E | AC | RA | E | E | AC | A");
  runp(kp<1, 1, 6, 98>, "E | E | AC | RRA");
  runp(kp<1, 1, 6, 162>, "E | E | AC | WRA");
  runp(kp<1, 1, 6, 226>, "E | E | AC | WRRA");
  runp(kp<1, 1, 6, 130>, "E | E | AC | WA");
  runp(kp<129, 1, 6, 2>, "WE | E | AC | A");
  runp(kp<257, 1, 6, 2>, "ME | E | AC | A");
  runp(kp<1, 1, 6, 258>, "E | E | AC | MA");
  runp(kp<1, 1, 6, 34, 1, 1, 6, 34>, "E | E | AC | RA (x2)");
  runp(kp<1, 1, 4, 512>, "E | E | C | D");
  runp(kp<1, 1, 4, 2048>, "E | E | C | D(vop3p)");
  runp(kp<1, 1, 4 | 1024, 0>, "E | E | CD | -");
  runp(kp<1, 4 | 1024, 1, 0>, "E | CD | E | -");
  runp(kp<1, 1, 4, 512 | 32>, "E | E | C | RD");
  runp(kp<1, 1, 4 | 1024, 32>, "E | E | CD | R");
  runp(kp<1, 1, 4 | 1024, 32 | 64>, "E | E | CD | RR");
  runp(kp<1, 1, 4 | 1024, 32 | 256>, "E | E | CD | RM");
  runp(kp<1, 1, 4 | 1024, 32, 1, 1, 4 | 1024, 0>, "E | E | CD | R | E | E | CD | -");
  runp(kp<1, 1, 6, 34, 1, 1, 6, 2>, "E | E | AC | RA | E | E | AC | A (present)");
  runp(kp<32, 0, 0, 0>, "R | - | - | -");
  runp(kp<32, 32, 32, 32>, "R | R | R | R");
  return 0;
}
