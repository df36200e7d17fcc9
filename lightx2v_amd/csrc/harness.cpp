// Native kernel harness: times the C-ABI entry points at real Wan shapes with HIP events and
// spot-checks results against a host fp32 computation.  No Python start-up, so a GPU-box call
// spends its minutes on kernels.   usage: kernel_harness [attn|gemm|ew|all] [S] [heads] [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <vector>

#include "../../include/lx2v_hip.h"

#define HIPCHK(x)                                                                 \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)
#define LXCHK(x)                                                    \
  do {                                                              \
    int s = (x);                                                    \
    if (s != 0) {                                                   \
      fprintf(stderr, "lx2v error %d: %s\n", s, lx2v_last_error()); \
      exit(3);                                                      \
    }                                                               \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

__global__ void fill_normal(uint16_t* p, size_t n, uint64_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    // counter hash -> two uniforms -> Box-Muller
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u1 = ((z & 0xffffff) + 1) * (1.0f / 16777217.0f);
    const float u2 = ((z >> 24) & 0xffffff) * (1.0f / 16777216.0f);
    const float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * scale;
    uint32_t u = __float_as_uint(g);
    u += 0x7fff + ((u >> 16) & 1);
    p[i] = (uint16_t)(u >> 16);
  }
}

static uint16_t* dev_normal(size_t n, uint64_t seed, float scale) {
  uint16_t* p;
  HIPCHK(hipMalloc(&p, n * 2));
  hipLaunchKernelGGL(fill_normal, dim3(2048), dim3(256), 0, 0, p, n, seed, scale);
  HIPCHK(hipDeviceSynchronize());
  return p;
}

template <class F>
static float time_ms(F&& fn, int iters, int warmup = 2) {
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a));
  HIPCHK(hipEventCreate(&b));
  for (int i = 0; i < warmup; ++i) fn();
  HIPCHK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) fn();
  HIPCHK(hipEventRecord(b, 0));
  HIPCHK(hipEventSynchronize(b));
  float ms;
  HIPCHK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

static void bench_attn(int S, int H, int iters, int skv = -1, bool rope = false) {
  if (skv < 0) skv = S;
  const size_t nq = (size_t)S * H * 128, nk = (size_t)skv * H * 128;
  uint16_t* q = dev_normal(nq, 1, 1.0f);
  uint16_t* k = dev_normal(nk, 2, 1.0f);
  uint16_t* v = dev_normal(nk, 3, 1.0f);
  uint16_t* o;
  HIPCHK(hipMalloc(&o, nq * 2));
  float* table = nullptr;
  if (rope) {
    std::vector<float> t((size_t)S * 128);
    for (int i = 0; i < S; ++i)
      for (int j = 0; j < 64; ++j) {
        const float ang = (float)(i % 97) * powf(10000.f, -(float)(j % 22) / 22.f);
        t[(size_t)i * 128 + j] = cosf(ang);
        t[(size_t)i * 128 + 64 + j] = sinf(ang);
      }
    HIPCHK(hipMalloc(&table, t.size() * 4));
    HIPCHK(hipMemcpy(table, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  }
  const float scale = 1.0f / sqrtf(128.f);
  const int64_t ld = (int64_t)H * 128;
  auto run = [&] { LXCHK(lx2v_attn_fwd_bf16(q, ld, k, ld, v, ld, o, ld, S, skv, H, 128, scale, table, nullptr)); };
  const float ms = time_ms(run, iters);
  const double flops = 4.0 * S * (double)skv * H * 128;
  printf("attn  S=%d Skv=%d H=%d rope=%d : %.3f ms  %.1f TFLOP/s (%.1f%% of 2500)\n", S, skv, H, (int)rope, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 25.0);

  // spot check a few rows of head H-1 against host fp32 softmax (no rope)
  if (!rope) {
    std::vector<uint16_t> hq(nq), hk(nk), hv(nk), ho(nq);
    HIPCHK(hipMemcpy(hq.data(), q, nq * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hk.data(), k, nk * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hv.data(), v, nk * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ho.data(), o, nq * 2, hipMemcpyDeviceToHost));
    double maxerr = 0;
    const int rows[4] = {0, S / 3, S / 2 + 17 < S ? S / 2 + 17 : S - 1, S - 1};
    for (int ri = 0; ri < 4; ++ri) {
      const int r = rows[ri], h = H - 1;
      std::vector<double> sc(skv);
      double mx = -1e300;
      for (int j = 0; j < skv; ++j) {
        double s = 0;
        for (int d = 0; d < 128; ++d) s += (double)bf2f(hq[((size_t)r * H + h) * 128 + d]) * bf2f(hk[((size_t)j * H + h) * 128 + d]);
        sc[j] = s * scale;
        mx = fmax(mx, sc[j]);
      }
      double den = 0;
      for (int j = 0; j < skv; ++j) {
        sc[j] = exp(sc[j] - mx);
        den += sc[j];
      }
      for (int d = 0; d < 128; ++d) {
        double acc = 0;
        for (int j = 0; j < skv; ++j) acc += sc[j] * bf2f(hv[((size_t)j * H + h) * 128 + d]);
        maxerr = fmax(maxerr, fabs(acc / den - bf2f(ho[((size_t)r * H + h) * 128 + d])));
      }
    }
    printf("      spot-check max|err| = %.4g %s\n", maxerr, maxerr < 2e-2 ? "OK" : "MISMATCH");
  }
  hipFree(q); hipFree(k); hipFree(v); hipFree(o);
  if (table) hipFree(table);
}

static void bench_attn_fp8(int S, int H, int iters) {
  const size_t n = (size_t)S * H * 128;
  uint16_t* q = dev_normal(n, 1, 1.0f);
  uint16_t* k = dev_normal(n, 2, 1.0f);
  uint16_t* v = dev_normal(n, 3, 1.0f);
  uint16_t* o;
  HIPCHK(hipMalloc(&o, n * 2));
  const int64_t wsb = lx2v_attn_fp8_workspace_bytes(S, S, H);
  void* ws;
  HIPCHK(hipMalloc(&ws, (size_t)wsb));
  const int64_t ld = (int64_t)H * 128;
  auto run = [&] { LXCHK(lx2v_attn_fwd_fp8(q, ld, k, ld, v, ld, o, ld, S, S, H, 128, 1.0f / sqrtf(128.f), ws, wsb, nullptr)); };
  const float ms = time_ms(run, iters);
  const double flops = 4.0 * S * (double)S * H * 128;
  printf("attn_fp8 S=%d H=%d (quantisation pre-pass included) : %.3f ms  %.1f TFLOP/s (%.1f%% of 5000)\n", S, H, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 50.0);
  hipFree(q); hipFree(k); hipFree(v); hipFree(o); hipFree(ws);
}

static void bench_gemm(int M, int N, int K, int epi, int iters) {
  uint16_t* a = dev_normal((size_t)M * K, 4, 1.0f);
  uint16_t* w = dev_normal((size_t)N * K, 5, 0.02f);
  uint16_t* c;
  HIPCHK(hipMalloc(&c, (size_t)M * N * 2));
  HIPCHK(hipMemset(c, 0, (size_t)M * N * 2));
  std::vector<float> hb(N), hg(N);
  for (int i = 0; i < N; ++i) {
    hb[i] = 0.01f * (float)((i * 37) % 19 - 9);
    hg[i] = 0.1f * (float)((i * 13) % 7 - 3);
  }
  float *bias, *gate;
  HIPCHK(hipMalloc(&bias, N * 4));
  HIPCHK(hipMalloc(&gate, N * 4));
  HIPCHK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(gate, hg.data(), N * 4, hipMemcpyHostToDevice));
  auto run = [&] { LXCHK(lx2v_gemm_bf16(a, K, nullptr, w, K, bias, c, N, M, N, K, epi, epi == 2 ? c : nullptr, N, epi == 2 ? gate : nullptr, nullptr)); };
  const float ms = time_ms(run, iters);
  const double flops = 2.0 * M * (double)N * K;
  printf("gemm  M=%d N=%d K=%d epi=%d : %.3f ms  %.1f TFLOP/s (%.1f%% of 2500)\n", M, N, K, epi, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 25.0);
  if (epi != 2) {
    LXCHK(lx2v_gemm_bf16(a, K, nullptr, w, K, bias, c, N, M, N, K, 0, nullptr, 0, nullptr, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<uint16_t> ha((size_t)K), hw((size_t)K);
    double maxerr = 0;
    const int rr[3] = {0, M / 2, M - 1}, cc[3] = {0, N / 3, N - 1};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        HIPCHK(hipMemcpy(ha.data(), a + (size_t)rr[i] * K, K * 2, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hw.data(), w + (size_t)cc[j] * K, K * 2, hipMemcpyDeviceToHost));
        double acc = hb[cc[j]];
        for (int kk = 0; kk < K; ++kk) acc += (double)bf2f(ha[kk]) * bf2f(hw[kk]);
        uint16_t got;
        HIPCHK(hipMemcpy(&got, c + (size_t)rr[i] * N + cc[j], 2, hipMemcpyDeviceToHost));
        maxerr = fmax(maxerr, fabs(acc - bf2f(got)) / (fabs(acc) + 1.0));
      }
    printf("      spot-check max rel err = %.4g %s\n", maxerr, maxerr < 1e-2 ? "OK" : "MISMATCH");
  }
  hipFree(a); hipFree(w); hipFree(c); hipFree(bias); hipFree(gate);
}

static void bench_gemm_fp8(int M, int N, int K, int epi, int iters) {
  uint16_t* a = dev_normal((size_t)M * K, 4, 1.0f);
  uint16_t* w = dev_normal((size_t)N * K, 5, 0.02f);
  uint8_t *a8, *w8;
  float *as, *ws, *bias;
  uint16_t* c;
  HIPCHK(hipMalloc(&a8, (size_t)M * K));
  HIPCHK(hipMalloc(&w8, (size_t)N * K));
  HIPCHK(hipMalloc(&as, (size_t)M * 4));
  HIPCHK(hipMalloc(&ws, (size_t)N * 4));
  HIPCHK(hipMalloc(&bias, (size_t)N * 4));
  HIPCHK(hipMemset(bias, 0, (size_t)N * 4));
  HIPCHK(hipMalloc(&c, (size_t)M * N * 2));
  LXCHK(lx2v_quant_fp8_rowwise(w, K, w8, K, ws, N, K, nullptr));
  const float qms = time_ms([&] { LXCHK(lx2v_quant_fp8_rowwise(a, K, a8, K, as, M, K, nullptr)); }, iters);
  printf("quant_fp8 rows=%d dim=%d : %.3f ms  %.0f GB/s\n", M, K, qms, 3.0 * M * K / qms * 1e-6);
  const float ms = time_ms([&] { LXCHK(lx2v_gemm_fp8(a8, K, as, w8, K, ws, bias, c, N, M, N, K, epi, epi == 2 ? c : nullptr, N, nullptr, nullptr)); }, iters);
  const double flops = 2.0 * M * (double)N * K;
  printf("gemm_fp8 M=%d N=%d K=%d epi=%d : %.3f ms  %.1f TFLOP/s (%.1f%% of 5000)\n", M, N, K, epi, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 50.0);
  hipFree(a); hipFree(w); hipFree(a8); hipFree(w8); hipFree(as); hipFree(ws); hipFree(bias); hipFree(c);
}

static void bench_ew(int S, int D, int iters) {
  uint16_t* x = dev_normal((size_t)S * D, 6, 1.0f);
  uint16_t* y = dev_normal((size_t)S * D, 7, 1.0f);
  std::vector<float> hv(D, 0.1f);
  float* vec;
  HIPCHK(hipMalloc(&vec, D * 4));
  HIPCHK(hipMemcpy(vec, hv.data(), D * 4, hipMemcpyHostToDevice));
  float* table;
  HIPCHK(hipMalloc(&table, (size_t)S * 128 * 4));
  HIPCHK(hipMemset(table, 0, (size_t)S * 128 * 4));
  const double bytes = (double)S * D * 2;
  float ms = time_ms([&] { LXCHK(lx2v_ln_modulate_bf16(x, D, nullptr, nullptr, vec, vec, y, D, S, D, 1e-6f, nullptr)); }, iters);
  printf("ln_modulate   S=%d D=%d : %.3f ms  %.0f GB/s\n", S, D, ms, 2 * bytes / ms * 1e-6);
  ms = time_ms([&] { LXCHK(lx2v_rmsnorm_rope_bf16(x, D, vec, 1, table, 128, y, D, S, D, 1e-6f, nullptr)); }, iters);
  printf("rmsnorm_rope  S=%d D=%d : %.3f ms  %.0f GB/s (+table)\n", S, D, ms, (2 * bytes + (double)S * 512) / ms * 1e-6);
  ms = time_ms([&] { LXCHK(lx2v_gate_residual_bf16(x, D, y, D, vec, S, D, nullptr)); }, iters);
  printf("gate_residual S=%d D=%d : %.3f ms  %.0f GB/s\n", S, D, ms, 3 * bytes / ms * 1e-6);
  ms = time_ms([&] { LXCHK(lx2v_gelu_tanh_bf16(x, y, (int64_t)S * D, nullptr)); }, iters);
  printf("gelu_tanh     n=%lld : %.3f ms  %.0f GB/s\n", (long long)S * D, ms, 2 * bytes / ms * 1e-6);
  hipFree(x); hipFree(y); hipFree(vec); hipFree(table);
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  const int S = argc > 2 ? atoi(argv[2]) : 32760;
  const int H = argc > 3 ? atoi(argv[3]) : 12;
  const int iters = argc > 4 ? atoi(argv[4]) : 3;
  const int D = H * 128;
  const int F = D == 1536 ? 8960 : (D == 5120 ? 13824 : 4 * D);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
  if (!strcmp(what, "selfattn")) bench_attn(S, H, iters);
  if (!strcmp(what, "selfattn8")) bench_attn_fp8(S, H, iters);
  if (!strcmp(what, "gemm1")) bench_gemm(S, D, D, 0, iters);
  if (!strcmp(what, "gemm8_1")) bench_gemm_fp8(S, D, D, 0, iters);
  if (!strcmp(what, "attn") || !strcmp(what, "all")) {
    bench_attn(S, H, iters);
    bench_attn(S, H, iters, 512);
    bench_attn(S, H, iters, -1, true);
  }
  if (!strcmp(what, "gemm") || !strcmp(what, "all")) {
    bench_gemm(S, 3 * D, D, 0, iters);
    bench_gemm(S, D, D, 2, iters);
    bench_gemm(S, F, D, 1, iters);
    bench_gemm(S, D, F, 2, iters);
    bench_gemm(512, 2 * D, D, 0, iters);
  }
  if (!strcmp(what, "gemm8") || !strcmp(what, "all")) {
    bench_gemm_fp8(S, 3 * D, D, 0, iters);
    bench_gemm_fp8(S, D, D, 2, iters);
    bench_gemm_fp8(S, F, D, 1, iters);
    bench_gemm_fp8(S, D, F, 2, iters);
  }
  if (!strcmp(what, "ew") || !strcmp(what, "all")) bench_ew(S, D, iters);
  return 0;
}
