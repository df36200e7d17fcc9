// Rounding and clamping of v_cvt_pk_u8_f32 on gfx950 (prints the byte for a few inputs).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_cvt_u8.hip -o build/probe_cvt_u8 && ./build/probe_cvt_u8
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, unsigned* y, int n) {
  int i = threadIdx.x;
  if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0);
}
int main() {
  const float h[] = {-3.f, -0.4f, 0.3f, 0.5f, 0.7f, 1.49f, 1.5f, 1.51f, 2.5f, 3.5f, 3.99f, 126.5f, 254.6f, 255.5f, 300.f, 1e9f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* dx; unsigned* dy; unsigned out[n];
  if (hipMalloc(&dx, sizeof(h)) != hipSuccess || hipMalloc(&dy, sizeof(out)) != hipSuccess) return 1;
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, dy, n);
  if (hipMemcpy(out, dy, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int i = 0; i < n; ++i) printf("%g -> %u\n", h[i], out[i]);
  return 0;
}
