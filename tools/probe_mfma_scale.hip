// Probe: which lane / byte of the scale operands v_mfma_scale_f32_32x32x64_f8f6f4 applies to which
// (row, 32-wide k block).  A and B are all ones (e4m3 0x38) restricted to one k block; the scale
// register of lane l carries the exponent l - 40 in one byte and 2^-127 in the others, so
// D[m][n] = 32 * 2^(l - 40) names the lane whose byte was used (or ~0 when that byte is not the one read).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OPSEL, bool SIDE_A>
__global__ void probe(float* out, int kblock, int byte) {
  const int l = threadIdx.x;
  const int ones = 0x38383838;
  i32x8 a, b;
  const bool mine = (l >> 5) == kblock;  // this lane holds k block l>>5 of its row / column
  for (int i = 0; i < 8; ++i) { a[i] = mine ? ones : 0; b[i] = mine ? ones : 0; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const int probe_scale = (127 - 40 + l) << (8 * byte);
  const int unit = 0x7f7f7f7f;
  f32x16 d = SIDE_A ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL, probe_scale, 0, unit)
                    : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, unit, OPSEL, probe_scale);
  for (int i = 0; i < 16; ++i) out[l * 16 + i] = d[i];
}

template <int OPSEL, bool SIDE_A>
void run(float* dout, int kblock, int byte) {
  float h[1024];
  probe<OPSEL, SIDE_A><<<1, 64>>>(dout, kblock, byte);
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  // D[m][n]: lane l holds n = l & 31, m = (i & 3) + 8 (i >> 2) + 4 (l >> 5)
  printf("%s opsel=%d kblock=%d byte=%d: ", SIDE_A ? "A" : "B", OPSEL, kblock, byte);
  for (int idx = 0; idx < 32; ++idx) {
    // A side: scale belongs to row m -> look at D[m = idx][n = 0]; B side: column n -> D[0][n = idx]
    int m = SIDE_A ? idx : 0, n = SIDE_A ? 0 : idx;
    int lane = n + 32 * ((m >> 2) & 1), i = (m & 3) + 4 * (m >> 3);
    float v = h[lane * 16 + i];
    if (v < 1e-20f) printf(" .");
    else printf(" %d", (int)lrintf(log2f(v / 32.f)) + 40);
  }
  printf("\n");
}

int main() {
  float* dout;
  hipMalloc(&dout, 4096);
  for (int kb = 0; kb < 2; ++kb)
    for (int byte = 0; byte < 4; ++byte) {
      run<0, true>(dout, kb, byte); run<1, true>(dout, kb, byte); run<2, true>(dout, kb, byte); run<3, true>(dout, kb, byte);
    }
  for (int kb = 0; kb < 2; ++kb)
    for (int byte = 0; byte < 4; ++byte) { run<0, false>(dout, kb, byte); run<1, false>(dout, kb, byte); }
  return 0;
}
