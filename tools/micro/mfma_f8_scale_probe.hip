// Probe of the per-lane E8M0 scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950: which lane's scale byte applies to
// which (row, 32-wide k block) of A, and to which (column, block) of B?  All operand bytes are e4m3 1.0; lane l carries scale
// 2^(l % 5) (byte 0 of its scale VGPR, opsel 0).  Hypothesis checked: lane l scales row / column l & 31, block l >> 5 (the
// operand layout: lane l holds k = 32 (l >> 5) + byte).  tools only.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/mfma_f8_scale_probe tools/micro/mfma_f8_scale_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ void probe(float* __restrict__ out, int which) {
  const int lane = threadIdx.x;
  i32x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
  f32x16_t c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  const int s = 127 + (lane % 5);
  if (which == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s, 0, 127);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, s);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

int main() {
  float* d; hipMalloc(&d, 64 * 16 * 4);
  std::vector<float> o(64 * 16);
  for (int which = 0; which < 2; ++which) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, which);
    hipMemcpy(o.data(), d, o.size() * 4, hipMemcpyDeviceToHost);
    // D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    int bad = 0;
    printf("%s scale per lane = 2^(lane %% 5): D[i][j] expected under 'lane l -> index l & 31, block l >> 5':\n", which ? "B" : "A");
    for (int i = 0; i < 32; ++i) {
      // value at (row i, col 0) for A; (row 0, col i) for B
      float got;
      if (which == 0) { const int half = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3); got = o[(32 * half + 0) * 16 + r]; }
      else got = o[i * 16 + 0];
      const float want = 32.f * (float)(1 << (i % 5)) + 32.f * (float)(1 << ((i + 32) % 5));
      if (got != want) ++bad;
      printf("  index %2d: got %7.1f want %7.1f%s\n", i, got, want, got == want ? "" : "   <-- differs");
    }
    printf("%s: %d of 32 differ\n", which ? "B" : "A", bad);
  }
  return 0;
}
