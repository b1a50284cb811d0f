// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands) on gfx950: which k index does byte j of lane l hold in the
// A and in the B operand, and what do the E8M0 scale operands do?  Exact small-integer data (cdna_hip_programming.md:
// "Other dtypes: check the map with exact integer data before relying on it").  tools only.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/mfma_f8_probe tools/micro/mfma_f8_probe.hip && tools/micro/mfma_f8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// e4m3 (OCP, bias 7): 1.0 = 0x38; small integers 1..16 exactly
__host__ __device__ inline unsigned char e4m3_of_int(int v) {
  // v in 0..16
  const unsigned char t[17] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50, 0x51, 0x52, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58};
  return t[v];
}

// out[test][lane][16]: accumulator registers
__global__ void probe(const unsigned char* __restrict__ abytes, const unsigned char* __restrict__ bbytes, float* __restrict__ out,
                      int ntests, int scale_a, int scale_b) {
  const int lane = threadIdx.x;
  for (int t = 0; t < ntests; ++t) {
    i32x8_t a, b;
    memcpy(&a, abytes + ((size_t)t * 64 + lane) * 32, 32);
    memcpy(&b, bbytes + ((size_t)t * 64 + lane) * 32, 32);
    f32x16_t c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0 /*A fp8*/, 0 /*B fp8*/, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 16; ++r) out[((size_t)t * 64 + lane) * 16 + r] = c[r];
  }
}

int main() {
  // test t = 0..63: A one-hot: lane la = 32*(t>>5) (row 0 of half t>>5), byte ja = t & 31, value 1.0.
  // B: every lane lb, byte jb: value (jb & 15) + 1 in tests 0..63; second bank of tests 64..127 with value (jb >> 4) + 1 + 2 * (lb >> 5)
  const int nt = 128;
  std::vector<unsigned char> A(nt * 64 * 32, 0), B(nt * 64 * 32, 0);
  for (int t = 0; t < nt; ++t) {
    const int tt = t & 63, la = 32 * (tt >> 5), ja = tt & 31;
    A[((size_t)t * 64 + la) * 32 + ja] = 0x38;
    for (int lb = 0; lb < 64; ++lb)
      for (int jb = 0; jb < 32; ++jb)
        B[((size_t)t * 64 + lb) * 32 + jb] = e4m3_of_int(t < 64 ? (jb & 15) + 1 : (jb >> 4) + 1 + 2 * (lb >> 5));
  }
  unsigned char *dA, *dB; float* dO;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dO, (size_t)nt * 64 * 16 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  std::vector<float> O((size_t)nt * 64 * 16);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO, nt, 0x7F7F7F7F, 0x7F7F7F7F);
  hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
  // D[row 0][col c]: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): row 0 = register 0 of lanes 0..31
  printf("A one-hot at (half, byte) -> B k-slot it meets: (byte&15)+1 code, then (byte>>4)+1+2*half code  [read at D[0][col 0]]\n");
  for (int tt = 0; tt < 64; ++tt) {
    const float lo = O[((size_t)tt * 64 + 0) * 16 + 0], hi = O[((size_t)(64 + tt) * 64 + 0) * 16 + 0];
    const int jb = ((int)lo - 1) + 16 * ((((int)hi - 1) % 2)), hb = ((int)hi - 1) / 2;
    printf("  A(h=%d, j=%2d) -> lo %4.1f hi %4.1f => B(h=%d, j=%2d)%s\n", tt >> 5, tt & 31, lo, hi, hb, jb,
           (hb == (tt >> 5) && jb == (tt & 31)) ? "" : "   <-- differs");
  }
  // which row does lane la of A address?  A one-hot at lane la (all 64), byte 0; B all ones; find nonzero rows
  std::vector<unsigned char> A2(64 * 64 * 32, 0), B2(64 * 64 * 32, 0x38);
  for (int t = 0; t < 64; ++t) A2[((size_t)t * 64 + t) * 32 + 0] = 0x38;
  hipMemcpy(dA, A2.data(), A2.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B2.data(), B2.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO, 64, 0x7F7F7F7F, 0x7F7F7F7F);
  hipMemcpy(O.data(), dO, (size_t)64 * 64 * 16 * 4, hipMemcpyDeviceToHost);
  printf("A lane -> output row (C/D standard layout assumed):");
  for (int t = 0; t < 64; ++t) {
    int row = -1;
    for (int l = 0; l < 64 && row < 0; ++l)
      for (int r = 0; r < 16; ++r)
        if (O[((size_t)t * 64 + l) * 16 + r] != 0.f && (l & 31) == 0) { row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); break; }
    printf(" %d", row);
  }
  printf("\n");
  // scales: A one-hot (lane 0, byte 0) = 1, B all ones: D[0][0] = 1 * scaleA * scaleB
  hipMemcpy(dA, A2.data(), 64 * 32, hipMemcpyHostToDevice);
  for (int sa = 0x7E; sa <= 0x81; ++sa) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO, 1, sa * 0x01010101, 0x7F7F7F7F);
    hipMemcpy(O.data(), dO, 64 * 16 * 4, hipMemcpyDeviceToHost);
    printf("scale_a byte 0x%02X, scale_b 0x7F: D[0][0] = %g\n", sa, O[0]);
  }
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO, 1, 0x7F7F7F7F, 0x82828282);
  hipMemcpy(O.data(), dO, 64 * 16 * 4, hipMemcpyDeviceToHost);
  printf("scale_a 0x7F, scale_b byte 0x82: D[0][0] = %g\n", O[0]);
  return 0;
}
