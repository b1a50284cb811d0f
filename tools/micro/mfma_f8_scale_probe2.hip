// Second probe of v_mfma_scale_f32_32x32x64_f8f6f4's per-lane scales: random small-integer fp8 operands, random per-lane scales
// on BOTH operands, result compared with a host MX computation under the layout "lane l: row / column l & 31, k block l >> 5".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
static const unsigned char T[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};   // e4m3 of 0..8
__global__ void probe(const unsigned char* A, const unsigned char* B, const int* sa, const int* sb, float* out) {
  const int lane = threadIdx.x;
  i32x8_t a, b;
  memcpy(&a, A + lane * 32, 32);
  memcpy(&b, B + lane * 32, 32);
  f32x16_t c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
int main() {
  srand(3);
  std::vector<unsigned char> A(64 * 32), B(64 * 32);
  std::vector<int> av(64 * 32), bv(64 * 32), sa(64), sb(64);
  for (int i = 0; i < 64 * 32; ++i) { av[i] = rand() % 9; bv[i] = rand() % 9; A[i] = T[av[i]]; B[i] = T[bv[i]]; }
  for (int l = 0; l < 64; ++l) { sa[l] = 127 + rand() % 6 - 3; sb[l] = 127 + rand() % 6 - 3; }
  unsigned char *dA, *dB; int *dsa, *dsb; float* dO;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dO, 64 * 16 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dO);
  std::vector<float> O(64 * 16);
  hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
  // hypotheses: which operand bytes form 32-wide block b of row i, and which lane's scale byte applies to it
  //   data 0: block b = the 32 bytes of lane i + 32 b                      (k = 32 (l >> 5) + byte)
  //   data 1: block b = bytes 16 b .. 16 b + 15 of lanes i and i + 32       (k = 32 (byte >> 4) + 16 (l >> 5) + (byte & 15))
  //   scale 0: block b scaled by lane i + 32 b;  scale 1: by lane i + 32 (1 - b)
  for (int dh = 0; dh < 2; ++dh)
    for (int sh = 0; sh < 2; ++sh) {
      int bad = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double want = 0;
          for (int blk = 0; blk < 2; ++blk) {
            double sum = 0;
            for (int k = 0; k < 32; ++k) {
              int la, ja;
              if (dh == 0) { la = 32 * blk; ja = k; } else { la = 32 * (k >> 4); ja = 16 * blk + (k & 15); }
              sum += av[(i + la) * 32 + ja] * bv[(j + la) * 32 + ja];
            }
            const int sl = sh == 0 ? 32 * blk : 32 * (1 - blk);
            want += sum * ldexp(1.0, sa[i + sl] - 127 + sb[j + sl] - 127);
          }
          const int half = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
          if (O[(32 * half + j) * 16 + r] != (float)want) ++bad;
        }
      printf("data hypothesis %d, scale hypothesis %d: %d of 1024 entries differ\n", dh, sh, bad);
    }
  return 0;
}
