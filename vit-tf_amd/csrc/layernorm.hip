// LayerNorm over the feature dim: fp32 residual stream in, 16-bit MFMA operand out.
//
// Replaces nn.LayerNorm(eps=1e-6) (norm1 / norm2 of every upstream DINO block; the reference reaches it through
// model(...) at infer.py:177).  HBM-bound: one wave per row, the row stays in registers between the two
// statistics passes (mean, then centred variance -- biased, as nn.LayerNorm), float4 loads / 8-byte stores,
// wave reductions by __shfl_xor over 64 lanes.
#include "vittf_common.h"

namespace {

constexpr int LN_MAX_V4 = 4;  // float4 per lane -> D <= 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int DT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, unsigned short* __restrict__ y,
                                                        int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + row * d);
  float4 v[LN_MAX_V4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int idx = lane + 64 * i;
    if (idx < nv) {
      v[i] = xr[idx];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int idx = lane + 64 * i;
    if (idx < nv) {
      const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  uint2* yr = reinterpret_cast<uint2*>(y + row * d);
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int idx = lane + 64 * i;
    if (idx < nv) {
      const float4 gg = g4[idx], bb = b4[idx];
      uint2 pk;
      pk.x = pack2_h16<DT>((v[i].x - mean) * rstd * gg.x + bb.x, (v[i].y - mean) * rstd * gg.y + bb.y);
      pk.y = pack2_h16<DT>((v[i].z - mean) * rstd * gg.z + bb.z, (v[i].w - mean) * rstd * gg.w + bb.w);
      yr[idx] = pk;
    }
  }
}

}  // namespace

extern "C" int vittf_layernorm(const float* x, const float* g, const float* b, void* y, int64_t rows, int32_t d,
                               float eps, int32_t dtype, void* stream) {
  if (!x || !g || !b || !y || rows <= 0 || d <= 0 || (d & 3) || d > 256 * LN_MAX_V4) return VITTF_ERR_INVALID_ARG;
  const int64_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VITTF_BF16)
    hipLaunchKernelGGL((layernorm_kernel<VITTF_BF16>), dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, (unsigned short*)y, rows, d, eps);
  else if (dtype == VITTF_FP16)
    hipLaunchKernelGGL((layernorm_kernel<VITTF_FP16>), dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, (unsigned short*)y, rows, d, eps);
  else
    return VITTF_ERR_INVALID_ARG;
  return vittf_check_launch();
}
