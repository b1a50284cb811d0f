// Feature-volume epilogue: slice-axis average pooling + token-major -> feature-major transpose, and the
// fp16 z + y + x sum over the per-axis pooled volumes gathered from all ranks.
//
// Replaces k[:, 1:].view(S, f0, f1, D).permute(0, 3, 1, 2).permute(permute_out) (infer.py:201-203),
// torch.nn.AdaptiveAvgPool3d(feat_out_sz) (infer.py:329; in-plane it is the identity because the token grid
// already equals feat_out_sz, only the slice axis is reduced) and the running fp16 sum of infer.py:330-332.
// Rounding follows the reference's CPU path bit for bit: the CPU kernel of AdaptiveAvgPool3d accumulates a
// window in the tensor dtype (fp16, one rounding per add, slice order) and rounds sum / count once more.
// HBM-bound byte shuffling: 16-byte loads along the feature dim, an LDS transpose, 128-byte store runs.
#include "vittf_common.h"

namespace {

constexpr int PT = 64;  // tile: 64 items of the fast output dim x 64 features

__device__ __forceinline__ float f16_round(float v) { return f16bits_to_f32(f32_to_f16bits(v)); }

struct PoolArgs {
  const unsigned short* k;
  int k_slice0, k_nslices, total_slices, n_out, win0, nwin, f0, f1, d;
  unsigned short* dst;
  int64_t sd, sw, sr, sc;
  int fast_is_win;  // 1: the window index is the contiguous output dim, 0: the token column is
  int tiles_fast;   // tiles along the fast dim
};

__global__ __launch_bounds__(256) void pool_kernel(PoolArgs a) {
  __shared__ unsigned short tile[PT][PT + 2];
  const int tid = threadIdx.x;
  // blockIdx.x = ((outer1 * outer2) * tiles_fast + ft) * d_tiles + dt
  const int d_tiles = a.d / PT;
  int bid = blockIdx.x;
  const int dt = bid % d_tiles; bid /= d_tiles;
  const int ft = bid % a.tiles_fast; bid /= a.tiles_fast;
  int row, col0, win0, n_items;
  if (a.fast_is_win) {        // outer = (row, col), items = windows
    col0 = bid % a.f1; row = bid / a.f1;
    win0 = ft * PT;
    n_items = min(PT, a.nwin - win0);
  } else {                    // outer = (row, window), items = columns
    win0 = bid % a.nwin; row = bid / a.nwin;
    col0 = ft * PT;
    n_items = min(PT, a.f1 - col0);
  }
  const int d0 = dt * PT;

  // ---- phase 1: window mean, 8 features per thread ----
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int item = (tid >> 3) + 32 * pass;
    const int dl = (tid & 7) * 8;
    if (item < n_items) {
      const int w = a.win0 + (a.fast_is_win ? win0 + item : win0);
      const int col = a.fast_is_win ? col0 : col0 + item;
      const int lo = (int)(((int64_t)w * a.total_slices) / a.n_out);
      const int hi = (int)((((int64_t)(w + 1)) * a.total_slices + a.n_out - 1) / a.n_out);
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      for (int s = lo; s < hi; ++s) {
        const int64_t off = (((int64_t)(s - a.k_slice0) * a.f0 + row) * a.f1 + col) * a.d + d0 + dl;
        const uint4 raw = *reinterpret_cast<const uint4*>(a.k + off);
        const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // the CPU AdaptiveAvgPool3d keeps its running sum in the tensor's own type: round to fp16 per add
          acc[2 * j] = f16_round(acc[2 * j] + f16bits_to_f32((unsigned short)(u[j] & 0xffff)));
          acc[2 * j + 1] = f16_round(acc[2 * j + 1] + f16bits_to_f32((unsigned short)(u[j] >> 16)));
        }
      }
      const float cnt = (float)(hi - lo);
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[item][dl + j] = f32_to_f16bits(acc[j] / cnt);
    }
  }
  __syncthreads();

  // ---- phase 2: feature-major stores, consecutive lanes = consecutive items of the contiguous output dim ----
  const int item = tid & 63;
  if (item < n_items) {
    int64_t base = (int64_t)row * a.sr;
    if (a.fast_is_win) base += (int64_t)col0 * a.sc + (int64_t)(win0 + item) * a.sw;
    else               base += (int64_t)win0 * a.sw + (int64_t)(col0 + item) * a.sc;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int dl = (tid >> 6) + 4 * j;
      a.dst[(int64_t)(d0 + dl) * a.sd + base] = tile[item][dl];
    }
  }
}

__global__ __launch_bounds__(256) void assemble_sum_kernel(const unsigned short* __restrict__ gz,
                                                           const unsigned short* __restrict__ gy,
                                                           const unsigned short* __restrict__ gx, int cz, int cy, int cx,
                                                           int d, int n0, int n1, int n2, unsigned short* __restrict__ out) {
  const int64_t total = (int64_t)d * n0 * n1 * n2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int64_t t = e;
    const int i2 = (int)(t % n2); t /= n2;
    const int i1 = (int)(t % n1); t /= n1;
    const int i0 = (int)(t % n0);
    const int f = (int)(t / n0);
    // slab layouts written by pool_kernel: rank-major, then (D, dims with the axis' dim cut to `chunk`)
    const int rz = i2 / cz, rx = i0 / cx, ry = i1 / cy;
    const int64_t oz = ((((int64_t)rz * d + f) * n0 + i0) * n1 + i1) * cz + (i2 - rz * cz);
    const int64_t oy = ((((int64_t)ry * d + f) * n0 + i0) * cy + (i1 - ry * cy)) * n2 + i2;
    const int64_t ox = ((((int64_t)rx * d + f) * cx + (i0 - rx * cx)) * n1 + i1) * n2 + i2;
    const float z = f16bits_to_f32(gz[oz]), y = f16bits_to_f32(gy[oy]), x = f16bits_to_f32(gx[ox]);
    // 0.0 + z is exact; each fp16 + fp16 add is rounded once (the fp32 sum of two halves is exact)
    out[e] = f32_to_f16bits(f16_round(z + y) + x);
  }
}

// The same sum, 8 consecutive i2 per thread (n2 % 8 == 0 and the z chunk a multiple of 8: the three slabs and the output
// are then 16-byte runs in i2), index split once per 8 values in 32-bit arithmetic: 0.63 -> ~0.2 ms for the 201 MB volume
// (the scalar form spends nine 64-bit divisions per value).  Same arithmetic per value, same bits.
__global__ __launch_bounds__(256) void assemble_sum8_kernel(const unsigned short* __restrict__ gz,
                                                            const unsigned short* __restrict__ gy,
                                                            const unsigned short* __restrict__ gx, int cz, int cy, int cx,
                                                            int d, int n0, int n1, int n2, unsigned short* __restrict__ out) {
  const int c2 = n2 >> 3;                                   // 8-value chunks per row
  const int64_t chunks = (int64_t)d * n0 * n1 * c2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < chunks; e += (int64_t)gridDim.x * 256) {
    const int64_t row = e / c2;                             // (f, i0, i1) flattened: < 2^31 (checked on the host)
    const int i2 = 8 * (int)(e - row * c2);
    const int r32 = (int)row;
    const int f = r32 / (n0 * n1), r2 = r32 - f * (n0 * n1);
    const int i0 = r2 / n1, i1 = r2 - i0 * n1;
    const int rz = i2 / cz, rx = i0 / cx, ry = i1 / cy;
    const int64_t oz = ((((int64_t)rz * d + f) * n0 + i0) * n1 + i1) * cz + (i2 - rz * cz);
    const int64_t oy = ((((int64_t)ry * d + f) * n0 + i0) * cy + (i1 - ry * cy)) * n2 + i2;
    const int64_t ox = ((((int64_t)rx * d + f) * cx + (i0 - rx * cx)) * n1 + i1) * n2 + i2;
    const uint4 z4 = *reinterpret_cast<const uint4*>(gz + oz), y4 = *reinterpret_cast<const uint4*>(gy + oy),
                x4 = *reinterpret_cast<const uint4*>(gx + ox);
    const unsigned zw[4] = {z4.x, z4.y, z4.z, z4.w}, yw[4] = {y4.x, y4.y, y4.z, y4.w}, xw[4] = {x4.x, x4.y, x4.z, x4.w};
    unsigned ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned w = 0;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float z = f16bits_to_f32((unsigned short)(zw[k] >> (16 * hh))), y = f16bits_to_f32((unsigned short)(yw[k] >> (16 * hh))),
                    x = f16bits_to_f32((unsigned short)(xw[k] >> (16 * hh)));
        w |= (unsigned)f32_to_f16bits(f16_round(z + y) + x) << (16 * hh);     // one rounding per fp16 add, as above
      }
      ow[k] = w;
    }
    *reinterpret_cast<uint4*>(out + row * n2 + i2) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}


// ---- nearest resize of a uint8 volume (labels / class masks) and fp16 -> fp32 widening of an uploaded volume ----
// F.interpolate(..., mode='nearest') on a 5-D tensor: src = min(floor(dst * (float)in / out), in - 1) per dim
// (predict_ntf.py:217-218 label up-sample to the volume size, evaluate_similarities.py:63 class-mask resize).
// One thread = 16 consecutive outputs of the fast dim (one 16-byte store); the byte gathers hit L1 / L2.
__global__ __launch_bounds__(256) void resize_nearest_u8_kernel(const unsigned char* __restrict__ src, int n0, int n1, int n2,
                                                                unsigned char* __restrict__ dst, int o0, int o1, int o2,
                                                                int equals) {
  const float s0 = (float)n0 / (float)o0, s1 = (float)n1 / (float)o1, s2 = (float)n2 / (float)o2;
  const int cz = (o2 + 15) / 16;                       // 16-byte chunks per output row
  const int64_t chunks = (int64_t)o0 * o1 * cz;
  const bool aligned = (o2 % 16 == 0) && (((uintptr_t)dst & 15) == 0);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < chunks; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % cz);
    int64_t t = e / cz;
    const int y = (int)(t % o1);
    const int x = (int)(t / o1);
    int sx = (int)floorf((float)x * s0), sy = (int)floorf((float)y * s1);
    sx = sx < n0 - 1 ? sx : n0 - 1; sy = sy < n1 - 1 ? sy : n1 - 1;
    const unsigned char* row = src + ((int64_t)sx * n1 + sy) * n2;
    unsigned char* orow = dst + ((int64_t)x * o1 + y) * o2 + 16 * c;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    const int nz = min(16, o2 - 16 * c);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < nz) {
        int sz = (int)floorf((float)(16 * c + j) * s2);
        sz = sz < n2 - 1 ? sz : n2 - 1;
        unsigned v = row[sz];
        if (equals >= 0) v = (v == (unsigned)equals) ? 1u : 0u;
        w[j >> 2] |= v << (8 * (j & 3));
      }
    }
    if (aligned) {
      *reinterpret_cast<uint4*>(orow) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      for (int j = 0; j < nz; ++j) orow[j] = (unsigned char)(w[j >> 2] >> (8 * (j & 3)));
    }
  }
}

__global__ __launch_bounds__(256) void widen_f16_kernel(const unsigned short* __restrict__ src, int64_t n,
                                                        float* __restrict__ dst) {
  const int64_t n8 = n >> 3;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
    const uint4 u = reinterpret_cast<const uint4*>(src)[e];
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
    float4 a, b;
    a.x = f16bits_to_f32((unsigned short)(w[0] & 0xffff)); a.y = f16bits_to_f32((unsigned short)(w[0] >> 16));
    a.z = f16bits_to_f32((unsigned short)(w[1] & 0xffff)); a.w = f16bits_to_f32((unsigned short)(w[1] >> 16));
    b.x = f16bits_to_f32((unsigned short)(w[2] & 0xffff)); b.y = f16bits_to_f32((unsigned short)(w[2] >> 16));
    b.z = f16bits_to_f32((unsigned short)(w[3] & 0xffff)); b.w = f16bits_to_f32((unsigned short)(w[3] >> 16));
    reinterpret_cast<float4*>(dst)[2 * e] = a;
    reinterpret_cast<float4*>(dst)[2 * e + 1] = b;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = (n8 << 3) + threadIdx.x;
    dst[i] = f16bits_to_f32(src[i]);
  }
}

}  // namespace

extern "C" int vittf_pool_slices(const uint16_t* k_slices, int32_t k_slice0, int32_t k_nslices, int32_t total_slices,
                                 int32_t n_out, int32_t win0, int32_t nwin, int32_t f0, int32_t f1, int32_t d,
                                 uint16_t* dst, int64_t dst_stride_d, int64_t dst_stride_win, int64_t dst_stride_row,
                                 int64_t dst_stride_col, void* stream) {
  if (!k_slices || !dst || total_slices <= 0 || n_out <= 0 || nwin <= 0 || win0 < 0 || win0 + nwin > n_out)
    return VITTF_ERR_INVALID_ARG;
  if (f0 <= 0 || f1 <= 0 || d <= 0 || d % PT != 0 || k_nslices <= 0 || k_slice0 < 0) return VITTF_ERR_INVALID_ARG;
  if (((uintptr_t)k_slices & 15) != 0) return VITTF_ERR_INVALID_ARG;
  // every slice touched by the requested windows must be resident
  const int64_t lo = ((int64_t)win0 * total_slices) / n_out;
  const int64_t hi = (((int64_t)(win0 + nwin)) * total_slices + n_out - 1) / n_out;
  if (lo < k_slice0 || hi > (int64_t)k_slice0 + k_nslices || hi > total_slices) return VITTF_ERR_INVALID_ARG;
  PoolArgs a;
  a.k = k_slices; a.k_slice0 = k_slice0; a.k_nslices = k_nslices; a.total_slices = total_slices; a.n_out = n_out;
  a.win0 = win0; a.nwin = nwin; a.f0 = f0; a.f1 = f1; a.d = d; a.dst = dst;
  a.sd = dst_stride_d; a.sw = dst_stride_win; a.sr = dst_stride_row; a.sc = dst_stride_col;
  a.fast_is_win = (dst_stride_win == 1 && dst_stride_col != 1) ? 1 : 0;
  int64_t blocks;
  if (a.fast_is_win) { a.tiles_fast = (nwin + PT - 1) / PT; blocks = (int64_t)f0 * f1 * a.tiles_fast * (d / PT); }
  else               { a.tiles_fast = (f1 + PT - 1) / PT;   blocks = (int64_t)f0 * nwin * a.tiles_fast * (d / PT); }
  if (blocks > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(pool_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return vittf_check_launch();
}

extern "C" int vittf_assemble_sum(const uint16_t* gz, const uint16_t* gy, const uint16_t* gx, int32_t nranks,
                                  const int32_t chunk[3], int32_t d, int32_t n0, int32_t n1, int32_t n2, uint16_t* out,
                                  void* stream) {
  if (!gz || !gy || !gx || !out || !chunk || nranks <= 0 || d <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0)
    return VITTF_ERR_INVALID_ARG;
  // chunk[a] windows per rank along volume dim a; nranks * chunk must cover the dim
  if (chunk[0] <= 0 || chunk[1] <= 0 || chunk[2] <= 0) return VITTF_ERR_INVALID_ARG;
  if ((int64_t)nranks * chunk[0] < n0 || (int64_t)nranks * chunk[1] < n1 || (int64_t)nranks * chunk[2] < n2)
    return VITTF_ERR_INVALID_ARG;
  const int64_t total = (int64_t)d * n0 * n1 * n2;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  const bool vec = n2 % 8 == 0 && chunk[2] % 8 == 0 && (int64_t)d * n0 * n1 < 0x7fffffff &&
                   (((uintptr_t)gz | (uintptr_t)gy | (uintptr_t)gx | (uintptr_t)out) & 15) == 0;
  if (vec) {
    blocks = (total / 8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(assemble_sum8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gz, gy, gx,
                       chunk[2], chunk[1], chunk[0], d, n0, n1, n2, out);
  } else {
    hipLaunchKernelGGL(assemble_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gz, gy, gx,
                       chunk[2], chunk[1], chunk[0], d, n0, n1, n2, out);
  }
  return vittf_check_launch();
}

extern "C" int vittf_resize_nearest_u8(const uint8_t* src, int32_t n0, int32_t n1, int32_t n2, uint8_t* dst, int32_t o0,
                                       int32_t o1, int32_t o2, int32_t equals, void* stream) {
  if (!src || !dst || src == dst || n0 <= 0 || n1 <= 0 || n2 <= 0 || o0 <= 0 || o1 <= 0 || o2 <= 0 || equals > 255)
    return VITTF_ERR_INVALID_ARG;
  const int64_t chunks = (int64_t)o0 * o1 * ((o2 + 15) / 16);
  int64_t blocks = (chunks + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(resize_nearest_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, n0, n1, n2,
                     dst, o0, o1, o2, equals);
  return vittf_check_launch();
}

extern "C" int vittf_widen_f16(const uint16_t* src, int64_t n, float* dst, void* stream) {
  if (!src || !dst || n <= 0 || ((uintptr_t)src & 15) != 0 || ((uintptr_t)dst & 15) != 0) return VITTF_ERR_INVALID_ARG;
  int64_t blocks = ((n >> 3) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(widen_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, n, dst);
  return vittf_check_launch();
}
