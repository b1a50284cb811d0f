// Front end of the slice loop: global min/max of the volume, and the fused
//   slice gather -> min-max normalise -> nearest resize -> patch-embed conv -> + CLS / position embedding.
//
// Replaces, without materialising the (S, 3, h, w) fp32 tensor the reference builds:
//   norm_minmax (infer.py:32-34), the 3-channel expand + ImageNet normalize (infer.py:154-155),
//   F.interpolate(..., mode='nearest') (infer.py:177) and PatchEmbed + prepare_tokens of the upstream ViT.
// The three input channels are the same grey value, so the conv is folded on the host to ONE input channel
// (P*P taps) plus a bias; the arithmetic is fp32 VALU FMAs in the generic kernel and, for ViT-S/8, split-fp16 products on
// the matrix cores that stay within 1e-6 of it (patch_embed_mfma_kernel): the precision of the first layer is kept.
#include "vittf_common.h"

namespace {

// ---------------------------------------------------------------- min / max
constexpr int MM_BLOCKS = 1024;

__device__ __forceinline__ void wave_minmax(float& lo, float& hi) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off));
    hi = fmaxf(hi, __shfl_xor(hi, off));
  }
}

__device__ __forceinline__ void block_minmax(float& lo, float& hi) {
  __shared__ float s_lo[4], s_hi[4];
  wave_minmax(lo, hi);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; }
  __syncthreads();
  lo = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
  hi = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
}

__global__ __launch_bounds__(256) void minmax_partial(const float* __restrict__ v, int64_t n, float* __restrict__ part) {
  float lo = INFINITY, hi = -INFINITY;
  const int64_t n4 = n >> 2;
  const float4* v4 = reinterpret_cast<const float4*>(v);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = v4[i];
    lo = fminf(fminf(lo, x.x), fminf(fminf(x.y, x.z), x.w));
    hi = fmaxf(fmaxf(hi, x.x), fmaxf(fmaxf(x.y, x.z), x.w));
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) { lo = fminf(lo, v[i]); hi = fmaxf(hi, v[i]); }
  block_minmax(lo, hi);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = lo; part[2 * blockIdx.x + 1] = hi; }
}

__global__ __launch_bounds__(256) void minmax_final(const float* __restrict__ part, int nblocks, float* __restrict__ out) {
  float lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < nblocks; i += 256) { lo = fminf(lo, part[2 * i]); hi = fmaxf(hi, part[2 * i + 1]); }
  block_minmax(lo, hi);
  if (threadIdx.x == 0) { out[0] = lo; out[1] = hi; }
}

// ---------------------------------------------------------------- patch embed
// One workgroup = TP patches of one slice x all D output features.
//   phase 1: gather the TP x P*P normalised grey values into LDS (fp32)
//   phase 2: thread d (and d + 256, ...) accumulates its feature for every patch; the folded weights are
//            k-major [P*P][D] so the per-thread weight reads are coalesced, the pixel reads are LDS broadcasts
constexpr int TP = 32;

template <int P>
__global__ __launch_bounds__(384) void patch_embed_kernel(vittf_slice_view view, int slice0, const float* __restrict__ w_t,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ cls_pos0,
                                                          const float* __restrict__ patch_pos, float* __restrict__ tokens_out,
                                                          int d, int f0, int f1) {
  constexpr int PP = P * P;
  __shared__ __attribute__((aligned(16))) float px[TP][PP + 4];   // row stride 16-byte aligned: phase 2 reads float4 along k
  const int npatch = f0 * f1;
  const int tokens = npatch + 1;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * TP;
  const float lo = view.minmax[0], hi = view.minmax[1];
  const float range = hi - lo;
  // F.interpolate(mode='nearest'): src = min(floor(dst * (in / out)), in - 1), scale in fp32
  const float sr = (float)view.in_rows / (float)view.out_rows;
  const float sc = (float)view.in_cols / (float)view.out_cols;
  const float* slice = view.vol + (int64_t)(slice0 + b) * view.stride_slice;

  for (int e = threadIdx.x; e < TP * PP; e += blockDim.x) {
    const int pl = e / PP, k = e - pl * PP;
    const int p = p0 + pl;
    float v = 0.f;
    if (p < npatch) {
      const int py = p / f1, pxx = p - py * f1;
      const int iy = py * P + k / P, ix = pxx * P + k % P;
      int ry = (int)floorf((float)iy * sr);
      int cx = (int)floorf((float)ix * sc);
      ry = ry < view.in_rows - 1 ? ry : view.in_rows - 1;
      cx = cx < view.in_cols - 1 ? cx : view.in_cols - 1;
      const float raw = slice[(int64_t)ry * view.stride_row + (int64_t)cx * view.stride_col];
      v = (raw - lo) / range;
    }
    px[pl][k] = v;
  }
  __syncthreads();

  float* out_b = tokens_out + (int64_t)b * tokens * d;
  for (int dd = threadIdx.x; dd < d; dd += blockDim.x) {
    float acc[TP];
#pragma unroll
    for (int i = 0; i < TP; ++i) acc[i] = 0.f;
    // four k per step: one 16-byte LDS broadcast read per patch instead of four 4-byte ones (the loop was LDS-issue
    // bound); every accumulator still sums its products in ascending k, so the result is bit-identical
    for (int k = 0; k < PP; k += 4) {
      const float w0 = w_t[(int64_t)k * d + dd], w1 = w_t[(int64_t)(k + 1) * d + dd];
      const float w2 = w_t[(int64_t)(k + 2) * d + dd], w3 = w_t[(int64_t)(k + 3) * d + dd];
#pragma unroll
      for (int i = 0; i < TP; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(&px[i][k]);
        acc[i] = fmaf(v.x, w0, acc[i]);
        acc[i] = fmaf(v.y, w1, acc[i]);
        acc[i] = fmaf(v.z, w2, acc[i]);
        acc[i] = fmaf(v.w, w3, acc[i]);
      }
    }
    const float bv = bias[dd];
#pragma unroll
    for (int i = 0; i < TP; ++i) {
      const int p = p0 + i;
      if (p < npatch) out_b[(int64_t)(1 + p) * d + dd] = acc[i] + bv + patch_pos[(int64_t)p * d + dd];
    }
    if (blockIdx.x == 0) out_b[dd] = cls_pos0[dd];
  }
}

// ---------------------------------------------------------------- patch embed on the matrix cores (D = 384, P = 8)
// The same sum, x[token][d] = sum_k px[token][k] w[k][d] + bias[d] + pos[token][d], with both operands split into an fp16 head
// and an fp16 tail (v = hi + lo to 2^-22 |v|; the products of fp16 values are exact in the fp32 accumulator) and three
// MFMAs per 16-wide k step: hi.hi + lo.hi + hi.lo -- 1.5 TFLOP per 512-slice launch instead of 0.1 TFLOP of fp32 FMAs at a
// fifth of the VALU's rate (1.67 ms per 256 slices, 1.5 % of the step).  Within 1e-6 of the fp32 chain relative to a row's
// largest value; per-row arithmetic, so the bits do not depend on the batching.
//   * persistent workgroups of 8 waves; the weights (12 output tiles x 4 k steps x {hi, lo} fragments of 1 KB = 96 KB)
//     are split once per workgroup into LDS in fragment order (lane-linear 16-byte reads);
//   * a wave owns 32 consecutive token rows: lane (row, half h) gathers the pixels of patch rows h, 2 + h, 4 + h, 6 + h --
//     exactly the B operand's k = 16 s + 8 h + e -- with sample_kernel's nearest-resize arithmetic, so no pixel goes through LDS;
//   * two passes of 6 output tiles (96 accumulator registers); epilogue from the accumulator layout: lane owns row j, columns
//     32 ot + 8 g + 4 h + {0 .. 3}: + bias + position embedding as 16-byte loads, 16-byte stores (48 of each per 32 rows: far
//     below this kernel's budget); the CLS row of a slice takes cls_token + pos[0].
constexpr int PEM_D = 384, PEM_K = 64, PEM_FRAGS = (PEM_D / 32) * (PEM_K / 16) * 2;
__global__ __launch_bounds__(512, 1) void patch_embed_mfma_kernel(vittf_slice_view view, int slice0, int batch,
                                                                  const float* __restrict__ w_t, const float* __restrict__ bias,
                                                                  const float* __restrict__ cls_pos0,
                                                                  const float* __restrict__ patch_pos,
                                                                  float* __restrict__ tokens_out, int f0, int f1) {
  __shared__ __attribute__((aligned(16))) unsigned short wlds[PEM_FRAGS * 512];      // 96 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  // fragment (ot, s, hl): lane (i = l31, half hh) holds w[16 s + 8 hh + e][32 ot + i], e = 0 .. 7
  for (int it = tid; it < (PEM_FRAGS / 2) * 64; it += 512) {
    const int fr = it >> 6, ln = it & 63;
    const int ot = fr >> 2, s = fr & 3;
    const int i = ln & 31, hh = ln >> 5;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = w_t[(int64_t)(16 * s + 8 * hh + e) * PEM_D + 32 * ot + i];
      const unsigned short hi = f32_to_f16bits(v);
      const unsigned short lo = f32_to_f16bits(v - f16bits_to_f32(hi));
      wlds[((fr * 2 + 0) * 64 + ln) * 8 + e] = hi;
      wlds[((fr * 2 + 1) * 64 + ln) * 8 + e] = lo;
    }
  }
  __syncthreads();
  const int npatch = f0 * f1, tokens = npatch + 1;
  const int64_t rows = (int64_t)batch * tokens;
  const float lo_v = view.minmax[0], range = view.minmax[1] - view.minmax[0];
  const float sr = (float)view.in_rows / (float)view.out_rows;
  const float sc = (float)view.in_cols / (float)view.out_cols;
  const int64_t ntile = (rows + 255) / 256;
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t r = tile * 256 + wave * 32 + l31;
    const bool valid = r < rows;
    const int b = valid ? (int)(r / tokens) : 0;
    const int t = valid ? (int)(r - (int64_t)b * tokens) : 0;
    const bool patch = valid && t > 0;
    const int p = patch ? t - 1 : 0;
    const int py = p / f1, pxx = p - py * f1;
    const float* slice = view.vol + (int64_t)(slice0 + b) * view.stride_slice;
    // the B operands: k step s = patch row 2 s + h, 8 columns
    s16x8_t bh[4], bl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int iy = py * 8 + 2 * s + h;
      int ry = (int)floorf((float)iy * sr);
      ry = ry < view.in_rows - 1 ? ry : view.in_rows - 1;
      unsigned short hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ix = pxx * 8 + e;
        int cx = (int)floorf((float)ix * sc);
        cx = cx < view.in_cols - 1 ? cx : view.in_cols - 1;
        float v = 0.f;
        if (patch) v = (slice[(int64_t)ry * view.stride_row + (int64_t)cx * view.stride_col] - lo_v) / range;
        hi[e] = f32_to_f16bits(v);
        lo[e] = f32_to_f16bits(v - f16bits_to_f32(hi[e]));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { bh[s][e] = (short)hi[e]; bl[s][e] = (short)lo[e]; }
    }
    float* orow = tokens_out + r * PEM_D;
    const float* prow = patch_pos + (int64_t)p * PEM_D;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      f32x16_t acc[6];
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int fr = (6 * pass + j) * 4 + s;
          const s16x8_t ah = *reinterpret_cast<const s16x8_t*>(&wlds[((fr * 2 + 0) * 64 + lane) * 8]);
          const s16x8_t al = *reinterpret_cast<const s16x8_t*>(&wlds[((fr * 2 + 1) * 64 + lane) * 8]);
          acc[j] = mfma32<VITTF_FP16>(ah, bh[s], acc[j]);
          acc[j] = mfma32<VITTF_FP16>(al, bh[s], acc[j]);
          acc[j] = mfma32<VITTF_FP16>(ah, bl[s], acc[j]);
        }
      if (valid) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c = 32 * (6 * pass + j) + 8 * g + 4 * h;
            float4 o;
            if (patch) {
              const float4 bv = *reinterpret_cast<const float4*>(bias + c);
              const float4 pv = *reinterpret_cast<const float4*>(prow + c);
              o.x = acc[j][4 * g + 0] + bv.x + pv.x; o.y = acc[j][4 * g + 1] + bv.y + pv.y;
              o.z = acc[j][4 * g + 2] + bv.z + pv.z; o.w = acc[j][4 * g + 3] + bv.w + pv.w;
            } else {
              o = *reinterpret_cast<const float4*>(cls_pos0 + c);
            }
            *reinterpret_cast<float4*>(orow + c) = o;
          }
      }
    }
  }
}

}  // namespace

extern "C" size_t vittf_minmax_workspace_bytes(void) { return 2 * MM_BLOCKS * sizeof(float); }

extern "C" int vittf_volume_minmax(const float* vol, int64_t n, float* out_minmax, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (!vol || !out_minmax || !ws || n <= 0) return VITTF_ERR_INVALID_ARG;
  if (ws_bytes < vittf_minmax_workspace_bytes()) return VITTF_ERR_WORKSPACE;
  if (((uintptr_t)vol & 15) != 0) return VITTF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  int64_t want = (n / 4 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > MM_BLOCKS ? MM_BLOCKS : want));
  hipLaunchKernelGGL(minmax_partial, dim3(blocks), dim3(256), 0, st, vol, n, (float*)ws);
  hipLaunchKernelGGL(minmax_final, dim3(1), dim3(256), 0, st, (const float*)ws, blocks, out_minmax);
  return vittf_check_launch();
}

extern "C" int vittf_patch_embed(const vittf_vit_config* cfg, const vittf_vit_weights* w, const vittf_pos_embed* pos,
                                 const vittf_slice_view* view, int32_t slice0, int32_t batch, float* tokens_out,
                                 void* stream) {
  if (!cfg || !w || !pos || !view || !tokens_out || batch <= 0 || slice0 < 0) return VITTF_ERR_INVALID_ARG;
  if (!w->pe_w_t || !w->pe_b || !pos->cls_plus_pos0 || !pos->patch_pos || !view->vol || !view->minmax)
    return VITTF_ERR_INVALID_ARG;
  const int p = cfg->patch;
  if ((p != 8 && p != 16) || view->out_rows % p || view->out_cols % p || view->out_rows <= 0 || view->out_cols <= 0)
    return VITTF_ERR_INVALID_ARG;
  if (view->in_rows <= 0 || view->in_cols <= 0 || batch > 65535) return VITTF_ERR_INVALID_ARG;
  const int f0 = view->out_rows / p, f1 = view->out_cols / p;
  const int nblk = (f0 * f1 + TP - 1) / TP;
  hipStream_t st = (hipStream_t)stream;
  if (cfg->embed_dim == PEM_D && p == 8 && (((uintptr_t)w->pe_b | (uintptr_t)pos->cls_plus_pos0 | (uintptr_t)pos->patch_pos | (uintptr_t)tokens_out) & 15) == 0) {
    // ViT-S/8: the conv on the matrix cores with fp16 head + tail operands (see patch_embed_mfma_kernel)
    const int cus = vittf_current_cus();
    if (cus <= 0) return VITTF_ERR_NO_DEVICE;
    const int64_t ntile = ((int64_t)batch * (f0 * f1 + 1) + 255) / 256;
    hipLaunchKernelGGL(patch_embed_mfma_kernel, dim3((unsigned)(ntile < cus ? ntile : cus)), dim3(512), 0, st, *view, slice0, batch,
                       w->pe_w_t, w->pe_b, pos->cls_plus_pos0, pos->patch_pos, tokens_out, f0, f1);
    return vittf_check_launch();
  }
  const int threads = cfg->embed_dim % 384 == 0 ? 384 : 256;   // one feature per thread in a single pass for D = 384 / 768
  if (p == 8) {
    hipLaunchKernelGGL((patch_embed_kernel<8>), dim3(nblk, batch), dim3(threads), 0, st, *view, slice0, w->pe_w_t, w->pe_b,
                       pos->cls_plus_pos0, pos->patch_pos, tokens_out, cfg->embed_dim, f0, f1);
  } else {
    hipLaunchKernelGGL((patch_embed_kernel<16>), dim3(nblk, batch), dim3(threads), 0, st, *view, slice0, w->pe_w_t, w->pe_b,
                       pos->cls_plus_pos0, pos->patch_pos, tokens_out, cfg->embed_dim, f0, f1);
  }
  return vittf_check_launch();
}
