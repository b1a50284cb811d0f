// Similarity query over the feature volume: query sampling, fused dot-product / threshold / power / class mean,
// global-max quantisation to uint8 with nearest resize, and label assignment.
//
// Replaces sample_features3d (infer.py:48-72), the einsum + where/pow/mean + quantise + F.interpolate chain of
// compute_similarities (predict_ntf.py:56-72, 95-100) and the running-max labels (predict_ntf.py:203-215).
//
// The feature volume is F-major fp16: voxel index contiguous, stride Nvox between features.  For the interactive
// regime (A <= a few dozen) the dot products are HBM-bound: every voxel's 2*F bytes are read ONCE per chunk of 16
// annotations, coalesced 4 B per lane, and the reference's A*4 B/voxel intermediate never exists.  Query vectors
// are wave-uniform (scalar loads of a transposed [F][16] copy); all arithmetic is fp32 like the reference CPU path.
#include "vittf_common.h"

#include <stdlib.h>

namespace {

// ---------------------------------------------------------------- query sampling (grid_sample 3-D)
template <bool HALF>
__device__ __forceinline__ float feat_at(const void* feat, int64_t idx) {
  if constexpr (HALF) return f16bits_to_f32(reinterpret_cast<const unsigned short*>(feat)[idx]);
  else return reinterpret_cast<const float*>(feat)[idx];
}

template <bool HALF>
__global__ __launch_bounds__(256) void sample_kernel(const void* __restrict__ feat, int f, int n0, int n1, int n2,
                                                     const float* __restrict__ rel, int na, int mode,
                                                     const float* __restrict__ vnorm, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)na * f) return;
  const int a = (int)(e / f), ff = (int)(e - (int64_t)a * f);
  // rel is in volume dim order; grid_sample's x <-> last volume dim (the flip of infer.py:67)
  const float r0 = rel[3 * a + 0], r1 = rel[3 * a + 1], r2 = rel[3 * a + 2];
  // unnormalise, align_corners=False: ((g + 1) * size - 1) / 2
  const float iz = ((r0 + 1.f) * (float)n0 - 1.f) / 2.f;   // along n0 ("depth" of grid_sample)
  const float iy = ((r1 + 1.f) * (float)n1 - 1.f) / 2.f;
  const float ix = ((r2 + 1.f) * (float)n2 - 1.f) / 2.f;   // along n2 (contiguous)
  const int64_t plane = (int64_t)n1 * n2;
  const int64_t fbase = (int64_t)ff * n0 * plane;
  float res = 0.f;
  if (mode == VITTF_SAMPLE_NEAREST) {
    const int x = (int)rintf(ix), y = (int)rintf(iy), z = (int)rintf(iz);
    if (x >= 0 && x < n2 && y >= 0 && y < n1 && z >= 0 && z < n0) {
      const int64_t vox = z * plane + (int64_t)y * n2 + x;
      res = feat_at<HALF>(feat, fbase + vox);
      if (vnorm) res = res / vnorm[vox];           // F.normalize(feat, dim=0): v / max(|v|, eps)
    }
  } else {
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx1 = ix - fx, wy1 = iy - fy, wz1 = iz - fz;
    const float wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy, wz0 = (fz + 1.f) - iz;
    // corner order and weight products of the CPU grid_sampler_3d: tnw, tne, tsw, tse, bnw, bne, bsw, bse
#pragma unroll
    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          const int x = x0 + cx, y = y0 + cy, z = z0 + cz;
          const float wgt = (cx ? wx1 : wx0) * (cy ? wy1 : wy0) * (cz ? wz1 : wz0);
          if (x >= 0 && x < n2 && y >= 0 && y < n1 && z >= 0 && z < n0) {
            const int64_t vox = z * plane + (int64_t)y * n2 + x;
            float val = feat_at<HALF>(feat, fbase + vox);
            if (vnorm) val = val / vnorm[vox];
            res = __fadd_rn(res, __fmul_rn(val, wgt));  // unfused, like the CPU op
          }
        }
  }
  out[e] = res;
}

// The interactive query's form (vittf_similarity_query): the relative coordinates arrive as a kernel argument (at most
// VITTF_QUERY_MAX_A annotations: no host -> device copy in front of the query), the arithmetic is sample_kernel's trilinear
// branch word for word, and the first threads zero the per-class maxima the accumulation kernels behind it add to (no memset).
struct RelArg { float v[3 * VITTF_QUERY_MAX_A]; };
__global__ __launch_bounds__(256) void sample_query_kernel(const unsigned short* __restrict__ feat, int f, int n0, int n1, int n2,
                                                           RelArg rel, int na, const float* __restrict__ vnorm,
                                                           float* __restrict__ out, unsigned* __restrict__ maxbits, int classes) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < classes) maxbits[e] = 0u;
  if (e >= (int64_t)na * f) return;
  const int a = (int)(e / f), ff = (int)(e - (int64_t)a * f);
  const float r0 = rel.v[3 * a + 0], r1 = rel.v[3 * a + 1], r2 = rel.v[3 * a + 2];
  const float iz = ((r0 + 1.f) * (float)n0 - 1.f) / 2.f;
  const float iy = ((r1 + 1.f) * (float)n1 - 1.f) / 2.f;
  const float ix = ((r2 + 1.f) * (float)n2 - 1.f) / 2.f;
  const int64_t plane = (int64_t)n1 * n2;
  const int64_t fbase = (int64_t)ff * n0 * plane;
  float res = 0.f;
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const float wx1 = ix - fx, wy1 = iy - fy, wz1 = iz - fz;
  const float wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy, wz0 = (fz + 1.f) - iz;
#pragma unroll
  for (int cz = 0; cz < 2; ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) {
        const int x = x0 + cx, y = y0 + cy, z = z0 + cz;
        const float wgt = (cx ? wx1 : wx0) * (cy ? wy1 : wy0) * (cz ? wz1 : wz0);
        if (x >= 0 && x < n2 && y >= 0 && y < n1 && z >= 0 && z < n0) {
          const int64_t vox = z * plane + (int64_t)y * n2 + x;
          float val = feat_at<true>(feat, fbase + vox);
          if (vnorm) val = val / vnorm[vox];
          res = __fadd_rn(res, __fmul_rn(val, wgt));
        }
      }
  out[e] = res;
}

// ---------------------------------------------------------------- per-voxel L2 norm (cosine similarity)
// max(|feat[:, v]|_2, 1e-12): the denominator of F.normalize(feat, dim=0) (compare_feat_sampling.py:45,
// tests/test_vishum.py:12).  Keeping the norms (1 MB) instead of a normalised fp32 copy of the volume (403 MB)
// leaves the similarity kernel on the fp16 volume: (sum_f x_f q_f) / |x| instead of sum_f (x_f / |x|) q_f.
__global__ __launch_bounds__(256) void voxel_norm_kernel(const unsigned short* __restrict__ feat, int f, int64_t nvox,
                                                         float* __restrict__ out) {
  const int64_t v0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (v0 >= nvox) return;
  const bool vec = (v0 + 2 <= nvox) && ((nvox & 1) == 0);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
  for (int ff = 0; ff < f; ++ff) {
    const unsigned short* p = feat + (int64_t)ff * nvox + v0;
    float x0, x1 = 0.f;
    if (vec) {
      const unsigned raw = *reinterpret_cast<const unsigned*>(p);
      x0 = f16bits_to_f32((unsigned short)(raw & 0xffff)); x1 = f16bits_to_f32((unsigned short)(raw >> 16));
    } else {
      x0 = f16bits_to_f32(p[0]);
      if (v0 + 1 < nvox) x1 = f16bits_to_f32(p[1]);
    }
    s0 = fmaf(x0, x0, s0); s1 = fmaf(x1, x1, s1);
  }
  out[v0] = fmaxf(sqrtf(s0), 1e-12f);
  if (v0 + 1 < nvox) out[v0 + 1] = fmaxf(sqrtf(s1), 1e-12f);
}

// The same norms with the feature axis split over the workgroup's four waves and 4 voxels per thread (f % 4 == 0,
// nvox % 4 == 0, 16-byte aligned volume): like the similarity pass, a 64^3 volume is too few voxels to keep enough loads
// in flight with one thread per voxel pair.  Partial sums of squares meet in wave 0 as ((p0 + p1) + p2) + p3.
__global__ __launch_bounds__(256) void voxel_norm_split_kernel(const unsigned short* __restrict__ feat, int f, int64_t nvox,
                                                               float* __restrict__ out) {
  __shared__ float part[3][4][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t v0 = ((int64_t)blockIdx.x * 64 + lane) * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const int fq = f >> 2, f0 = wave * fq;
  if (v0 < nvox) {
#pragma unroll 8
    for (int ff = f0; ff < f0 + fq; ++ff) {
      const uint2 raw = *reinterpret_cast<const uint2*>(feat + (int64_t)ff * nvox + v0);
      const float x0 = f16bits_to_f32((unsigned short)(raw.x & 0xffff)), x1 = f16bits_to_f32((unsigned short)(raw.x >> 16));
      const float x2 = f16bits_to_f32((unsigned short)(raw.y & 0xffff)), x3 = f16bits_to_f32((unsigned short)(raw.y >> 16));
      s[0] = fmaf(x0, x0, s[0]); s[1] = fmaf(x1, x1, s[1]); s[2] = fmaf(x2, x2, s[2]); s[3] = fmaf(x3, x3, s[3]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) part[wave - 1][j][lane] = s[j];
  }
  __syncthreads();
  if (wave != 0 || v0 >= nvox) return;
  float4 o;
  float* op = &o.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) op[j] = fmaxf(sqrtf(((s[j] + part[0][j][lane]) + part[1][j][lane]) + part[2][j][lane]), 1e-12f);
  *reinterpret_cast<float4*>(out + v0) = o;
}

// ---------------------------------------------------------------- similarity
constexpr int ACH = 16;   // annotations per pass over the volume
constexpr int VPT = 2;    // voxels per thread (4-byte loads): 64^3 voxels -> 2048 waves, 8 per CU (4 per thread left the
                          // kernel latency-bound at 0.85 TB/s with one wave per SIMD)
constexpr int MAXC = 8;   // classes handled by one pass (more classes: several launches)

struct SimChunk {
  int n_ann;          // annotations in this chunk (<= ACH)
  int cls[ACH];       // class of each annotation, relative to c0
  int c0, nc;         // classes [c0, c0 + nc) are touched by this chunk
  int first[MAXC];    // chunk holds the first annotations of that class -> overwrite instead of accumulate
  int last[MAXC];     // chunk holds the last annotations of that class  -> finalise (mean, max)
  float count[MAXC];    // annotations of that class (fp32, exact)
};

// qf_t: [F][ACH] fp32 transposed query block for this chunk (zero padded)
__global__ __launch_bounds__(256) void transpose_queries(const float* __restrict__ qf, int f, int a0, int n_ann,
                                                         float* __restrict__ qf_t) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= f * ACH) return;
  const int ff = e / ACH, a = e - ff * ACH;
  qf_t[e] = a < n_ann ? qf[(int64_t)(a0 + a) * f + ff] : 0.f;
}

__device__ __forceinline__ float thresh_pow(float s) {
  // where(s >= 0.25, s, 0) ** 2.5   (predict_ntf.py:71)
  return s >= 0.25f ? s * s * sqrt_cr_normal(s) : 0.f;
}

// ACT 0: where(s >= 0.25, s, 0) ** 2.5 (predict_ntf.py:71);  ACT 1: clamp(s, 0, 1) ** expo (infer.py:104, resample_topk)
template <int ACT>
__device__ __forceinline__ float activate(float s, float expo) {
  if constexpr (ACT == 0) return thresh_pow(s);
  const float c = fminf(fmaxf(s, 0.f), 1.f);
  return c > 0.f ? __builtin_amdgcn_exp2f(expo * __builtin_amdgcn_logf(c)) : 0.f;   // v_log_f32 is log2
}

// Everything after the dot products, for the NV voxels v0.. of one thread: cosine normalisation, per-class reduction over
// the chunk's annotations, accumulate / finalise the class maps, per-class maximum.
template <bool BIG, int ACT, int NV>
__device__ __forceinline__ void sim_finish(float (&acc)[ACH][NV], int64_t v0, int64_t nvox, const SimChunk& ch,
                                           const float* __restrict__ vnorm, float expo, float* __restrict__ sim,
                                           unsigned* __restrict__ maxbits) {
  if (vnorm) {     // cosine similarity: the volume is normalised per voxel (the queries were sampled from it normalised)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float nv = (v0 + j < nvox) ? vnorm[v0 + j] : 1.f;
#pragma unroll
      for (int a = 0; a < ACH; ++a) acc[a][j] = acc[a][j] / nv;
    }
  }
  // per-class reduction over the chunk's annotations: they are sorted by class, so a running sum in annotation order is
  // handed over whenever the next annotation belongs to another class (wave-uniform branches: the activation is
  // evaluated once per value and at most nc hand-overs run)
  // the activation first, in place (the same for every class)
  if constexpr (!BIG) {
#pragma unroll
    for (int a = 0; a < ACH; ++a)
#pragma unroll
      for (int j = 0; j < NV; ++j) acc[a][j] = activate<ACT>(acc[a][j], expo);
  }
  // per-class reduction over the chunk's annotations, one class per trip of a RUN-TIME loop: the annotations are sorted by
  // class, so class c's values are summed in annotation order by skipping the others (wave-uniform scalar branches; acc
  // keeps static indices).  One copy of the hand-over code instead of sixteen: the unrolled form kept ~170 VGPRs and 22
  // SGPRs in scratch around its branches (23 MB of scratch writes per pass: profiles/r02b_similarity_pmc.txt).
  for (int c = 0; c < ch.nc; ++c) {
    float run[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) run[j] = 0.f;
#pragma unroll
    for (int a = 0; a < ACH; ++a) {
      if (ch.cls[a] == c) {            // (padding annotations carry class -1)
#pragma unroll
        for (int j = 0; j < NV; ++j) run[j] += acc[a][j];
      }
    }
    float* dst = sim + (int64_t)(ch.c0 + c) * nvox + v0;
    const bool first = ch.first[c] != 0, last = ch.last[c] != 0;
    const float count = ch.count[c];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (v0 + j < nvox) {
        float t = run[j];
        if (!first) t += dst[j];
        if (last) {
          t = t / count;                      // mean = sum / count (predict_ntf.py:72 / :63), true division
          if (BIG) t = activate<ACT>(t, expo);
          m = fmaxf(m, t);
        }
        dst[j] = t;
      }
    }
    if (last) {   // sims are >= 0, so the uint bit pattern orders like the float: one atomicMax per wave and class
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      if ((threadIdx.x & 63) == 0) atomic_max_nonneg(maxbits + ch.c0 + c, m);
    }
  }
}

template <bool BIG, int ACT, bool HALF>
__global__ __launch_bounds__(256) void sim_accumulate(const void* __restrict__ feat_v, int f, int64_t nvox,
                                                      const float* __restrict__ qf_t, SimChunk ch,
                                                      const float* __restrict__ vnorm, float expo, float* __restrict__ sim,
                                                      unsigned* __restrict__ maxbits) {
  const int64_t v0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VPT;
  float acc[ACH][VPT];
#pragma unroll
  for (int a = 0; a < ACH; ++a)
#pragma unroll
    for (int j = 0; j < VPT; ++j) acc[a][j] = 0.f;
  const bool full = v0 + VPT <= nvox;
  if (v0 < nvox) {
    const bool vec = full && ((nvox & 1) == 0);
#pragma unroll 8
    for (int ff = 0; ff < f; ++ff) {
      float x[VPT];
      if constexpr (HALF) {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(feat_v) + (int64_t)ff * nvox + v0;
        if (vec) {
          const unsigned raw = *reinterpret_cast<const unsigned*>(p);
          x[0] = f16bits_to_f32((unsigned short)(raw & 0xffff)); x[1] = f16bits_to_f32((unsigned short)(raw >> 16));
        } else {
#pragma unroll
          for (int j = 0; j < VPT; ++j) x[j] = (v0 + j < nvox) ? f16bits_to_f32(p[j]) : 0.f;
        }
      } else {
        const float* p = reinterpret_cast<const float*>(feat_v) + (int64_t)ff * nvox + v0;
#pragma unroll
        for (int j = 0; j < VPT; ++j) x[j] = (v0 + j < nvox) ? p[j] : 0.f;
      }
      const float* q = qf_t + ff * ACH;   // wave-uniform -> scalar loads
#pragma unroll
      for (int a = 0; a < ACH; ++a) {
        const float qa = q[a];
#pragma unroll
        for (int j = 0; j < VPT; ++j) acc[a][j] = fmaf(x[j], qa, acc[a][j]);
      }
    }
  }
  sim_finish<BIG, ACT, VPT>(acc, v0, nvox, ch, vnorm, expo, sim, maxbits);
}

// The same pass with the feature axis split over the workgroup's four waves (f % 4 == 0, nvox % 4 == 0): a 64^3 volume
// has only 2 waves per SIMD worth of voxels at 2 voxels per thread, too few loads in flight for the 201 MB stream
// (181 us = 1.1 TB/s).  Here a thread takes 4 voxels (8- / 16-byte loads) of one quarter of the features, so four times as many
// independent load streams are in flight; the partial dot products meet in wave 0 through LDS as (p0 + p2) + (p1 + p3),
// and wave 0 finishes as above: 56-62 us = 3.5 TB/s.
constexpr int SVPT = 4;
template <bool BIG, int ACT, bool HALF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void sim_accumulate_split(
    const void* __restrict__ feat_v, int f, int64_t nvox, const float* __restrict__ qf_t, SimChunk ch,
    const float* __restrict__ vnorm, float expo, float* __restrict__ sim, unsigned* __restrict__ maxbits) {
  __shared__ float part[2][ACH * SVPT][64];                 // 32 KB: [slot][annotation * 4 + voxel][lane]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t v0 = ((int64_t)blockIdx.x * 64 + lane) * SVPT;
  float acc[ACH][SVPT];
#pragma unroll
  for (int a = 0; a < ACH; ++a)
#pragma unroll
    for (int j = 0; j < SVPT; ++j) acc[a][j] = 0.f;
  const int fq = f >> 2, f0 = wave * fq;
  if (v0 < nvox) {                                            // (nvox % 4 == 0: all four voxels exist)
#pragma unroll 8
    for (int ff = f0; ff < f0 + fq; ++ff) {
      float x[SVPT];
      if constexpr (HALF) {
        const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(feat_v) + (int64_t)ff * nvox + v0);
        x[0] = f16bits_to_f32((unsigned short)(raw.x & 0xffff)); x[1] = f16bits_to_f32((unsigned short)(raw.x >> 16));
        x[2] = f16bits_to_f32((unsigned short)(raw.y & 0xffff)); x[3] = f16bits_to_f32((unsigned short)(raw.y >> 16));
      } else {
        const float4 raw = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(feat_v) + (int64_t)ff * nvox + v0);
        x[0] = raw.x; x[1] = raw.y; x[2] = raw.z; x[3] = raw.w;
      }
      const float* q = qf_t + ff * ACH;   // wave-uniform -> scalar loads
#pragma unroll
      for (int a = 0; a < ACH; ++a) {
        const float qa = q[a];
#pragma unroll
        for (int j = 0; j < SVPT; ++j) acc[a][j] = fmaf(x[j], qa, acc[a][j]);
      }
    }
  }
  // (p0 + p2) + (p1 + p3) in two steps through a 32 KB buffer, 128 VGPRs: four workgroups per CU, so the 1024
  // workgroups of a 64^3 volume are resident at once (all three partials at once = 48 KB and 144 VGPRs = three per CU
  // and a second, mostly empty round: 70-79 us against 56-62 us on the same box)
#define SIM_PART_PUT(SLOT)                                                                   \
  _Pragma("unroll") for (int a = 0; a < ACH; ++a)                                            \
      _Pragma("unroll") for (int j = 0; j < SVPT; ++j) part[SLOT][a * SVPT + j][lane] = acc[a][j];
  // (adds in groups of four annotations behind scheduling fences: with all 64 LDS reads hoisted in front of the adds, 64
  // temporaries + 64 accumulators filled the 128-VGPR budget and ~170 registers went to scratch in wave 0's finish)
#define SIM_PART_ADD(SLOT)                                                                   \
  _Pragma("unroll") for (int a = 0; a < ACH; ++a) {                                          \
    _Pragma("unroll") for (int j = 0; j < SVPT; ++j) acc[a][j] += part[SLOT][a * SVPT + j][lane]; \
    if ((a & 3) == 3) __builtin_amdgcn_sched_barrier(0);                                     \
  }
  // (every wave reaches all three barriers)
  if (wave >= 2) SIM_PART_PUT(wave - 2)
  __syncthreads();
  if (wave < 2) SIM_PART_ADD(wave)
  __syncthreads();
  if (wave == 1) SIM_PART_PUT(0)
  __syncthreads();
  if (wave != 0) return;
  SIM_PART_ADD(0)
#undef SIM_PART_PUT
#undef SIM_PART_ADD
  sim_finish<BIG, ACT, SVPT>(acc, v0, nvox, ch, vnorm, expo, sim, maxbits);
}

__global__ __launch_bounds__(256) void sim_quantize(const float* __restrict__ sim, const unsigned* __restrict__ maxbits,
                                                    int classes, int n0, int n1, int n2, int o0, int o1, int o2,
                                                    unsigned char* __restrict__ out) {
  const int64_t per = (int64_t)o0 * o1 * o2;
  const int64_t total = per * classes;
  const float s0 = (float)n0 / (float)o0, s1 = (float)n1 / (float)o1, s2 = (float)n2 / (float)o2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e / per);
    int64_t t = e - (int64_t)c * per;
    const int z = (int)(t % o2); t /= o2;
    const int y = (int)(t % o1);
    const int x = (int)(t / o1);
    int sx = (int)floorf((float)x * s0), sy = (int)floorf((float)y * s1), sz = (int)floorf((float)z * s2);
    sx = sx < n0 - 1 ? sx : n0 - 1; sy = sy < n1 - 1 ? sy : n1 - 1; sz = sz < n2 - 1 ? sz : n2 - 1;
    const float v = sim[(int64_t)c * n0 * n1 * n2 + ((int64_t)sx * n1 + sy) * n2 + sz];
    const float quant = 0.99f * __uint_as_float(maxbits[c]);     // 0.99 * sim.max()      predict_ntf.py:98
    const float q = (255.0f / quant) * v;                        // 255 / quant * sim     predict_ntf.py:99
    // .to(torch.uint8) on x86: truncate toward zero, keep the low byte (257 -> 1); NaN (max == 0) -> 0
    const int qi = (q == q) ? (int)q : 0;
    out[e] = (unsigned char)(qi & 255);
  }
}

// The same arithmetic, 16 consecutive outputs of one row per thread (o2 % 16 == 0, 16-byte aligned output): the index
// split is done once per 16 values in 32-bit arithmetic (the form above spends three 64-bit divisions per BYTE: 69 us for
// the 16.7 MB of one 256^3 map, more than the accumulation over the 201 MB volume) and the 16 bytes leave as one store.
__global__ __launch_bounds__(256) void sim_quantize16(const float* __restrict__ sim, const unsigned* __restrict__ maxbits,
                                                      int classes, int n0, int n1, int n2, int o0, int o1, int o2,
                                                      unsigned char* __restrict__ out) {
  const int zc = o2 >> 4;                                   // 16-value chunks per output row
  const int rows = classes * o0 * o1;                       // (checked on the host: fits 31 bits, as does rows * zc)
  const float s0 = (float)n0 / (float)o0, s1 = (float)n1 / (float)o1, s2 = (float)n2 / (float)o2;
  for (int64_t e64 = (int64_t)blockIdx.x * 256 + threadIdx.x; e64 < (int64_t)rows * zc; e64 += (int64_t)gridDim.x * 256) {
    const int e = (int)e64;
    const int row = e / zc, zi = e - row * zc;
    const int c = row / (o0 * o1), r2 = row - c * (o0 * o1);
    const int x = r2 / o1, y = r2 - x * o1;
    int sx = (int)floorf((float)x * s0), sy = (int)floorf((float)y * s1);
    sx = sx < n0 - 1 ? sx : n0 - 1; sy = sy < n1 - 1 ? sy : n1 - 1;
    const float* src = sim + (int64_t)c * n0 * n1 * n2 + ((int64_t)sx * n1 + sy) * n2;
    const float scale = 255.0f / (0.99f * __uint_as_float(maxbits[c]));      // 255 / (0.99 * sim.max())  predict_ntf.py:98-99
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned word = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int z = 16 * zi + 4 * k + j;
        int sz = (int)floorf((float)z * s2);
        sz = sz < n2 - 1 ? sz : n2 - 1;
        const float q = scale * src[sz];
        const int qi = (q == q) ? (int)q : 0;               // truncate toward zero, keep the low byte; NaN (max == 0) -> 0
        word |= (unsigned)(qi & 255) << (8 * j);
      }
      w[k] = word;
    }
    *reinterpret_cast<uint4*>(out + (int64_t)row * o2 + 16 * zi) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// The same bytes when the output's last dim is a power-of-two multiple (R2 = 1 << SH <= 16) of the map's -- the nearest resize to
// vol.shape // 2 of a 512^3 volume's 64^3 maps: R2 = 4 -- : floorf(z * (1 / R2)) = z >> SH exactly, so a thread's 16 outputs
// are 16 / R2 consecutive map values (one 16-, 8- or 4-byte load) each repeated R2 times, and the row split needs no division
// per value (the general form above: 20 us for one 256^3 map, more than half of the accumulation over the 201 MB volume).
template <int SH>
__global__ __launch_bounds__(256) void sim_quantize16_pow2(const float* __restrict__ sim, const unsigned* __restrict__ maxbits,
                                                           int classes, int n0, int n1, int n2, int o0, int o1, int o2,
                                                           unsigned char* __restrict__ out) {
  constexpr int R2 = 1 << SH, NS = 16 / R2;               // outputs per map value, map values per thread
  const int zc = o2 >> 4;
  const int rows = classes * o0 * o1;
  const float s0 = (float)n0 / (float)o0, s1 = (float)n1 / (float)o1;
  for (int64_t e64 = (int64_t)blockIdx.x * 256 + threadIdx.x; e64 < (int64_t)rows * zc; e64 += (int64_t)gridDim.x * 256) {
    const int e = (int)e64;
    const int row = e / zc, zi = e - row * zc;
    const int c = row / (o0 * o1), r2 = row - c * (o0 * o1);
    const int x = r2 / o1, y = r2 - x * o1;
    int sx = (int)floorf((float)x * s0), sy = (int)floorf((float)y * s1);
    sx = sx < n0 - 1 ? sx : n0 - 1; sy = sy < n1 - 1 ? sy : n1 - 1;
    const float* src = sim + (int64_t)c * n0 * n1 * n2 + ((int64_t)sx * n1 + sy) * n2 + zi * NS;
    const float scale = 255.0f / (0.99f * __uint_as_float(maxbits[c]));
    float v[NS];
    if constexpr (NS == 4) { const float4 t = *reinterpret_cast<const float4*>(src); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if constexpr (NS == 2) { const float2 t = *reinterpret_cast<const float2*>(src); v[0] = t.x; v[1] = t.y; }
    else {
#pragma unroll
      for (int i = 0; i < NS; ++i) v[i] = src[i];
    }
    unsigned char b[16];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const float q = scale * v[i];
      const int qi = (q == q) ? (int)q : 0;
#pragma unroll
      for (int j = 0; j < R2; ++j) b[i * R2 + j] = (unsigned char)(qi & 255);
    }
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (unsigned)b[4 * k] | ((unsigned)b[4 * k + 1] << 8) | ((unsigned)b[4 * k + 2] << 16) | ((unsigned)b[4 * k + 3] << 24);
    *reinterpret_cast<uint4*>(out + (int64_t)row * o2 + 16 * zi) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

struct LabelArgs { int classes; int thr[16]; };

__global__ __launch_bounds__(256) void labels_kernel(const unsigned char* __restrict__ sims, int64_t n, LabelArgs a,
                                                     unsigned char* __restrict__ labels) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    int pred = 0, best = 0;
    for (int c = 0; c < a.classes; ++c) {
      const int s = sims[(int64_t)c * n + e];
      if (s > a.thr[c] && s > best) { pred = c + 1; best = s; }
    }
    labels[e] = (unsigned char)pred;
  }
}

}  // namespace

extern "C" int vittf_sample_features(const void* feat, int32_t feat_is_fp16, int32_t f, int32_t n0, int32_t n1, int32_t n2,
                                     const float* rel, int32_t a, int32_t mode, const float* voxel_norm, float* out,
                                     void* stream) {
  if (!feat || !rel || !out || f <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0 || a <= 0) return VITTF_ERR_INVALID_ARG;
  if (mode != VITTF_SAMPLE_NEAREST && mode != VITTF_SAMPLE_TRILINEAR) return VITTF_ERR_INVALID_ARG;
  const int64_t total = (int64_t)a * f;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (feat_is_fp16)
    hipLaunchKernelGGL((sample_kernel<true>), dim3(blocks), dim3(256), 0, st, feat, f, n0, n1, n2, rel, a, mode, voxel_norm, out);
  else
    hipLaunchKernelGGL((sample_kernel<false>), dim3(blocks), dim3(256), 0, st, feat, f, n0, n1, n2, rel, a, mode, voxel_norm, out);
  return vittf_check_launch();
}

extern "C" int vittf_voxel_norm(const uint16_t* feat, int32_t f, int64_t nvox, float* out, void* stream) {
  if (!feat || !out || f <= 0 || nvox <= 0 || ((uintptr_t)feat & 3) != 0) return VITTF_ERR_INVALID_ARG;
  if (f % 4 == 0 && nvox % 4 == 0 && (((uintptr_t)feat | (uintptr_t)out) & 15) == 0) {
    hipLaunchKernelGGL(voxel_norm_split_kernel, dim3((unsigned)((nvox / 4 + 63) / 64)), dim3(256), 0, (hipStream_t)stream, feat, f,
                       nvox, out);
    return vittf_check_launch();
  }
  const int64_t threads = (nvox + 1) / 2;
  hipLaunchKernelGGL(voxel_norm_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat, f, nvox,
                     out);
  return vittf_check_launch();
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t vittf_sim_mfma_workspace_bytes(int32_t classes, int32_t annotations);                      // sim_mfma.hip
bool vittf_sim_mfma_applies(int32_t f, int32_t classes, int32_t total_a, const void* ws, size_t ws_bytes, const void* feat,
                            int64_t nvox);
int vittf_sim_mfma_maps(const unsigned short* feat, int32_t f, int64_t nvox, const float* qf, const int32_t* class_start_host,
                        int32_t classes, const float* voxel_norm, float* sim, unsigned* maxbits, void* ws, size_t ws_bytes,
                        hipStream_t st);                                                         // 1 = not applicable

namespace {
struct SimWs { size_t maxbits, qf_t, mfma, mfma_bytes, maps, total; };
SimWs sim_ws_layout(int32_t classes, int64_t nvox, int32_t annotations) {
  // [max bits per class | transposed query chunk (F <= 4096) | padded fp16 query images (many annotations) | fp32 class maps]
  SimWs L{};
  L.maxbits = 0;
  L.qf_t = align256((size_t)classes * 4);
  L.mfma = L.qf_t + align256((size_t)4096 * ACH * 4);
  L.mfma_bytes = annotations >= 1 ? align256(vittf_sim_mfma_workspace_bytes(classes, annotations)) : 0;
  L.maps = L.mfma + L.mfma_bytes;
  L.total = L.maps + (size_t)classes * (size_t)nvox * 4;
  return L;
}
}  // namespace

extern "C" size_t vittf_similarity_workspace_bytes(int32_t classes, int64_t nvox, int32_t annotations) {
  if (classes <= 0 || nvox < 0 || annotations < 0) return 0;
  return sim_ws_layout(classes, nvox, annotations).total;
}

namespace {
// argument checks + the chunk walk shared by the two entry points: fp32 class maps [classes][nvox] into `sim`
// mode 0: predict_ntf (threshold / power per annotation, class mean); 1: its single-class A > 1024 variant (mean of the
// dots first); 2: resample_topk (clamp(0, 1) ** expo per query, group mean)
int accumulate_class_maps(const void* feat, bool half, int32_t f, int64_t nvox, const float* qf,
                          const int32_t* class_start_host, int32_t classes, int32_t mode, float expo,
                          const float* voxel_norm, float* sim, unsigned* maxbits, float* qf_t, void* mfma_ws,
                          size_t mfma_ws_bytes, hipStream_t st, bool maxbits_zeroed = false) {
  if (mode < 0 || mode > 2 || (!half && mode != 2)) return VITTF_ERR_INVALID_ARG;
  if (class_start_host[0] != 0) return VITTF_ERR_INVALID_ARG;
  for (int c = 0; c < classes; ++c)
    if (class_start_host[c + 1] <= class_start_host[c]) return VITTF_ERR_INVALID_ARG;  // empty class: caller drops it
  if (((uintptr_t)feat & 3) != 0) return VITTF_ERR_INVALID_ARG;
  if (!maxbits_zeroed && hipMemsetAsync(maxbits, 0, (size_t)classes * 4, st) != hipSuccess) return VITTF_ERR_LAUNCH;

  const int total_a = class_start_host[classes];
  if (mode == 0 && half && mfma_ws_bytes) {   // many annotations, F = 384 / 768: the volume is read once (sim_mfma.hip)
    static const bool use_mfma = [] { const char* e = getenv("VITTF_SIM_MFMA"); return !e || atoi(e) != 0; }();
    if (use_mfma && vittf_sim_mfma_applies(f, classes, total_a, mfma_ws, mfma_ws_bytes, feat, nvox)) {   // (asked first: an empty scope would count as a launch)
      ProfScope ps(VITTF_KERNEL_SIMILARITY, st);
      const int rc = vittf_sim_mfma_maps((const unsigned short*)feat, f, nvox, qf, class_start_host, classes, voxel_norm, sim,
                                         maxbits, mfma_ws, mfma_ws_bytes, st);
      if (rc != 1) return rc;
    }
  }
  const int64_t threads = (nvox + VPT - 1) / VPT;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  // walk the annotation list in chunks of ACH; a chunk may span several (at most MAXC) classes
  int a0 = 0;
  while (a0 < total_a) {
    SimChunk ch;
    int c_first = 0;
    while (class_start_host[c_first + 1] <= a0) ++c_first;
    ch.c0 = c_first;
    int n = 0, c = c_first;
    while (n < ACH && a0 + n < total_a) {
      while (class_start_host[c + 1] <= a0 + n) ++c;
      if (c - c_first >= MAXC) break;
      ch.cls[n] = c - c_first;
      ++n;
    }
    ch.n_ann = n;
    for (int i = n; i < ACH; ++i) ch.cls[i] = -1;
    ch.nc = ch.cls[n - 1] + 1;
    for (int i = 0; i < MAXC; ++i) { ch.first[i] = ch.last[i] = 0; ch.count[i] = 1.f; }
    for (int i = 0; i < ch.nc; ++i) {
      const int cc = c_first + i;
      ch.first[i] = class_start_host[cc] >= a0;
      ch.last[i] = class_start_host[cc + 1] <= a0 + n;
      ch.count[i] = (float)(class_start_host[cc + 1] - class_start_host[cc]);
    }
    hipLaunchKernelGGL(transpose_queries, dim3((f * ACH + 255) / 256), dim3(256), 0, st, qf, f, a0, n, qf_t);
    // split-feature kernel when the shapes allow its 8- / 16-byte loads (VITTF_SIM_SPLIT=0: always the one-wave-per-voxel-pair kernel)
    const char* split_env = getenv("VITTF_SIM_SPLIT");   // (read per call: the tests switch it)
    const bool use_split = !split_env || atoi(split_env) != 0;
    const bool split = use_split && f % 4 == 0 && nvox % 4 == 0 && ((uintptr_t)feat & 15) == 0;
    const unsigned sblocks = (unsigned)((nvox / SVPT + 63) / 64);
#define SIM_LAUNCH(BIG, ACT, HALF)                                                                                       \
  {                                                                                                                      \
    if (split) hipLaunchKernelGGL((sim_accumulate_split<BIG, ACT, HALF>), dim3(sblocks), dim3(256), 0, st, feat, f, nvox, \
                                  qf_t, ch, voxel_norm, expo, sim, maxbits);                                             \
    else hipLaunchKernelGGL((sim_accumulate<BIG, ACT, HALF>), dim3(blocks), dim3(256), 0, st, feat, f, nvox, qf_t, ch,   \
                            voxel_norm, expo, sim, maxbits);                                                             \
  }
    {
      vittf_note_kernel(VITTF_KERNEL_SIMILARITY, split ? "sim_accumulate_split" : "sim_accumulate");
      ProfScope ps(VITTF_KERNEL_SIMILARITY, st);
      if (mode == 1) SIM_LAUNCH(true, 0, true)
      else if (mode == 0) SIM_LAUNCH(false, 0, true)
      else if (half) SIM_LAUNCH(false, 1, true)
      else SIM_LAUNCH(false, 1, false)
    }
#undef SIM_LAUNCH
    a0 += n;
  }
  return VITTF_OK;
}
}  // namespace

namespace {
// the quantise + resize launch behind the accumulation (predict_ntf.py:95-100)
int quantize_maps(const float* sim, const unsigned* maxbits, int32_t classes, int32_t n0, int32_t n1, int32_t n2, int32_t o0,
                  int32_t o1, int32_t o2, uint8_t* out, hipStream_t st) {
  const int64_t total_out = (int64_t)classes * o0 * o1 * o2;
  int64_t qblocks = (total_out + 255) / 256;
  if (qblocks > 8192) qblocks = 8192;
  const char* q16_env = getenv("VITTF_SIM_QUANT16");   // (read per call: the tests compare the kernels; 2 = the general 16-value form)
  const int q16 = q16_env ? atoi(q16_env) : 1;
  if (q16 != 0 && o2 % 16 == 0 && ((uintptr_t)out & 15) == 0 && total_out / 16 < 0x7fffffff && (int64_t)classes * o0 * o1 < 0x7fffffff) {
    qblocks = (total_out / 16 + 255) / 256;
    if (qblocks > 16384) qblocks = 16384;
    const int r2 = o2 % n2 == 0 ? o2 / n2 : 0;
    const bool pow2 = q16 == 1 && (r2 == 1 || r2 == 2 || r2 == 4 || r2 == 8 || r2 == 16) && ((uintptr_t)sim & 15) == 0 && n2 % (16 / r2) == 0 &&
                      ((int64_t)n2 * 4) % 16 == 0;
#define SIM_Q16P(SH) hipLaunchKernelGGL((sim_quantize16_pow2<SH>), dim3((unsigned)qblocks), dim3(256), 0, st, sim, maxbits, classes, n0, n1, n2, o0, o1, o2, out)
    if (pow2 && r2 == 1) SIM_Q16P(0);
    else if (pow2 && r2 == 2) SIM_Q16P(1);
    else if (pow2 && r2 == 4) SIM_Q16P(2);
    else if (pow2 && r2 == 8) SIM_Q16P(3);
    else if (pow2 && r2 == 16) SIM_Q16P(4);
    else
      hipLaunchKernelGGL(sim_quantize16, dim3((unsigned)qblocks), dim3(256), 0, st, sim, maxbits, classes, n0, n1, n2, o0, o1,
                         o2, out);
#undef SIM_Q16P
  } else {
    hipLaunchKernelGGL(sim_quantize, dim3((unsigned)qblocks), dim3(256), 0, st, sim, maxbits, classes, n0, n1, n2, o0, o1,
                       o2, out);
  }
  return vittf_check_launch();
}
}  // namespace

extern "C" int vittf_similarity(const uint16_t* feat, int32_t f, int32_t n0, int32_t n1, int32_t n2, const float* qf,
                                const int32_t* class_start_host, int32_t classes, int32_t big_a_mean,
                                const float* voxel_norm, int32_t o0, int32_t o1, int32_t o2, uint8_t* out, void* ws,
                                size_t ws_bytes, void* stream) {
  if (!feat || !qf || !class_start_host || !out || !ws) return VITTF_ERR_INVALID_ARG;
  if (f <= 0 || f > 4096 || n0 <= 0 || n1 <= 0 || n2 <= 0 || classes <= 0 || o0 <= 0 || o1 <= 0 || o2 <= 0)
    return VITTF_ERR_INVALID_ARG;
  const int64_t nvox = (int64_t)n0 * n1 * n2;
  if (class_start_host[classes] < 0) return VITTF_ERR_INVALID_ARG;
  const SimWs L = sim_ws_layout(classes, nvox, class_start_host[classes]);
  if (ws_bytes < L.total) return VITTF_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* wsb = (char*)ws;
  unsigned* maxbits = (unsigned*)wsb;
  float* qf_t = (float*)(wsb + L.qf_t);
  float* sim = (float*)(wsb + L.maps);
  const int rc = accumulate_class_maps(feat, true, f, nvox, qf, class_start_host, classes, big_a_mean ? 1 : 0, 0.f,
                                       voxel_norm, sim, maxbits, qf_t, wsb + L.mfma, L.mfma_bytes, st);
  if (rc != VITTF_OK) return rc;
  return quantize_maps(sim, maxbits, classes, n0, n1, n2, o0, o1, o2, out, st);
}

extern "C" size_t vittf_similarity_query_workspace_bytes(int32_t classes, int64_t nvox, int32_t annotations, int32_t f) {
  if (classes <= 0 || nvox < 0 || annotations <= 0 || f <= 0) return 0;
  return sim_ws_layout(classes, nvox, annotations).total + align256((size_t)annotations * f * 4);
}

extern "C" int vittf_similarity_query(const uint16_t* feat, int32_t f, int32_t n0, int32_t n1, int32_t n2, const float* rel_host,
                                      const int32_t* class_start_host, int32_t classes, int32_t big_a_mean,
                                      const float* voxel_norm, int32_t o0, int32_t o1, int32_t o2, uint8_t* out, void* ws,
                                      size_t ws_bytes, void* stream) {
  if (!feat || !rel_host || !class_start_host || !out || !ws) return VITTF_ERR_INVALID_ARG;
  if (f <= 0 || f > 4096 || n0 <= 0 || n1 <= 0 || n2 <= 0 || classes <= 0 || o0 <= 0 || o1 <= 0 || o2 <= 0)
    return VITTF_ERR_INVALID_ARG;
  const int a = class_start_host[classes];
  if (a <= 0 || a > VITTF_QUERY_MAX_A || classes > 256) return VITTF_ERR_INVALID_ARG;
  const int64_t nvox = (int64_t)n0 * n1 * n2;
  const SimWs L = sim_ws_layout(classes, nvox, a);
  if (ws_bytes < L.total + align256((size_t)a * f * 4)) return VITTF_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* wsb = (char*)ws;
  unsigned* maxbits = (unsigned*)wsb;
  float* qf = (float*)(wsb + L.total);
  RelArg rel;
  for (int i = 0; i < 3 * VITTF_QUERY_MAX_A; ++i) rel.v[i] = i < 3 * a ? rel_host[i] : 0.f;
  const int64_t total = (int64_t)a * f;
  hipLaunchKernelGGL(sample_query_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, feat, f, n0, n1, n2, rel, a,
                     voxel_norm, qf, maxbits, classes);
  const int rc = accumulate_class_maps(feat, true, f, nvox, qf, class_start_host, classes, big_a_mean ? 1 : 0, 0.f, voxel_norm,
                                       (float*)(wsb + L.maps), maxbits, (float*)(wsb + L.qf_t), wsb + L.mfma, L.mfma_bytes, st,
                                       /*maxbits_zeroed=*/true);
  if (rc != VITTF_OK) return rc;
  return quantize_maps((const float*)(wsb + L.maps), maxbits, classes, n0, n1, n2, o0, o1, o2, out, st);
}

extern "C" int vittf_similarity_maps_f32(const void* feat, int32_t feat_is_fp16, int32_t f, int32_t n0, int32_t n1,
                                         int32_t n2, const float* qf, const int32_t* class_start_host, int32_t classes,
                                         int32_t mode, float exponent, const float* voxel_norm, float* maps_out, void* ws,
                                         size_t ws_bytes, void* stream) {
  if (!feat || !qf || !class_start_host || !maps_out || !ws) return VITTF_ERR_INVALID_ARG;
  if (f <= 0 || f > 4096 || n0 <= 0 || n1 <= 0 || n2 <= 0 || classes <= 0) return VITTF_ERR_INVALID_ARG;
  const int64_t nvox = (int64_t)n0 * n1 * n2;
  if (class_start_host[classes] < 0) return VITTF_ERR_INVALID_ARG;
  const SimWs L = sim_ws_layout(classes, 0, class_start_host[classes]);
  if (ws_bytes < L.total) return VITTF_ERR_WORKSPACE;
  char* wsb = (char*)ws;
  const int rc = accumulate_class_maps(feat, feat_is_fp16 != 0, f, nvox, qf, class_start_host, classes, mode, exponent,
                                       voxel_norm, maps_out, (unsigned*)wsb, (float*)(wsb + L.qf_t), wsb + L.mfma,
                                       L.mfma_bytes, (hipStream_t)stream);
  return rc != VITTF_OK ? rc : vittf_check_launch();
}

extern "C" int vittf_assign_labels(const uint8_t* sims, int32_t classes, int64_t n, const int32_t* thr_host,
                                   uint8_t* labels, void* stream) {
  if (!sims || !thr_host || !labels || classes <= 0 || classes > 16 || n <= 0) return VITTF_ERR_INVALID_ARG;
  LabelArgs a;
  a.classes = classes;
  for (int i = 0; i < 16; ++i) a.thr[i] = i < classes ? thr_host[i] : 0;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(labels_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sims, n, a, labels);
  return vittf_check_launch();
}
