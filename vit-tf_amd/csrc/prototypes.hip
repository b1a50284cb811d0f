// Query refinement helpers of the interactive loop: top-K voxels of a similarity map and the
// mean pairwise distance used to prune prototypes.
//
// Replaces, in resample_topk (infer.py:94-97):
//     kth = torch.topk(s.flatten(), K, largest=True, sorted=True).values[-1];  (s >= kth).nonzero()[:K]
// and in take_most_dissimilar (infer.py:118-121):
//     1 - F.cosine_similarity(f[None], f[:, None], dim=-1).mean(0)      /      torch.cdist(f, f).mean(0)
// Both are HBM-/latency-bound integer and fp32 work on small inputs (maps of 64^3..256^3 voxels, a few thousand
// feature vectors): one 1024-thread workgroup per map / row block, no sort.
#include "vittf_common.h"

namespace {

// order-preserving map float -> uint32 (larger float <=> larger key); NaN sorts above +inf like torch.topk
__device__ __forceinline__ unsigned order_key(float v) {
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// One workgroup per map.  Radix select (4 passes of 8 bits, 256-bin LDS histogram) finds the key of the k-th largest
// value; an ordered compaction then writes the first k indices whose key is >= it.
__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ maps, int64_t nvox, int k,
                                                    int* __restrict__ idx_out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_prefix, sel_remaining;
  __shared__ int wave_count[16];
  __shared__ int written;
  const float* s = maps + (int64_t)blockIdx.x * nvox;
  int* out = idx_out + (int64_t)blockIdx.x * k;
  const int tid = threadIdx.x;
  unsigned prefix = 0, remaining = (unsigned)k;      // keys matching `prefix` in the bits above `shift` are candidates
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    for (int64_t i = tid; i < nvox; i += 1024) {
      const unsigned key = order_key(s[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (tid == 0) {      // walk the bins from the largest digit down until the k-th largest falls inside one
      unsigned rem = remaining;
      int d = 255;
      for (; d > 0; --d) {
        if (hist[d] >= rem) break;
        rem -= hist[d];
      }
      sel_prefix = prefix | ((unsigned)d << shift);
      sel_remaining = rem;
    }
    __syncthreads();
    prefix = sel_prefix;
    remaining = sel_remaining;
    __syncthreads();
  }
  const unsigned kth = prefix;
  // ordered compaction: chunks of 1024 consecutive voxels, wave ballots + a 16-entry scan
  if (tid == 0) written = 0;
  __syncthreads();
  for (int64_t base = 0; base < nvox; base += 1024) {
    const int64_t i = base + tid;
    const bool hit = i < nvox && order_key(s[i]) >= kth;
    const unsigned long long bal = __ballot(hit);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) wave_count[wave] = __popcll(bal);
    __syncthreads();
    int before = written;
    for (int w = 0; w < wave; ++w) before += wave_count[w];
    const int pos = before + __popcll(bal & ((1ull << lane) - 1));
    if (hit && pos < k) out[pos] = (int)i;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wave_count[w];
      written += tot;
    }
    __syncthreads();
    if (written >= k) break;
  }
}

// dist[i] over all j; one workgroup per row i, the 256 threads stride over j; row i cached in LDS.
template <int MEASURE>
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ x, int n, int f, float* __restrict__ dist) {
  extern __shared__ float xi[];
  __shared__ double red[256];
  const int i = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < f; c += 256) xi[c] = x[(int64_t)i * f + c];
  __syncthreads();
  float nii = 0.f;
  for (int c = 0; c < f; ++c) nii = fmaf(xi[c], xi[c], nii);
  double acc = 0.0;
  for (int j = tid; j < n; j += 256) {
    const float* xj = x + (int64_t)j * f;
    float dot = 0.f, njj = 0.f, d2 = 0.f;
    for (int c = 0; c < f; ++c) {
      const float a = xi[c], b = xj[c];
      if (MEASURE == 0) { dot = fmaf(a, b, dot); njj = fmaf(b, b, njj); }
      else { const float d = a - b; d2 = fmaf(d, d, d2); }
    }
    if (MEASURE == 0) acc += (double)(dot / (fmaxf(sqrtf(nii), 1e-8f) * fmaxf(sqrtf(njj), 1e-8f)));   // F.cosine_similarity, eps = 1e-8
    else acc += (double)sqrtf(d2);
  }
  red[tid] = acc;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (tid < s2) red[tid] += red[tid + s2];
    __syncthreads();
  }
  if (tid == 0) {
    const float mean = (float)(red[0] / (double)n);
    dist[i] = MEASURE == 0 ? 1.0f - mean : mean;
  }
}

}  // namespace

extern "C" int vittf_topk_voxels(const float* maps, int32_t nmaps, int64_t nvox, int32_t k, int32_t* idx_out, void* stream) {
  if (!maps || !idx_out || nmaps <= 0 || nvox <= 0 || k <= 0 || k > nvox || nvox > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(topk_kernel, dim3(nmaps), dim3(1024), 0, (hipStream_t)stream, maps, nvox, k, idx_out);
  return vittf_check_launch();
}

extern "C" int vittf_mean_pairwise_distance(const float* x, int32_t n, int32_t f, int32_t measure, float* dist, void* stream) {
  if (!x || !dist || n <= 0 || f <= 0 || f > 8192 || (measure != 0 && measure != 1)) return VITTF_ERR_INVALID_ARG;
  if (measure == 0)
    hipLaunchKernelGGL((pairwise_kernel<0>), dim3(n), dim3(256), (size_t)f * 4, (hipStream_t)stream, x, n, f, dist);
  else
    hipLaunchKernelGGL((pairwise_kernel<1>), dim3(n), dim3(256), (size_t)f * 4, (hipStream_t)stream, x, n, f, dist);
  return vittf_check_launch();
}
