// Label-volume helpers either side of the similarity query: the candidate masks the
// annotation samplers draw from, and the confusion matrix every reported score derives from.
//
// Replaces, in sample_surface (compare_feat_sampling.py:19-24):
//     outer = binary_erosion(vol, generate_binary_structure(3, dist_from_surface))
//     inner = binary_erosion(outer, generate_binary_structure(3, 1));   np.logical_xor(inner, outer)
// (scipy.ndimage defaults: one iteration, origin 0, border_value 0 -- voxels outside the volume count as unset), and in
// predict_ntf.py:228-246 / evaluate_similarities.py:63-68 the sklearn confusion_matrix the precision / recall / F1 /
// IoU / accuracy figures come from.  Both are byte streams over up to 512^3 voxels, HBM-bound when done right: 8 voxels
// of the fast axis per thread with byte-parallel set membership (erosion; a one-voxel-per-thread kernel covers ragged
// shapes), 16 voxels per thread and a lane-replicated LDS histogram (confusion matrix).
#include "vittf_common.h"

#include <stdlib.h>

namespace {

// structuring element of generate_binary_structure(3, c): offsets with |d0| + |d1| + |d2| <= c; bit (d0+1)*9 + (d1+1)*3 + (d2+1)
unsigned structure_bits(int connectivity) {
  unsigned bits = 0;
  for (int a = -1; a <= 1; ++a)
    for (int b = -1; b <= 1; ++b)
      for (int c = -1; c <= 1; ++c)
        if (abs(a) + abs(b) + abs(c) <= connectivity) bits |= 1u << ((a + 1) * 9 + (b + 1) * 3 + (c + 1));
  return bits;
}

// SHELL = false: dst = erosion of the set {src == cls} (cls >= 0) or {src != 0} (cls < 0)
// SHELL = true : dst = src_set AND NOT erosion(src_set)   (= XOR, the erosion being a subset: the element has its centre)
template <bool SHELL>
__global__ __launch_bounds__(256) void erode_kernel(const unsigned char* __restrict__ src, int n0, int n1, int n2, int cls,
                                                    unsigned bits, unsigned char* __restrict__ dst) {
  const int i2 = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i1 = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int i0 = blockIdx.z;
  if (i2 >= n2 || i1 >= n1) return;
  const int64_t s1 = n2, s0 = (int64_t)n1 * n2;
  const int64_t v = i0 * s0 + i1 * s1 + i2;
  bool all = true;
#pragma unroll
  for (int a = -1; a <= 1; ++a)
#pragma unroll
    for (int b = -1; b <= 1; ++b)
#pragma unroll
      for (int c = -1; c <= 1; ++c) {
        if (!(bits & (1u << ((a + 1) * 9 + (b + 1) * 3 + (c + 1))))) continue;   // (uniform)
        const int j0 = i0 + a, j1 = i1 + b, j2 = i2 + c;
        bool set = false;
        if (j0 >= 0 && j0 < n0 && j1 >= 0 && j1 < n1 && j2 >= 0 && j2 < n2) {
          const unsigned char x = src[v + a * s0 + b * s1 + c];
          set = cls >= 0 ? x == (unsigned char)cls : x != 0;
        }
        all = all && set;
      }
  if (SHELL) {
    const unsigned char x = src[v];
    const bool centre = cls >= 0 ? x == (unsigned char)cls : x != 0;
    dst[v] = centre && !all;
  } else {
    dst[v] = all;
  }
}

// Same result, 8 voxels of the fast axis per thread (n2 % 8 == 0, 8-byte aligned volumes): one 8-byte load per
// neighbouring row instead of three byte loads per voxel and row.  A row's membership is kept as one 0x80 flag per
// byte; the two bytes beyond the thread's eight come from the neighbouring lanes (from memory at the wave's ends).
__device__ __forceinline__ unsigned member_flags(unsigned x, int cls) {
  const unsigned y = cls >= 0 ? x ^ (0x01010101u * (unsigned)cls) : x;
  const unsigned zero = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & 0x80808080u;   // 0x80 exactly where the byte is 0
  return cls >= 0 ? zero : zero ^ 0x80808080u;
}

template <bool SHELL>
__global__ __launch_bounds__(256) void erode8_kernel(const unsigned char* __restrict__ src, int n0, int n1, int n2, int cls,
                                                     int connectivity, unsigned char* __restrict__ dst) {
  const int chunks = n2 >> 3;                                   // per row
  const int q = blockIdx.x * 256 + threadIdx.x;                 // chunk of the (i1, i2) plane
  const int i0 = blockIdx.y;
  const bool live = q < n1 * chunks;
  const int i1 = live ? q / chunks : 0, c8 = live ? q - i1 * chunks : 0;
  const int lane = threadIdx.x & 63;
  const int64_t s1 = n2, s0 = (int64_t)n1 * n2;
  const int64_t v = i0 * s0 + i1 * s1 + 8 * c8;
  unsigned long long all = 0x8080808080808080ull, centre = 0;
#pragma unroll
  for (int a = -1; a <= 1; ++a)
#pragma unroll
    for (int b = -1; b <= 1; ++b) {
      const int reach = connectivity - (a != 0) - (b != 0);     // offsets |c| <= reach of this row belong to the element
      if (reach < 0) continue;                                  // (uniform)
      const int j0 = i0 + a, j1 = i1 + b;
      const bool row_in = live && j0 >= 0 && j0 < n0 && j1 >= 0 && j1 < n1;
      unsigned long long f = 0;
      const unsigned char* row = src + v + a * s0 + b * s1;
      if (row_in) {
        const uint2 w = *reinterpret_cast<const uint2*>(row);
        f = (unsigned long long)member_flags(w.x, cls) | ((unsigned long long)member_flags(w.y, cls) << 32);
      }
      if (a == 0 && b == 0) centre = f;
      if (reach >= 1) {
        // the voxel before the first and after the last of the eight: the neighbouring chunk of the same row
        unsigned lo = (unsigned)f, hi = (unsigned)(f >> 32);
        unsigned left = __shfl_up(hi, 1) >> 31, right = (__shfl_down(lo, 1) >> 7) & 1u;
        if (lane == 0) left = (row_in && c8 > 0) ? (member_flags(row[-1], cls) >> 7) & 1u : 0;
        if (lane == 63) right = (row_in && c8 + 1 < chunks) ? (member_flags(row[8], cls) >> 7) & 1u : 0;
        if (c8 == 0) left = 0;
        if (c8 + 1 == chunks) right = 0;
        if (!row_in) { left = 0; right = 0; }
        const unsigned long long fl = (f << 8) | ((unsigned long long)left << 7);
        const unsigned long long fr = (f >> 8) | ((unsigned long long)right << 63);
        f &= fl & fr;
      }
      all &= f;
    }
  if (!live) return;
  const unsigned long long r = SHELL ? (centre & ~all) : all;
  const unsigned long long bytes = r >> 7;                      // 0x80 flags -> 0 / 1 bytes
  *reinterpret_cast<uint2*>(dst + v) = make_uint2((unsigned)bytes, (unsigned)(bytes >> 32));
}

constexpr int CM_MAX_CLASSES = 16, CM_COPIES = 32;

// counts[t * classes + p] += 1 for every voxel; counts[classes * classes] counts voxels with a value >= classes
__global__ __launch_bounds__(256) void confusion_kernel(const unsigned char* __restrict__ target,
                                                        const unsigned char* __restrict__ pred, int64_t n, int classes,
                                                        unsigned long long* __restrict__ counts) {
  // one copy of the histogram per lane (mod 32): a lane's LDS atomics hit its own bank, whatever the data
  __shared__ unsigned hist[(CM_MAX_CLASSES * CM_MAX_CLASSES + 1) * CM_COPIES];
  const int tid = threadIdx.x, copy = tid & (CM_COPIES - 1);
  const int bins = classes * classes + 1;
  for (int i = tid; i < bins * CM_COPIES; i += 256) hist[i] = 0;
  __syncthreads();
  const int64_t chunks = n / 16;
  const bool aligned = ((((size_t)target) | ((size_t)pred)) & 15) == 0;
  if (aligned) {
    for (int64_t c = (int64_t)blockIdx.x * 256 + tid; c < chunks; c += (int64_t)gridDim.x * 256) {
      const uint4 t4 = reinterpret_cast<const uint4*>(target)[c];
      const uint4 p4 = reinterpret_cast<const uint4*>(pred)[c];
      const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int t = (tw[w] >> (8 * b)) & 255, p = (pw[w] >> (8 * b)) & 255;
          const int bin = (t < classes && p < classes) ? t * classes + p : classes * classes;
          atomicAdd(&hist[bin * CM_COPIES + copy], 1u);
        }
    }
  }
  // the tail (and everything, when the pointers are not 16-byte aligned)
  for (int64_t i = (aligned ? chunks * 16 : 0) + (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = target[i], p = pred[i];
    const int bin = (t < classes && p < classes) ? t * classes + p : classes * classes;
    atomicAdd(&hist[bin * CM_COPIES + copy], 1u);
  }
  __syncthreads();
  for (int b = tid; b < bins; b += 256) {
    unsigned long long s = 0;
    for (int c = 0; c < CM_COPIES; ++c) s += hist[b * CM_COPIES + c];
    if (s) atomicAdd(counts + b, s);
  }
}

}  // namespace

namespace {
template <bool SHELL>
void launch_erode(const unsigned char* src, int n0, int n1, int n2, int cls, int connectivity, unsigned char* dst, hipStream_t st) {
  if (connectivity < 1) connectivity = 1;
  if (connectivity > 3) connectivity = 3;
  static const bool wide = [] { const char* e = getenv("VITTF_ERODE_WIDE"); return !e || atoi(e) != 0; }();
  if (wide && n2 % 8 == 0 && ((((size_t)src) | ((size_t)dst)) & 7) == 0) {
    const dim3 grid((unsigned)(((int64_t)n1 * (n2 / 8) + 255) / 256), n0);
    hipLaunchKernelGGL(erode8_kernel<SHELL>, grid, dim3(256), 0, st, src, n0, n1, n2, cls, connectivity, dst);
  } else {
    const dim3 grid((n2 + 63) / 64, (n1 + 3) / 4, n0);
    hipLaunchKernelGGL(erode_kernel<SHELL>, grid, dim3(256), 0, st, src, n0, n1, n2, cls, structure_bits(connectivity), dst);
  }
}
}  // namespace

extern "C" size_t vittf_surface_shell_workspace_bytes(int32_t n0, int32_t n1, int32_t n2) {
  if (n0 <= 0 || n1 <= 0 || n2 <= 0) return 0;
  return (((size_t)n0 * n1 * n2) + 255) & ~(size_t)255;
}

extern "C" int vittf_erode_mask(const uint8_t* src, int32_t n0, int32_t n1, int32_t n2, int32_t class_id, int32_t connectivity,
                                uint8_t* dst, void* stream) {
  if (!src || !dst || n0 <= 0 || n1 <= 0 || n2 <= 0 || n0 > 65535 || class_id > 255 || src == dst) return VITTF_ERR_INVALID_ARG;
  if ((n1 + 3) / 4 > 65535) return VITTF_ERR_INVALID_ARG;
  launch_erode<false>(src, n0, n1, n2, class_id, connectivity, dst, (hipStream_t)stream);
  return vittf_check_launch();
}

extern "C" int vittf_surface_shell(const uint8_t* labels, int32_t n0, int32_t n1, int32_t n2, int32_t class_id,
                                   int32_t connectivity, uint8_t* shell, void* ws, size_t ws_bytes, void* stream) {
  if (!labels || !shell || !ws || n0 <= 0 || n1 <= 0 || n2 <= 0 || n0 > 65535 || class_id > 255) return VITTF_ERR_INVALID_ARG;
  if (ws_bytes < vittf_surface_shell_workspace_bytes(n0, n1, n2)) return VITTF_ERR_WORKSPACE;
  if ((n1 + 3) / 4 > 65535) return VITTF_ERR_INVALID_ARG;
  unsigned char* outer = (unsigned char*)ws;
  launch_erode<false>(labels, n0, n1, n2, class_id, connectivity, outer, (hipStream_t)stream);
  launch_erode<true>(outer, n0, n1, n2, -1, 1, shell, (hipStream_t)stream);
  return vittf_check_launch();
}

extern "C" int vittf_confusion_matrix(const uint8_t* target, const uint8_t* pred, int64_t n, int32_t classes, int64_t* counts,
                                      void* stream) {
  if (!counts || n < 0 || classes < 1 || classes > CM_MAX_CLASSES || (n > 0 && (!target || !pred))) return VITTF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(counts, 0, ((size_t)classes * classes + 1) * 8, st) != hipSuccess) return VITTF_ERR_LAUNCH;
  if (n == 0) return VITTF_OK;
  int64_t blocks = (n / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), 0, st, target, pred, n, classes,
                     (unsigned long long*)counts);
  return vittf_check_launch();
}
