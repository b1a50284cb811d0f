// 3-D bilateral-solver refinement of one class similarity map, on the device.
//
// Replaces the optional post-process of predict_ntf.compute_similarities (predict_ntf.py:73-96) and what it calls
// in bilateral_solver3d.py: trilinear resizes (F.interpolate), the uint8 reference (norm_minmax, infer.py:32-34),
// crop_pad (:183-204), the Sobel-magnitude confidence (:176-181, 233-237), BilateralGrid (:37-105, SciPy CSR
// splat / blur matrices from float64-hashed coordinates + np.unique), bistochastize (:107-118), the
// Jacobi-preconditioned conjugate gradient of BilateralSolver.solve (:128-154, scipy.sparse.linalg.cg, <= 25
// iterations), slice, nan_to_num (:245) and write_crop_into (:206-209).
//
// Device formulation.  The reference volume is grey, so the two chroma coordinates are constant and a bilateral
// vertex is (x, y, z, luma); np.unique's sorted-hash order is the lexicographic order of (luma, z, y, x).  That key
// space is small (luma bins x spatial bins of the crop box), so there is no sort and no sparse matrix:
//   occupancy table over the key space -> exclusive prefix sum = vertex ids in the reference's order ->
//   per-vertex index of the 8 neighbours (+-1 in x, y, z, luma; -1 = absent).
// splat = atomic adds (int counts, fp64 sums), blur(y)_i = 12 y_i + sum of present neighbours, slice = gather.
// Bistochastisation and the CG are one launch per phase over all vertices (fp64; per-workgroup tree reductions whose
// partials every workgroup then adds in the same fixed order, so dot products are deterministic; convergence latched
// in a device flag, no host round trip).  All arithmetic that the reference does in fp64 is fp64 here;
// the fp32 parts (resize, Sobel) use unfused multiply / add in the CPU operators' order.
#include "vittf_common.h"

#include <math.h>
#include <stdlib.h>

namespace {

constexpr int BT = 256;

// ---- F.interpolate(mode='trilinear', align_corners=False), fp32 ------------------------------------------------
__device__ __forceinline__ void lin_coord(int o, float scale, int in_size, int& i0, int& i1, float& w0, float& w1) {
  float src = __fsub_rn(__fmul_rn(scale, (float)o + 0.5f), 0.5f);   // area_pixel_compute_source_index
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 < in_size - 1 ? i0 + 1 : i0;
  w1 = __fsub_rn(src, (float)i0);
  w0 = __fsub_rn(1.f, w1);
}

__global__ __launch_bounds__(BT) void trilinear_kernel(const float* __restrict__ src, int s0, int s1, int s2,
                                                       float* __restrict__ dst, int d0, int d1, int d2) {
  const int64_t idx = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (idx >= (int64_t)d0 * d1 * d2) return;
  const int x = (int)(idx % d2), y = (int)((idx / d2) % d1), z = (int)(idx / ((int64_t)d1 * d2));
  int z0, z1, y0, y1, x0, x1;
  float wz0, wz1, wy0, wy1, wx0, wx1;
  lin_coord(z, (float)s0 / (float)d0, s0, z0, z1, wz0, wz1);
  lin_coord(y, (float)s1 / (float)d1, s1, y0, y1, wy0, wy1);
  lin_coord(x, (float)s2 / (float)d2, s2, x0, x1, wx0, wx1);
  auto at = [&](int a, int b, int c) { return src[((int64_t)a * s1 + b) * s2 + c]; };
  auto row = [&](int a, int b) { return __fadd_rn(__fmul_rn(wx0, at(a, b, x0)), __fmul_rn(wx1, at(a, b, x1))); };
  auto plane = [&](int a) { return __fadd_rn(__fmul_rn(wy0, row(a, y0)), __fmul_rn(wy1, row(a, y1))); };
  dst[idx] = __fadd_rn(__fmul_rn(wz0, plane(z0)), __fmul_rn(wz1, plane(z1)));
}

// (255 * (v - min) / (max - min)).to(uint8): predict_ntf.py:84-85 with norm_minmax (infer.py:32-34)
__global__ __launch_bounds__(BT) void to_u8_kernel(const float* __restrict__ v, int64_t n,
                                                   const float* __restrict__ minmax, unsigned char* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  const float mi = minmax[0], ma = minmax[1];
  const float t = __fmul_rn(255.0f, __fdiv_rn(__fsub_rn(v[i], mi), __fsub_rn(ma, mi)));
  out[i] = (unsigned char)(int)t;
}

// bounding box of sim > thresh: bounds = {min0, min1, min2, max0, max1, max2} (initialised to INT_MAX / -1)
__global__ __launch_bounds__(BT) void bbox_kernel(const float* __restrict__ sim, int d0, int d1, int d2, float thresh,
                                                  int* __restrict__ bounds) {
  __shared__ int red[6][BT / 64];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
  const int64_t n = (int64_t)d0 * d1 * d2;
  for (int64_t idx = (int64_t)blockIdx.x * BT + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * BT) {
    if (sim[idx] > thresh) {
      const int x = (int)(idx % d2), y = (int)((idx / d2) % d1), z = (int)(idx / ((int64_t)d1 * d2));
      lo[0] = min(lo[0], z); lo[1] = min(lo[1], y); lo[2] = min(lo[2], x);
      hi[0] = max(hi[0], z); hi[1] = max(hi[1], y); hi[2] = max(hi[2], x);
    }
  }
  // thread, wave, then workgroup reduction: 6 atomics per workgroup of a 1024-workgroup grid instead of 6 per voxel
  // above the threshold (4 ms at 256^3)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[i] = min(lo[i], __shfl_xor(lo[i], off));
      hi[i] = max(hi[i], __shfl_xor(hi[i], off));
    }
    if ((threadIdx.x & 63) == 0) { red[i][threadIdx.x >> 6] = lo[i]; red[3 + i][threadIdx.x >> 6] = hi[i]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    int v = red[threadIdx.x][0];
    for (int w = 1; w < BT / 64; ++w) v = threadIdx.x < 3 ? min(v, red[threadIdx.x][w]) : max(v, red[threadIdx.x][w]);
    if (threadIdx.x < 3) { if (v != 0x7fffffff) atomicMin(bounds + threadIdx.x, v); }
    else if (v >= 0) atomicMax(bounds + threadIdx.x, v);
  }
}

struct Box { int lo0, lo1, lo2, c0, c1, c2, d1, d2; };   // crop origin / extent inside a (., d1, d2) volume

// Sobel magnitude of ref / 255 inside the crop (zero padding at the crop faces), fp32, the three squares added in the
// reference's order (last dim first); also the maximum (as float bits: the values are >= 0)
__global__ __launch_bounds__(BT) void sobel_kernel(const unsigned char* __restrict__ ref, Box b, float* __restrict__ g,
                                                   unsigned* __restrict__ gmax_bits) {
  const int64_t idx = (int64_t)blockIdx.x * BT + threadIdx.x;
  const int64_t n = (int64_t)b.c0 * b.c1 * b.c2;
  float val = 0.f;
  if (idx < n) {
    const int x = (int)(idx % b.c2), y = (int)((idx / b.c2) % b.c1), z = (int)(idx / ((int64_t)b.c1 * b.c2));
    auto at = [&](int zz, int yy, int xx) -> float {
      if (zz < 0 || zz >= b.c0 || yy < 0 || yy >= b.c1 || xx < 0 || xx >= b.c2) return 0.f;
      return __fdiv_rn((float)ref[((int64_t)(b.lo0 + zz) * b.d1 + (b.lo1 + yy)) * b.d2 + (b.lo2 + xx)], 255.0f);
    };
    auto diff = [&](float hi, float lo) { return __fsub_rn(__fmul_rn(0.5f, hi), __fmul_rn(0.5f, lo)); };
    const float dx = diff(at(z, y, x + 1), at(z, y, x - 1));
    const float dy = diff(at(z, y + 1, x), at(z, y - 1, x));
    const float dz = diff(at(z + 1, y, x), at(z - 1, y, x));
    val = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    g[idx] = val;
  }
  // block maximum -> one atomic
  __shared__ float red[BT];
  red[threadIdx.x] = val;
  __syncthreads();
  for (int s = BT / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(gmax_bits, __float_as_uint(red[0]));
}

struct KeyDims { int nl, nz, ny, nx; };

// bilateral-space key of every crop voxel + occupancy of the key space
__global__ __launch_bounds__(BT) void key_kernel(const unsigned char* __restrict__ ref, Box b, const int* __restrict__ luma_bin,
                                                 double sigma_spatial, KeyDims kd, unsigned* __restrict__ key,
                                                 int* __restrict__ occupied) {
  const int64_t idx = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (idx >= (int64_t)b.c0 * b.c1 * b.c2) return;
  const int x = (int)(idx % b.c2), y = (int)((idx / b.c2) % b.c1), z = (int)(idx / ((int64_t)b.c1 * b.c2));
  const int cx = (int)((double)x / sigma_spatial), cy = (int)((double)y / sigma_spatial), cz = (int)((double)z / sigma_spatial);
  const int cl = luma_bin[ref[((int64_t)(b.lo0 + z) * b.d1 + (b.lo1 + y)) * b.d2 + (b.lo2 + x)]];
  const unsigned k = (unsigned)(((cl * kd.nz + cz) * kd.ny + cy) * kd.nx + cx);
  key[idx] = k;
  occupied[k] = 1;
}

// exclusive prefix sum of `occupied` (0 / 1): rank[k] = vertex id of key k; total -> *nvert.  Three launches: sums of
// blocks of RANK_CHUNK keys, a one-workgroup scan of those sums, then the ranks inside every block (one workgroup over
// the ~2 M keys of a 256^3 crop took 1.7 ms).
constexpr int RANK_CHUNK = 1024;
__global__ __launch_bounds__(BT) void rank_sums_kernel(const int* __restrict__ occupied, int nkeys, int* __restrict__ block_sum) {
  __shared__ int red[BT / 64];
  const int base = blockIdx.x * RANK_CHUNK;
  int s = 0;
#pragma unroll
  for (int j = 0; j < RANK_CHUNK / BT; ++j) {
    const int k = base + j * BT + threadIdx.x;
    if (k < nkeys) s += occupied[k];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) block_sum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// in place: block_sum[b] -> sum of the blocks before b; *nvert = total
__global__ __launch_bounds__(1024) void rank_scan_kernel(int* __restrict__ block_sum, int nblocks, int* __restrict__ nvert) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (nblocks + 1023) / 1024;
  const int lo = tid * per, hi = min(nblocks, lo + per);
  int s = 0;
  for (int k = lo; k < hi; ++k) s += block_sum[k];
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {     // Hillis-Steele inclusive scan
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - s;
  for (int k = lo; k < hi; ++k) { const int v = block_sum[k]; block_sum[k] = run; run += v; }
  if (tid == 1023) *nvert = part[1023];
}

__global__ __launch_bounds__(BT) void rank_kernel(const int* __restrict__ occupied, int nkeys, const int* __restrict__ block_off,
                                                  int* __restrict__ rank) {
  __shared__ int wave_tot[BT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int run = block_off[blockIdx.x];
#pragma unroll 1
  for (int j = 0; j < RANK_CHUNK / BT; ++j) {
    const int k = blockIdx.x * RANK_CHUNK + j * BT + threadIdx.x;
    const int v = k < nkeys ? occupied[k] : 0;
    const unsigned long long mask = __ballot(v != 0);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(mask);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) { if (w < wave) off += wave_tot[w]; tot += wave_tot[w]; }
    if (k < nkeys) rank[k] = run + off + before;
    run += tot;
    __syncthreads();
  }
}

// splat: per-vertex voxel count, confidence sum and confidence-weighted target sum; the voxel -> vertex map
__global__ __launch_bounds__(BT) void splat_kernel(const unsigned* __restrict__ key, const int* __restrict__ rank,
                                                   const float* __restrict__ g, const unsigned* __restrict__ gmax_bits,
                                                   const float* __restrict__ sim, Box b, int64_t n,
                                                   int* __restrict__ vertex_of_voxel, int* __restrict__ count,
                                                   double* __restrict__ w_splat, double* __restrict__ b_splat) {
  const int64_t idx = (int64_t)blockIdx.x * BT + threadIdx.x;
  const bool live = idx < n;
  int v = -1;
  int cnt = 0;
  double c = 0.0, tc = 0.0;
  if (live) {
    v = rank[key[idx]];
    vertex_of_voxel[idx] = v;
    const int x = (int)(idx % b.c2), y = (int)((idx / b.c2) % b.c1), z = (int)(idx / ((int64_t)b.c1 * b.c2));
    const double t = (double)sim[((int64_t)(b.lo0 + z) * b.d1 + (b.lo1 + y)) * b.d2 + (b.lo2 + x)];
    c = (double)__fsub_rn(__uint_as_float(*gmax_bits), g[idx]);   // confidence = max - Sobel, fp32 then double
    tc = t * c;
    cnt = 1;
  }
  // neighbours along x share a grid cell (and mostly a luma bin): runs of equal vertex ids inside the wave are summed
  // in the lanes (segmented suffix sums) and only the head of a run goes to memory -- a fifth of the atomics
  const int lane = threadIdx.x & 63;
  const int vnext = __shfl_down(v, 1);
  int ended = (lane == 63 || vnext != v) ? 1 : 0;      // my partial sum already reaches the end of my run
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {             // segmented suffix sums by doubling
    const int co = __shfl_down(cnt, off), eo = __shfl_down(ended, off);
    const double cc = __shfl_down(c, off), tt = __shfl_down(tc, off);
    if (!ended && lane + off < 64) { cnt += co; c += cc; tc += tt; ended = eo; }
  }
  const int vprev = __shfl_up(v, 1);
  const bool head = live && (lane == 0 || vprev != v);
  if (head) {
    atomicAdd(count + v, cnt);
    atomicAdd(w_splat + v, c);
    atomicAdd(b_splat + v, tc);
  }
}

// neighbour table: nb[dir][vertex], dir = 2 * dim + (step > 0), dims x, y, z, luma
__global__ __launch_bounds__(BT) void neighbour_kernel(const int* __restrict__ occupied, const int* __restrict__ rank,
                                                       KeyDims kd, int nkeys, int nvert_cap, int* __restrict__ nb) {
  const int k = blockIdx.x * BT + threadIdx.x;
  if (k >= nkeys || !occupied[k]) return;
  const int v = rank[k];
  int rem = k;
  const int cx = rem % kd.nx; rem /= kd.nx;
  const int cy = rem % kd.ny; rem /= kd.ny;
  const int cz = rem % kd.nz;
  const int cl = rem / kd.nz;
  const int coord[4] = {cx, cy, cz, cl};
  const int size[4] = {kd.nx, kd.ny, kd.nz, kd.nl};
  const int stride[4] = {1, kd.nx, kd.nx * kd.ny, kd.nx * kd.ny * kd.nz};
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = coord[d] + (s ? 1 : -1);
      int id = -1;
      if (c >= 0 && c < size[d]) {
        const int k2 = k + (s ? stride[d] : -stride[d]);
        if (occupied[k2]) id = rank[k2];
      }
      nb[(int64_t)(2 * d + s) * nvert_cap + v] = id;
    }
}

struct SolveArgs {
  int nvert, nvert_cap;
  const int* count;          // voxels per vertex (splat of ones)
  const int* nb;             // [8][nvert_cap]
  const double* w_splat;     // splat(confidence)
  const double* b;           // splat(target * confidence)
  double *m, *n, *x, *r, *p, *q, *inv_diag, *tmp;
  double lam, a_diag_min, rtol;
  int maxiter, bistoch_iters;
};

__device__ __forceinline__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
  return out;
}

__device__ __forceinline__ double blur_at(const double* __restrict__ y, const int* __restrict__ nb, int cap, int i) {
  double out = 12.0 * y[i];     // 2 * dim with dim = 6 (x, y, z, luma, u, v)
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const int j = nb[(int64_t)d * cap + i];
    if (j >= 0) out += y[j];
  }
  return out;
}

// bistochastize (bilateral_solver3d.py:107-118) + BilateralSolver.solve (:128-154) in one workgroup
__global__ __launch_bounds__(1024) void solve_kernel(SolveArgs a) {
  __shared__ double red[1024];
  const int tid = threadIdx.x, nv = a.nvert, cap = a.nvert_cap;
  // ---- n = 1; 10 x: n = sqrt(n * m0 / blur(n));  m = n * blur(n) ----
  for (int i = tid; i < nv; i += 1024) a.n[i] = 1.0;
  __syncthreads();
  for (int it = 0; it < a.bistoch_iters; ++it) {
    for (int i = tid; i < nv; i += 1024) a.tmp[i] = blur_at(a.n, a.nb, cap, i);
    __syncthreads();
    for (int i = tid; i < nv; i += 1024) a.n[i] = sqrt(a.n[i] * (double)a.count[i] / a.tmp[i]);
    __syncthreads();
  }
  for (int i = tid; i < nv; i += 1024) a.m[i] = a.n[i] * blur_at(a.n, a.nb, cap, i);
  __syncthreads();
  // A y = lam * (m y - n blur(n y)) + w_splat y;  diag(A) = lam * (m - 12 n^2) + w_splat
  auto apply_a = [&](const double* y, double* out) {
    for (int i = tid; i < nv; i += 1024) a.tmp[i] = a.n[i] * y[i];
    __syncthreads();
    for (int i = tid; i < nv; i += 1024)
      out[i] = a.lam * (a.m[i] * y[i] - a.n[i] * blur_at(a.tmp, a.nb, cap, i)) + a.w_splat[i] * y[i];
    __syncthreads();
  };
  double bb = 0.0, anyx = 0.0;
  for (int i = tid; i < nv; i += 1024) {
    const double d = a.lam * (a.m[i] - a.n[i] * 12.0 * a.n[i]) + a.w_splat[i];
    a.inv_diag[i] = 1.0 / fmax(d, a.a_diag_min);
    const double x0 = a.b[i] / a.w_splat[i];     // flat initialisation splat(x w) / splat(w)
    a.x[i] = x0;
    bb += a.b[i] * a.b[i];
    anyx += (x0 != 0.0) ? 1.0 : 0.0;             // NaN counts as "any", like ndarray.any()
  }
  const double bnorm = sqrt(block_sum(bb, red));
  anyx = block_sum(anyx, red);
  if (bnorm == 0.0) {                            // scipy: returns b (all zeros) at once
    for (int i = tid; i < nv; i += 1024) a.x[i] = a.b[i];
    return;
  }
  const double atol = a.rtol * bnorm;
  if (anyx != 0.0) {
    apply_a(a.x, a.q);
    for (int i = tid; i < nv; i += 1024) a.r[i] = a.b[i] - a.q[i];
  } else {
    for (int i = tid; i < nv; i += 1024) a.r[i] = a.b[i];
  }
  __syncthreads();
  double rho_prev = 0.0;
  for (int it = 0; it < a.maxiter; ++it) {
    double rr = 0.0;
    for (int i = tid; i < nv; i += 1024) rr += a.r[i] * a.r[i];
    rr = block_sum(rr, red);
    if (sqrt(rr) < atol) break;                  // NaN compares false: the loop runs on, as in the reference
    double rho = 0.0;
    for (int i = tid; i < nv; i += 1024) rho += a.r[i] * (a.inv_diag[i] * a.r[i]);
    rho = block_sum(rho, red);
    const double beta = it > 0 ? rho / rho_prev : 0.0;
    for (int i = tid; i < nv; i += 1024) {
      const double z = a.inv_diag[i] * a.r[i];
      a.p[i] = it > 0 ? a.p[i] * beta + z : z;
    }
    __syncthreads();
    apply_a(a.p, a.q);
    double pq = 0.0;
    for (int i = tid; i < nv; i += 1024) pq += a.p[i] * a.q[i];
    pq = block_sum(pq, red);
    const double alpha = rho / pq;
    for (int i = tid; i < nv; i += 1024) {
      a.x[i] += alpha * a.p[i];
      a.r[i] -= alpha * a.q[i];
    }
    __syncthreads();
    rho_prev = rho;
  }
}

// ---- the same solver spread over the chip: one launch per phase instead of one workgroup for everything ----------
// (77 k vertices, ~45 dependent vector passes: 16.5 ms in one workgroup.)  Every dot product is reduced in two fixed
// steps -- a tree inside each workgroup, then every workgroup adds the per-workgroup partials in the same order -- so the
// result does not depend on scheduling.  Convergence (SciPy's test, at the top of an iteration) is decided identically
// by every workgroup from those partials and latched in `done`, after which the remaining launches return at once.
struct SolveScratch {
  double *bb, *anyx, *rr[2], *rho[2], *pq;   // per-workgroup partial sums
  int* done;
  int nblk;
};

__device__ __forceinline__ double wg_sum(double v, double* red) {   // BT threads, fixed tree
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
#pragma unroll
  for (int s = BT / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
  return out;
}
// `done` as ONE value per workgroup (workgroup 0 may set it while others start the same kernel: every thread of a
// workgroup must take the same branch in front of the barriers below)
__device__ __forceinline__ bool wg_done(const int* done) {
  __shared__ int flag;
  if (threadIdx.x == 0) flag = *done;
  __syncthreads();
  const bool d = flag != 0;
  __syncthreads();
  return d;
}
__device__ __forceinline__ double all_sum(const double* __restrict__ part, int nblk, double* red) {
  double v = 0.0;
  for (int i = threadIdx.x; i < nblk; i += BT) v += part[i];
  return wg_sum(v, red);
}

__global__ __launch_bounds__(BT) void bs_step_kernel(SolveArgs a, const double* __restrict__ n_in, double* __restrict__ n_out,
                                                     int first) {
  const int i = blockIdx.x * BT + threadIdx.x;
  if (i >= a.nvert) return;
  if (first) { n_out[i] = 1.0; return; }
  n_out[i] = sqrt(n_in[i] * (double)a.count[i] / blur_at(n_in, a.nb, a.nvert_cap, i));
}

// m = n blur(n); diag(A), flat initialisation, |b|^2 and any(x0) partials; tmp = n x0 for the first A x
__global__ __launch_bounds__(BT) void cg_setup_kernel(SolveArgs a, SolveScratch sc) {
  __shared__ double red[BT];
  const int i = blockIdx.x * BT + threadIdx.x;
  double bb = 0.0, anyx = 0.0;
  if (i < a.nvert) {
    const double ni = a.n[i];
    const double mi = ni * blur_at(a.n, a.nb, a.nvert_cap, i);
    a.m[i] = mi;
    const double d = a.lam * (mi - ni * 12.0 * ni) + a.w_splat[i];
    a.inv_diag[i] = 1.0 / fmax(d, a.a_diag_min);
    const double x0 = a.b[i] / a.w_splat[i];     // flat initialisation splat(x w) / splat(w)
    a.x[i] = x0;
    a.tmp[i] = ni * x0;
    bb = a.b[i] * a.b[i];
    anyx = (x0 != 0.0) ? 1.0 : 0.0;              // NaN counts as "any", like ndarray.any()
  }
  bb = wg_sum(bb, red);
  anyx = wg_sum(anyx, red);
  if (threadIdx.x == 0) { sc.bb[blockIdx.x] = bb; sc.anyx[blockIdx.x] = anyx; }
  if (blockIdx.x == 0 && threadIdx.x == 0) *sc.done = 0;
}

// r = b - A x0 (or b when x0 is all zero); |b| = 0: x = b and done (scipy returns b at once); partials of r.r, r.Dr
__global__ __launch_bounds__(BT) void cg_residual_kernel(SolveArgs a, SolveScratch sc) {
  __shared__ double red[BT];
  const double bnorm = sqrt(all_sum(sc.bb, sc.nblk, red));
  const double anyx = all_sum(sc.anyx, sc.nblk, red);
  const int i = blockIdx.x * BT + threadIdx.x;
  if (bnorm == 0.0) {
    if (i < a.nvert) a.x[i] = a.b[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *sc.done = 1;
    return;
  }
  double rr = 0.0, rho = 0.0;
  if (i < a.nvert) {
    double r = a.b[i];
    if (anyx != 0.0)
      r -= a.lam * (a.m[i] * a.x[i] - a.n[i] * blur_at(a.tmp, a.nb, a.nvert_cap, i)) + a.w_splat[i] * a.x[i];
    a.r[i] = r;
    rr = r * r;
    rho = r * (a.inv_diag[i] * r);
  }
  rr = wg_sum(rr, red);
  rho = wg_sum(rho, red);
  if (threadIdx.x == 0) { sc.rr[0][blockIdx.x] = rr; sc.rho[0][blockIdx.x] = rho; }
}

// top of iteration `it`: convergence test, then p = z + beta p and tmp = n p
__global__ __launch_bounds__(BT) void cg_p_kernel(SolveArgs a, SolveScratch sc, int it) {
  __shared__ double red[BT];
  if (wg_done(sc.done)) return;
  const int cur = it & 1;
  const double rr = all_sum(sc.rr[cur], sc.nblk, red);
  const double atol = a.rtol * sqrt(all_sum(sc.bb, sc.nblk, red));
  if (sqrt(rr) < atol) {                         // NaN compares false: the loop runs on, as in the reference
    if (blockIdx.x == 0 && threadIdx.x == 0) *sc.done = 1;
    return;
  }
  const double rho = all_sum(sc.rho[cur], sc.nblk, red);
  const double beta = it > 0 ? rho / all_sum(sc.rho[cur ^ 1], sc.nblk, red) : 0.0;
  const int i = blockIdx.x * BT + threadIdx.x;
  if (i < a.nvert) {
    const double z = a.inv_diag[i] * a.r[i];
    const double p = it > 0 ? a.p[i] * beta + z : z;
    a.p[i] = p;
    a.tmp[i] = a.n[i] * p;
  }
}

// q = A p and the partials of p.q
__global__ __launch_bounds__(BT) void cg_q_kernel(SolveArgs a, SolveScratch sc) {
  __shared__ double red[BT];
  if (wg_done(sc.done)) return;
  const int i = blockIdx.x * BT + threadIdx.x;
  double pq = 0.0;
  if (i < a.nvert) {
    const double p = a.p[i];
    const double q = a.lam * (a.m[i] * p - a.n[i] * blur_at(a.tmp, a.nb, a.nvert_cap, i)) + a.w_splat[i] * p;
    a.q[i] = q;
    pq = p * q;
  }
  pq = wg_sum(pq, red);
  if (threadIdx.x == 0) sc.pq[blockIdx.x] = pq;
}

// x += alpha p, r -= alpha q and the partials of the new r.r, r.Dr (for iteration it + 1)
__global__ __launch_bounds__(BT) void cg_x_kernel(SolveArgs a, SolveScratch sc, int it) {
  __shared__ double red[BT];
  if (wg_done(sc.done)) return;
  const int cur = it & 1;
  const double alpha = all_sum(sc.rho[cur], sc.nblk, red) / all_sum(sc.pq, sc.nblk, red);
  const int i = blockIdx.x * BT + threadIdx.x;
  double rr = 0.0, rho = 0.0;
  if (i < a.nvert) {
    a.x[i] += alpha * a.p[i];
    const double r = a.r[i] - alpha * a.q[i];
    a.r[i] = r;
    rr = r * r;
    rho = r * (a.inv_diag[i] * r);
  }
  rr = wg_sum(rr, red);
  rho = wg_sum(rho, red);
  if (threadIdx.x == 0) { sc.rr[cur ^ 1][blockIdx.x] = rr; sc.rho[cur ^ 1][blockIdx.x] = rho; }
}

// slice + torch.nan_to_num + write_crop_into
__global__ __launch_bounds__(BT) void slice_kernel(const double* __restrict__ x, const int* __restrict__ vertex_of_voxel, Box b,
                                                   int64_t n, float* __restrict__ sim) {
  const int64_t idx = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (idx >= n) return;
  float v = (float)x[vertex_of_voxel[idx]];
  if (isnan(v)) v = 0.f;
  else if (isinf(v)) v = v > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  const int xx = (int)(idx % b.c2), yy = (int)((idx / b.c2) % b.c1), zz = (int)(idx / ((int64_t)b.c1 * b.c2));
  sim[((int64_t)(b.lo0 + zz) * b.d1 + (b.lo1 + yy)) * b.d2 + (b.lo2 + xx)] = v;
}

// (255 / (0.99 max) * sim).to(uint8) with the x86 wrap-around (predict_ntf.py:95-96): max via float bits (sim >= 0)
__global__ __launch_bounds__(BT) void absmax_kernel(const float* __restrict__ sim, int64_t n, float* __restrict__ mx) {
  __shared__ float red[BT];
  float v = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * BT + threadIdx.x; i < n; i += (int64_t)gridDim.x * BT) v = fmaxf(v, sim[i]);
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = BT / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {   // float max by compare-and-swap (values may be negative after the solver)
    unsigned* p = reinterpret_cast<unsigned*>(mx);
    unsigned old = *p;
    while (red[0] > __uint_as_float(old)) {
      const unsigned seen = atomicCAS(p, old, __float_as_uint(red[0]));
      if (seen == old) break;
      old = seen;
    }
  }
}
__global__ __launch_bounds__(BT) void quantize_kernel(const float* __restrict__ sim, int64_t n, const float* __restrict__ mx,
                                                      unsigned char* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  const float quant = __fmul_rn(0.99f, *mx);
  const float s = __fmul_rn(__fdiv_rn(255.0f, quant), sim[i]);
  out[i] = (unsigned char)((long long)s & 255);     // truncate toward zero, then wrap like the x86 cast
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
unsigned blocks_for(int64_t n) { return (unsigned)((n + BT - 1) / BT); }

struct Layout {
  size_t vol_r, vol_u8, bounds, gmax, minmax, mm_ws, nvert, luma, g, key, vov, occupied, rank, rank_blocks, count, nb, vecs, partials, total;
  int64_t nvox; int nkeys_cap; int nvert_cap;
};

Layout make_layout(int o0, int o1, int o2, double sigma_spatial, int luma_bins) {
  Layout L{};
  L.nvox = (int64_t)o0 * o1 * o2;
  const int64_t k0 = (int64_t)((o0 - 1) / sigma_spatial) + 1, k1 = (int64_t)((o1 - 1) / sigma_spatial) + 1,
                k2 = (int64_t)((o2 - 1) / sigma_spatial) + 1;
  const int64_t nkeys = (int64_t)luma_bins * k0 * k1 * k2;
  L.nkeys_cap = nkeys > 0x7fffffff ? 0 : (int)nkeys;
  L.nvert_cap = (int)(nkeys < L.nvox ? nkeys : L.nvox);
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
  L.vol_r = take(L.nvox * 4); L.vol_u8 = take(L.nvox); L.bounds = take(6 * 4); L.gmax = take(8); L.minmax = take(8);
  L.mm_ws = take(vittf_minmax_workspace_bytes()); L.nvert = take(4); L.luma = take(256 * 4);
  L.g = take(L.nvox * 4); L.key = take(L.nvox * 4); L.vov = take(L.nvox * 4);
  L.occupied = take((size_t)L.nkeys_cap * 4); L.rank = take((size_t)L.nkeys_cap * 4);
  L.count = take((size_t)L.nvert_cap * 4); L.nb = take((size_t)L.nvert_cap * 8 * 4);
  L.vecs = take((size_t)L.nvert_cap * 8 * 10);   // w_splat, b, m, n, x, r, p, q, inv_diag, tmp
  L.rank_blocks = take(((size_t)L.nkeys_cap / RANK_CHUNK + 1) * 4);
  L.partials = take(((size_t)L.nvert_cap / BT + 1) * 8 * 7 + 256);   // bb, anyx, rr[2], rho[2], pq per workgroup + done
  L.total = off;
  return L;
}

constexpr int LUMA_BINS_MAX = 256;

}  // namespace

extern "C" size_t vittf_bilateral_workspace_bytes(int32_t o0, int32_t o1, int32_t o2, double sigma_spatial,
                                                  int32_t luma_bins) {
  if (o0 <= 0 || o1 <= 0 || o2 <= 0 || !(sigma_spatial > 0) || luma_bins <= 0 || luma_bins > LUMA_BINS_MAX) return 0;
  const Layout L = make_layout(o0, o1, o2, sigma_spatial, luma_bins);
  return L.nkeys_cap ? L.total : 0;
}

extern "C" int vittf_bilateral_refine(const float* sim_in, int32_t n0, int32_t n1, int32_t n2, const float* volume, int32_t v0,
                                      int32_t v1, int32_t v2, int32_t o0, int32_t o1, int32_t o2,
                                      const int32_t* luma_bin_host, int32_t luma_bins, const vittf_bilateral_params* prm,
                                      float* sim_out, int32_t* info_host, void* ws, size_t ws_bytes, void* stream) {
  if (!sim_in || !volume || !luma_bin_host || !prm || !sim_out || !ws) return VITTF_ERR_INVALID_ARG;
  if (n0 <= 0 || n1 <= 0 || n2 <= 0 || v0 <= 0 || v1 <= 0 || v2 <= 0 || o0 <= 0 || o1 <= 0 || o2 <= 0)
    return VITTF_ERR_INVALID_ARG;
  if (!(prm->sigma_spatial > 0) || prm->cg_maxiter < 0 || prm->pad < 0 || luma_bins <= 0 || luma_bins > LUMA_BINS_MAX)
    return VITTF_ERR_INVALID_ARG;
  for (int i = 0; i < 256; ++i)
    if (luma_bin_host[i] < 0 || luma_bin_host[i] >= luma_bins) return VITTF_ERR_INVALID_ARG;
  const Layout L = make_layout(o0, o1, o2, prm->sigma_spatial, luma_bins);
  if (!L.nkeys_cap) return VITTF_ERR_INVALID_ARG;
  if (ws_bytes < L.total) return VITTF_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)ws;
  float* vol_r = (float*)(w + L.vol_r);
  unsigned char* vol_u8 = (unsigned char*)(w + L.vol_u8);
  int* bounds = (int*)(w + L.bounds);
  unsigned* gmax = (unsigned*)(w + L.gmax);
  float* minmax = (float*)(w + L.minmax);
  int* nvert_d = (int*)(w + L.nvert);
  int* luma_d = (int*)(w + L.luma);
  const int64_t nvox = L.nvox;

  // ---- reference volume -> sim_shape (trilinear), min-max, uint8 (predict_ntf.py:80-85) ----
  hipLaunchKernelGGL(trilinear_kernel, dim3(blocks_for(nvox)), dim3(BT), 0, st, volume, v0, v1, v2, vol_r, o0, o1, o2);
  int rc = vittf_volume_minmax(vol_r, nvox, minmax, w + L.mm_ws, vittf_minmax_workspace_bytes(), stream);
  if (rc != VITTF_OK) return rc;
  hipLaunchKernelGGL(to_u8_kernel, dim3(blocks_for(nvox)), dim3(BT), 0, st, vol_r, nvox, minmax, vol_u8);
  // ---- similarity -> sim_shape (predict_ntf.py:86-88) ----
  if (n0 == o0 && n1 == o1 && n2 == o2) {
    if (hipMemcpyAsync(sim_out, sim_in, nvox * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return VITTF_ERR_LAUNCH;
  } else {
    hipLaunchKernelGGL(trilinear_kernel, dim3(blocks_for(nvox)), dim3(BT), 0, st, sim_in, n0, n1, n2, sim_out, o0, o1, o2);
  }
  // ---- crop_pad (bilateral_solver3d.py:183-204) ----
  const int init[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1, -1, -1};
  if (hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) return VITTF_ERR_LAUNCH;
  hipLaunchKernelGGL(bbox_kernel, dim3(blocks_for(nvox) < 1024 ? blocks_for(nvox) : 1024), dim3(BT), 0, st, sim_out, o0, o1, o2,
                     prm->crop_threshold, bounds);
  int hb[6];
  if (hipMemcpyAsync(hb, bounds, sizeof(hb), hipMemcpyDeviceToHost, st) != hipSuccess) return VITTF_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return VITTF_ERR_LAUNCH;
  if (info_host) { info_host[0] = 0; info_host[1] = 0; }
  if (hb[3] < 0) return vittf_check_launch();       // nothing above the threshold: the map is returned unrefined
  const int dims[3] = {o0, o1, o2};
  Box b{};
  int lo[3], hi[3];
  for (int i = 0; i < 3; ++i) {
    lo[i] = hb[i] - prm->pad < 0 ? 0 : hb[i] - prm->pad;
    hi[i] = hb[3 + i] + prm->pad + 1 > dims[i] ? dims[i] : hb[3 + i] + prm->pad + 1;
  }
  b.lo0 = lo[0]; b.lo1 = lo[1]; b.lo2 = lo[2];
  b.c0 = hi[0] - lo[0]; b.c1 = hi[1] - lo[1]; b.c2 = hi[2] - lo[2];
  b.d1 = o1; b.d2 = o2;
  const int64_t n = (int64_t)b.c0 * b.c1 * b.c2;
  KeyDims kd{luma_bins, (int)((b.c0 - 1) / prm->sigma_spatial) + 1, (int)((b.c1 - 1) / prm->sigma_spatial) + 1,
             (int)((b.c2 - 1) / prm->sigma_spatial) + 1};
  const int nkeys = kd.nl * kd.nz * kd.ny * kd.nx;      // <= nkeys_cap (the crop is inside the volume)

  float* g = (float*)(w + L.g);
  unsigned* key = (unsigned*)(w + L.key);
  int* vov = (int*)(w + L.vov);
  int* occupied = (int*)(w + L.occupied);
  int* rank = (int*)(w + L.rank);
  int* count = (int*)(w + L.count);
  int* nb = (int*)(w + L.nb);
  double* vecs = (double*)(w + L.vecs);
  const int cap = L.nvert_cap;

  if (hipMemcpyAsync(luma_d, luma_bin_host, 256 * 4, hipMemcpyHostToDevice, st) != hipSuccess) return VITTF_ERR_LAUNCH;
  (void)hipMemsetAsync(gmax, 0, 8, st);
  (void)hipMemsetAsync(occupied, 0, (size_t)nkeys * 4, st);
  hipLaunchKernelGGL(sobel_kernel, dim3(blocks_for(n)), dim3(BT), 0, st, vol_u8, b, g, gmax);
  hipLaunchKernelGGL(key_kernel, dim3(blocks_for(n)), dim3(BT), 0, st, vol_u8, b, luma_d, prm->sigma_spatial, kd, key, occupied);
  {
    int* block_off = (int*)(w + L.rank_blocks);
    const int rb = (nkeys + RANK_CHUNK - 1) / RANK_CHUNK;
    hipLaunchKernelGGL(rank_sums_kernel, dim3(rb), dim3(BT), 0, st, occupied, nkeys, block_off);
    hipLaunchKernelGGL(rank_scan_kernel, dim3(1), dim3(1024), 0, st, block_off, rb, nvert_d);
    hipLaunchKernelGGL(rank_kernel, dim3(rb), dim3(BT), 0, st, occupied, nkeys, block_off, rank);
  }
  int nvert = 0;
  if (hipMemcpyAsync(&nvert, nvert_d, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return VITTF_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return VITTF_ERR_LAUNCH;
  if (nvert <= 0 || nvert > cap) return VITTF_ERR_LAUNCH;
  if (info_host) { info_host[0] = nvert; info_host[1] = (int)n; }
  (void)hipMemsetAsync(count, 0, (size_t)nvert * 4, st);
  (void)hipMemsetAsync(vecs, 0, (size_t)cap * 8 * 2, st);          // w_splat, b
  SolveArgs a{};
  a.nvert = nvert; a.nvert_cap = cap; a.count = count; a.nb = nb;
  a.w_splat = vecs; a.b = vecs + cap;
  a.m = vecs + 2 * (size_t)cap; a.n = vecs + 3 * (size_t)cap; a.x = vecs + 4 * (size_t)cap; a.r = vecs + 5 * (size_t)cap;
  a.p = vecs + 6 * (size_t)cap; a.q = vecs + 7 * (size_t)cap; a.inv_diag = vecs + 8 * (size_t)cap; a.tmp = vecs + 9 * (size_t)cap;
  a.lam = prm->lam; a.a_diag_min = prm->a_diag_min; a.rtol = prm->cg_tol; a.maxiter = prm->cg_maxiter;
  a.bistoch_iters = prm->bistochastize_iters;
  hipLaunchKernelGGL(splat_kernel, dim3(blocks_for(n)), dim3(BT), 0, st, key, rank, g, gmax, sim_out, b, n, vov, count,
                     vecs, vecs + cap);
  hipLaunchKernelGGL(neighbour_kernel, dim3(blocks_for(nkeys)), dim3(BT), 0, st, occupied, rank, kd, nkeys, cap, nb);
  static const bool wide_solver = [] { const char* e = getenv("VITTF_BLS_SOLVER"); return !e || atoi(e) != 0; }();
  if (!wide_solver) {
    hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(1024), 0, st, a);     // everything in one workgroup (first version)
  } else {
    SolveScratch sc{};
    sc.nblk = (nvert + BT - 1) / BT;
    const size_t pstride = (size_t)cap / BT + 1;
    double* pbase = (double*)(w + L.partials);
    sc.bb = pbase; sc.anyx = pbase + pstride; sc.rr[0] = pbase + 2 * pstride; sc.rr[1] = pbase + 3 * pstride;
    sc.rho[0] = pbase + 4 * pstride; sc.rho[1] = pbase + 5 * pstride; sc.pq = pbase + 6 * pstride;
    sc.done = (int*)(pbase + 7 * pstride);
    const dim3 grid(sc.nblk), block(BT);
    // n = 1; bistoch_iters x: n = sqrt(n m0 / blur(n)), ping-pong between a.n and a.p (free until the CG starts)
    double* nbuf[2] = {a.n, a.p};
    hipLaunchKernelGGL(bs_step_kernel, grid, block, 0, st, a, nbuf[1], nbuf[0], 1);
    int cur = 0;
    for (int it = 0; it < a.bistoch_iters; ++it, cur ^= 1)
      hipLaunchKernelGGL(bs_step_kernel, grid, block, 0, st, a, nbuf[cur], nbuf[cur ^ 1], 0);
    if (cur == 1 && hipMemcpyAsync(a.n, a.p, (size_t)nvert * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) return VITTF_ERR_LAUNCH;
    hipLaunchKernelGGL(cg_setup_kernel, grid, block, 0, st, a, sc);
    hipLaunchKernelGGL(cg_residual_kernel, grid, block, 0, st, a, sc);
    for (int it = 0; it < a.maxiter; ++it) {
      hipLaunchKernelGGL(cg_p_kernel, grid, block, 0, st, a, sc, it);
      hipLaunchKernelGGL(cg_q_kernel, grid, block, 0, st, a, sc);
      hipLaunchKernelGGL(cg_x_kernel, grid, block, 0, st, a, sc, it);
    }
  }
  hipLaunchKernelGGL(slice_kernel, dim3(blocks_for(n)), dim3(BT), 0, st, a.x, vov, b, n, sim_out);
  return vittf_check_launch();
}

extern "C" int vittf_quantize_wrap_u8(const float* sim, int64_t n, uint8_t* out, float* max_scratch, void* stream) {
  if (!sim || !out || !max_scratch || n <= 0) return VITTF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const float ninf = -INFINITY;
  if (hipMemcpyAsync(max_scratch, &ninf, 4, hipMemcpyHostToDevice, st) != hipSuccess) return VITTF_ERR_LAUNCH;
  const unsigned nb = blocks_for(n) < 1024 ? blocks_for(n) : 1024;
  hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(BT), 0, st, sim, n, max_scratch);
  hipLaunchKernelGGL(quantize_kernel, dim3(blocks_for(n)), dim3(BT), 0, st, sim, n, max_scratch, out);
  return vittf_check_launch();
}
