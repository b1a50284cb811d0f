// Per-class similarity maps on the matrix cores (F = 384 and 768): the interactive few-query case and BASELINE configs[4].
//
// Same arithmetic as sim_accumulate (similarity.hip; predict_ntf.py:65, 71-72): for every voxel and class,
// mean over the class's annotations of where(q . x >= 0.25, q . x, 0) ** 2.5.  The VALU kernel spends 16 FMAs per
// 2-byte feature element (A = 16: 41 us of VALU issue for a 30 us stream) and re-reads the whole feature volume for
// every 16 annotations (A = 5 x 1024: 320 passes over 201 MB); here the volume is read ONCE at the speed of the stream:
//   * workgroup = 8 waves = 256 voxels; a wave keeps the 384 features of its 32 voxels as 24 MFMA B-operands
//     (96 VGPRs) for the whole kernel -- the volume is the stationary operand, the queries stream;
//   * the volume is F-major ([feature][voxel]), the B operand wants 8 consecutive FEATURES of one voxel per lane: the
//     workgroup's [384][256] tile comes in by LDS-DMA as whole 512-byte feature rows (four parts of 96 rows through the
//     two halves of the ring, 16-byte chunks XOR-swizzled by the row on the SOURCE side) and every wave picks its
//     fragments up with ds_read_b64_tr_b16 (4 rows x 16 voxels per 16 lanes, conflict-free); volumes whose rows are not
//     16-byte aligned (nvox % 8) fall back to 2-byte strided loads;
//   * the queries are prepared once per call (sim_mfma_prep) as fp16 hi + lo halves (q = hi + lo to 2^-22; the
//     volume is fp16 exactly, accumulation is fp32), padded per class to a multiple of 32 with zero rows (a zero
//     query contributes where(0 >= 0.25) = 0) and stored in the LDS image order, so a 32-query chunk (48 KB) is
//     a linear LDS-DMA copy into a 2-deep ring;
//   * per chunk and wave 48 x v_mfma_f32_32x32x16_f16 (rows = queries, columns = voxels), then the activation and
//     the in-lane sum over the 16 query rows a lane holds; one barrier per chunk;
//   * the class table travels as a kernel argument (up to 32 classes; more: device tables filled by copies);
//   * round 3: the streaming unit stays one 48 KB image of [32 queries][384 features] (hi + lo), but a wave's 192 operand
//     registers hold either TWO 32-voxel blocks of 384 features (workgroup = 512 voxels: every query fragment read from LDS
//     feeds four MFMAs instead of two, and every staged chunk serves twice the voxels -- the 5 x 1024-query preset was bound
//     by exactly these two streams) or ONE block of 768 features (ViT-B/8 volumes: a 32-query chunk is two units, the
//     accumulators run over both before the activation).
#include "vittf_common.h"

#include <stdlib.h>
#include <vector>

namespace {

constexpr int SM_F = 384, SM_KS = 24, SM_THREADS = 512, SM_VOX = 256;
constexpr int SM_PART = 6 * 4096;       // one [32 queries][384] fp16 image: six [32][64] sub-images
constexpr int SM_CHUNK = 2 * SM_PART;   // hi + lo: 48 KB

constexpr int SM_MIN_A = 8;           // fewer annotations: the VALU kernels of similarity.hip (VITTF_SIM_MFMA_MIN overrides)
constexpr int SM_MAXC = 32;
struct SmClasses { int n; int start[SM_MAXC + 1]; };   // n = 0: the tables are in device memory instead

// padded query p (class-major, each class padded to a multiple of 32) <- source row (-1: zero row): from the class table
// in the argument, or from src_row[p] when there are more than SM_MAXC classes
__global__ __launch_bounds__(256) void sim_mfma_prep(const float* __restrict__ qf, SmClasses cl, const int* __restrict__ src_row,
                                                     int padded, char* __restrict__ img, int ku) {
  const int e = blockIdx.x * 256 + threadIdx.x;          // one 16-byte chunk of one padded row (48 ku chunks per row)
  if (e >= padded * 48 * ku) return;
  const int p = e / (48 * ku), cw = e - 48 * ku * p;     // cw: chunk inside the whole row
  const int u = cw / 48, c = cw - 48 * u;                // unit (384-feature half of a 768-wide row), chunk inside the unit
  const int F = SM_F * ku;
  int src = -1;
  if (cl.n > 0) {
    int p0 = 0;
    for (int k = 0; k < cl.n; ++k) {
      const int nk = cl.start[k + 1] - cl.start[k], pk = (nk + 31) & ~31;
      if (p >= p0 && p < p0 + pk) src = p - p0 < nk ? cl.start[k] + (p - p0) : -1;
      p0 += pk;
    }
  } else {
    src = src_row[p];
  }
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float q0 = 0.f, q1 = 0.f;
    if (src >= 0) { q0 = qf[(int64_t)src * F + 8 * cw + 2 * j]; q1 = qf[(int64_t)src * F + 8 * cw + 2 * j + 1]; }
    const unsigned short h0 = f32_to_f16bits(q0), h1 = f32_to_f16bits(q1);
    const unsigned short l0 = f32_to_f16bits(q0 - f16bits_to_f32(h0)), l1 = f32_to_f16bits(q1 - f16bits_to_f32(h1));
    hi[j] = (unsigned)h0 | ((unsigned)h1 << 16);
    lo[j] = (unsigned)l0 | ((unsigned)l1 << 16);
  }
  char* dst = img + ((int64_t)(p >> 5) * ku + u) * SM_CHUNK + (c >> 3) * 4096 + tile_off(p & 31, c & 7);
  *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  *reinterpret_cast<uint4*>(dst + SM_PART) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

__device__ __forceinline__ float sm_thresh_pow(float s) { return s >= 0.25f ? s * s * sqrt_cr_normal(s) : 0.f; }   // (vittf_common.h: same bits as sqrtf, a third of its instructions)

// KU = 384-feature units per query row (1: F = 384, 2: F = 768); VB = 32-voxel blocks per wave (workgroup = 256 VB voxels).
// KU x VB <= 2: the voxel operands fill 96 KU VB registers.
template <bool DMA, int KU, int VB>
__global__ __launch_bounds__(SM_THREADS) void sim_mfma_kernel(const unsigned short* __restrict__ feat, int64_t nvox,
                                                              const char* __restrict__ qimg, SmClasses cl,
                                                              const int* __restrict__ chunk_start,
                                                              const float* __restrict__ counts, int classes, int total,
                                                              const float* __restrict__ vnorm, float* __restrict__ sim,
                                                              unsigned* __restrict__ maxbits) {
  static_assert(KU * VB <= 2, "register budget");
  constexpr int F = SM_F * KU, KST = SM_KS * KU;             // features, 16-wide k steps per voxel
  constexpr int ROWS_PART = 96 / VB;                         // feature rows per staged part and 256-voxel half: 48 KB per part
  constexpr int PARTS = F / ROWS_PART, KS_PART = ROWS_PART / 16;
  __shared__ __attribute__((aligned(16))) char ring[2 * SM_CHUNK];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int64_t v0 = (int64_t)blockIdx.x * (SM_VOX * VB);
  int64_t v[VB];
  bool valid[VB];
#pragma unroll
  for (int b = 0; b < VB; ++b) { v[b] = v0 + b * SM_VOX + wave * 32 + l31; valid[b] = v[b] < nvox; }
  const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(ring);

  // the wave's VB x 32 voxels x F features as B operands: xf[b][s] = features 16 s + 8 h .. + 7 of voxel v[b]
  s16x8_t xf[VB][KST];
  if constexpr (DMA) {
    // Part p = feature rows ROWS_PART p .. of the workgroup's VB halves of 256 voxels, 512 B per row and half, the halves
    // one behind the other in ring half p & 1.  One DMA instruction of a wave = two rows (lanes 0-31 / 32-63): LDS position
    // (lane & 31) of row r holds the row's 16-byte chunk (lane & 31) ^ 4 (r & 3) -- swizzled on the source side, the
    // destination of an LDS-DMA is lane-linear -- so that the four rows of a transposing read sit on four different 64-byte
    // bank groups.  A chunk beyond the end of the row (partial last workgroup) reads the row's last chunk instead: its
    // voxels are never stored.
    const int64_t last_chunk = nvox - 8;
#define SM_STAGE_ROWS(P)                                                                                   \
    {                                                                                                      \
      _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                      \
        const int j_ = wave * 6 + i;                          /* 1 KB piece of the 48 KB part */           \
        const int hf_ = j_ / (ROWS_PART / 2), r_ = 2 * (j_ % (ROWS_PART / 2)) + h;                         \
        int64_t vs_ = v0 + hf_ * SM_VOX + 8 * (l31 ^ (4 * (r_ & 3)));                                      \
        vs_ = vs_ < last_chunk ? vs_ : last_chunk;                                                         \
        lds_dma16_flat(feat + (int64_t)((P) * ROWS_PART + r_) * nvox + vs_,                                \
                       ring_lds + ((P) & 1) * SM_CHUNK + j_ * 1024);                                       \
      }                                                                                                    \
    }
    // transposing read: 16-lane group g = lane >> 4 covers voxels 16 (g & 1) .. + 15 of the wave's 32 and k-group g >> 1;
    // lane 4 q + pp of the group addresses row q, voxels 4 pp .. 4 pp + 3 (8 bytes); it receives 4 features of ITS voxel
    const int grp = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int tr_chunk = (4 * (wave ^ qq)) | (2 * (grp & 1)) | (pp >> 1);       // swizzled 16-byte chunk of row (.. + qq)
    const int tr_off = (8 * (grp >> 1) + qq) * 512 + tr_chunk * 16 + 8 * (pp & 1);
    SM_STAGE_ROWS(0)
    SM_STAGE_ROWS(1)
#pragma unroll
    for (int p = 0; p < PARTS; ++p) {
      if (p + 1 < PARTS) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // part p landed (6 pieces of p + 1 may be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
#pragma unroll
      for (int b = 0; b < VB; ++b) {
        const char* buf = ring + (p & 1) * SM_CHUNK + b * (ROWS_PART * 512) + tr_off;
#pragma unroll
        for (int s = 0; s < KS_PART; ++s) {
          const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(buf + (16 * s) * 512));
          const s16x4_t c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(buf + (16 * s + 4) * 512));
          s16x8_t t;
          t[0] = a[0]; t[1] = a[1]; t[2] = a[2]; t[3] = a[3]; t[4] = c[0]; t[5] = c[1]; t[6] = c[2]; t[7] = c[3];
          xf[b][KS_PART * p + s] = t;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                      // everybody has read ring half p & 1
      if (p + 2 < PARTS) SM_STAGE_ROWS(p + 2)
    }
#undef SM_STAGE_ROWS
  } else {
#pragma unroll
    for (int b = 0; b < VB; ++b)
#pragma unroll
      for (int s = 0; s < KST; ++s) {
        s16x8_t t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = valid[b] ? (short)feat[(int64_t)(16 * s + 8 * h + j) * nvox + v[b]] : (short)0;
        xf[b][s] = t;
      }
  }
  float nv[VB];
#pragma unroll
  for (int b = 0; b < VB; ++b) nv[b] = (vnorm && valid[b]) ? vnorm[v[b]] : 1.f;

  // the query stream: units of 48 KB ([32 queries][384 features], hi | lo), KU per 32-query chunk
  const int units = total * KU;
  const i32x4_t rsrc = lds_dma_rsrc(qimg, (unsigned)((int64_t)units * SM_CHUNK < 0x7fffffff ? (int64_t)units * SM_CHUNK : 0x7fffffff));
  const int voff = lane * 16;
  // unit g -> ring[g & 1]: 48 pieces of 1 KB, 6 per wave
#define SM_STAGE(G)                                                                                     \
  {                                                                                                     \
    const unsigned dst_ = ring_lds + ((G) & 1) * SM_CHUNK + wave * 6144;                                \
    const int src_ = (G) * SM_CHUNK + wave * 6144;                                                      \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) lds_dma16(rsrc, dst_ + i * 1024, voff, src_ + i * 1024); \
  }
  const int aoff0 = tile_off(l31, h);
  if (units > 0) SM_STAGE(0)
  int g = 0;                              // chunk
  int g_end = 0;
  for (int c = 0; c < classes; ++c) {
    float csum[VB];
#pragma unroll
    for (int b = 0; b < VB; ++b) csum[b] = 0.f;
    const int n_c = cl.n > 0 ? cl.start[c + 1] - cl.start[c] : 0;
    g_end = cl.n > 0 ? g_end + ((n_c + 31) >> 5) : chunk_start[c + 1];
    const float count = cl.n > 0 ? (float)n_c : counts[c];
    for (; g < g_end; ++g) {
      f32x16_t acc[VB];
#pragma unroll
      for (int b = 0; b < VB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int gu = g * KU + u;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // unit gu landed; ring[(gu + 1) & 1] is free
        if (gu + 1 < units) SM_STAGE(gu + 1)
        const char* buf = ring + (gu & 1) * SM_CHUNK;
#pragma unroll
        for (int s = 0; s < SM_KS; ++s) {
          const int off = (s >> 2) * 4096 + (aoff0 ^ (32 * (s & 3)));
          const s16x8_t qh = *reinterpret_cast<const s16x8_t*>(buf + off);
          const s16x8_t ql = *reinterpret_cast<const s16x8_t*>(buf + SM_PART + off);
#pragma unroll
          for (int b = 0; b < VB; ++b) acc[b] = mfma32<VITTF_FP16>(qh, xf[b][SM_KS * u + s], acc[b]);
#pragma unroll
          for (int b = 0; b < VB; ++b) acc[b] = mfma32<VITTF_FP16>(ql, xf[b][SM_KS * u + s], acc[b]);
        }
      }
#pragma unroll
      for (int b = 0; b < VB; ++b) {
        float part0 = 0.f, part1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          part0 += sm_thresh_pow(vnorm ? acc[b][r] / nv[b] : acc[b][r]);
          part1 += sm_thresh_pow(vnorm ? acc[b][r + 1] / nv[b] : acc[b][r + 1]);
        }
        csum[b] += part0 + part1;
      }
    }
    // the two lane halves hold the other 16 query rows of every chunk
    float m = 0.f;
#pragma unroll
    for (int b = 0; b < VB; ++b) {
      const unsigned cb = __float_as_uint(csum[b]);
      const auto sw = __builtin_amdgcn_permlane32_swap(cb, cb, false, false);
      const float mean = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) / count;
      if (valid[b] && h == 0) sim[(int64_t)c * nvox + v[b]] = mean;
      m = fmaxf(m, valid[b] ? mean : 0.f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) atomic_max_nonneg(maxbits + c, m);
  }
#undef SM_STAGE
}

// The interactive query: ONE class of at most 32 annotations = one query chunk.  Persistent workgroups (one per CU) walk
// the 256-voxel tiles.  The queries are split into their fp16 hi + lo images by the workgroup itself (48 KB of LDS, once per
// workgroup: no preparation launch), and the volume stream never stops: parts of 64 feature rows (32 KB) go round a
// 3-deep ring with ONE barrier per part -- the barrier that publishes part p also says that everybody has finished reading
// part p - 1, whose buffer takes part p + 2 -- so two parts (64 KB per CU) are in flight while a third is read, across
// tile boundaries as well: a tile's 48 MFMAs + activation run under the first parts of the workgroup's next tile.
constexpr int SM_FEW_ROWS = 64, SM_FEW_PART = SM_FEW_ROWS * 512, SM_FEW_PARTS = SM_F / SM_FEW_ROWS;   // 32 KB, 6 per tile
static_assert(3 * SM_FEW_PART == 2 * SM_CHUNK && SM_FEW_PARTS % 3 == 0, "three parts fill the ring; a tile is a whole number of ring turns");

__global__ __launch_bounds__(SM_THREADS) void sim_mfma_few_kernel(const unsigned short* __restrict__ feat, int64_t nvox,
                                                                  const float* __restrict__ qf, int n_q, int ntiles,
                                                                  const float* __restrict__ vnorm, float* __restrict__ sim,
                                                                  unsigned* __restrict__ maxbits) {
  __shared__ __attribute__((aligned(16))) char ring[2 * SM_CHUNK];
  __shared__ __attribute__((aligned(16))) char qbuf[SM_CHUNK];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(ring);
  const int64_t last_chunk = nvox - 8;
  // (layout of a part, source-side swizzle and the transposing read: sim_mfma_kernel above; FEW_PIECES DMA pieces per wave and
  //  part: the counted wait behind the query loads below is derived from the same constant)
  constexpr int FEW_PIECES = 4;
  static_assert(FEW_PIECES * 8 * 1024 == SM_FEW_PART, "a part = 8 waves x FEW_PIECES pieces of 1 KB");
#define SM_STAGE_ROWS(T, P)                                                                                \
  {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < FEW_PIECES; ++i) {                                                     \
      const int r_ = 2 * (wave * FEW_PIECES + i) + h;                                                      \
      int64_t vs_ = (int64_t)(T) * SM_VOX + 8 * (l31 ^ (4 * (r_ & 3)));                                    \
      vs_ = vs_ < last_chunk ? vs_ : last_chunk;                                                           \
      lds_dma16_flat(feat + (int64_t)((P) * SM_FEW_ROWS + r_) * nvox + vs_,                                \
                     ring_lds + ((P) % 3) * SM_FEW_PART + (wave * FEW_PIECES + i) * 1024);                       \
    }                                                                                                      \
  }
  int t = blockIdx.x;
  const int stride = gridDim.x;
  if (t >= ntiles) return;      // (never: the grid is min(ntiles, CUs); what follows relies on every workgroup having a first tile)
  // The query rows are requested AHEAD of the first volume parts, by loads hipcc does not see (round 4): its own wait for
  // them would be vmcnt(0), i.e. it would also wait for the 8 LDS-DMA pieces of parts 0 and 1 -- the memory counter retires
  // in order -- and the query images would only be built once 64 KB of volume had come in from HBM.  Issued first and waited
  // for with a counted vmcnt(8), they arrive and are converted while the parts are still on their way.  A thread owns three
  // of the 32 x 48 chunks of 8 features; rows >= n_q read row n_q - 1 and are zeroed.
  f32x4_t qa[3], qb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = tid + i * SM_THREADS;
    const int p = e / 48, c = e - 48 * p;
    const float* src = qf + (int64_t)(p < n_q ? p : n_q - 1) * SM_F + 8 * c;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(qa[i]), "=&v"(qb[i]) : "v"(src) : "memory");
  }
  SM_STAGE_ROWS(t, 0) SM_STAGE_ROWS(t, 1)
  // (two parts' pieces are younger than the query loads; tests/test_host_cpu.py checks on the disassembly that nothing touches
  //  the loaded registers between the loads and this wait)
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(qa[0]), "+v"(qb[0]), "+v"(qa[1]), "+v"(qb[1]), "+v"(qa[2]), "+v"(qb[2]) : "n"(2 * FEW_PIECES) : "memory");
  // the query images (sim_mfma_prep's arithmetic): 32 rows x 48 chunks of 8 features; rows >= n_q are zero
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = tid + i * SM_THREADS;
    const int p = e / 48, c = e - 48 * p;
    const float qv[8] = {qa[i][0], qa[i][1], qa[i][2], qa[i][3], qb[i][0], qb[i][1], qb[i][2], qb[i][3]};
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float q0 = p < n_q ? qv[2 * j] : 0.f, q1 = p < n_q ? qv[2 * j + 1] : 0.f;
      const unsigned short h0 = f32_to_f16bits(q0), h1 = f32_to_f16bits(q1);
      const unsigned short l0 = f32_to_f16bits(q0 - f16bits_to_f32(h0)), l1 = f32_to_f16bits(q1 - f16bits_to_f32(h1));
      hi[j] = (unsigned)h0 | ((unsigned)h1 << 16);
      lo[j] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    char* dst = qbuf + (c >> 3) * 4096 + tile_off(p, c & 7);
    *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(dst + SM_PART) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
  const int grp = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int tr_chunk = (4 * (wave ^ qq)) | (2 * (grp & 1)) | (pp >> 1);
  const int tr_off = (8 * (grp >> 1) + qq) * 512 + tr_chunk * 16 + 8 * (pp & 1);
  const int aoff0 = tile_off(l31, h);
  const float count = (float)n_q;
  float wmax = 0.f;
  // The activation of a tile (16 x the 2.5-th power per lane: VALU only) is spread over the part loop of the workgroup's
  // NEXT tile, three values behind every part's fragment reads -- between two tiles' streams it cost 19 of 59 us.
  f32x16_t pend;                      // the previous tile's dot products
  int64_t pend_v = -1;                // its voxel (-1: nothing pending)
  float pend_nv = 1.f, part0 = 0.f, part1 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) pend[r] = 0.f;
  auto activate = [&](int r) {        // registers 4 g .. 4 g + 3 are query rows 8 g .. 8 g + 7; rows >= n_q are zero queries: +0
    if (8 * (r >> 2) < n_q) {
      const float a = sm_thresh_pow(vnorm ? pend[r] / pend_nv : pend[r]);
      if (r & 1) part1 += a; else part0 += a;
    }
  };
  auto finish = [&]() {
    const unsigned cb = __float_as_uint(part0 + part1);
    const auto sw = __builtin_amdgcn_permlane32_swap(cb, cb, false, false);   // the other 16 query rows sit in the other lane half
    const float mean = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) / count;
    const bool valid = pend_v < nvox;
    if (valid && h == 0) sim[pend_v] = mean;
    wmax = fmaxf(wmax, valid ? mean : 0.f);
    part0 = 0.f; part1 = 0.f;
  };
  for (; t < ntiles; t += stride) {
    const bool more = t + stride < ntiles;
    // the tile's dot products grow part by part (8 MFMAs behind every part's fragment reads): with all 48 behind the last
    // part nothing new was requested for the length of that phase, once per tile -- 16 % of the tile time
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int p = 0; p < SM_FEW_PARTS; ++p) {
      // part p landed: of everything issued after it only the 4 pieces of part p + 1 (and, conservatively counted, an older
      // store of the previous tile) may still be in flight -- the memory counter retires in order.  The barrier also says
      // that every wave has the fragments of part p - 1 in its registers (and, the first time, that the query images are
      // written): its buffer is refilled right behind the barrier.
      if (p == SM_FEW_PARTS - 1 && !more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (p + 2 < SM_FEW_PARTS) SM_STAGE_ROWS(t, p + 2)
      else if (more) SM_STAGE_ROWS(t + stride, p + 2 - SM_FEW_PARTS)
      const char* buf = ring + (p % 3) * SM_FEW_PART + tr_off;
      s16x8_t xf[SM_FEW_ROWS / 16];
#pragma unroll
      for (int s = 0; s < SM_FEW_ROWS / 16; ++s) {
        const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(buf + (16 * s) * 512));
        const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(buf + (16 * s + 4) * 512));
        s16x8_t x;
        x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
        xf[s] = x;
      }
#pragma unroll
      for (int s4 = 0; s4 < SM_FEW_ROWS / 16; ++s4) {
        const int s = (SM_FEW_ROWS / 16) * p + s4;
        const int off = (s >> 2) * 4096 + (aoff0 ^ (32 * (s & 3)));
        const s16x8_t qh = *reinterpret_cast<const s16x8_t*>(qbuf + off);
        const s16x8_t ql = *reinterpret_cast<const s16x8_t*>(qbuf + SM_PART + off);
        acc = mfma32<VITTF_FP16>(qh, xf[s4], acc);
        acc = mfma32<VITTF_FP16>(ql, xf[s4], acc);
      }
      if (pend_v >= 0) {              // (wave-uniform) the previous tile's values 3 p .. 3 p + 2; the last part takes value 15
#pragma unroll
        for (int r = 3 * p; r < (p == SM_FEW_PARTS - 1 ? 16 : 3 * p + 3); ++r) activate(r);
      }
    }
    if (pend_v >= 0) finish();
    const int64_t v = (int64_t)t * SM_VOX + wave * 32 + l31;
    pend_nv = (vnorm && v < nvox) ? vnorm[v] : 1.f;
    pend = acc;
    pend_v = v;
  }
  if (pend_v >= 0) {                  // the workgroup's last tile
#pragma unroll
    for (int r = 0; r < 16; ++r) activate(r);
    finish();
  }
#undef SM_STAGE_ROWS
  // ONE atomic per workgroup: the persistent workgroups all finish together, and 2048 wave-level atomics on one address
  // took 18 us of a 59 us launch (the L2 serialises them at ~9 ns each)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
  float* wm = reinterpret_cast<float*>(qbuf);          // (the query images are dead: every wave is past its last MFMA ...
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   //  ... once it is past this barrier)
  if (lane == 0) wm[wave] = wmax;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (tid == 0) {
    float m = wm[0];
#pragma unroll
    for (int w = 1; w < SM_THREADS / 64; ++w) m = fmaxf(m, wm[w]);
    atomic_max_nonneg(maxbits, m);
  }
}

}  // namespace

size_t vittf_sim_mfma_workspace_bytes(int32_t classes, int32_t annotations) {
  // worst case: every class padded by 31 rows, 768-wide rows (two 48 KB units per chunk); + the row map, chunk table and
  // class counts
  const size_t chunks = ((size_t)annotations + 31 * (size_t)classes + 31) / 32;
  return chunks * 2 * SM_CHUNK + ((chunks * 32 * 4 + 255) & ~(size_t)255) + (((size_t)classes + 1) * 8 + 255 & ~(size_t)255) + 256;
}

// does the matrix-core path take this query set?  (asked before the profiler scope is opened: an empty scope would count
// as a launch of the class)
static bool sm_rows_aligned(const void* feat, int64_t nvox) { return nvox % 8 == 0 && nvox >= 8 && ((uintptr_t)feat & 15) == 0; }

bool vittf_sim_mfma_applies(int32_t f, int32_t classes, int32_t total_a, const void* ws, size_t ws_bytes, const void* feat,
                            int64_t nvox) {
  const char* e = getenv("VITTF_SIM_MFMA_MIN");   // (read per call: the tests switch it)
  const int min_a = e ? atoi(e) : SM_MIN_A;
  // 768-wide rows need the whole-row LDS-DMA (16-byte aligned feature rows): their strided-load form has no registers left
  if (!(f == SM_F || (f == 2 * SM_F && sm_rows_aligned(feat, nvox)))) return false;
  return total_a >= min_a && ws && ws_bytes >= vittf_sim_mfma_workspace_bytes(classes, total_a);
}

// fp32 class maps [classes][nvox] + per-class maxima; returns 1 when this path does not apply
int vittf_sim_mfma_maps(const unsigned short* feat, int32_t f, int64_t nvox, const float* qf, const int32_t* class_start_host,
                        int32_t classes, const float* voxel_norm, float* sim, unsigned* maxbits, void* ws, size_t ws_bytes,
                        hipStream_t st) {
  const int total_a = class_start_host[classes];
  if (!vittf_sim_mfma_applies(f, classes, total_a, ws, ws_bytes, feat, nvox)) return 1;
  SmClasses cl;
  cl.n = classes <= SM_MAXC ? classes : 0;
  for (int c = 0; c <= SM_MAXC; ++c) cl.start[c] = c <= classes && cl.n ? class_start_host[c] : 0;
  size_t chunks = 0;
  for (int c = 0; c < classes; ++c) chunks += (size_t)(class_start_host[c + 1] - class_start_host[c] + 31) / 32;
  const int padded = (int)(chunks * 32);
  const int ku = f / SM_F;
  char* w = (char*)ws;
  char* img = w;
  int* src_d = (int*)(w + chunks * ku * SM_CHUNK);
  int* chunk_d = (int*)((char*)src_d + (((size_t)padded * 4 + 255) & ~(size_t)255));
  float* counts_d = (float*)(chunk_d + classes + 1);
  std::vector<int> src_row, chunk_start;
  std::vector<float> counts;
  if (!cl.n) {   // more classes than the argument holds: the tables go through device memory
    chunk_start.assign(classes + 1, 0);
    counts.resize(classes);
    for (int c = 0; c < classes; ++c) {
      const int n = class_start_host[c + 1] - class_start_host[c];
      counts[c] = (float)n;
      for (int i = 0; i < ((n + 31) & ~31); ++i) src_row.push_back(i < n ? class_start_host[c] + i : -1);
      chunk_start[c + 1] = (int)(src_row.size() / 32);
    }
    if (hipMemcpyAsync(src_d, src_row.data(), (size_t)padded * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(chunk_d, chunk_start.data(), ((size_t)classes + 1) * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(counts_d, counts.data(), (size_t)classes * 4, hipMemcpyHostToDevice, st) != hipSuccess)
      return VITTF_ERR_LAUNCH;
  }
  const unsigned blocks = (unsigned)((nvox + SM_VOX - 1) / SM_VOX);
  // whole-row LDS-DMA needs 16-byte aligned rows; other volumes take the strided loads
  const bool dma = sm_rows_aligned(feat, nvox);
  const int cus = vittf_current_cus();        // of the CURRENT device, per call (the rank's device, not device 0)
  if (cus <= 0) return VITTF_ERR_NO_DEVICE;
  if (dma && ku == 1 && cl.n == 1 && chunks == 1) {   // the interactive query: one class, one chunk; the kernel prepares the queries itself
    vittf_note_kernel(VITTF_KERNEL_SIMILARITY, "sim_mfma_few_kernel");
    hipLaunchKernelGGL(sim_mfma_few_kernel, dim3(blocks < (unsigned)cus ? blocks : (unsigned)cus), dim3(SM_THREADS), 0, st, feat, nvox,
                       qf + (size_t)class_start_host[0] * SM_F, total_a, (int)blocks, voxel_norm, sim, maxbits);
    return vittf_check_launch();
  }
  hipLaunchKernelGGL(sim_mfma_prep, dim3((padded * 48 * ku + 255) / 256), dim3(256), 0, st, qf, cl, src_d, padded, img, ku);
  // two voxel blocks per wave (512-voxel workgroups) when that still fills the chip; VITTF_SIM_MFMA_VB=1 forces one (read per
  // call: the tests run both)
  const char* evb = getenv("VITTF_SIM_MFMA_VB");
  const int vb = (ku == 2 || !dma) ? 1 : (evb ? (atoi(evb) == 1 ? 1 : 2) : ((nvox + 2 * SM_VOX - 1) / (2 * SM_VOX) >= cus ? 2 : 1));
  const unsigned wgs = (unsigned)((nvox + (int64_t)SM_VOX * vb - 1) / ((int64_t)SM_VOX * vb));
#define SM_LAUNCH(DMAV, KUV, VBV)                                                                                     \
  hipLaunchKernelGGL((sim_mfma_kernel<DMAV, KUV, VBV>), dim3(wgs), dim3(SM_THREADS), 0, st, feat, nvox, img, cl, chunk_d, counts_d, \
                     classes, (int)chunks, voxel_norm, sim, maxbits)
  if (ku == 2) {
    vittf_note_kernel(VITTF_KERNEL_SIMILARITY, "sim_mfma_kernel<F 768>");
    SM_LAUNCH(true, 2, 1);
  } else if (vb == 2) {
    vittf_note_kernel(VITTF_KERNEL_SIMILARITY, "sim_mfma_kernel<2 voxel blocks>");
    SM_LAUNCH(true, 1, 2);
  } else {
    vittf_note_kernel(VITTF_KERNEL_SIMILARITY, "sim_mfma_kernel");
    if (dma) SM_LAUNCH(true, 1, 1); else SM_LAUNCH(false, 1, 1);
  }
#undef SM_LAUNCH
  if (!cl.n && hipStreamSynchronize(st) != hipSuccess) return VITTF_ERR_LAUNCH;   // the host vectors above must outlive the copies
  return vittf_check_launch();
}
