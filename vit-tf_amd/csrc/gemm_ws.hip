// Weight-stationary MFMA GEMM for the wide K = 384 linears of ViT-S (attn.qkv, mlp.fc1 + GELU):
//   out = epilogue(A[rows][384] . W[N][384]^T + bias[N]),  N a multiple of 384, 16-bit output
//
// Why a second GEMM: the tiled kernel (gemm.hip) moves 192 KB from L2 into LDS for every 128 x 128 output tile
// (both operand panels, K = 384), 1.8 - 2.4 GB per launch at 131 k rows, and its MFMA, epilogue (GELU is ~60
// VALU cycles per value and wave) and store phases only overlap across workgroups.  With K = 384 a 32-row slice
// of W is only 24 MFMA A-operands = 96 VGPRs, so here the weights never pass through LDS at all:
//
//   * workgroup = 12 waves (768 threads, 3 per SIMD, one workgroup per CU, persistent); wave w keeps
//     W[n0 + 32 w .. +32][0..384) in registers, so the workgroup owns a 384-column panel of the output and only
//     the ACTIVATIONS stream: 32-row tiles (24 KB, whole K) arrive by LDS-DMA into a 2-deep ring.
//     L2->LDS traffic per launch = rows x 768 B x (N / 384): 0.3 - 0.4 GB instead of 1.8 - 2.4 GB.
//   * software pipeline across tiles: while the 24 MFMAs of tile t issue (C^T = W . A^T: W fragment = A operand,
//     activations = B operand, one ds_read_b128 each), the wave runs the activation function of tile t - 1, whose
//     16 fp32 values (bias and q scale already applied) wait in registers; then tile t - 1 is packed into a padded
//     LDS tile and leaves as whole 768-byte row segments through bounds-checked buffer stores that stay in flight
//     under tile t + 1 (vmcnt retires in issue order, so the counted wait at the top covers only the DMA).
//   * 3 waves per SIMD share the 512 registers: 168 each, 96 of them weights, so the issue order is pinned per
//     K step (sched_barrier) and the LDS read runs one step ahead only.
#include "vittf_common.h"

#include <stdlib.h>

namespace {

// __syncthreads() is fence + barrier: hipcc drains vmcnt(0) in front of it, i.e. every store of the previous tile
// and the DMA of the next one -- exactly what this kernel keeps in flight.  Workgroup-local data only moves through
// LDS here, so: wait for this wave's LDS operations, then the bare barrier.
#define WS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

constexpr int WK = 384, WKS = 24, WNT = 384, WBM = 32, WTHREADS = 768;
constexpr int WSUB = WBM * 64 * 2;               // one [32 rows][64 k] sub-image: 4 KB
constexpr int WABYTES = 6 * WSUB;                // activation tile, whole K: 24 KB
constexpr int WCS = WNT * 2 + 16;                // staging row stride (payload 768 B)
constexpr int WCBYTES = WBM * WCS;

// Issue the LDS-DMA of one [32][384] activation tile (six [32][64] sub-images of 256 16-byte chunks) from the six
// loader waves, 4 pieces per lane: waves 0..3 take chunk (tid & 255) of sub-images 0..3, waves 4..5 take chunks
// c and c + 128 (c = tid - 256) of sub-images 4 and 5.  A sub-image further = 128 B further in the row, so each lane
// needs two source offsets (chunk c, chunk c + 128) and the rest is scalar.  The tile is addressed through a buffer
// descriptor that starts at its first row and ends with the matrix: rows past the end read as zeros (and are never
// stored).  (Not a template: hipcc's host pass drops function templates whose bodies name the descriptor type.)
__device__ __forceinline__ void ws_stage(const unsigned short* __restrict__ A, int64_t m0, int64_t rows, int wave,
                                         int voff_a, int voff_b, unsigned tile_lds) {
  const int64_t left = (rows - m0) * (WK * 2);
  const i32x4_t rsrc = lds_dma_rsrc(A + m0 * WK, (unsigned)(left < WABYTES ? left : WABYTES));
  if (wave < 4) {
    const unsigned dst = tile_lds + wave * 1024;    // + lane * 16 by the hardware
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_dma16(rsrc, dst + i * WSUB, voff_a, i * 128);
  } else {
    const unsigned dst = tile_lds + 4 * WSUB + (wave - 4) * 1024;
    lds_dma16(rsrc, dst, voff_a, 4 * 128);
    lds_dma16(rsrc, dst + 2048, voff_b, 4 * 128);
    lds_dma16(rsrc, dst + WSUB, voff_a, 5 * 128);
    lds_dma16(rsrc, dst + WSUB + 2048, voff_b, 5 * 128);
  }
}

template <int DT, int EPI>
__global__ __launch_bounds__(WTHREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_ws_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, const float* __restrict__ bias,
    unsigned short* __restrict__ out, int64_t rows, int n, int n_mt, int total) {
  // Separate LDS objects on purpose: hipcc orders ds_write / ds_read against outstanding LDS-DMA with vmcnt(0)
  // unless alias scopes (one per LDS variable) prove that they touch different memory.
  __shared__ __attribute__((aligned(16))) char smem[2 * WABYTES];    // the two activation tiles (LDS-DMA targets)
  __shared__ __attribute__((aligned(16))) char cbuf[WCBYTES];        // staging tile
  __shared__ __attribute__((aligned(16))) float bias_s[WNT];         // the panel's bias
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
  const int h = lane >> 5, l31 = lane & 31;
  const float* const bias_l = bias_s;

  const int G = gridDim.x;
  const int per = total / G, rem = total - per * G, bx = blockIdx.x;   // contiguous, balanced item ranges
  const int it0 = bx * per + (bx < rem ? bx : rem), it1 = it0 + per + (bx < rem ? 1 : 0);
  if (it0 >= it1) return;

  // per-lane read offset inside a [32][64] sub-image: row l31, chunk 2 j + h -> aoff0 ^ 32 j
  const int aoff0 = tile_off(l31, h);
  // Division of the vector-memory work (vmcnt retires in issue order, stores included, and a store takes
  // microseconds to retire under load): waves 0..5 issue all LDS-DMA and nothing else, so their counted wait at the
  // top of a tile covers DMA only; waves 6..11 issue all stores and never wait for them.
  const bool loader = wave < 6;
  int voff_a, voff_b;     // loader lanes: source byte offsets of chunk c and chunk c + 128 of a sub-image
  {
    int r, c;
    tile_pos(tid & 255, r, c);
    voff_a = r * (WK * 2) + c * 16;
    tile_pos((tid & 127) + 128, r, c);
    voff_b = r * (WK * 2) + c * 16;
  }
  const int ts = tid - 384;                          // storer thread index (waves 6..11)
  const int r48 = ts / 48, c48 = ts - 48 * r48;     // staging tile -> global: row 8 i + r48, 16-byte chunk c48
  const int st_lds = r48 * WCS + c48 * 16;           // (+ 8 i rows: immediates / scalar offsets)
  const int st_glb = r48 * n * 2 + c48 * 16;

  // the finished-but-not-stored tile: fp32 values with bias (and q scale) applied, and where it goes
  float prev[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) prev[i] = 0.f;
  int64_t prev_m0 = 0;
  int prev_n0 = -1;            // -1: nothing pending

  // pack `prev` into the staging tile ...
#define WS_PACK()                                                                                               \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                               \
    uint2 pk;                                                                                                   \
    pk.x = pack2_h16<DT>(prev[4 * g + 0], prev[4 * g + 1]);                                                     \
    pk.y = pack2_h16<DT>(prev[4 * g + 2], prev[4 * g + 3]);                                                     \
    *reinterpret_cast<uint2*>(cbuf + l31 * WCS + (32 * wave + 8 * g + 4 * h) * 2) = pk;                         \
  }
  // ... and (after a barrier) the storer waves write the tile (M0, N0) as whole row segments; rows past the end
  // fall outside the descriptor and are dropped by the hardware
#define WS_STORE(M0, N0)                                                                                        \
  if (!loader) {                                                                                                \
    unsigned short* o16 = out + (M0) * n + (N0);                                                                \
    const int64_t left = ((rows - (M0)) * n - (N0)) * 2;                                                        \
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(                                                       \
        o16, 0, (int)(left < (int64_t)WBM * n * 2 ? left : (int64_t)WBM * n * 2), 0x00020000);                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
      const u32x4_t d = *reinterpret_cast<const u32x4_t*>(cbuf + st_lds + i * (8 * WCS));                       \
      __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, st_glb, i * 16 * n, 0);                                  \
    }                                                                                                           \
  }

  const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(smem);   // LDS byte address of the activation ring
  if (loader) ws_stage(A, (int64_t)(it0 % n_mt) * WBM, rows, wave, voff_a, voff_b, ring_lds);

  for (int it = it0; it < it1;) {
  // ---- one 384-column panel: this wave's 32 weight rows, whole K, stay in registers for all its row tiles ----
  const int nt = it / n_mt;
  const int it_end = (nt + 1) * n_mt < it1 ? (nt + 1) * n_mt : it1;
  const int n0 = nt * WNT;
  s16x8_t wf[WKS];
  {
    const unsigned short* wrow = W + (int64_t)(n0 + 32 * wave + l31) * WK + 8 * h;
#pragma unroll
    for (int s = 0; s < WKS; ++s) wf[s] = *reinterpret_cast<const s16x8_t*>(wrow + 16 * s);
    // the bias goes through LDS: a global load inside the loop would make hipcc drain vmcnt (stores + next DMA).
    // Its last readers (the previous panel's tiles) are behind a barrier; visible after the next barrier.
    if (tid < WNT / 4) reinterpret_cast<float4*>(bias_s)[tid] = reinterpret_cast<const float4*>(bias + n0)[tid];
  }
  // the q third carries the softmax scale and the exp -> exp2 base change: one rounding, like plain q
  const float qsc = (EPI == VITTF_EPI_BIAS_QKV && n0 + 32 * wave < n / 3) ? 0.125f * 1.44269504088896340736f : 1.0f;
  for (; it < it_end; ++it) {
    const int par = (it - it0) & 1;
    const char* abuf = smem + par * WABYTES;
    if (loader) {
      if (it + 1 < it1) {
        ws_stage(A, (int64_t)((it + 1) % n_mt) * WBM, rows, wave, voff_a, voff_b, ring_lds + (par ^ 1) * WABYTES);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but the 4 DMA pieces of the next tile
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    WS_BARRIER();   // the tile of this item is in LDS for everybody; the staging tile is free
    __builtin_amdgcn_sched_barrier(0);

    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // Issue order pinned per K step: LDS read of step s + 1, MFMA of step s, one activation of the previous tile.
    // Left alone, hipcc hoists the reads (24 fragments = 96 more registers than the 168 there are).
#define WS_READ(S) (*reinterpret_cast<const s16x8_t*>(abuf + ((S) >> 2) * WSUB + (aoff0 ^ (32 * ((S) & 3)))))
    s16x8_t f[2];
    f[0] = WS_READ(0);
#pragma unroll
    for (int s = 0; s < WKS; ++s) {
      if (s + 1 < WKS) f[(s + 1) & 1] = WS_READ(s + 1);
      acc = mfma32<DT>(wf[s], f[s & 1], acc);
      if constexpr (EPI == VITTF_EPI_BIAS_GELU) {
        if (s < 16) prev[s] = gelu_poly(prev[s]);   // (zeros / stale values while nothing is pending)
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // the read (one step ahead) first,
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // then the MFMA,
      __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);  // then the previous tile's activation in its shadow
      __builtin_amdgcn_sched_barrier(0);
    }
#undef WS_READ

    // tile t - 1 -> staging; accumulators of tile t -> prev (the lane owns activation row l31 and output columns
    // 32 wave + 8 g + 4 h + {0..3}); every read of the panel's bias happens before the barrier below
    const bool pending = prev_n0 >= 0;
    WS_PACK()   // unconditional (the staging tile is free): keeps the activation above inside the MFMA loop
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *reinterpret_cast<const float4*>(bias_l + 32 * wave + 8 * g + 4 * h);
      prev[4 * g + 0] = (acc[4 * g + 0] + bv.x) * qsc;
      prev[4 * g + 1] = (acc[4 * g + 1] + bv.y) * qsc;
      prev[4 * g + 2] = (acc[4 * g + 2] + bv.z) * qsc;
      prev[4 * g + 3] = (acc[4 * g + 3] + bv.w) * qsc;
    }
    WS_BARRIER();
    if (pending) WS_STORE(prev_m0, prev_n0)
    prev_m0 = (int64_t)(it - nt * n_mt) * WBM;
    prev_n0 = n0;
  }   // row tiles of the panel
  }   // panels

  // the last tile: its activation was not overlapped with anything
  if constexpr (EPI == VITTF_EPI_BIAS_GELU) {
#pragma unroll
    for (int s = 0; s < 16; ++s) prev[s] = gelu_poly(prev[s]);
  }
  WS_BARRIER();   // the previous tile's readers of the staging tile
  WS_PACK()
  WS_BARRIER();
  WS_STORE(prev_m0, prev_n0)
#undef WS_PACK
#undef WS_STORE
}

template <int DT>
int launch_ws(const void* a, const void* w, const float* bias, void* out, int64_t rows, int n, int epi, hipStream_t st) {
  static const int cus = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    return v > 0 ? v : 256;
  }();
  const int n_mt = (int)((rows + WBM - 1) / WBM);
  const int total = n_mt * (n / WNT);
  const int grid = total < cus ? total : cus;
  const unsigned short* A = (const unsigned short*)a;
  const unsigned short* Wp = (const unsigned short*)w;
#define VITTF_WS_CASE(E)                                                                                     \
  case E:                                                                                                    \
    hipLaunchKernelGGL((gemm_ws_kernel<DT, E>), dim3(grid), dim3(WTHREADS), 0, st, A, Wp, bias,              \
                       (unsigned short*)out, rows, n, n_mt, total);                                          \
    break;
  switch (epi) {
    VITTF_WS_CASE(VITTF_EPI_BIAS)
    VITTF_WS_CASE(VITTF_EPI_BIAS_GELU)
    VITTF_WS_CASE(VITTF_EPI_BIAS_QKV)
    default: return 1;
  }
#undef VITTF_WS_CASE
  return vittf_check_launch();
}

}  // namespace

// Called by vittf_gemm (gemm.hip) for the shapes this kernel covers; returns 1 (not handled) otherwise: the fp32
// residual epilogues (HBM-bound read-modify-write) and the K-feature epilogue stay on the tiled kernel.
int vittf_gemm_ws(const void* a, const void* w, const float* bias, void* out, int64_t rows, int32_t n, int32_t k,
                  int32_t epilogue, int32_t dtype, hipStream_t st) {
  if (k != WK || n % WNT != 0) return 1;
  if ((rows + WBM - 1) / WBM * (n / WNT) > (1 << 30)) return 1;
  if (dtype == VITTF_BF16) return launch_ws<VITTF_BF16>(a, w, bias, out, rows, n, epilogue, st);
  if (dtype == VITTF_FP16) return launch_ws<VITTF_FP16>(a, w, bias, out, rows, n, epilogue, st);
  return VITTF_ERR_INVALID_ARG;
}
