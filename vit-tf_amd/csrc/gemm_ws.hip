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

// LNF: the activations are LayerNorm(x) of an fp32 residual stream x[rows][384], computed here instead of by a separate
// kernel: waves 0..3 fetch their 8 rows of the next tile as raw fp32 (12 KB each, linear LDS-DMA) and, after the
// tile's stores have been handed to the storer waves, normalise them (8 lanes per row: mean, centred variance, affine;
// the arithmetic of layernorm.hip) straight into the 16-bit operand image of the ring.
template <int DT, int EPI, bool LNF>
__global__ __launch_bounds__(WTHREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_ws_kernel(
    const void* __restrict__ a_in, const unsigned short* __restrict__ W, const float* __restrict__ bias,
    unsigned short* __restrict__ out, int64_t rows, int n, int n_mt, int total, const float* __restrict__ ln_g,
    const float* __restrict__ ln_b, float ln_eps) {
  const unsigned short* A = reinterpret_cast<const unsigned short*>(a_in);
  const float* X = reinterpret_cast<const float*>(a_in);
  __shared__ __attribute__((aligned(16))) char raw[LNF ? WBM * WK * 4 : 16];    // LNF: the next tile's fp32 rows
  __shared__ __attribute__((aligned(16))) float gb_s[LNF ? 2 * WK : 4];         // LNF: gamma | beta
  // Separate LDS objects on purpose: hipcc orders ds_write / ds_read against outstanding LDS-DMA with vmcnt(0)
  // unless alias scopes (one per LDS variable) prove that they touch different memory.
  constexpr int RING = LNF ? 2 : 3;    // activation tiles in the ring: with one barrier per tile the DMA of a tile can only be
                                       // issued after the barrier behind the last reader of its slot; depth 3 gives it two tiles to land
  __shared__ __attribute__((aligned(16))) char smem[RING * WABYTES];    // the activation ring (LDS-DMA targets)
  __shared__ __attribute__((aligned(16))) char cbuf2[2 * WCBYTES];   // two staging tiles (packed one tile, stored the next)
  __shared__ __attribute__((aligned(16))) float bias_s2[2 * WNT];    // the panel's bias, by panel parity
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
  const int h = lane >> 5, l31 = lane & 31;

  // Work split: items = (panel, row tile), panel-major, contiguous balanced ranges.  (Giving the P panels of one range
  // of row tiles to P workgroups of one XCD, so that an activation tile is fetched from HBM once, measured slower:
  // qkv 0.129 -> 0.164 ms.)
  const int G = gridDim.x, bx = blockIdx.x;
  const int per = total / G, rem = total - per * G;
  const int it0 = bx * per + (bx < rem ? bx : rem), it1 = it0 + per + (bx < rem ? 1 : 0);
  if (it0 >= it1) return;

  // per-lane read offset inside a [32][64] sub-image: row l31, chunk 2 j + h -> aoff0 ^ 32 j
  const int aoff0 = tile_off(l31, h);
  // Division of the vector-memory work (vmcnt retires in issue order, stores included, and a store takes
  // microseconds to retire under load): waves 0..5 issue all LDS-DMA and nothing else, so their counted wait at the
  // top of a tile covers DMA only; waves 6..11 issue all stores and never wait for them.
  const bool loader = wave < (LNF ? 4 : 6);
  const bool storer = wave >= 6;
  int voff_a, voff_b;     // loader lanes: source byte offsets of chunk c and chunk c + 128 of a sub-image
  {
    int r, c;
    tile_pos(tid & 255, r, c);
    voff_a = r * (WK * 2) + c * 16;
    tile_pos((tid & 127) + 128, r, c);
    voff_b = r * (WK * 2) + c * 16;
  }

  // the finished-but-not-stored tile: fp32 values with bias (and q scale) applied, and where it goes
  float prev[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) prev[i] = 0.f;
  int64_t prev_m0 = 0;
  int prev_n0 = -1;            // -1: nothing pending
  int64_t st_m0 = 0;           // the tile packed into the staging slot last iteration, stored after the next barrier
  int st_n0 = -1;

  // pack `prev` into the staging tile ...
#define WS_PACK(SLOT)                                                                                           \
  {                                                                                                             \
    int ln_ = lane;                                                                                             \
    asm volatile("" : "+v"(ln_));   /* lane-derived offsets are recomputed per tile: as loop invariants they get spilled, */ \
    const int pk_off_ = (ln_ & 31) * WCS + (32 * wave + 4 * (ln_ >> 5)) * 2;   /* and a scratch reload drains vmcnt */ \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                             \
      uint2 pk;                                                                                                 \
      pk.x = pack2_h16<DT>(prev[4 * g + 0], prev[4 * g + 1]);                                                   \
      pk.y = pack2_h16<DT>(prev[4 * g + 2], prev[4 * g + 3]);                                                   \
      *reinterpret_cast<uint2*>(cbuf2 + (SLOT) * WCBYTES + pk_off_ + 16 * g) = pk;                              \
    }                                                                                                           \
  }
  // ... and (after a barrier) the storer waves write the tile (M0, N0) as whole row segments; rows past the end
  // fall outside the descriptor and are dropped by the hardware
#define WS_STORE(M0, N0, SLOT)                                                                                  \
  if (storer) {                                                                                                 \
    int ts_ = tid - 384;                              /* storer thread index (waves 6..11) */                   \
    asm volatile("" : "+v"(ts_));                     /* (recomputed per tile: see WS_LN_TRANSFORM) */          \
    const int r48 = ts_ / 48, c48 = ts_ - 48 * r48;  /* staging tile -> global: row 8 i + r48, 16-byte chunk c48 */ \
    const int st_lds = r48 * WCS + c48 * 16;          /* (+ 8 i rows: immediates / scalar offsets) */           \
    const int st_glb = r48 * n * 2 + c48 * 16;                                                                  \
    unsigned short* o16 = out + (M0) * n + (N0);                                                                \
    const int64_t left = ((rows - (M0)) * n - (N0)) * 2;                                                        \
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(                                                       \
        o16, 0, (int)(left < (int64_t)WBM * n * 2 ? left : (int64_t)WBM * n * 2), 0x00020000);                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
      const u32x4_t d = *reinterpret_cast<const u32x4_t*>(cbuf2 + (SLOT) * WCBYTES + st_lds + i * (8 * WCS));   \
      __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, st_glb, i * 16 * n, 0);                                  \
    }                                                                                                           \
  }

  const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(smem);   // LDS byte address of the activation ring
  const unsigned raw_lds = (unsigned)(size_t)LDS_PTR(raw);
  // LNF: raw rows of tile MT -> LDS (waves 0..3, 8 rows = 12 pieces of 1 KB each, rows past the end read as zeros) ...
#define WS_STAGE_RAW(MT)                                                                                        \
  {                                                                                                             \
    const int64_t m0_ = (int64_t)(MT) * WBM;                                                                    \
    const int64_t left_ = (rows - m0_) * (WK * 4);                                                              \
    const i32x4_t rs_ = lds_dma_rsrc(X + m0_ * WK, (unsigned)(left_ < WBM * WK * 4 ? left_ : WBM * WK * 4));    \
    _Pragma("unroll") for (int i = 0; i < 12; ++i)                                                              \
      lds_dma16(rs_, raw_lds + wave * 12288 + i * 1024, lane * 16, wave * 12288 + i * 1024);                    \
  }
  // ... and (once landed) this wave's 8 rows -> LayerNorm -> 16-bit operand image BUF of the ring.  Lane = (row
  // lane >> 3, segment lane & 7); a lane owns the 8-column chunks sg + 8 j, j = 0..5, i.e. chunk sg of sub-image j.
#define WS_LN_TRANSFORM(BUF)                                                                                    \
  {                                                                                                             \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
    int lane_ = lane;                                                                                           \
    asm volatile("" : "+v"(lane_));   /* recompute the addresses here: as loop invariants they would be spilled */ \
    const int r_ = 8 * wave + (lane_ >> 3), sg_ = lane_ & 7;                                                    \
    const char* rr_ = raw + r_ * (WK * 4) + sg_ * 32;                                                           \
    /* three passes over the lane's 48 values, each software-pipelined one chunk ahead (the LDS round trip would   */ \
    /* otherwise be exposed 18 times); no deeper: 112 registers are taken by the weights and the pending tile      */ \
    float4 ua_[2], ub_[2];                                                                                      \
    float s_ = 0.f;                                                                                             \
    ua_[0] = *reinterpret_cast<const float4*>(rr_); ub_[0] = *reinterpret_cast<const float4*>(rr_ + 16);        \
    _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                             \
      if (j + 1 < 6) { ua_[(j + 1) & 1] = *reinterpret_cast<const float4*>(rr_ + (j + 1) * 256);                \
                       ub_[(j + 1) & 1] = *reinterpret_cast<const float4*>(rr_ + (j + 1) * 256 + 16); }         \
      const float4 u0 = ua_[j & 1], u1 = ub_[j & 1];                                                            \
      s_ += ((u0.x + u0.y) + (u0.z + u0.w)) + ((u1.x + u1.y) + (u1.z + u1.w));                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                           \
    s_ += __shfl_xor(s_, 1); s_ += __shfl_xor(s_, 2); s_ += __shfl_xor(s_, 4);                                  \
    const float mean_ = s_ / (float)WK;                                                                         \
    float q_ = 0.f;                                                                                             \
    ua_[0] = *reinterpret_cast<const float4*>(rr_); ub_[0] = *reinterpret_cast<const float4*>(rr_ + 16);        \
    _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                             \
      if (j + 1 < 6) { ua_[(j + 1) & 1] = *reinterpret_cast<const float4*>(rr_ + (j + 1) * 256);                \
                       ub_[(j + 1) & 1] = *reinterpret_cast<const float4*>(rr_ + (j + 1) * 256 + 16); }         \
      const float4 u0 = ua_[j & 1], u1 = ub_[j & 1];                                                            \
      const float a0 = u0.x - mean_, a1 = u0.y - mean_, a2 = u0.z - mean_, a3 = u0.w - mean_;                   \
      const float a4 = u1.x - mean_, a5 = u1.y - mean_, a6 = u1.z - mean_, a7 = u1.w - mean_;                   \
      q_ += ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((a4 * a4 + a5 * a5) + (a6 * a6 + a7 * a7));          \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                           \
    q_ += __shfl_xor(q_, 1); q_ += __shfl_xor(q_, 2); q_ += __shfl_xor(q_, 4);                                  \
    const float rstd_ = 1.0f / sqrtf(q_ / (float)WK + ln_eps);                                                  \
    char* dst_ = smem + (BUF) * WABYTES + tile_off(r_, sg_);                                                    \
    const float* gp_ = gb_s + 8 * sg_;                                                                          \
    float4 ga_[2], gc_[2], ba_[2], bc_[2];                                                                      \
    ua_[0] = *reinterpret_cast<const float4*>(rr_); ub_[0] = *reinterpret_cast<const float4*>(rr_ + 16);        \
    ga_[0] = *reinterpret_cast<const float4*>(gp_); gc_[0] = *reinterpret_cast<const float4*>(gp_ + 4);         \
    ba_[0] = *reinterpret_cast<const float4*>(gp_ + WK); bc_[0] = *reinterpret_cast<const float4*>(gp_ + WK + 4); \
    _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                             \
      if (j + 1 < 6) {                                                                                          \
        ua_[(j + 1) & 1] = *reinterpret_cast<const float4*>(rr_ + (j + 1) * 256);                               \
        ub_[(j + 1) & 1] = *reinterpret_cast<const float4*>(rr_ + (j + 1) * 256 + 16);                          \
        ga_[(j + 1) & 1] = *reinterpret_cast<const float4*>(gp_ + 64 * (j + 1));                                \
        gc_[(j + 1) & 1] = *reinterpret_cast<const float4*>(gp_ + 64 * (j + 1) + 4);                            \
        ba_[(j + 1) & 1] = *reinterpret_cast<const float4*>(gp_ + WK + 64 * (j + 1));                           \
        bc_[(j + 1) & 1] = *reinterpret_cast<const float4*>(gp_ + WK + 64 * (j + 1) + 4);                       \
      }                                                                                                         \
      const float4 u0 = ua_[j & 1], u1 = ub_[j & 1], g0 = ga_[j & 1], g1 = gc_[j & 1], b0 = ba_[j & 1], b1 = bc_[j & 1]; \
      uint4 pk_;                                                                                                \
      pk_.x = pack2_h16<DT>((u0.x - mean_) * rstd_ * g0.x + b0.x, (u0.y - mean_) * rstd_ * g0.y + b0.y);        \
      pk_.y = pack2_h16<DT>((u0.z - mean_) * rstd_ * g0.z + b0.z, (u0.w - mean_) * rstd_ * g0.w + b0.w);        \
      pk_.z = pack2_h16<DT>((u1.x - mean_) * rstd_ * g1.x + b1.x, (u1.y - mean_) * rstd_ * g1.y + b1.y);        \
      pk_.w = pack2_h16<DT>((u1.z - mean_) * rstd_ * g1.z + b1.z, (u1.w - mean_) * rstd_ * g1.w + b1.w);        \
      *reinterpret_cast<uint4*>(dst_ + j * WSUB) = pk_;                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                           \
  }
  if constexpr (LNF) {
    for (int i = tid; i < 2 * WK; i += WTHREADS) gb_s[i] = i < WK ? ln_g[i] : ln_b[i - WK];
    if (loader) WS_STAGE_RAW(it0 % n_mt)
    WS_BARRIER();                       // gamma / beta visible
    if (loader) {
      WS_LN_TRANSFORM(0)
      if (it0 + 1 < it1) WS_STAGE_RAW((it0 + 1) % n_mt)      // (the transform's LDS reads are complete: it ends on writes
    }                                                          //  that consumed them, and the DMA is issued after those)
  } else {
    if (loader) {
      ws_stage(A, (int64_t)(it0 % n_mt) * WBM, rows, wave, voff_a, voff_b, ring_lds);
      if (it0 + 1 < it1) ws_stage(A, (int64_t)((it0 + 1) % n_mt) * WBM, rows, wave, voff_a, voff_b, ring_lds + WABYTES);
    }
  }

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
    // Two copies by panel parity: a wave may still hand over the previous panel's last tile; visible after the next barrier.
    if (tid < WNT / 4) reinterpret_cast<float4*>(bias_s2 + (nt & 1) * WNT)[tid] = reinterpret_cast<const float4*>(bias + n0)[tid];
  }
  // the q third carries the softmax scale and the exp -> exp2 base change: one rounding, like plain q
  const float qsc = (EPI == VITTF_EPI_BIAS_QKV && n0 + 32 * wave < n / 3) ? 0.125f * 1.44269504088896340736f : 1.0f;
  for (; it < it_end; ++it) {
    const int par = (it - it0) % RING;
    const char* abuf = smem + par * WABYTES;
    if constexpr (!LNF) {
      // this tile's DMA pieces (issued two tiles ago) have landed; the next tile's 4 pieces may still be in flight
      if (loader) {
        if (it + 1 < it1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    WS_BARRIER();   // the ONE barrier per tile: this tile's operands are in LDS for everybody, and so is the staging
                    // slot packed during the previous tile, which the storer waves now send off (the other waves go on)
    __builtin_amdgcn_sched_barrier(0);
    const int slot = (it - it0) & 1;
    if constexpr (!LNF) {
      // the DMA of tile it + 2 goes out only now: its ring slot was read by the MFMAs of tile it - 1, and with one
      // barrier per tile a slow wave is still in them until it reaches the barrier above
      if (loader && it + 2 < it1)
        ws_stage(A, (int64_t)((it + 2) % n_mt) * WBM, rows, wave, voff_a, voff_b, ring_lds + ((par + 2) % RING) * WABYTES);
    }
    if (st_n0 >= 0) WS_STORE(st_m0, st_n0, slot ^ 1)
    if constexpr (LNF) {
      // Loader waves first normalise the NEXT tile (its raw rows were requested a whole tile ago) into ring[par ^ 1] --
      // last read one tile ago, two barriers back -- and request the raw rows of the tile after it; the other two waves
      // of each SIMD run their MFMAs meanwhile, the loader's own MFMAs follow in their shadow.
      if (loader && it + 1 < it1) {
        WS_LN_TRANSFORM(par ^ 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the raw rows have been read: the buffer may be overwritten
        if (it + 2 < it1) WS_STAGE_RAW((it + 2) % n_mt)
      }
    }

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
    WS_PACK(slot)   // (also when nothing is pending: keeps the activation above inside the MFMA loop; never stored then)
    st_m0 = prev_m0;
    st_n0 = prev_n0;
    int lb_ = lane;
    asm volatile("" : "+v"(lb_));
    const float* bias_w = bias_s2 + (nt & 1) * WNT + 32 * wave + 4 * (lb_ >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *reinterpret_cast<const float4*>(bias_w + 8 * g);
      prev[4 * g + 0] = (acc[4 * g + 0] + bv.x) * qsc;
      prev[4 * g + 1] = (acc[4 * g + 1] + bv.y) * qsc;
      prev[4 * g + 2] = (acc[4 * g + 2] + bv.z) * qsc;
      prev[4 * g + 3] = (acc[4 * g + 3] + bv.w) * qsc;
    }
    prev_m0 = (int64_t)(it - nt * n_mt) * WBM;
    prev_n0 = n0;
  }   // row tiles of the panel
  }   // panels

  // drain: the tile packed in the last iteration, then the last tile itself (its activation overlapped with nothing)
  const int last_slot = (it1 - 1 - it0) & 1;
  WS_BARRIER();
  if (st_n0 >= 0) WS_STORE(st_m0, st_n0, last_slot)
  if constexpr (EPI == VITTF_EPI_BIAS_GELU) {
#pragma unroll
    for (int s = 0; s < 16; ++s) prev[s] = gelu_poly(prev[s]);
  }
  WS_PACK(last_slot ^ 1)
  WS_BARRIER();
  WS_STORE(prev_m0, prev_n0, last_slot ^ 1)
#undef WS_PACK
#undef WS_STORE
#undef WS_STAGE_RAW
#undef WS_LN_TRANSFORM
}

template <int DT, bool LNF>
int launch_ws(const void* a, const void* w, const float* bias, void* out, int64_t rows, int n, int epi, const float* ln_g,
              const float* ln_b, float ln_eps, hipStream_t st) {
  const int cus = vittf_current_cus();        // per call, of the current device
  if (cus <= 0) return VITTF_ERR_NO_DEVICE;
  const int n_mt = (int)((rows + WBM - 1) / WBM);
  const int total = n_mt * (n / WNT);
  const int grid = total < cus ? total : cus;
  const unsigned short* Wp = (const unsigned short*)w;
#define VITTF_WS_CASE(E)                                                                                     \
  case E:                                                                                                    \
    hipLaunchKernelGGL((gemm_ws_kernel<DT, E, LNF>), dim3(grid), dim3(WTHREADS), 0, st, a, Wp, bias,         \
                       (unsigned short*)out, rows, n, n_mt, total, ln_g, ln_b, ln_eps);                      \
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
  if (dtype == VITTF_BF16) return launch_ws<VITTF_BF16, false>(a, w, bias, out, rows, n, epilogue, nullptr, nullptr, 0.f, st);
  if (dtype == VITTF_FP16) return launch_ws<VITTF_FP16, false>(a, w, bias, out, rows, n, epilogue, nullptr, nullptr, 0.f, st);
  return VITTF_ERR_INVALID_ARG;
}

// LayerNorm fused into the activation loader (see LNF above): out = epilogue(LN(x) . W^T + bias), x fp32 [rows][384].
extern "C" int vittf_ln_gemm(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const void* w,
                             const float* bias, void* out, int64_t rows, int32_t n, int32_t k, int32_t epilogue,
                             int32_t dtype, void* stream) {
  if (!x || !ln_g || !ln_b || !w || !bias || !out || rows <= 0) return VITTF_ERR_INVALID_ARG;
  if (k != WK || n <= 0 || n % WNT != 0 || (rows + WBM - 1) / WBM * (n / WNT) > (1 << 30)) return VITTF_ERR_INVALID_ARG;
  int rc = VITTF_ERR_INVALID_ARG;
  if (dtype == VITTF_BF16) rc = launch_ws<VITTF_BF16, true>(x, w, bias, out, rows, n, epilogue, ln_g, ln_b, ln_eps, (hipStream_t)stream);
  if (dtype == VITTF_FP16) rc = launch_ws<VITTF_FP16, true>(x, w, bias, out, rows, n, epilogue, ln_g, ln_b, ln_eps, (hipStream_t)stream);
  return rc == 1 ? VITTF_ERR_INVALID_ARG : rc;
}
