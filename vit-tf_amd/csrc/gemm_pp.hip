// Ping-pong MFMA GEMM for the long-K linears of ViT-B (D = 768: attn.qkv, attn.proj, mlp.fc1 + GELU, mlp.fc2):
//     out = epilogue(A[rows][K] . W[N][K]^T + bias[N]),   K % 32 == 0, K >= 768, N % 256 == 0
//
// Round 4 (BASELINE configs[3]).  gemm.hip's 128 x 128 x 64 tiles (one stage, four workgroups per CU) run these shapes at
// 740-870 TFLOP/s; a first pipelined 256 x 128 kernel with two workgroups per CU (3-deep ring, one barrier per 32-wide K step,
// 6 LDS-DMA pieces per wave and 16 MFMAs) was SLOWER (800): every wave of it alternates between issuing memory work and
// issuing MFMAs, and an LDS-DMA piece costs 60-180 issue cycles.  This kernel separates the two in time and pairs them in
// space (MI355X_MICROARCH.md "Two waves per SIMD"; the 8-wave ping-pong schedule):
//   * workgroup = 8 waves = 2 groups of 4 on a 256 x 256 output tile; wave (g, c) owns rows 128 g .. + 127 and columns
//     64 c .. + 63 = 4 x 2 accumulator tiles (128 VGPRs), computed transposed like the other GEMMs (weights = MFMA A operand,
//     activations = B operand: a lane owns one activation row and 4 consecutive columns per register quad).  Waves w and w + 4
//     share a SIMD: the two groups are SIMD partners.
//   * a K step of 32 = a stage: [256][32] activation image + [256][32] weight image = 32 KB in the "double-row" layout of
//     gemm_rows.hip (conflict-free ds_read_b128); 4-deep ring = 128 KB, one workgroup per CU.
//   * per stage a wave runs a LOAD segment -- its 12 fragment reads of stage t, its 4 LDS-DMA pieces of stage t + 3, the counted
//     wait for its pieces of stage t + 1 -- and an MFMA segment (16 MFMAs, s_setprio 1), each closed by a workgroup barrier;
//     group 1 runs one barrier behind group 0, so while one group's waves issue MFMAs their SIMD partners issue memory work:
//     the matrix pipe of a SIMD always has a wave in its MFMA segment.  0.75 LDS reads and 0.25 DMA pieces per MFMA.
//   * hazards by construction: stage t is read in slots 2 t (group 0) and 2 t + 1 (group 1); every wave has waited for its
//     pieces of stage t in its load segment of stage t - 1 (slots 2 t - 2, 2 t - 1), in front of a barrier the readers pass;
//     the pieces of stage t + 3 overwrite the slot of stage t - 1, last read in slot 2 t - 1 with lgkmcnt(0) in front of that
//     slot's closing barrier, and are issued in slots 2 t and 2 t + 1.
//   * epilogue through LDS (the ring is dead) in row halves / quarters so that every global access is a whole row segment.
//   * a partial last row tile reads zeros for its missing rows (descriptor bounds) and drops their outputs: a row's bits do
//     not depend on where it sits in a launch.
#include "vittf_common.h"

#include <stdlib.h>

namespace {

constexpr int PBM = 256, PBN = 256, PBK = 32;
constexpr int PIMG = 256 * PBK * 2;              // one operand image: 16 KB
constexpr int PSTAGE = 2 * PIMG;                 // 32 KB
constexpr int PSTAGES = 4;
constexpr int PLDS = PSTAGES * PSTAGE;           // 128 KB
constexpr int PCS = 512 + 16;                    // staging row stride: 128 rows x 528 B = 67,584 B
static_assert(128 * PCS <= PLDS, "the C tile re-uses the ring");

#define PP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- timing-only variants for tools/pp_variants.sh (never in libvittf.so: the Makefile does not define PP_VARIANT) ----
#ifndef PP_VARIANT
#define PP_VARIANT 0
#endif
constexpr bool PV_NO_DMA = PP_VARIANT & 1;       // no LDS-DMA inside the K loop (wrong results)
constexpr bool PV_KEEP_M0 = PP_VARIANT & 2;      // LDS-DMA pieces that leave M0 pointing at their destination (no save / restore)
constexpr bool PV_NO_READS = PP_VARIANT & 4;     // fragments read once, before the K loop (wrong results)
constexpr bool PV_SPLIT_DMA = PP_VARIANT & 8;    // two of the four pieces of a stage issued at the head of the MFMA segment
constexpr bool PV_NO_PRIO = PP_VARIANT & 16;     // no s_setprio around the MFMA segment
constexpr bool PV_NO_EPI = PP_VARIANT & 32;      // no epilogue (wrong results)
constexpr bool PV_STAMPS = PP_VARIANT & 64;      // s_memtime at the segment boundaries, summed over a tile's K loop
#if PP_VARIANT & 64
__device__ unsigned long long g_pp_stamps[8 /*workgroups*/][8 /*waves*/][6];
#define PP_T(k) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tacc[k] += t_ - tprev; tprev = t_; }
#else
#define PP_T(k)
#endif

__device__ __forceinline__ void pp_dma16_keep(i32x4_t rsrc, unsigned lds_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void pp_dma16(i32x4_t rsrc, unsigned lds_addr, int voff, int soff) {
  if constexpr (PV_KEEP_M0) pp_dma16_keep(rsrc, lds_addr, voff, soff);
  else lds_dma16(rsrc, lds_addr, voff, soff);
}

// byte offset of 16-byte k-chunk kc (0..3) of row r inside a [R][32] operand image (two rows per 128-byte tile_off row)
__device__ __forceinline__ int p_img_off(int r, int kc) { return tile_off(r >> 1, ((r & 1) << 2) | kc); }

template <int DT, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W,
                                                         const float* __restrict__ bias, void* __restrict__ out, int64_t rows,
                                                         int n, int k, int tokens, int n_tiles, int total_tiles) {
  __shared__ __attribute__((aligned(16))) char smem[PLDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int grp = wave >> 2, wc = wave & 3;      // waves w and w + 4 share a SIMD: the two groups are SIMD partners

  const int tile = xcd_remap(blockIdx.x, total_tiles);
  const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
  const int64_t m0 = (int64_t)mt * PBM;
  const int n0 = nt * PBN;

  // ---- LDS-DMA: chunk q = i * 512 + tid of an image <- (row, k chunk) by the inverse of p_img_off; 2 + 2 pieces per wave ----
  int voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int dr, c;
    tile_pos(i * 512 + tid, dr, c);
    voff[i] = (2 * dr + (c >> 2)) * k * 2 + (c & 3) * 16;
  }
  const int rows_here = (int)(rows - m0 < PBM ? rows - m0 : PBM);
  const i32x4_t rsrc_a = lds_dma_rsrc(A + m0 * k, (unsigned)((int64_t)rows_here * k * 2));
  const i32x4_t rsrc_w = lds_dma_rsrc(W + (int64_t)n0 * k, (unsigned)(PBN * k * 2));
  const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(smem);
  const unsigned dma_wave = ring_lds + wave * 1024;
#define PP_STAGE_A(T, BUF)                                                  \
  {                                                                         \
    const unsigned dst_ = dma_wave + (BUF) * PSTAGE;                        \
    const int so_ = (T) * (PBK * 2);                                        \
    pp_dma16(rsrc_a, dst_, voff[0], so_);                                   \
    pp_dma16(rsrc_a, dst_ + 8192, voff[1], so_);                            \
  }
#define PP_STAGE_W(T, BUF)                                                  \
  {                                                                         \
    const unsigned dst_ = dma_wave + (BUF) * PSTAGE;                        \
    const int so_ = (T) * (PBK * 2);                                        \
    pp_dma16(rsrc_w, dst_ + PIMG, voff[0], so_);                            \
    pp_dma16(rsrc_w, dst_ + PIMG + 8192, voff[1], so_);                     \
  }
#define PP_STAGE(T, BUF) { PP_STAGE_A(T, BUF) PP_STAGE_W(T, BUF) }

  // ---- fragment addresses inside a stage ----
  int aoff[4][2], woff[2][2];                    // [32-row block][k16 step]
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) aoff[mi][s] = p_img_off(grp * 128 + mi * 32 + l31, 2 * s + h);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) woff[ni][s] = PIMG + p_img_off(wc * 64 + ni * 32 + l31, 2 * s + h);
  }

  f32x16_t acc[2][4];                            // [ni][mi]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = k / PBK;                        // >= 24
  PP_STAGE(0, 0);
  PP_STAGE(1, 1);
  PP_STAGE(2, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // this wave's pieces of stage 0
  PP_BARRIER();                                        // ... everybody's
  if (grp == 1) PP_BARRIER();                          // group 1 runs one barrier behind
  int buf = 0;
  s16x8_t af[4][2], wf[2][2];
  [[maybe_unused]] unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
  if constexpr (PV_STAMPS) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
  for (int t = 0; t < nk; ++t) {
    // ---- load segment: fragments of stage t, requests for stage t + 3, wait for this wave's pieces of stage t + 1 ----
    const char* st = smem + (PV_NO_READS ? 0 : buf) * PSTAGE;
    if (!PV_NO_READS || t == 0)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) wf[ni][s] = *reinterpret_cast<const s16x8_t*>(st + woff[ni][s]);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi][s] = *reinterpret_cast<const s16x8_t*>(st + aoff[mi][s]);
    }
    const int nb = buf == 0 ? 3 : buf - 1;             // the slot of stage t - 1 = (t + 3) % 4
    if (PV_NO_DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (PV_SPLIT_DMA) {
      // (the two weight pieces of stage t + 3 go out at the head of the MFMA segment: the counted waits see 2 pieces fewer
      // of the newest stage here and the same numbers otherwise)
      if (t + 3 < nk) { PP_STAGE_A(t + 3, nb); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
      else if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (t + 3 < nk) {
      PP_STAGE(t + 3, nb);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); // all but stages t + 2, t + 3
    } else if (t + 2 < nk) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); // all but stage t + 2
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_T(0)                                            // reads issued + DMA issued + counted wait
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_T(1)                                            // fragments arrived
    PP_BARRIER();
    PP_T(2)                                            // waited at the barrier that closes the load segment
    // ---- MFMA segment ----
    __builtin_amdgcn_sched_barrier(0);
    if (PV_SPLIT_DMA && !PV_NO_DMA && t + 3 < nk) PP_STAGE_W(t + 3, nb);
    if (!PV_NO_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma32<DT>(wf[ni][s], af[mi][s], acc[ni][mi]);
    if (!PV_NO_PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    PP_T(3)                                            // 16 MFMAs issued
    PP_BARRIER();
    PP_T(4)                                            // waited at the barrier that closes the MFMA segment
    buf = buf == 3 ? 0 : buf + 1;
  }
  if (grp == 0) PP_BARRIER();                          // (the barrier group 1 spent up front)
#undef PP_STAGE
#undef PP_STAGE_A
#undef PP_STAGE_W
#if PP_VARIANT & 64
  if (blockIdx.x >= 8 && blockIdx.x < 16 && lane == 0) {     // (second round of XCD 0..7's first workgroups? no: blocks 8..15)
    unsigned long long t_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
    for (int q = 0; q < 5; ++q) g_pp_stamps[blockIdx.x - 8][wave][q] = tacc[q];
    g_pp_stamps[blockIdx.x - 8][wave][5] = t_;
  }
#endif
  if constexpr (PV_NO_EPI) {                           // (timing-only: keep the accumulators alive)
    float sink = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sink += acc[i][j][r];
    if (sink == 1.2345f) reinterpret_cast<float*>(out)[tid] = sink;
    return;
  }

  // ---- epilogue: the accumulators hold C^T -- a lane owns activation row (per mi) and columns 32 ni + 8 g + 4 h + {0..3}.
  //      The ring is dead (the last fragment reads finished in front of a barrier everybody has passed). ----
  if constexpr (EPI == VITTF_EPI_BIAS_RESIDUAL) {
    // fp32 read-modify-write of the residual stream: four passes of 128 rows x 128 columns (512-byte row segments)
    float* xo = reinterpret_cast<float*>(out);
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
      const int pg = pass >> 1, pc = pass & 1;         // row half, column half
      if (pass) PP_BARRIER();
      if (grp == pg && (wc >> 1) == pc) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int ml = mi * 32 + l31;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int nl = (wc & 1) * 64 + ni * 32 + 8 * g + 4 * h;     // column inside this 128-wide half
              const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + pc * 128 + nl);
              float4 v;
              v.x = acc[ni][mi][4 * g + 0] + bv.x; v.y = acc[ni][mi][4 * g + 1] + bv.y;
              v.z = acc[ni][mi][4 * g + 2] + bv.z; v.w = acc[ni][mi][4 * g + 3] + bv.w;
              *reinterpret_cast<float4*>(smem + ml * PCS + nl * 4) = v;
            }
          }
        }
      }
      PP_BARRIER();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = i * 16 + (tid >> 5);            // 32 lanes per row: 512 bytes
        const int64_t m = m0 + pg * 128 + rl;
        if (m >= rows) continue;
        const float4 d = *reinterpret_cast<const float4*>(smem + rl * PCS + (tid & 31) * 16);
        float4* p = reinterpret_cast<float4*>(xo + m * n + n0 + pc * 128 + (tid & 31) * 4);
        float4 x = *p;
        x.x += d.x; x.y += d.y; x.z += d.z; x.w += d.w;
        *p = x;
      }
    }
  } else {
    // 16-bit outputs: two passes of 128 rows x 256 columns (512-byte row segments)
    unsigned short* o16 = reinterpret_cast<unsigned short*>(out);
#pragma unroll 1
    for (int pg = 0; pg < 2; ++pg) {
      if (pg) PP_BARRIER();
      if (grp == pg) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int ml = mi * 32 + l31;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int nl = wc * 64 + ni * 32 + 8 * g + 4 * h;
              const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + nl);
              float v0 = acc[ni][mi][4 * g + 0] + bv.x;
              float v1 = acc[ni][mi][4 * g + 1] + bv.y;
              float v2 = acc[ni][mi][4 * g + 2] + bv.z;
              float v3 = acc[ni][mi][4 * g + 3] + bv.w;
              if constexpr (EPI == VITTF_EPI_BIAS_GELU) {
                v0 = gelu_poly(v0); v1 = gelu_poly(v1); v2 = gelu_poly(v2); v3 = gelu_poly(v3);
              }
              if constexpr (EPI == VITTF_EPI_BIAS_QKV) {
                // the q third carries the softmax scale and the exp -> exp2 base change: one rounding, like plain q
                const float sc = (n0 + nl) < n / 3 ? 0.125f * 1.44269504088896340736f : 1.0f;
                v0 *= sc; v1 *= sc; v2 *= sc; v3 *= sc;
              }
              uint2 pk;
              if constexpr (EPI == VITTF_EPI_KFEAT) {
                pk.x = pack2_h16<VITTF_FP16>(v0, v1);
                pk.y = pack2_h16<VITTF_FP16>(v2, v3);
              } else {
                pk.x = pack2_h16<DT>(v0, v1);
                pk.y = pack2_h16<DT>(v2, v3);
              }
              *reinterpret_cast<uint2*>(smem + ml * PCS + nl * 2) = pk;
            }
          }
        }
      }
      PP_BARRIER();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = i * 16 + (tid >> 5);            // 32 lanes per row: 512 bytes
        const int64_t m = m0 + pg * 128 + rl;
        if (m >= rows) continue;
        int64_t orow = m;
        if constexpr (EPI == VITTF_EPI_KFEAT) {
          const int64_t b = m / tokens;
          const int tok = (int)(m - b * tokens);
          if (tok == 0) continue;                      // CLS row dropped (infer.py:202 k[:, 1:])
          orow = b * (tokens - 1) + tok - 1;
        }
        const uint4 v = *reinterpret_cast<const uint4*>(smem + rl * PCS + (tid & 31) * 16);
        *reinterpret_cast<uint4*>(o16 + orow * n + n0 + (tid & 31) * 8) = v;
      }
    }
  }
}

template <int DT>
int launch_pp(const void* a, const void* w, const float* bias, void* out, int64_t rows, int n, int k, int epi, int tokens,
              hipStream_t st) {
  const int64_t m_tiles = (rows + PBM - 1) / PBM;
  const int n_tiles = n / PBN;
  const int64_t total64 = m_tiles * n_tiles;
  if (total64 > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  const int total = (int)total64;
  const unsigned short* A = (const unsigned short*)a;
  const unsigned short* Wp = (const unsigned short*)w;
#define VITTF_PP_CASE(E)                                                                                              \
  case E:                                                                                                             \
    hipLaunchKernelGGL((gemm_pp_kernel<DT, E>), dim3(total), dim3(512), 0, st, A, Wp, bias, out, rows, n, k, tokens,  \
                       n_tiles, total);                                                                               \
    break;
  switch (epi) {
    VITTF_PP_CASE(VITTF_EPI_BIAS)
    VITTF_PP_CASE(VITTF_EPI_BIAS_GELU)
    VITTF_PP_CASE(VITTF_EPI_BIAS_RESIDUAL)
    VITTF_PP_CASE(VITTF_EPI_KFEAT)
    VITTF_PP_CASE(VITTF_EPI_BIAS_QKV)
    default: return VITTF_ERR_INVALID_ARG;
  }
#undef VITTF_PP_CASE
  return vittf_check_launch();
}

}  // namespace

#ifdef PP_STANDALONE      // tools/pp_variants.sh builds this file alone
void vittf_note_kernel(int, const char*) {}
#endif
#if PP_VARIANT & 64
extern "C" int vittf_pp_stamps(unsigned long long* out) {      // [8][8][6], host memory
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_stamps), sizeof(g_pp_stamps)) == hipSuccess ? 0 : -1;
}
#endif

// 1 = shape not covered (the caller falls back to gemm.hip's 128 x 128 tiles)
#ifdef PP_STANDALONE
extern "C"
#endif
int vittf_gemm_pp(const void* a, const void* w, const float* bias, void* out, int64_t rows, int32_t n, int32_t k,
                  int32_t epilogue, int32_t tokens, int32_t dtype, hipStream_t st) {
  if (k < 768 || k % PBK != 0 || n % PBN != 0) return 1;
  if ((int64_t)k * 2 * PBM > 0x7fffffff) return 1;      // per-lane source offsets are 32-bit
  if ((((uintptr_t)a | (uintptr_t)w | (uintptr_t)out) & 15) != 0) return 1;
  vittf_note_kernel(VITTF_KERNEL_GEMM, "gemm_pp_kernel");
  if (dtype == VITTF_BF16) return launch_pp<VITTF_BF16>(a, w, bias, out, rows, n, k, epilogue, tokens, st);
  if (dtype == VITTF_FP16) return launch_pp<VITTF_FP16>(a, w, bias, out, rows, n, k, epilogue, tokens, st);
  return VITTF_ERR_INVALID_ARG;
}
