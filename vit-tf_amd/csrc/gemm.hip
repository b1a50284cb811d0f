// MFMA GEMM for the ViT linears:  out = epilogue(A[rows][K] . W[N][K]^T + bias[N])
//
// Replaces nn.Linear (attn.qkv, attn.proj, mlp.fc1 + GELU, mlp.fc2) of the upstream DINO blocks that the
// reference runs through torch (infer.py:177 -> model(...)); the fused epilogues replace the separate
// bias / GELU / residual-add passes.
//
// Shape of the machine mapping (gfx950):
//   * 128 x 128 output tile per 256-thread workgroup, 4 waves as 2 (m) x 2 (n), each wave 64 x 64
//   * K step 64; A and W tiles are [128][64 x 16 bit] images (16 KB each) filled by global_load_lds_dwordx4
//     (no VGPR staging); the XOR swizzle of tile_off() is applied on the per-lane SOURCE address because the
//     LDS destination of an LDS-DMA is lane-linear; one stage + 4 workgroups per CU (the K = 384 shapes have
//     only six K steps, so latency is hidden across workgroups, not inside one)
//   * v_mfma_f32_32x32x16 with W as the A operand and the activations as the B operand, i.e. the wave
//     computes C^T: a lane then owns ONE activation row and 4 consecutive output columns per register quad,
//     so bias loads are float4 and stores are 8 B (16-bit out) or 16 B (fp32 residual) per lane
//   * workgroups are remapped so that consecutive tiles (which share the A panel) run on one XCD's L2
#include "vittf_common.h"

#include <stdlib.h>

namespace {

// LDS-only synchronisation for the epilogue: __syncthreads() also drains vmcnt(0), i.e. the residual read-modify-write
// of the first half tile would be waited for before the second half is even staged
#define GEMM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KB

// issue the LDS-DMA of one [128][64] operand tile: 1024 16-byte chunks, 4 per thread (asm pieces, see lds_dma16:
// with the builtin, hipcc drains vmcnt(0) before the fragment reads of the tile being multiplied, which is what
// made the two-stage pipeline slower than the single stage)
template <typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ src, int64_t ld, int64_t row0, int64_t row_max,
                                           int k0, unsigned lds_tile, int tid, const int (&voff)[4]) {
  // descriptor over the tile's rows [row0, min(row0 + 127, row_max)]: rows past the end read as zeros (never stored)
  const int64_t nrows = row_max - row0 + 1 < 128 ? row_max - row0 + 1 : 128;
  const i32x4_t rsrc = lds_dma_rsrc(src + row0 * ld, (unsigned)(nrows * ld * 2));
#pragma unroll
  for (int i = 0; i < 4; ++i) lds_dma16(rsrc, lds_tile + i * 4096, voff[i], k0 * 2);
}

template <int DT, int EPI, int NSTAGE>
__global__ __launch_bounds__(256, NSTAGE == 1 ? 4 : 2) void gemm_kernel(const unsigned short* __restrict__ A,
                                                      const unsigned short* __restrict__ W,
                                                      const float* __restrict__ bias, void* __restrict__ out,
                                                      int64_t rows, int n, int k, int tokens, int n_tiles,
                                                      int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A tile | W tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;

  const int tile = xcd_remap(blockIdx.x, total_tiles);
  const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nt * BN;

  f32x16_t acc[2][2];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = k / BK;
  int voff[4];   // LDS-DMA source offsets: chunk q = i * 256 + tid of a tile image <- (row, 16-byte chunk) = tile_pos(q)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r, c;
    tile_pos(i * 256 + tid, r, c);
    voff[i] = (r * k + c * 8) * 2;
  }
  const unsigned lds_wave = (unsigned)(size_t)LDS_PTR(smem) + (__builtin_amdgcn_readfirstlane(tid & ~63) << 4);
  auto compute_tile = [&](const char* a_t, const char* w_t) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 2 * s + h;
      s16x8_t af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const s16x8_t*>(a_t + tile_off(wm * 64 + i * 32 + l31, c));
        wf[i] = *reinterpret_cast<const s16x8_t*>(w_t + tile_off(wn * 64 + i * 32 + l31, c));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = mfma32<DT>(wf[ni], af[mi], acc[ni][mi]);
    }
  };
  if constexpr (NSTAGE == 1) {
    // one 32 KB stage, four workgroups per CU: the other workgroups' MFMAs cover this one's load latency
    for (int t = 0; t < nk; ++t) {
      stage_tile(A, k, m0, rows - 1, t * BK, lds_wave, tid, voff);
      stage_tile(W, k, n0, n - 1, t * BK, lds_wave + TILE_BYTES, tid, voff);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute_tile(smem, smem + TILE_BYTES);
      __syncthreads();
    }
  } else {
    stage_tile(A, k, m0, rows - 1, 0, lds_wave, tid, voff);
    stage_tile(W, k, n0, n - 1, 0, lds_wave + TILE_BYTES, tid, voff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      char* cur = smem + (t & 1) * 2 * TILE_BYTES;
      if (t + 1 < nk) {
        const unsigned nxt = lds_wave + ((t + 1) & 1) * 2 * TILE_BYTES;
        stage_tile(A, k, m0, rows - 1, (t + 1) * BK, nxt, tid, voff);
        stage_tile(W, k, n0, n - 1, (t + 1) * BK, nxt + TILE_BYTES, tid, voff);
      }
      compute_tile(cur, cur + TILE_BYTES);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue ----
  // The accumulators hold C^T: a lane owns activation row m (per mi) and columns nb + 8g + 4h + {0..3}.
  if constexpr (EPI == VITTF_EPI_BIAS_RESIDUAL) {
    // fp32 read-modify-write of the residual stream, also re-tiled through LDS so that every global access is a
    // whole 256-byte row segment: two passes of 64 columns (the 128 x 64 fp32 half tile is 32 KB + padding)
    constexpr int CSF = 64 * 4 + 16;
    float* xo = reinterpret_cast<float*>(out);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half) GEMM_LDS_BARRIER();
      if (wn == half) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int ml = wm * 64 + mi * 32 + l31;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int nl = ni * 32 + 8 * g + 4 * h;          // column inside this 64-wide half
              const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + half * 64 + nl);
              float4 v;
              v.x = acc[ni][mi][4 * g + 0] + bv.x; v.y = acc[ni][mi][4 * g + 1] + bv.y;
              v.z = acc[ni][mi][4 * g + 2] + bv.z; v.w = acc[ni][mi][4 * g + 3] + bv.w;
              *reinterpret_cast<float4*>(smem + ml * CSF + nl * 4) = v;
            }
          }
        }
      }
      GEMM_LDS_BARRIER();
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int rl = pass * 16 + (tid >> 4);
        const int64_t m = m0 + rl;
        if (m >= rows) continue;
        const float4 d = *reinterpret_cast<const float4*>(smem + rl * CSF + (tid & 15) * 16);
        float4* p = reinterpret_cast<float4*>(xo + m * n + n0 + half * 64 + (tid & 15) * 4);
        float4 x = *p;
        x.x += d.x; x.y += d.y; x.z += d.z; x.w += d.w;
        *p = x;
      }
    }
  } else {
    // 16-bit outputs go through LDS so that global stores are whole 256-byte rows (16 lanes x 16 B): storing the
    // register fragments directly puts 16-byte pieces on 32 different rows per instruction and halves the GEMM's
    // speed (measured: 0.23 -> 0.12 ms for the qkv shape with the stores removed).
    // The K loop ended on a barrier, so the operand tiles are dead; C tile rows are padded to 272 B.
    constexpr int CS = BN * 2 + 16;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int ml = wm * 64 + mi * 32 + l31;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * 64 + ni * 32 + 8 * g + 4 * h;
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
          *reinterpret_cast<uint2*>(smem + ml * CS + nl * 2) = pk;
        }
      }
    }
    GEMM_LDS_BARRIER();
    unsigned short* o16 = reinterpret_cast<unsigned short*>(out);
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int rl = pass * 16 + (tid >> 4);
      const int64_t m = m0 + rl;
      if (m >= rows) continue;
      int64_t orow = m;
      if constexpr (EPI == VITTF_EPI_KFEAT) {
        const int64_t b = m / tokens;
        const int tok = (int)(m - b * tokens);
        if (tok == 0) continue;  // CLS row dropped (infer.py:202 k[:, 1:])
        orow = b * (tokens - 1) + tok - 1;
      }
      const uint4 v = *reinterpret_cast<const uint4*>(smem + rl * CS + (tid & 15) * 16);
      *reinterpret_cast<uint4*>(o16 + orow * n + n0 + (tid & 15) * 8) = v;
    }
  }
}

template <int DT>
int launch_gemm(const void* a, const void* w, const float* bias, void* out, int64_t rows, int n, int k, int epi,
                int tokens, hipStream_t st) {
  const int m_tiles = (int)((rows + BM - 1) / BM), n_tiles = n / BN;
  const int total = m_tiles * n_tiles;
  // 1 = one 32 KB operand stage and four workgroups per CU (default: +16 % on the K = 384 shapes, whose six
  // K steps are too short for a two-stage pipeline to cover the load latency); 2 = double buffer, two per CU
  constexpr int nstage = 1;      // (one 32 KB stage x four workgroups per CU; the double-buffered form measured slower on every shape)
  const size_t lds = nstage == 1 ? (size_t)BM * (BN * 2 + 16) : (size_t)4 * TILE_BYTES;   // stage(s) / padded C tile (16-bit: 128 x 272 B; fp32 half tile: 128 x 272 B)
  const unsigned short* A = (const unsigned short*)a;
  const unsigned short* Wp = (const unsigned short*)w;
#define VITTF_GEMM_CASE(E)                                                                                   \
  case E: {                                                                                                  \
    static bool attr_set = false;                                                                            \
    if (!attr_set) {                                                                                         \
      (void)hipFuncSetAttribute((const void*)gemm_kernel<DT, E, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)(4 * TILE_BYTES));                                                      \
      attr_set = true;                                                                                       \
    }                                                                                                        \
    if (nstage == 1)                                                                                         \
      hipLaunchKernelGGL((gemm_kernel<DT, E, 1>), dim3(total), dim3(256), lds, st, A, Wp, bias, out, rows, n, k, \
                         tokens, n_tiles, total);                                                            \
    else                                                                                                     \
      hipLaunchKernelGGL((gemm_kernel<DT, E, 2>), dim3(total), dim3(256), lds, st, A, Wp, bias, out, rows, n, k, \
                         tokens, n_tiles, total);                                                            \
    break;                                                                                                   \
  }
  switch (epi) {
    VITTF_GEMM_CASE(VITTF_EPI_BIAS)
    VITTF_GEMM_CASE(VITTF_EPI_BIAS_GELU)
    VITTF_GEMM_CASE(VITTF_EPI_BIAS_RESIDUAL)
    VITTF_GEMM_CASE(VITTF_EPI_KFEAT)
    VITTF_GEMM_CASE(VITTF_EPI_BIAS_QKV)
    default: return VITTF_ERR_INVALID_ARG;
  }
#undef VITTF_GEMM_CASE
  return vittf_check_launch();
}

}  // namespace

int vittf_gemm_rows(const void* a, const void* w, const float* bias, float* x, int64_t rows, int32_t n, int32_t k,
                    int32_t dtype, const float* ln_g, const float* ln_b, float ln_eps, void* h, hipStream_t st);   // gemm_rows.hip

int vittf_gemm_pp(const void* a, const void* w, const float* bias, void* out, int64_t rows, int32_t n, int32_t k,
                  int32_t epilogue, int32_t tokens, int32_t dtype, hipStream_t st);   // gemm_pp.hip; 1 = not covered

// K >= 768 with N % 256 == 0 (the ViT-B linears) run on the persistent 256 x 256 kernel of gemm_pp.hip.  Residual linears with
// 768 output columns take it from K = 3072 on (fc2: the LayerNorm behind it then runs as its own launch instead of in the
// whole-row kernel's epilogue: 1.56 against 1.87 ms per 64 slices); proj, K = 768, stays on the whole-row kernel.
static bool use_pp_residual(int k) { return k >= 3072; }

extern "C" int vittf_gemm(const void* a, const void* w, const float* bias, void* out, int64_t rows, int32_t n,
                          int32_t k, int32_t epilogue, int32_t tokens, int32_t dtype, void* stream) {
  if (!a || !w || !bias || !out || rows <= 0 || n <= 0 || k <= 0) return VITTF_ERR_INVALID_ARG;
  if (n % BN != 0 || k % BK != 0) return VITTF_ERR_INVALID_ARG;
  if (epilogue == VITTF_EPI_KFEAT && tokens < 2) return VITTF_ERR_INVALID_ARG;
  if (rows / BM + 1 > (1 << 20)) return VITTF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  // residual epilogue with 384 / 768 output columns (ViT-S / ViT-B proj and fc2): whole-row kernel
  const bool pp_first = use_pp_residual(k) && n == 768 && k >= 768;
  if (!pp_first && epilogue == VITTF_EPI_BIAS_RESIDUAL && (n == 384 || n == 768)) {
    const int rc = vittf_gemm_rows(a, w, bias, (float*)out, rows, n, k, dtype, nullptr, nullptr, 0.f, nullptr, st);
    if (rc != 1) return rc;
  }
  {
    const int rc = vittf_gemm_pp(a, w, bias, out, rows, n, k, epilogue, tokens, dtype, st);
    if (rc != 1) return rc;
  }
  if (dtype == VITTF_BF16) return launch_gemm<VITTF_BF16>(a, w, bias, out, rows, n, k, epilogue, tokens, st);
  if (dtype == VITTF_FP16) return launch_gemm<VITTF_FP16>(a, w, bias, out, rows, n, k, epilogue, tokens, st);
  return VITTF_ERR_INVALID_ARG;
}

// x += a . w^T + bias (fp32 residual stream, n = 384), then h = LayerNorm(x; g, b) as the 16-bit operand of the next
// GEMM: whole-row kernel with the LayerNorm in its epilogue; shapes it does not cover take the two separate kernels.
extern "C" int vittf_layernorm(const float* x, const float* g, const float* b, void* y, int64_t rows, int32_t d, float eps,
                               int32_t dtype, void* stream);
extern "C" int vittf_gemm_residual_ln(const void* a, const void* w, const float* bias, float* x, int64_t rows, int32_t n,
                                      int32_t k, int32_t dtype, const float* ln_g, const float* ln_b, float ln_eps, void* h,
                                      void* stream) {
  if (!a || !w || !bias || !x || !ln_g || !ln_b || !h || rows <= 0 || n <= 0 || k <= 0) return VITTF_ERR_INVALID_ARG;
  if (dtype != VITTF_BF16 && dtype != VITTF_FP16) return VITTF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if ((n == 384 || n == 768) && !(use_pp_residual(k) && n == 768 && k >= 768)) {
    const int rc = vittf_gemm_rows(a, w, bias, x, rows, n, k, dtype, ln_g, ln_b, ln_eps, h, st);
    if (rc != 1) return rc;
  }
  const int rc = vittf_gemm(a, w, bias, x, rows, n, k, VITTF_EPI_BIAS_RESIDUAL, 0, dtype, stream);
  if (rc != VITTF_OK) return rc;
  return vittf_layernorm(x, ln_g, ln_b, h, rows, n, ln_eps, dtype, stream);
}
