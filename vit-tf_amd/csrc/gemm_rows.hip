// Whole-row MFMA GEMM for the residual linears of ViT-S (attn.proj, mlp.fc2):
//     x[rows][384] (fp32) += A[rows][K] . W[384][K]^T + bias          (K = 384 or 1536)
// and, optionally in the same pass, the next LayerNorm:  h[rows][384] (16 bit) = LayerNorm(x_new; g, b).
//
// Why next to gemm.hip: a 128 x 128 tile moves 32 KB of operands from L2 into LDS per 64-wide K step, 2.36 GB per fc2
// launch, and its run time follows L2->LDS bytes / ~17 TB/s + HBM bytes / ~5 TB/s.  Here a workgroup owns 256 whole
// rows: 16 KB (activations) + 24 KB (all 384 weight rows) per 32-wide K step for 256 x 384 outputs = 0.98 GB per
// launch (-58 %), and because the rows are whole the LayerNorm that follows every residual add can be computed on
// the way out instead of by a separate kernel that re-reads the 201 MB stream.
//
//   * 8 waves (2 per SIMD, 256 VGPRs) as 2 (rows) x 4 (columns): a wave owns 128 x 96 outputs = 12 accumulator tiles
//     (192 VGPRs), computed transposed (weights = MFMA A operand, activations = B operand) like the other GEMMs;
//     per 16-wide K slice 4 + 3 fragment reads feed 12 MFMAs.
//   * operands arrive by LDS-DMA (asm pieces) into a 3-deep ring of 40 KB stages, one bare barrier per K step; a
//     [R][32] operand image is stored as R / 2 "double rows" of 128 B in the tile_off() layout (rows 2 j and 2 j + 1
//     side by side), which keeps the 16-byte fragment reads conflict-free.
//   * epilogue in four rounds of 64 rows through a padded fp32 LDS tile (it reuses the ring), each round in two
//     passes of 32 rows: 16 lanes per row do the read-modify-write as 256-byte runs with the x of the next pass
//     already in flight, and -- with the whole row in those 16 lanes' registers -- the LayerNorm statistics,
//     affine transform and 16-bit store.
//   * a partial last tile reads zeros for its missing rows and drops their outputs, so a row's bits do not depend on
//     where it sits in a launch (1 rank and N ranks batch the slices differently and must write the same file).
#include "vittf_common.h"

namespace {

constexpr int RN = 384, RBM = 256, RBK = 32, RTHREADS = 512;
constexpr int RA_BYTES = RBM * RBK * 2;           // 16 KB
constexpr int RW_BYTES = RN * RBK * 2;            // 24 KB
constexpr int RSTAGE = RA_BYTES + RW_BYTES;       // 40 KB
constexpr int RSTAGES = 3;
constexpr int RCS = RN * 4 + 16;                  // fp32 staging row stride
constexpr int RBIAS_OFF = 64 * RCS;               // bias copy behind the 64-row staging tile (99,328 B)
constexpr int RLDS = RSTAGES * RSTAGE;            // 122,880 B
static_assert(RBIAS_OFF + 3 * RN * 4 <= RLDS, "staging tile + bias / gamma / beta must fit in the ring");

#define ROWS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// byte offset of 16-byte k-chunk kc (0..3) of row r inside a [R][32] operand image
__device__ __forceinline__ int img_off(int r, int kc) { return tile_off(r >> 1, ((r & 1) << 2) | kc); }

template <int DT, bool LN>
__global__ __launch_bounds__(RTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_rows_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ X, int64_t rows, int k, const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    unsigned short* __restrict__ H) {
  __shared__ __attribute__((aligned(16))) char smem[RLDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t m0 = (int64_t)blockIdx.x * RBM;

  // ---- LDS-DMA: 5 pieces per thread and stage (2 of the activation image, 3 of the weight image) ----
  int voff[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int q = (i < 2 ? i : i - 2) * RTHREADS + tid;
    int dr, c;
    tile_pos(q, dr, c);
    voff[i] = (2 * dr + (c >> 2)) * k * 2 + (c & 3) * 16;
  }
  // the last tile may be partial: its missing activation rows read as zeros (descriptor bounds), their outputs are
  // computed and dropped -- every row takes the same arithmetic wherever it sits in the launch
  const int rows_here = (int)(rows - m0 < RBM ? rows - m0 : RBM);
  const i32x4_t rsrc_a = lds_dma_rsrc(A + m0 * k, (unsigned)(rows_here * k * 2));
  const i32x4_t rsrc_w = lds_dma_rsrc(W, (unsigned)(RN * k * 2));
  const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(smem);
  const unsigned dma_wave = ring_lds + wave * 1024;
#define ROWS_STAGE(T, BUF)                                                                      \
  {                                                                                             \
    const unsigned dst_ = dma_wave + (BUF) * RSTAGE;                                            \
    const int so_ = (T) * (RBK * 2);                                                            \
    lds_dma16(rsrc_a, dst_, voff[0], so_);                                                      \
    lds_dma16(rsrc_a, dst_ + 8192, voff[1], so_);                                               \
    lds_dma16(rsrc_w, dst_ + RA_BYTES, voff[2], so_);                                           \
    lds_dma16(rsrc_w, dst_ + RA_BYTES + 8192, voff[3], so_);                                    \
    lds_dma16(rsrc_w, dst_ + RA_BYTES + 16384, voff[4], so_);                                   \
  }

  // ---- fragment addresses: block b (32 rows further) = + 2048 B and the swizzle bit flips when b is odd ----
  int aoff[2], woff[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    aoff[s2] = img_off(128 * wm + l31, 2 * s2 + h);
    woff[s2] = RA_BYTES + img_off(96 * wn + l31, 2 * s2 + h);
  }

  f32x16_t acc[3][4];   // [column block nb][row block mb]
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

  const int nk = k / RBK;
  ROWS_STAGE(0, 0)
  if (nk > 1) ROWS_STAGE(1, 1)
  for (int t = 0; t < nk; ++t) {
    // this wave's pieces of stage t have landed (the 5 youngest may belong to stage t + 1)
    if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ROWS_BARRIER();   // everybody's pieces have; and everybody is done with stage t - 1, whose buffer is refilled now
    if (t + 2 < nk) ROWS_STAGE(t + 2, (t + 2) % RSTAGES)
    const char* buf = smem + (t % RSTAGES) * RSTAGE;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      s16x8_t af[4], wf[3];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        af[mb] = *reinterpret_cast<const s16x8_t*>(buf + (aoff[s2] ^ ((mb & 1) << 7)) + mb * 2048);
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
        wf[nb] = *reinterpret_cast<const s16x8_t*>(buf + (woff[s2] ^ ((nb & 1) << 7)) + nb * 2048);
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = mfma32<DT>(wf[nb], af[mb], acc[nb][mb]);
    }
  }
#undef ROWS_STAGE

  // ---- epilogue: four rounds of 64 rows (row block mb of both row halves) through a padded fp32 LDS tile ----
  ROWS_BARRIER();                                  // the last stage has been read by everybody: the ring is free
  float* const sbias = reinterpret_cast<float*>(smem + RBIAS_OFF);
  for (int i = tid; i < (LN ? 3 : 1) * RN; i += RTHREADS)
    sbias[i] = i < RN ? bias[i] : (i < 2 * RN ? ln_g[i - RN] : ln_b[i - 2 * RN]);
  // read-modify-write in passes of 32 rows (one row half of the round): 16 lanes per row, 16-byte chunks seg + 16 j.
  // The x loads of the NEXT pass are issued before the stores of this one: the memory counter retires in issue order,
  // so a load waited for behind earlier stores would also wait for those stores.
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));   // opaque: keeps the epilogue's lane-derived addresses out of the main loop's registers
  const int row_p = tid_e >> 4, seg = tid_e & 15;
  float4 xn[6];
#define ROWS_LOADX(MB, P)                                                                       \
  {                                                                                             \
    const int rl_ = (P) * 128 + 32 * (MB) + row_p;                                              \
    const float* xr_ = X + (m0 + (rl_ < rows_here ? rl_ : rows_here - 1)) * RN + seg * 4;       \
    _Pragma("unroll") for (int j = 0; j < 6; ++j) xn[j] = *reinterpret_cast<const float4*>(xr_ + 64 * j); \
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    ROWS_BARRIER();                                // bias visible (first round) / previous round's readers are done
    // the lane owns row 128 wm + 32 mb + l31 and columns 96 wn + 32 nb + 8 g + 4 h + {0..3}
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 96 * wn + 32 * nb + 8 * g + 4 * h;
        const float4 bv = *reinterpret_cast<const float4*>(sbias + n);
        float4 v;
        v.x = acc[nb][mb][4 * g + 0] + bv.x; v.y = acc[nb][mb][4 * g + 1] + bv.y;
        v.z = acc[nb][mb][4 * g + 2] + bv.z; v.w = acc[nb][mb][4 * g + 3] + bv.w;
        *reinterpret_cast<float4*>(smem + (32 * wm + l31) * RCS + n * 4) = v;
      }
    if (mb == 0) {   // (not earlier: all 192 accumulator registers are live until the first round has been staged)
      __builtin_amdgcn_sched_barrier(0);
      ROWS_LOADX(0, 0)
    }
    ROWS_BARRIER();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int64_t gm = m0 + p * 128 + 32 * mb + row_p;
      const bool live = p * 128 + 32 * mb + row_p < rows_here;
      const char* sr = smem + (32 * p + row_p) * RCS + seg * 16;
      float4 x[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) x[j] = xn[j];
      if (p == 0) ROWS_LOADX(mb, 1)
      else if (mb < 3) ROWS_LOADX(mb + 1, 0)
      __builtin_amdgcn_sched_barrier(0);
      float* xw = X + gm * RN + seg * 4;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float4 d = *reinterpret_cast<const float4*>(sr + 256 * j);
        x[j].x += d.x; x[j].y += d.y; x[j].z += d.z; x[j].w += d.w;
        if (live) *reinterpret_cast<float4*>(xw + 64 * j) = x[j];
        s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
      }
      __builtin_amdgcn_sched_barrier(0);   // (keeps the LayerNorm's LDS reads from being hoisted into the loop above: spills)
      if constexpr (LN) {
        // the row's 384 new values sit in these 16 lanes (24 each): statistics by 4 shuffles, arithmetic of layernorm.hip
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        const float mean = s / (float)RN;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          x[j].x -= mean; x[j].y -= mean; x[j].z -= mean; x[j].w -= mean;
          q += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
        }
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
        const float rstd = 1.0f / sqrtf(q / (float)RN + ln_eps);
        unsigned short* hr = H + gm * RN + seg * 4;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float4 gg = *reinterpret_cast<const float4*>(sbias + RN + seg * 4 + 64 * j);
          const float4 bb = *reinterpret_cast<const float4*>(sbias + 2 * RN + seg * 4 + 64 * j);
          uint2 pk;
          pk.x = pack2_h16<DT>(x[j].x * rstd * gg.x + bb.x, x[j].y * rstd * gg.y + bb.y);
          pk.y = pack2_h16<DT>(x[j].z * rstd * gg.z + bb.z, x[j].w * rstd * gg.w + bb.w);
          if (live) *reinterpret_cast<uint2*>(hr + 64 * j) = pk;
        }
      }
    }
  }
#undef ROWS_LOADX
}

}  // namespace

// 1 = shape not covered
int vittf_gemm_rows(const void* a, const void* w, const float* bias, float* x, int64_t rows, int32_t n, int32_t k,
                    int32_t dtype, const float* ln_g, const float* ln_b, float ln_eps, void* h, hipStream_t st) {
  if (n != RN || k % RBK != 0 || k < 2 * RBK || rows <= 0) return 1;
  if ((int64_t)RBM * k * 2 > 0x7fffffff || rows / RBM + 1 > 0x7fffffff) return 1;
  const dim3 grid((unsigned)((rows + RBM - 1) / RBM)), block(RTHREADS);
  const unsigned short* A = (const unsigned short*)a;
  const unsigned short* Wp = (const unsigned short*)w;
  const bool ln = ln_g && ln_b && h;
  if (dtype == VITTF_BF16) {
    if (ln) hipLaunchKernelGGL((gemm_rows_kernel<VITTF_BF16, true>), grid, block, 0, st, A, Wp, bias, x, rows, k, ln_g, ln_b, ln_eps, (unsigned short*)h);
    else hipLaunchKernelGGL((gemm_rows_kernel<VITTF_BF16, false>), grid, block, 0, st, A, Wp, bias, x, rows, k, ln_g, ln_b, ln_eps, (unsigned short*)h);
  } else if (dtype == VITTF_FP16) {
    if (ln) hipLaunchKernelGGL((gemm_rows_kernel<VITTF_FP16, true>), grid, block, 0, st, A, Wp, bias, x, rows, k, ln_g, ln_b, ln_eps, (unsigned short*)h);
    else hipLaunchKernelGGL((gemm_rows_kernel<VITTF_FP16, false>), grid, block, 0, st, A, Wp, bias, x, rows, k, ln_g, ln_b, ln_eps, (unsigned short*)h);
  } else {
    return VITTF_ERR_INVALID_ARG;
  }
  return vittf_check_launch();
}
