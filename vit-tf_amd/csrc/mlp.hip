// Fused transformer MLP for D = 384:   x += fc2( gelu_erf( fc1(h) + b1 ) ) + b2
//
// Replaces Mlp.forward + the residual add of the upstream DINO block (reached through model(...), infer.py:177).
// The unfused path writes the [rows][4D] hidden activation to HBM and reads it back (806 MB per 32-slice batch,
// more than all other traffic of the two GEMMs together); here it never leaves the registers.
//
// Machine mapping (gfx950), "activations in registers, weights through an LDS ring":
//   * one 256-thread workgroup = 4 waves = 128 rows; ONE wave per SIMD with the whole 512-entry register file:
//     a wave owns 32 rows and keeps, per lane, its rows' h fragments (24 x 4 VGPR), the fc1 accumulators of the
//     current 128-wide hidden chunk (64) and the full 384-wide fc2 accumulators (192)
//   * both products are computed transposed (weights as the MFMA A operand, activations as the B operand), so the
//     fc1 accumulator tile -- bias + GELU applied, converted pairwise to 16 bit -- IS the B operand of fc2
//     (k order 16s + 8(j>>2) + 4h + (j&3); W2's hidden dim is stored in that order by the host, fc2_w_perm)
//   * W1 / W2 stream through LDS as uniform [128][64] tiles (16 KB, tile_off swizzle, global_load_lds_dwordx4):
//     144 tiles per workgroup through an 8-slot ring, 6 tiles in flight (counted vmcnt, one raw s_barrier per
//     tile), so L2 latency is hidden by the ring, not by occupancy; every CU streams the same 2.4 MB of weights
//   * the output is a 16-byte fp32 read-modify-write of the residual stream per lane and register quad
//
// Status (r01): bit-for-bit the arithmetic of the two-GEMM path and parity-tested, but NOT yet faster -- 0.68 ms vs
// 0.63 ms for 131 k rows.  With one wave per SIMD nothing overlaps unless the instruction stream itself
// interleaves it: per 16-MFMA step (512 cycles of matrix work) the wave also spends ~300 cycles waiting for its 16
// fragment reads, ~200 on the DMA address arithmetic, ~400 (amortised) on the erf-GELU and a barrier -- all serial.
// The engine therefore keeps the two-GEMM path by default (HipViT(fused_mlp=True) / VITTF_FUSED_MLP=1 opts in);
// the next step is a half-step software pipeline (fragment reads of the next half under the MFMAs of this one).
#include "vittf_common.h"

namespace {

constexpr int D = 384, HID = 4 * D;
constexpr int CH = 128;                 // hidden units per chunk
constexpr int NCHUNK = HID / CH;        // 12
constexpr int KT1 = D / 64;             // 6 W1 tiles per chunk ([128 hidden][64 k])
constexpr int NT2 = D / 128;            // 3 output column tiles
constexpr int STEPS = KT1 + 2 * NT2;    // 12 tiles per chunk
constexpr int TOTAL = NCHUNK * STEPS;   // 144
constexpr int TILE = 128 * 64 * 2;      // 16 KB
constexpr int RING = 8;                 // slots
constexpr int DEPTH = 6;                // tiles in flight

__device__ __forceinline__ float gelu_erf(float x) {
  // exact-erf GELU; erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7), same as gemm.hip
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = 1.0f - p * __expf(-z * z);
  const float erfv = x < 0.f ? -e : e;
  return 0.5f * x * (1.0f + erfv);
}

// LDS-DMA of weight tile `seq` of the workgroup's stream into its ring slot (4 x 16 B per thread)
__device__ __forceinline__ void issue_tile(int seq, const unsigned short* __restrict__ w1,
                                           const unsigned short* __restrict__ w2p, char* smem, int tid) {
  const int hc = seq / STEPS, r = seq - hc * STEPS;
  const unsigned short* src;
  int ld;
  if (r < KT1) {                 // W1[hc*128 + row][r*64 + k]
    src = w1 + (int64_t)hc * CH * D + r * 64;
    ld = D;
  } else {                       // W2p[nt2*128 + row][hc*128 + kt2*64 + k], kt2-major: r' = kt2 * NT2 + nt2
    const int rp = r - KT1, kt2 = rp / NT2, nt2 = rp - kt2 * NT2;
    src = w2p + (int64_t)nt2 * 128 * HID + hc * CH + kt2 * 64;
    ld = HID;
  }
  char* slot = smem + (seq & (RING - 1)) * TILE;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid;
    int row, c;
    tile_pos(q, row, c);
    const unsigned short* g = src + (int64_t)row * ld + c * 8;
    // (asm piece, see lds_dma16: with the builtin hipcc added its own, stricter vmcnt waits in front of the fragment reads)
    lds_dma16_flat(g, (unsigned)(size_t)LDS_PTR(slot) + ((i * 256 + __builtin_amdgcn_readfirstlane(tid & ~63)) << 4));
  }
}

template <int DT>
__global__ __launch_bounds__(256, 1) void mlp_kernel(const unsigned short* __restrict__ hbuf,
                                                     const unsigned short* __restrict__ w1, const float* __restrict__ b1,
                                                     const unsigned short* __restrict__ w2p, const float* __restrict__ b2,
                                                     float* __restrict__ x, int64_t rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // RING x 16 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int64_t m = (int64_t)blockIdx.x * 128 + wave * 32 + l31;
  const int64_t m_c = m < rows ? m : rows - 1;

  // ---- this lane's h fragments (B operand): H[row][16 s + 8 h .. +7], s = 0..23 ----
  s16x8_t hf[D / 16];
  {
    const unsigned short* hp = hbuf + m_c * D + 8 * h;
#pragma unroll
    for (int s = 0; s < D / 16; ++s) hf[s] = *reinterpret_cast<const s16x8_t*>(hp + 16 * s);
  }
  // retire the loads before any LDS-DMA is in flight (hipcc otherwise drains the whole DMA queue at their first use)
#pragma unroll
  for (int s = 0; s < D / 16; ++s) asm volatile("" : "+v"(hf[s]));

  f32x16_t xacc[D / 32];     // fc2 accumulators: out tile ot (32 columns) x this lane's row
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) xacc[i][r] = 0.f;

  // per-lane part of the fragment address inside a [128][64] tile, per k-step s (see tile_off)
  const int p_l = l31 >> 1;
  const int bslot = (((l31 & 1) << 3) | h) ^ (p_l & 15);
  const int fa0 = (p_l << 8) + ((bslot ^ 0) << 4);
  const int fa1 = (p_l << 8) + ((bslot ^ 2) << 4);
  const int fa2 = (p_l << 8) + ((bslot ^ 4) << 4);
  const int fa3 = (p_l << 8) + ((bslot ^ 6) << 4);

#pragma unroll
  for (int s = 0; s < DEPTH; ++s) issue_tile(s, w1, w2p, smem, tid);

  for (int hc = 0; hc < NCHUNK; ++hc) {
    f32x16_t gacc[4];        // fc1 accumulators of this chunk: hidden tile i (32 units) x this lane's row
    s16x8_t gf[4][2];        // the same after bias + GELU, as fc2 B-operand fragments
#pragma unroll
    for (int r = 0; r < STEPS; ++r) {
      const int seq = hc * STEPS + r;
      // tile `seq` has landed once at most DEPTH-1 younger tiles (4 DMA instructions each) are still in flight
      if (seq + DEPTH - 1 < TOTAL) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 * (DEPTH - 1)) : "memory");
      else                         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");   // everyone's share landed; everyone is done with the slot refilled below
      if (seq + DEPTH < TOTAL) issue_tile(seq + DEPTH, w1, w2p, smem, tid);
      const char* tb = smem + (seq & (RING - 1)) * TILE;
      const char* a0 = tb + fa0;
      const char* a1 = tb + fa1;
      const char* a2 = tb + fa2;
      const char* a3 = tb + fa3;
      // all sixteen weight fragments of the tile first (one wave per SIMD: nothing else hides the LDS latency, and
      // hipcc serialises read -> wait -> MFMA through one register quad when the reads are written next to their use)
      s16x8_t wf[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wf[i][0] = *reinterpret_cast<const s16x8_t*>(a0 + 4096 * i);
        wf[i][1] = *reinterpret_cast<const s16x8_t*>(a1 + 4096 * i);
        wf[i][2] = *reinterpret_cast<const s16x8_t*>(a2 + 4096 * i);
        wf[i][3] = *reinterpret_cast<const s16x8_t*>(a3 + 4096 * i);
      }
      if (r < KT1) {
        // ---- fc1: G^T[hidden tile i][row] += W1 tile . h^T, k tile r ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (r == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) gacc[i][q] = 0.f;
          }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) gacc[i] = mfma32<DT>(wf[i][s4], hf[4 * r + s4], gacc[i]);
        }
        if (r == KT1 - 1) {
          // bias + GELU + pack: registers 8 s' .. 8 s' + 7 of tile i become the fragment of fc2 k-step (i, s')
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
              typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
              u32x4_t u;
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const int hid = hc * CH + 32 * i + 16 * sp + 8 * g + 4 * h;      // acc_row(8 sp + 4 g + j, h)
                const float4 bv = *reinterpret_cast<const float4*>(b1 + hid);
                const float v0 = gelu_erf(gacc[i][8 * sp + 4 * g + 0] + bv.x);
                const float v1 = gelu_erf(gacc[i][8 * sp + 4 * g + 1] + bv.y);
                const float v2 = gelu_erf(gacc[i][8 * sp + 4 * g + 2] + bv.z);
                const float v3 = gelu_erf(gacc[i][8 * sp + 4 * g + 3] + bv.w);
                u[2 * g] = pack2_h16<DT>(v0, v1);
                u[2 * g + 1] = pack2_h16<DT>(v2, v3);
              }
              gf[i][sp] = __builtin_bit_cast(s16x8_t, u);
            }
          }
        }
      } else {
        // ---- fc2: X^T[out tile][row] += W2 tile . G, tile order kt2-major ----
        const int rp = r - KT1;
        const int kt2 = rp / NT2, nt2 = rp - kt2 * NT2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ot = nt2 * 4 + i;
          xacc[ot] = mfma32<DT>(wf[i][0], gf[2 * kt2 + 0][0], xacc[ot]);
          xacc[ot] = mfma32<DT>(wf[i][1], gf[2 * kt2 + 0][1], xacc[ot]);
          xacc[ot] = mfma32<DT>(wf[i][2], gf[2 * kt2 + 1][0], xacc[ot]);
          xacc[ot] = mfma32<DT>(wf[i][3], gf[2 * kt2 + 1][1], xacc[ot]);
        }
      }
    }
  }

  // ---- x[row][col .. col+3] += acc + b2 : lane owns row m, columns 32 ot + 8 g + 4 h + {0..3} ----
  if (m < rows) {
    float* xr = x + m * D + 4 * h;
#pragma unroll
    for (int ot = 0; ot < D / 32; ++ot) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = 32 * ot + 8 * g;
        const float4 bv = *reinterpret_cast<const float4*>(b2 + col + 4 * h);
        float4* p = reinterpret_cast<float4*>(xr + col);
        float4 v = *p;
        v.x += xacc[ot][4 * g + 0] + bv.x; v.y += xacc[ot][4 * g + 1] + bv.y;
        v.z += xacc[ot][4 * g + 2] + bv.z; v.w += xacc[ot][4 * g + 3] + bv.w;
        *p = v;
      }
    }
  }
}

}  // namespace

extern "C" int vittf_mlp_fused(const void* h, const void* w1, const float* b1, const void* w2_perm, const float* b2,
                               float* x, int64_t rows, int32_t d, int32_t dtype, void* stream) {
  if (!h || !w1 || !b1 || !w2_perm || !b2 || !x || rows <= 0) return VITTF_ERR_INVALID_ARG;
  if (d != D) return VITTF_ERR_INVALID_ARG;          // register budget is sized for ViT-S
  const int64_t blocks = (rows + 127) / 128;
  if (blocks > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  const size_t lds = (size_t)RING * TILE;
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* hp = (const unsigned short*)h;
  const unsigned short* w1p = (const unsigned short*)w1;
  const unsigned short* w2p = (const unsigned short*)w2_perm;
  if (dtype == VITTF_BF16) {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)mlp_kernel<VITTF_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    hipLaunchKernelGGL((mlp_kernel<VITTF_BF16>), dim3((unsigned)blocks), dim3(256), lds, st, hp, w1p, b1, w2p, b2, x, rows);
  } else if (dtype == VITTF_FP16) {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)mlp_kernel<VITTF_FP16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    hipLaunchKernelGGL((mlp_kernel<VITTF_FP16>), dim3((unsigned)blocks), dim3(256), lds, st, hp, w1p, b1, w2p, b2, x, rows);
  } else {
    return VITTF_ERR_INVALID_ARG;
  }
  return vittf_check_launch();
}
