// Activation-stationary GEMM for the K = 384 linears with a wide output (the qkv projection of ViT-S):
//     out[rows][n] = epilogue( A[rows][384] . W[n][384]^T + bias[n] ),   16-bit output, epilogue = bias | bias + q scale
// Replaces Attention.qkv of the upstream DINO block (the hooked tensor of /root/reference/infer.py:133-135, reached through
// model(...) at infer.py:177).
//
// The weight-stationary kernel this supersedes for qkv (gemm_ws.hip) re-read its 0.8 GB of activations once per 384-column
// panel (4.78 GB per 256-slice launch by PMC against 3.2 GB algorithmic) and ran one barrier per 32-row tile over 12 waves.
// Here the ACTIVATIONS are the stationary operand: a wave keeps its 32 rows x 384 k as 24 B fragments (96 registers) for all
// n / 32 output tiles, and the 0.9 MB of weights stream through an LDS-DMA ring as 24 KB images in consumption order
// (weights.pack_row_images: every piece 1 KB of contiguous, L2-resident memory), shared by the workgroup's 8 waves = 256 rows.
// Every activation byte is fetched once; two waves per SIMD run their dependent 24-MFMA chains side by side (the rate the main
// steps of the block tail reach: ~41 cycles per MFMA and SIMD, profiles/r05a, r05b).  Per unit (32 output columns): ONE bare
// barrier, 24 MFMAs, the previous unit biased / q-scaled / packed into staging rows in the gaps, stores of 128-byte runs.
// vmcnt retires in order, so the waves that wait for the ring every unit must not have stores in their queue (a first version in
// which every wave did both ran at the store LATENCY: 3000 cycles per unit): waves 0 .. 3 issue all LDS-DMA and never store,
// waves 4 .. 7 store their own rows and, through double-buffered staging rows, those of their SIMD partner, and never wait for
// vector memory inside a tile.  The sums are those of the weight-stationary kernel (one ascending-k chain from zero, then
// (acc + bias) * scale): the outputs are bit-equal.
#include "vittf_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

#ifndef AS_VARIANT      // timing-only builds (tools/gemm_as_ab.py; never in libvittf.so): 1 no stores, 2 no staging, 4 no LDS-DMA in the loop
#define AS_VARIANT 0
#endif
#ifndef AS_STORE_AUX      // cache policy of the output stores: 2 = non-temporal (every byte is written once; the weights live in L2)
#define AS_STORE_AUX 0
#endif
constexpr bool V_NO_STORE = AS_VARIANT & 1, V_NO_STAGE = AS_VARIANT & 2, V_NO_DMA = AS_VARIANT & 4;
constexpr int K = 384;
constexpr int SB = 24576;                    // bytes of one weight image = one ring step: [32 output columns][384 k]
constexpr int NSLOT = 4, AHEAD = NSLOT - 1;
constexpr int PIECES = SB / 1024 / 4;        // LDS-DMA pieces per loader wave (waves 0 .. 3) and step
constexpr int NF = 4;                        // weight fragments in flight per wave
constexpr int STG_ROW = 144, STG = 32 * STG_ROW;      // a wave's staging rows: 128 bytes + 16 (36 banks)
constexpr int STG_OFF = NSLOT * SB;
constexpr int BIAS_OFF = STG_OFF + 12 * STG;   // staging: two buffers per loader wave (pairs alternate), one per storer wave
constexpr int MAXN = 1536;
constexpr int NEXT_OFF = BIAS_OFF + MAXN * 4;
constexpr int LDS_BYTES = NEXT_OFF + 16;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((address_space(3))) const s16x8_t* lds_frag_ptr;
typedef __attribute__((address_space(3))) const f32x4_t* lds_f4_ptr;
typedef __attribute__((address_space(3))) u32x2_t* lds_w2_ptr;
typedef __attribute__((address_space(3))) volatile unsigned* lds_u32_ptr;

__device__ __forceinline__ void lds_dma16_keep(i32x4_t rsrc, unsigned lds_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <int DT>
__global__ __launch_bounds__(512, 1) void gemm_as_kernel(const unsigned short* __restrict__ abuf, const unsigned short* __restrict__ wpk,
                                                         const float* __restrict__ bias, unsigned short* __restrict__ out, int64_t rows,
                                                         int n, int qcols, int ntiles, unsigned* __restrict__ tile_ctr) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31, q8 = lane >> 3, c8 = lane & 7;
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
  float* const cst = reinterpret_cast<float*>(smem + BIAS_OFF);
  for (int i = tid; i < n; i += 512) cst[i] = bias[i];
  const unsigned nxt = lds0 + NEXT_OFF;
  if (tid == 0) *(lds_u32_ptr)nxt = atomicAdd(tile_ctr, 1u);
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane((int)*(lds_u32_ptr)nxt);
  if (tile >= ntiles) return;
  const int units = n >> 5;                    // 32-column output tiles = steps of the weight stream per row tile
  const i32x4_t rsrc = lds_dma_rsrc(wpk, (unsigned)(units * SB));
  const bool storer = wave >= 4;
  const int src0 = (wave & 3) * (PIECES * 1024);
  // the first AHEAD images
  if (!storer) {
#pragma unroll
    for (int u = 0; u < AHEAD; ++u)
#pragma unroll
      for (int i = 0; i < PIECES; ++i)
        lds_dma16(rsrc, lds0 + u * SB + src0 + i * 1024, lane * 16, u * SB + src0 + i * 1024);
  }
  int g = 0, slot = 0;                         // stream position / ring slot of the unit being computed
  const int aoff0 = tile_off(l31, h);
  unsigned base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) base[i] = lds0 + (aoff0 ^ (32 * i));
  // staging rows: loader wave a: buffers (pair & 1) at (2 a + (pair & 1)) STG; storer wave 4 + b: (8 + b) STG
  const unsigned stg = lds0 + STG_OFF + (storer ? 8 + (wave - 4) : 2 * wave) * STG;
  const unsigned stg_w = stg + l31 * STG_ROW + 8 * h;      // a row per lane: + 64 (unit & 1) + 16 g (+ STG for a loader's odd pairs)
  const unsigned stg_r = stg + q8 * STG_ROW + c8 * 16;     // eight lanes per row: + 8 i rows
  const unsigned stg_rp = lds0 + STG_OFF + 2 * (wave & 3) * STG + q8 * STG_ROW + c8 * 16;      // (storer) its partner's buffers
  unsigned cl = lds0 + BIAS_OFF + 16 * h;
  asm volatile("" : "+v"(cl));
  constexpr float QS = 0.125f * 1.44269504088896340736f;   // log2(e) / 8 on the q third (the attention kernels work in exp2)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  s16x8_t wf[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) wf[f] = *(lds_frag_ptr)(base[f & 3] + (f >> 2) * 4096);
  const int row_bytes_out = n * 2;

  while (true) {
    unsigned next_v = 0;
    if (tid == 256) next_v = atomicAdd(tile_ctr, 1u);
    // this wave's 32 rows as 24 B fragments: A[row][16 s + 8 h .. + 7]; rows past the end read as zero / are not written
    const int64_t first = (int64_t)tile * 256, left = rows - first;
    const int nrows = left < 256 ? (int)left : 256;
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(abuf) + first * K, 0, nrows * (K * 2), 0x00020000);
    const auto ro = __builtin_amdgcn_make_buffer_rsrc(out + first * n, 0, nrows * row_bytes_out, 0x00020000);
    s16x8_t hq[K / 16];
#pragma unroll
    for (int s = 0; s < K / 16; ++s)
      hq[s] = __builtin_bit_cast(s16x8_t, __builtin_amdgcn_raw_buffer_load_b128(ra, (wave * 32 + l31) * (K * 2) + 16 * h, 32 * s, 0));
    const int oo = (wave * 32 + q8) * row_bytes_out + c8 * 16;      // rows 8 i + lane / 8, byte column 128 (unit >> 1) + 16 (lane % 8)
    f32x16_t qa = {}, qb = {};
    // a finished unit: (acc + bias) * scale -> 16 bit -> this wave's staging rows (columns 64 (unit & 1) ..)
    // (its bias tile -- register r of lane half h = bias[32 u + (r & 3) + 8 (r >> 2) + 4 h] -- was read from LDS at the start of
    //  the unit itself, a whole unit before it is used: a read inside the gap would cost the in-order wave an LDS round trip)
    f32x4_t ba[4], bb[4];
    auto load_bias = [&](int u, f32x4_t (&b)[4]) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) b[gq] = *(lds_f4_ptr)(cl + 4 * (32 * u + 8 * gq));
    };
    auto stage = [&](const f32x16_t& q, const f32x4_t (&bq)[4], int u, int gq) {
      const f32x4_t b = bq[gq];
      const float qsc = 32 * u < qcols ? QS : 1.f;
      u32x2_t pk;
      pk[0] = pack2_h16<DT>((q[4 * gq + 0] + b[0]) * qsc, (q[4 * gq + 1] + b[1]) * qsc);
      pk[1] = pack2_h16<DT>((q[4 * gq + 2] + b[2]) * qsc, (q[4 * gq + 3] + b[3]) * qsc);
      *(lds_w2_ptr)(stg_w + (storer ? 0 : ((u >> 1) & 1) * STG) + 64 * (u & 1) + 16 * gq) = pk;
    };
    // (storer waves) units 2 m, 2 m + 1 = 64 columns = 128 bytes of a row: the wave's own rows, or its partner's (32 rows up)
    auto store_pair = [&](int m, bool partner) {
      if (V_NO_STORE || !storer) return;
      const unsigned rd = partner ? stg_rp + (m & 1) * STG : stg_r;
      f32x4_t rbk[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) rbk[i] = *(lds_f4_ptr)(rd + i * 8 * STG_ROW);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, rbk[i]), ro, oo + (partner ? -128 * row_bytes_out : 0) + m * 128,
                                               i * 8 * row_bytes_out, AS_STORE_AUX);
    };
    // one unit: 24 MFMAs acc = W(u) . A^T from zero; the unit before it staged in the gaps.  WAITN: vector-memory operations
    // of this wave that may be in flight at its start (the pieces of the step before, stores, the row loads of a new tile).
    auto unit = [&](auto WAITc, f32x16_t& acc, f32x4_t (&bcur)[4], const f32x16_t& prevq, const f32x4_t (&bprev)[4], int u, bool stage_prev) {
      constexpr int WAITN = decltype(WAITc)::value;
      load_bias(u, bcur);
      if (storer) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(V_NO_DMA ? 0 : WAITN) : "memory");
      if (u >= 3 && (u & 1)) store_pair((u - 3) >> 1, true);      // the partner's pair, staged during the last two units
      const int g_next = g + AHEAD < units ? g + AHEAD : g + AHEAD - units;
      const int slot_free = slot == 0 ? NSLOT - 1 : slot - 1;
#pragma unroll
      for (int j = 0; j < 24; ++j) {
        acc = mfma32<DT>(wf[j % NF], hq[j], j == 0 ? f32x16_t{} : acc);
        if (j == 24 - NF) {                    // from here on the refills read the next unit's image
          const int d_ = slot == NSLOT - 1 ? -(NSLOT - 1) * SB : SB;
#pragma unroll
          for (int i = 0; i < 4; ++i) { base[i] += d_; asm volatile("" : "+v"(base[i])); }
        }
        wf[j % NF] = *(lds_frag_ptr)(base[((j + NF) % 24) & 3] + (((j + NF) % 24) >> 2) * 4096);
        if (!V_NO_DMA && !storer && j % 4 == 3) lds_dma16_keep(rsrc, lds0 + slot_free * SB + src0 + (j >> 2) * 1024, lane * 16, g_next * SB + src0 + (j >> 2) * 1024);
        if (!V_NO_STAGE && stage_prev && j % 6 == 2) stage(prevq, bprev, u - 1, j / 6);
        __builtin_amdgcn_sched_barrier(0);
      }
      g = g + 1 == units ? 0 : g + 1;
      slot = slot + 1 == NSLOT ? 0 : slot + 1;
    };
    // Loader waves, at the start of unit u: the pieces issued during unit u - 2 (the image of unit u + 1, whose first fragments
    // are read behind this unit's last MFMAs) have landed; the six of unit u - 1 may be in flight (+ a new tile's row loads).
    using W6 = std::integral_constant<int, PIECES>;
    using WT = std::integral_constant<int, PIECES + K / 16>;
    unit(WT{}, qa, ba, qb, bb, 0, false);
    unit(W6{}, qb, bb, qa, ba, 1, true);
    for (int u = 2; u < units; u += 2) {
      unit(W6{}, qa, ba, qb, bb, u, true);     // unit u - 1 staged: the pair u - 2, u - 1 is complete
      store_pair((u - 2) >> 1, false);
      unit(W6{}, qb, bb, qa, ba, u + 1, true);
    }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) stage(qb, bb, units - 1, gq);
    store_pair((units - 2) >> 1, false);
    if (tid == 256) *(lds_u32_ptr)nxt = next_v;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    store_pair((units - 2) >> 1, true);       // the partner's last pair (its second unit was staged behind the last unit)
    const int next = __builtin_amdgcn_readfirstlane((int)*(lds_u32_ptr)nxt);
    if (next >= ntiles) break;
    tile = next;      // (thread 0 writes the word again a whole tile of barriers later)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the images requested beyond the last unit: land before the LDS goes away
}

}  // namespace

#ifdef AS_STANDALONE
void vittf_note_kernel(int, const char*) {}
#endif

extern "C" size_t vittf_gemm_as_workspace_bytes(void) { return sizeof(unsigned); }

// out = epilogue(a . w^T + bias) for k = 384, n a multiple of 64 up to 1536; w_packed = the n / 32 images of 24 KB
// (weights.pack_row_images); epilogue VITTF_EPI_BIAS or VITTF_EPI_BIAS_QKV (the first n / 3 columns times log2(e) / 8).
// tile_counter: vittf_gemm_as_workspace_bytes() of caller-owned device memory (zeroed by the call), one per concurrent stream.
extern "C" int vittf_gemm_as(const void* a, const void* w_packed, const float* bias, void* out, int64_t rows, int32_t n, int32_t k,
                             int32_t epilogue, int32_t dtype, void* tile_counter, void* stream) {
  if (!a || !w_packed || !bias || !out || rows <= 0) return VITTF_ERR_INVALID_ARG;
  if (k != K || n <= 0 || n % 64 != 0 || n > MAXN) return VITTF_ERR_INVALID_ARG;
  if (epilogue != VITTF_EPI_BIAS && epilogue != VITTF_EPI_BIAS_QKV) return VITTF_ERR_INVALID_ARG;
  if (epilogue == VITTF_EPI_BIAS_QKV && n % 96 != 0) return VITTF_ERR_INVALID_ARG;
  if ((((uintptr_t)a | (uintptr_t)w_packed | (uintptr_t)out | (uintptr_t)bias) & 15) != 0) return VITTF_ERR_INVALID_ARG;
  if (!tile_counter || ((uintptr_t)tile_counter & 3) != 0) return VITTF_ERR_INVALID_ARG;
  const int64_t tiles = (rows + 255) / 256;
  if (tiles > 0x7fffff) return VITTF_ERR_INVALID_ARG;
  const int cus = vittf_current_cus();
  if (cus <= 0) return VITTF_ERR_NO_DEVICE;
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  hipStream_t st = (hipStream_t)stream;
  unsigned* ctr = (unsigned*)tile_counter;
  if (hipMemsetAsync(ctr, 0, sizeof(unsigned), st) != hipSuccess) return VITTF_ERR_LAUNCH;
  const int qcols = epilogue == VITTF_EPI_BIAS_QKV ? n / 3 : 0;
#define AS_LAUNCH(DTV)                                                                                               \
  hipLaunchKernelGGL((gemm_as_kernel<DTV>), dim3(grid), dim3(512), 0, st, (const unsigned short*)a,                  \
                     (const unsigned short*)w_packed, bias, (unsigned short*)out, rows, (int)n, qcols, (int)tiles, ctr)
  if (dtype == VITTF_BF16) AS_LAUNCH(VITTF_BF16);
  else if (dtype == VITTF_FP16) AS_LAUNCH(VITTF_FP16);
  else return VITTF_ERR_INVALID_ARG;
#undef AS_LAUNCH
  vittf_note_kernel(VITTF_KERNEL_GEMM_QKV, "gemm_as_kernel");
  return vittf_check_launch();
}
