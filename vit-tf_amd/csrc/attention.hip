// Flash-style multi-head self-attention forward, head dim 64, for the DINO ViT blocks.
//
// Replaces Attention.forward of the upstream model the reference calls (infer.py:177):
//   softmax(q k^T / sqrt(64)) v  per head, without materialising the N x N score matrix.
//
// Machine mapping (gfx950, wave64):
//   * one 256-thread workgroup = 4 waves = 128 query rows of one (slice, head); each wave owns 32 query rows
//   * K/V tiles of 64 keys are register-staged (buffer_load_dwordx4 issued before the tile's MFMAs, ds_write
//     after them) into two LDS buffers, one barrier per tile; the buffer descriptor's range check zero-fills rows
//     past the end of the slice, the tile offset rides in the scalar offset, so the prefetch costs no VALU
//   * scores are computed TRANSPOSED, S^T = K Q^T with v_mfma_f32_32x32x16 (K rows as the A operand, Q rows as
//     the B operand, Q fragments live in registers for the whole kernel), so a lane owns one query column:
//     the row maximum / row sum of the online softmax are in-lane reductions plus ONE v_permlane32_swap
//   * the S^T accumulator registers, converted pairwise to 16 bit (v_cvt_pk), are directly the B operand of the
//     second product O^T = V^T P^T (k order 16s + 8(j>>2) + 4h + (j&3)); the matching V^T A-fragments come from
//     ds_read_b64_tr_b16 transposed reads of a row-major V image (8-row x 32-col subtiles, conflict free)
//   * the K image uses the tile_off() swizzle shared with the GEMM and is read with ds_read_b128
//   * every LDS address is a per-lane base computed once + an immediate (the two buffers are two instantiations
//     of the tile body): the kernel is VALU-bound at head dim 64 (PMC: VALU 75 % busy vs MFMA 38 %), so address
//     arithmetic inside the loop is what was removed first
//   * exp2 with the softmax scale folded into one FMA: p = exp2(s*c - m*c), c = log2(e)/8
//   * token count need not be tile aligned (N = f0*f1 + 1): the last key tile is masked to -inf, query rows
//     past the end are clamped on load and their stores are guarded
//   * workgroups are remapped so that the q-tiles of one (slice, head) share an XCD's L2 (K/V re-reads)
#include "attn_common.h"

#include <stdlib.h>

namespace {

constexpr int QT = 128;   // query rows per workgroup
constexpr int KT = 64;    // keys per tile
constexpr int KV_TILE_BYTES = KT * 64 * 2;  // 8 KB
constexpr int BUF_BYTES = 2 * KV_TILE_BYTES; // K | V

// One 64-key tile for this wave's 32 query rows.  BUF selects the LDS buffer at compile time so that every
// ds_read offset is an immediate on one of six per-lane base registers.  LAST masks keys >= tokens.
template <int DT, int BUF, bool LAST>
__device__ __forceinline__ void attn_tile(const char* ka0, const char* ka1, const char* ka2, const char* ka3,
                                          const char* va0, const char* va1, const s16x8_t& q0, const s16x8_t& q1,
                                          const s16x8_t& q2, const s16x8_t& q3, f32x16_t& o0, f32x16_t& o1,
                                          float& m_run, float& l_run, int t, int tokens, int h, float c) {
  // Two 32-key halves, each carried from S^T to O^T: the score registers (16) and the P fragments (8) of one half are
  // (nearly) all that is live.  The halves are NOT fenced off from each other any more: within the register budget of
  // the launch bounds hipcc sinks the O^T MFMAs of one half into the exp2 stream of the next (an `s_nop 10` behind the
  // S^T chain otherwise idles the wave): -1.5 % per launch in the pipeline, no spills (168 / 128 VGPRs).
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    constexpr int kb = BUF * BUF_BYTES;
    f32x16_t sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    sacc = mfma32<DT>(*reinterpret_cast<const s16x8_t*>(ka0 + kb + 4096 * kt), q0, sacc);
    sacc = mfma32<DT>(*reinterpret_cast<const s16x8_t*>(ka1 + kb + 4096 * kt), q1, sacc);
    sacc = mfma32<DT>(*reinterpret_cast<const s16x8_t*>(ka2 + kb + 4096 * kt), q2, sacc);
    sacc = mfma32<DT>(*reinterpret_cast<const s16x8_t*>(ka3 + kb + 4096 * kt), q3, sacc);
    if constexpr (LAST) {  // ragged last tile: keys >= tokens contribute nothing
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * KT + 32 * kt + acc_row(r, h);
        if (key >= tokens) sacc[r] = -INFINITY;
      }
    }

    float p[16];
    // ---- online softmax (lane = one query column; 16 of the half's 32 keys are in this lane) ----
    float tmax = max3_f32(sacc[0], sacc[1], sacc[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = max3_f32(tmax, sacc[r], sacc[r + 1]);
    tmax = max3_f32(tmax, sacc[15], sacc[15]);
    float m_new;
    {
      const unsigned tb = __float_as_uint(tmax);
      const auto sw = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);   // one of the two holds the other half
      m_new = max3_f32(m_run, __uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float mc = m_new * c;
    if (!__all(m_new == m_run)) {   // rare after the first tiles: rescale what was accumulated at the old maximum
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      m_run = m_new;
    }
    float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      p[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], c, -mc));
      p[r + 1] = __builtin_amdgcn_exp2f(fmaf(sacc[r + 1], c, -mc));
      psum0 += p[r];
      psum1 += p[r + 1];
    }
    l_run += psum0 + psum1;
    s16x8_t pf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      u32x4_t u;
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = pack2_h16<DT>(p[8 * s2 + 2 * j], p[8 * s2 + 2 * j + 1]);
      pf[s2] = __builtin_bit_cast(s16x8_t, u);
    }

    // ---- O^T += V^T P^T for these 32 keys ----
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int dvt = 0; dvt < 2; ++dvt) {
        constexpr int vb = BUF * BUF_BYTES + KV_TILE_BYTES;
        const int imm = vb + 4096 * kt + 2048 * s2 + 512 * dvt;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(va0 + imm));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(va1 + imm + 1024));
        const s16x8_t vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        if (dvt == 0) o0 = mfma32<DT>(vf, pf[s2], o0);
        else          o1 = mfma32<DT>(vf, pf[s2], o1);
      }
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256, 4) void attn_kernel(const unsigned short* __restrict__ qkv,
                                                      unsigned short* __restrict__ out, int tokens, int heads,
                                                      int q_tiles, int total, float c) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];  // [buffer][K | V]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;

  const int item = xcd_remap(blockIdx.x, total);
  const int qt = item % q_tiles;
  const int bh = item / q_tiles;
  const int hd = bh % heads, b = bh / heads;
  const int dmodel = heads * 64;
  const int ld = 3 * dmodel;                                   // elements per token row of qkv
  const unsigned short* base = qkv + (int64_t)b * tokens * ld;

  // buffer descriptor over this slice's qkv rows: loads past the last token return 0 (no clamping VALU)
  const i32x4_t rsrc = lds_dma_rsrc(base, (unsigned)((int64_t)tokens * ld * 2));

  // ---- Q fragments (B operand): lane holds Q[row l31][16 s + 8 h .. +7] ----
  const int qrow = qt * QT + wave * 32 + l31;
  const int qrow_c = qrow < tokens ? qrow : tokens - 1;
  const unsigned short* qp = base + (int64_t)qrow_c * ld + hd * 64 + 8 * h;
  s16x8_t q0 = *reinterpret_cast<const s16x8_t*>(qp);
  s16x8_t q1 = *reinterpret_cast<const s16x8_t*>(qp + 16);
  s16x8_t q2 = *reinterpret_cast<const s16x8_t*>(qp + 32);
  s16x8_t q3 = *reinterpret_cast<const s16x8_t*>(qp + 48);

  // ---- staging: LDS-DMA (buffer_load_dwordx4 ... lds), no staging registers and no ds_write ----
  // A tile image is 512 16-byte chunks per operand; wave-instruction i of wave w fills linear chunks
  // [i*256 + w*64, +64).  The LDS destination of an LDS-DMA is lane-linear, so the swizzles of tile_off() /
  // v_off() are applied by choosing which (row, chunk) each lane FETCHES.
  int voff_k0, voff_k1, voff_v0, voff_v1;
  {
    int r, cc;
    tile_pos(tid, r, cc);
    voff_k0 = (r * ld + dmodel + hd * 64 + cc * 8) * 2;
    tile_pos(256 + tid, r, cc);
    voff_k1 = (r * ld + dmodel + hd * 64 + cc * 8) * 2;
    // inverse of v_off: chunk position q = 64 (key>>3) + 32 (ch>>2) + 4 (key&7) + ((ch&3) ^ ((key>>2)&3))
    auto v_src = [&](int q) {
      const int kg = q >> 6, half = (q >> 5) & 1, k7 = (q >> 2) & 7, x = q & 3;
      const int key = 8 * kg + k7;
      const int ch = 4 * half + (x ^ ((key >> 2) & 3));
      return (key * ld + 2 * dmodel + hd * 64 + ch * 8) * 2;
    };
    voff_v0 = v_src(tid);
    voff_v1 = v_src(256 + tid);
  }
  const int tile_stride = KT * ld * 2;
  const int nt = (tokens + KT - 1) / KT;
  // wave-uniform LDS byte address; the hardware adds lane * 16.  (asm pieces, see lds_dma16: with the builtin hipcc
  // put s_waitcnt vmcnt(0) in front of the V reads in the middle of every tile.)
  const unsigned dma_dst = (unsigned)(size_t)LDS_PTR(smem) + (__builtin_amdgcn_readfirstlane(tid & ~63) << 4);
  // The LAST tile (the only one that can reach past the slice's rows) carries its tile offset in the per-lane voffset:
  // that is the operand the descriptor's range check is documented to cover, so rows >= tokens arrive as zeros whatever
  // lies behind the slice (the next slice's rows, or uninitialised workspace whose NaN / Inf bit patterns would turn
  // P = 0 times V into NaN).  Every other tile keeps the offset in the scalar operand: no VALU on the hot path.
#define ATTN_STAGE_TILE(t, BUFI)                                                                              \
  {                                                                                                           \
    const int so_ = (t) * tile_stride;                                                                        \
    if ((t) == nt - 1) {                                                                                      \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES, voff_k0 + so_, 0);                                           \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + 4096, voff_k1 + so_, 0);                                    \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES, voff_v0 + so_, 0);                           \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES + 4096, voff_v1 + so_, 0);                    \
    } else {                                                                                                  \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES, voff_k0, so_);                                               \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + 4096, voff_k1, so_);                                        \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES, voff_v0, so_);                               \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES + 4096, voff_v1, so_);                        \
    }                                                                                                         \
  }

  // ---- per-lane LDS read bases (see tile_off / v_off: the kt, s2, dvt, jj and buffer terms are immediates) ----
  const int p_l = l31 >> 1;
  const int bslot = (((l31 & 1) << 3) | h) ^ (p_l & 15);
  const char* const ka0 = smem + (p_l << 8) + ((bslot ^ 0) << 4);
  const char* const ka1 = smem + (p_l << 8) + ((bslot ^ 2) << 4);
  const char* const ka2 = smem + (p_l << 8) + ((bslot ^ 4) << 4);
  const char* const ka3 = smem + (p_l << 8) + ((bslot ^ 6) << 4);
  const int g16 = lane >> 4;                 // 16-lane group 0..3
  const int tr_q = (lane & 15) >> 2;         // row inside the 4-row block
  const int tr_p = lane & 3;
  const int tr_ch = 2 * (g16 & 1) + (tr_p >> 1);
  const int vl0 = 64 * (4 * h + tr_q) + 16 * (tr_ch ^ h) + 8 * (tr_p & 1);
  const char* const va0 = smem + vl0;          // jj = 0
  const char* const va1 = smem + (vl0 ^ 32);   // jj = 1: (key >> 2) & 3 gains 2 -> chunk index ^ 2

  f32x16_t o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  ATTN_STAGE_TILE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // Retire the Q loads here: otherwise hipcc's waitcnt pass re-waits for the Q registers inside the loop.
  asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));

  // N = f0*f1 + 1 leaves the last q-tile with a single valid row: waves whose 32 rows are all past the end keep
  // staging and synchronising but skip the arithmetic (3 of 4 waves in 1 of 33 workgroups at N = 4097)
  const bool active = __builtin_amdgcn_readfirstlane(qt * QT + wave * 32) < tokens;
  int t = 0;
  if (!active) {   // a separate loop: with `if (active)` around the tile body hipcc keeps the 32 output accumulators in
                   // two register sets and copies them at every loop head (32 v_mov per tile on the hot path)
    for (; t + 1 < nt; ++t) {
      if (t & 1) ATTN_STAGE_TILE(t + 1, 0) else ATTN_STAGE_TILE(t + 1, 1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    return;
  }
  // every tile but the last: the DMA of tile t + 1 flies under the MFMAs of t.  Two tiles per trip, straight-line (an
  // `if (t & 1)` diamond makes hipcc give the two ring-buffer variants different accumulator registers + copies)
#define ATTN_STEP(BUFC, BUFN)                                                                                          \
  {                                                                                                                    \
    ATTN_STAGE_TILE(t + 1, BUFN)                                                                                       \
    attn_tile<DT, BUFC, false>(ka0, ka1, ka2, ka3, va0, va1, q0, q1, q2, q3, o0, o1, m_run, l_run, t, tokens, h, c); \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* this wave's DMA pieces have landed ... */                    \
    __syncthreads();                                    /* ... and everybody's have, and everybody is done reading */  \
    ++t;                                                                                                               \
  }
  while (t + 2 < nt) {
    ATTN_STEP(0, 1)
    ATTN_STEP(1, 0)
  }
  if (t + 1 < nt) ATTN_STEP(0, 1)
#undef ATTN_STEP
  if (t & 1) attn_tile<DT, 1, true>(ka0, ka1, ka2, ka3, va0, va1, q0, q1, q2, q3, o0, o1, m_run, l_run, t, tokens, h, c);
  else       attn_tile<DT, 0, true>(ka0, ka1, ka2, ka3, va0, va1, q0, q1, q2, q3, o0, o1, m_run, l_run, t, tokens, h, c);

  // ---- normalise and store: lane owns query row `qrow`, columns 32 dvt + 8 g + 4 h + {0..3} ----
  float l_tot;
  {
    const unsigned lb = __float_as_uint(l_run);
    const auto sw = __builtin_amdgcn_permlane32_swap(lb, lb, false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.0f / l_tot;
  if (qrow < tokens) {
    unsigned short* orow = out + ((int64_t)b * tokens + qrow) * dmodel + hd * 64 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 pk;
      pk.x = pack2_h16<DT>(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv);
      pk.y = pack2_h16<DT>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(orow + 8 * g) = pk;
      pk.x = pack2_h16<DT>(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv);
      pk.y = pack2_h16<DT>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(orow + 32 + 8 * g) = pk;
    }
  }
#undef ATTN_STAGE_TILE
}

}  // namespace

int vittf_attention_pp64(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype,
                         hipStream_t st);                                                          // attention_pp64.hip

extern "C" int vittf_attention(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads,
                               int32_t dtype, int32_t q_prescaled, void* stream) {
  if (!qkv || !out || batch <= 0 || tokens <= 0 || heads <= 0) return VITTF_ERR_INVALID_ARG;
  // 32-bit byte offsets inside one slice's qkv rows (buffer addressing)
  if ((int64_t)(tokens + KT) * heads * 64 * 3 * 2 > 0x7fffffffLL) return VITTF_ERR_INVALID_ARG;
  const int q_tiles = (tokens + QT - 1) / QT;
  const int64_t total64 = (int64_t)batch * heads * q_tiles;
  if (total64 > (1 << 30)) return VITTF_ERR_INVALID_ARG;
  const int total = (int)total64;
  const float c = 0.125f * 1.44269504088896340736f;
  hipStream_t st = (hipStream_t)stream;
  if (q_prescaled) {
    // pre-scaled q (what the engine runs): attention_pp64.hip, two 32-row query blocks per wave taking turns, two waves per SIMD
    if (dtype != VITTF_BF16 && dtype != VITTF_FP16) return VITTF_ERR_INVALID_ARG;
    vittf_note_kernel(VITTF_KERNEL_ATTENTION, "attn_pp64_kernel");
    return vittf_attention_pp64(qkv, out, batch, tokens, heads, dtype, st);
  }
  // q as the model produces it: the online-maximum kernel of this file
#define VITTF_ATTN_LAUNCH(DTV)                                                                              \
  hipLaunchKernelGGL((attn_kernel<DTV>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv,  \
                     (unsigned short*)out, tokens, heads, q_tiles, total, c)
  vittf_note_kernel(VITTF_KERNEL_ATTENTION, "attn_kernel<online maximum>");
  if (dtype == VITTF_BF16) VITTF_ATTN_LAUNCH(VITTF_BF16);
  else if (dtype == VITTF_FP16) VITTF_ATTN_LAUNCH(VITTF_FP16);
  else return VITTF_ERR_INVALID_ARG;
#undef VITTF_ATTN_LAUNCH
  return vittf_check_launch();
}
