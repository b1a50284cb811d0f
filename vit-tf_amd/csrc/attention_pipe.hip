// Software-pipelined flash attention forward, head dim 64, pre-scaled q (the engine's default since round 2).
//
// Same contract as attention.hip (softmax(q k^T / 8) v per head; Attention.forward of the upstream model the
// reference calls, infer.py:177), same LDS images, same lazy running maximum -- but the three stages of a 32-key
// half step no longer run one after the other inside a wave.  At head dim 64 a 32 x 32 score block costs 8 MFMAs
// (256 matrix-pipe cycles) against 16 v_exp + 16 adds + 8 v_cvt_pk + the MFMAs' own issue slots (~290 VALU-port
// cycles): whichever pipe a wave is not using idles unless ANOTHER wave happens to be in the opposite phase (round 1:
// 52 cycles per MFMA, the two pipes co-executing 21 % of the time).  Here every wave carries two independent
// instruction streams through each slot h:
//
//     matrix pipe :  S(h+1) = K(h+1) Q^T - M        and        O += V(h-1)^T P(h-1)^T
//     VALU        :  P(h) = exp2(S(h)), row sums, the overflow check, 16-bit packing
//
// so the scores of the next half step and the output product of the previous one run under the softmax arithmetic
// of the current one (S and the packed P are double-buffered in registers: +24 VGPRs).
//
//   * K/V tiles of 64 keys in a 3-deep LDS ring (48 KB per workgroup): while tile t is being
//     soft-maxed, the score MFMAs already read tile t+1 and the output MFMAs still read tile t-1.  ONE barrier per
//     tile, placed between its two slots: behind it nobody reads tile t-1 any more, so the LDS-DMA of tile t+2 is
//     issued there into the same buffer and has a whole tile of arithmetic to land.
//   * the lazy maximum's slow path (a row sum says the 16-bit P would overflow: rare, wave-uniform) sits at the END of
//     a slot, so that between two checks there is one long basic block the scheduler can interleave.  It rebuilds the
//     half step from LDS (raw scores, true maximum), rescales O, l and the -M tile, and shifts the already computed
//     S(h+1) to the new M.
//   * everything else as in attention.hip: S^T = K Q^T so a lane owns one query column; the P registers feed
//     O^T = V^T P^T directly; V^T fragments by ds_read_b64_tr_b16; the ragged last tile is range-checked by the
//     buffer descriptor (its tile offset in the per-lane voffset) and masked to -inf.
#include "attn_common.h"

namespace {

constexpr int QT = 128;   // query rows per workgroup (4 waves x 32)
constexpr int KT = 64;    // keys per tile
constexpr int KV_TILE_BYTES = KT * 64 * 2;   // 8 KB
constexpr int BUF_BYTES = 2 * KV_TILE_BYTES;  // K | V
constexpr int NBUF = 3;
#ifndef PIPE_WAVES
#define PIPE_WAVES 2
#endif

struct LdsBases {
  const char *ka0, *ka1, *ka2, *ka3, *va0, *va1;
};

// S^T(32 keys x 32 queries) = K(half HALF of ring buffer BUF) Q^T + c
template <int DT, int BUF, int HALF>
__device__ __forceinline__ f32x16_t score_mfma(const LdsBases& b, const s16x8_t& q0, const s16x8_t& q1, const s16x8_t& q2,
                                               const s16x8_t& q3, f32x16_t c) {
  constexpr int off = BUF * BUF_BYTES + 4096 * HALF;
  c = mfma32<DT>(*reinterpret_cast<const s16x8_t*>(b.ka0 + off), q0, c);
  c = mfma32<DT>(*reinterpret_cast<const s16x8_t*>(b.ka1 + off), q1, c);
  c = mfma32<DT>(*reinterpret_cast<const s16x8_t*>(b.ka2 + off), q2, c);
  c = mfma32<DT>(*reinterpret_cast<const s16x8_t*>(b.ka3 + off), q3, c);
  return c;
}

// O^T(64 dims x 32 queries) += V^T(half HALF of ring buffer BUF) P^T
template <int DT, int BUF, int HALF>
__device__ __forceinline__ void out_mfma(const LdsBases& b, const s16x8_t& pf0, const s16x8_t& pf1, f32x16_t& o0,
                                         f32x16_t& o1) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
    for (int dvt = 0; dvt < 2; ++dvt) {
      constexpr int vb = BUF * BUF_BYTES + KV_TILE_BYTES + 4096 * HALF;
      const int imm = vb + 2048 * s2 + 512 * dvt;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(b.va0 + imm));
      const s16x4_t hi =
          __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(b.va1 + imm + 1024));
      const s16x8_t vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      if (dvt == 0) o0 = mfma32<DT>(vf, s2 == 0 ? pf0 : pf1, o0);
      else          o1 = mfma32<DT>(vf, s2 == 0 ? pf0 : pf1, o1);
    }
  }
}

template <int DT> __device__ __forceinline__ void pack_p(const float (&p)[16], s16x8_t& pf0, s16x8_t& pf1) {
  u32x4_t u0, u1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u0[j] = pack2_h16<DT>(p[2 * j], p[2 * j + 1]);
    u1[j] = pack2_h16<DT>(p[8 + 2 * j], p[8 + 2 * j + 1]);
  }
  pf0 = __builtin_bit_cast(s16x8_t, u0);
  pf1 = __builtin_bit_cast(s16x8_t, u1);
}

// per-wave running state
struct AttnState {
  f32x16_t o0, o1, negm;
  float l_run;
};

// Slot h.  Matrix pipe: S(h+1) from (KBUF, KHALF) into s_next [DO_S], O += P(h-1) V(h-1) from (VBUF, VHALF) [DO_O].
// VALU: softmax of s_cur = S(h) (its K half is (CBUF, CHALF), needed again only on the slow path) -> p, then packed
// into pf_cur by the caller-visible tail.  MASK: the half belongs to the ragged last tile.
template <int DT, int CBUF, int CHALF, int KBUF, int KHALF, int VBUF, int VHALF, bool DO_S, bool DO_O, bool MASK>
__device__ __forceinline__ void attn_slot(const LdsBases& b, const s16x8_t& q0, const s16x8_t& q1, const s16x8_t& q2,
                                          const s16x8_t& q3, AttnState& st, f32x16_t& s_cur, f32x16_t& s_next,
                                          const s16x8_t& pfp0, const s16x8_t& pfp1, s16x8_t& pfc0, s16x8_t& pfc1, int key0,
                                          int tokens, int h) {
  constexpr float THR = DT == VITTF_FP16 ? 8192.f : 1073741824.f;
  if constexpr (DO_S) s_next = score_mfma<DT, KBUF, KHALF>(b, q0, q1, q2, q3, st.negm);
  if constexpr (DO_O) out_mfma<DT, VBUF, VHALF>(b, pfp0, pfp1, st.o0, st.o1);
  if constexpr (MASK) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (key0 + acc_row(r, h) >= tokens) s_cur[r] = -INFINITY;
  }
  float p[16];
  float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    p[r] = __builtin_amdgcn_exp2f(s_cur[r]);
    p[r + 1] = __builtin_amdgcn_exp2f(s_cur[r + 1]);
    psum0 += p[r];
    psum1 += p[r + 1];
  }
  float ps = psum0 + psum1;
  if (__builtin_expect(__any(!(ps <= THR)), 0)) {
    // ---- slow path: the half's values have outgrown the 16-bit P at the current M.  Raw scores again from LDS, the
    // true row maximum, everything accumulated so far rescaled to the new M. ----
    f32x16_t zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    f32x16_t raw = score_mfma<DT, CBUF, CHALF>(b, q0, q1, q2, q3, zero);
    if constexpr (MASK) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (key0 + acc_row(r, h) >= tokens) raw[r] = -INFINITY;
    }
    float tmax = max3_f32(raw[0], raw[1], raw[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = max3_f32(tmax, raw[r], raw[r + 1]);
    tmax = fmaxf(tmax, raw[15]);
    const unsigned tb = __float_as_uint(tmax);
    const auto sw = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);
    tmax = max3_f32(tmax, __uint_as_float(sw[0]), __uint_as_float(sw[1]));      // both lane halves agree
    const float m_old = -st.negm[0];
    const float m_new = fmaxf(tmax, m_old);
    const float delta = m_new - m_old;                                          // >= 0, per query column
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    st.l_run *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st.o0[r] *= alpha;
      st.o1[r] *= alpha;
      st.negm[r] = -m_new;
      if constexpr (DO_S) s_next[r] -= delta;                                   // S(h+1) was formed with the old M
    }
    psum0 = 0.f; psum1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      p[r] = __builtin_amdgcn_exp2f(raw[r] - m_new);
      p[r + 1] = __builtin_amdgcn_exp2f(raw[r + 1] - m_new);
      psum0 += p[r];
      psum1 += p[r + 1];
    }
    ps = psum0 + psum1;
  }
  st.l_run += ps;
  pack_p<DT>(p, pfc0, pfc1);
}

template <int DT>
__global__ __launch_bounds__(256, PIPE_WAVES) void attn_pipe_kernel(const unsigned short* __restrict__ qkv,
                                                           unsigned short* __restrict__ out, int tokens, int heads,
                                                           int q_tiles, int total) {
  __shared__ __attribute__((aligned(16))) char smem[NBUF * BUF_BYTES];  // [ring slot][K | V]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;

  const int item = xcd_remap(blockIdx.x, total);
  const int qt = item % q_tiles;
  const int bh = item / q_tiles;
  const int hd = bh % heads, bi = bh / heads;
  const int dmodel = heads * 64;
  const int ld = 3 * dmodel;                                   // elements per token row of qkv
  const unsigned short* base = qkv + (int64_t)bi * tokens * ld;

  // buffer descriptor over this slice's qkv rows: loads past the last token return 0
  const i32x4_t rsrc = lds_dma_rsrc(base, (unsigned)((int64_t)tokens * ld * 2));

  // ---- Q fragments (B operand): lane holds Q[row l31][16 s + 8 h .. +7] ----
  const int qrow = qt * QT + wave * 32 + l31;
  const int qrow_c = qrow < tokens ? qrow : tokens - 1;
  const unsigned short* qp = base + (int64_t)qrow_c * ld + hd * 64 + 8 * h;
  s16x8_t q0 = *reinterpret_cast<const s16x8_t*>(qp);
  s16x8_t q1 = *reinterpret_cast<const s16x8_t*>(qp + 16);
  s16x8_t q2 = *reinterpret_cast<const s16x8_t*>(qp + 32);
  s16x8_t q3 = *reinterpret_cast<const s16x8_t*>(qp + 48);

  // ---- LDS-DMA staging: which (row, chunk) each lane fetches so that the lane-linear destination is the image ----
  int voff_k0, voff_k1, voff_v0, voff_v1;
  {
    int r, cc;
    tile_pos(tid, r, cc);
    voff_k0 = (r * ld + dmodel + hd * 64 + cc * 8) * 2;
    tile_pos(256 + tid, r, cc);
    voff_k1 = (r * ld + dmodel + hd * 64 + cc * 8) * 2;
    int key, ch;
    v_pos(tid, key, ch);
    voff_v0 = (key * ld + 2 * dmodel + hd * 64 + ch * 8) * 2;
    v_pos(256 + tid, key, ch);
    voff_v1 = (key * ld + 2 * dmodel + hd * 64 + ch * 8) * 2;
  }
  const int tile_stride = KT * ld * 2;
  const int nt = (tokens + KT - 1) / KT;
  const unsigned dma_dst = (unsigned)(size_t)LDS_PTR(smem) + (__builtin_amdgcn_readfirstlane(tid & ~63) << 4);
  // the last tile carries its offset in the range-checked voffset (see attention.hip)
#define PIPE_STAGE_TILE(t, BUFI)                                                                    \
  {                                                                                                 \
    const int so_ = (t) * tile_stride;                                                              \
    if ((t) == nt - 1) {                                                                            \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES, voff_k0 + so_, 0);                                 \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + 4096, voff_k1 + so_, 0);                          \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES, voff_v0 + so_, 0);                 \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES + 4096, voff_v1 + so_, 0);          \
    } else {                                                                                        \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES, voff_k0, so_);                                     \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + 4096, voff_k1, so_);                              \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES, voff_v0, so_);                     \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES + 4096, voff_v1, so_);              \
    }                                                                                               \
  }

  // ---- per-lane LDS read bases (tile_off / v_off: buffer, half, s2, dvt, jj terms are immediates) ----
  LdsBases b;
  {
    const int p_l = l31 >> 1;
    const int bslot = (((l31 & 1) << 3) | h) ^ (p_l & 15);
    b.ka0 = smem + (p_l << 8) + ((bslot ^ 0) << 4);
    b.ka1 = smem + (p_l << 8) + ((bslot ^ 2) << 4);
    b.ka2 = smem + (p_l << 8) + ((bslot ^ 4) << 4);
    b.ka3 = smem + (p_l << 8) + ((bslot ^ 6) << 4);
    const int g16 = lane >> 4;
    const int tr_q = (lane & 15) >> 2;
    const int tr_p = lane & 3;
    const int tr_ch = 2 * (g16 & 1) + (tr_p >> 1);
    const int vl0 = 64 * (4 * h + tr_q) + 16 * (tr_ch ^ h) + 8 * (tr_p & 1);
    b.va0 = smem + vl0;
    b.va1 = smem + (vl0 ^ 32);
  }

  // ---- prologue: tiles 0 and 1 on their way; wait for tile 0 (and the Q loads in front of it) ----
  PIPE_STAGE_TILE(0, 0)
  if (nt > 1) {
    PIPE_STAGE_TILE(1, 1)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));   // Q loads retired here, not re-waited inside the loop

  const bool active = __builtin_amdgcn_readfirstlane(qt * QT + wave * 32) < tokens;
  if (!active) {   // all 32 rows past the end: keep staging and synchronising, skip the arithmetic
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 2 < nt) {
        const int bi2 = (t + 2) % NBUF;
        if (bi2 == 0) PIPE_STAGE_TILE(t + 2, 0) else if (bi2 == 1) PIPE_STAGE_TILE(t + 2, 1) else PIPE_STAGE_TILE(t + 2, 2)
      }
    }
    return;
  }

  AttnState st;
  f32x16_t sA, sB;                 // S of even / odd half steps
  s16x8_t pa0 = {}, pa1 = {}, pb0 = {}, pb1 = {};      // packed P of even / odd half steps
#pragma unroll
  for (int r = 0; r < 16; ++r) { st.o0[r] = 0.f; st.o1[r] = 0.f; st.negm[r] = 0.f; }
  st.l_run = 0.f;
  {
    // S(0) and the first maximum: M is fixed by the first 32 keys (key 0 is always valid)
    sA = score_mfma<DT, 0, 0>(b, q0, q1, q2, q3, st.negm);
    if (nt == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (acc_row(r, h) >= tokens) sA[r] = -INFINITY;
    }
    float tmax = max3_f32(sA[0], sA[1], sA[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = max3_f32(tmax, sA[r], sA[r + 1]);
    tmax = fmaxf(tmax, sA[15]);
    const unsigned tb = __float_as_uint(tmax);
    const auto sw = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);
    tmax = max3_f32(tmax, __uint_as_float(sw[0]), __uint_as_float(sw[1]));
#pragma unroll
    for (int r = 0; r < 16; ++r) { st.negm[r] = -tmax; sA[r] -= tmax; }
  }

  // One tile: slot 2t (softmax of half 0; S of half 1 of this tile; O of the previous tile's half 1), the barrier that
  // retires tile t-1 and publishes tile t+1, the DMA of tile t+2, slot 2t+1 (softmax of half 1; S of the next tile's
  // half 0; O of this tile's half 0).  B = t % 3.
#define PIPE_TILE(B, FIRST, LASTT)                                                                                     \
  {                                                                                                                    \
    constexpr int BN = ((B) + 1) % NBUF, BP = ((B) + 2) % NBUF;                                                        \
    attn_slot<DT, B, 0, B, 1, BP, 1, true, !(FIRST), LASTT>(b, q0, q1, q2, q3, st, sA, sB, pb0, pb1, pa0, pa1, t * KT,     \
                                                           tokens, h);                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
    __syncthreads();                                                                                                   \
    if (!(LASTT) && t + 2 < nt) PIPE_STAGE_TILE(t + 2, BP)                                                             \
    attn_slot<DT, B, 1, BN, 0, B, 0, !(LASTT), true, LASTT>(b, q0, q1, q2, q3, st, sB, sA, pa0, pa1, pb0, pb1,             \
                                                          t * KT + 32, tokens, h);                                     \
    ++t;                                                                                                               \
  }
  int t = 0;
  if (nt == 1) {
    PIPE_TILE(0, true, true)
  } else {
    PIPE_TILE(0, true, false)
    while (t + 3 <= nt - 1) {
      PIPE_TILE(1, false, false)
      PIPE_TILE(2, false, false)
      PIPE_TILE(0, false, false)
    }
    const int rem = (nt - 1) - t;        // 0..2 more full tiles in front of the last one; t % 3 == 1 here
    if (rem >= 1) PIPE_TILE(1, false, false)
    if (rem >= 2) PIPE_TILE(2, false, false)
    if (rem == 0) PIPE_TILE(1, false, true)
    else if (rem == 1) PIPE_TILE(2, false, true)
    else PIPE_TILE(0, false, true)
  }
#undef PIPE_TILE
  // the output product of the very last half step (t == nt now; its V sits in ring slot (nt - 1) % 3, half 1)
  {
    const int bl = (nt - 1) % NBUF;
    if (bl == 0) out_mfma<DT, 0, 1>(b, pb0, pb1, st.o0, st.o1);
    else if (bl == 1) out_mfma<DT, 1, 1>(b, pb0, pb1, st.o0, st.o1);
    else out_mfma<DT, 2, 1>(b, pb0, pb1, st.o0, st.o1);
  }

  // ---- normalise and store: lane owns query row `qrow`, columns 32 dvt + 8 g + 4 h + {0..3} ----
  float l_tot;
  {
    const unsigned lb = __float_as_uint(st.l_run);
    const auto sw = __builtin_amdgcn_permlane32_swap(lb, lb, false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.0f / l_tot;
  if (qrow < tokens) {
    unsigned short* orow = out + ((int64_t)bi * tokens + qrow) * dmodel + hd * 64 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 pk;
      pk.x = pack2_h16<DT>(st.o0[4 * g + 0] * inv, st.o0[4 * g + 1] * inv);
      pk.y = pack2_h16<DT>(st.o0[4 * g + 2] * inv, st.o0[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(orow + 8 * g) = pk;
      pk.x = pack2_h16<DT>(st.o1[4 * g + 0] * inv, st.o1[4 * g + 1] * inv);
      pk.y = pack2_h16<DT>(st.o1[4 * g + 2] * inv, st.o1[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(orow + 32 + 8 * g) = pk;
    }
  }
#undef PIPE_STAGE_TILE
}

}  // namespace

// C++ linkage: called by vittf_attention (attention.hip) for q_prescaled = 1 unless VITTF_ATTN_PIPE=0
int vittf_attention_pipe(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype,
                         hipStream_t st) {
  const int q_tiles = (tokens + QT - 1) / QT;
  const int total = batch * heads * q_tiles;
  if (dtype == VITTF_BF16)
    hipLaunchKernelGGL((attn_pipe_kernel<VITTF_BF16>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv,
                       (unsigned short*)out, tokens, heads, q_tiles, total);
  else
    hipLaunchKernelGGL((attn_pipe_kernel<VITTF_FP16>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv,
                       (unsigned short*)out, tokens, heads, q_tiles, total);
  return vittf_check_launch();
}
