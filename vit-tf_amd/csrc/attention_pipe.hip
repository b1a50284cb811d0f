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

#include <stdlib.h>

namespace {

constexpr int QT = 128;   // query rows per workgroup (4 waves x 32)
constexpr int KT = 64;    // keys per tile
constexpr int KV_TILE_BYTES = KT * 64 * 2;   // 8 KB
constexpr int BUF_BYTES = 2 * KV_TILE_BYTES;  // K | V
constexpr int NBUF = 3;
#ifndef PIPE_WAVES
#define PIPE_WAVES 2
#endif
#ifndef PIPE_SCHED
#define PIPE_SCHED 1
#endif

// Diagnostic build only (ABL == 5): shader-clock and 100 MHz real-time stamps around the tile loop of the first waves,
// written to a buffer no other code reads (MI355X_MICROARCH.md, DVFS item 6).
constexpr int STAMP_WAVES = 4096;
__device__ unsigned long long g_pipe_stamps[STAMP_WAVES * 4];

struct LdsBases {
  const char *ka0, *ka1, *ka2, *ka3, *va0, *va1;
};

// MFMA operand fragments of one 32-key half step, fetched from LDS one slot before they are used
struct KFrag { s16x8_t k0, k1, k2, k3; };          // K rows (A operand of S^T = K Q^T), one per 16-wide d chunk
struct VFrag { s16x8_t v00, v01, v10, v11; };      // V^T (A operand of O^T = V^T P^T): [k-step s2][d half dvt]

template <int BUF, int HALF> __device__ __forceinline__ void load_k(const LdsBases& b, KFrag& f) {
  constexpr int off = BUF * BUF_BYTES + 4096 * HALF;
  f.k0 = *reinterpret_cast<const s16x8_t*>(b.ka0 + off);
  f.k1 = *reinterpret_cast<const s16x8_t*>(b.ka1 + off);
  f.k2 = *reinterpret_cast<const s16x8_t*>(b.ka2 + off);
  f.k3 = *reinterpret_cast<const s16x8_t*>(b.ka3 + off);
}

__device__ __forceinline__ s16x8_t load_vt(const char* a0, const char* a1, int imm) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + imm));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a1 + imm + 1024));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int BUF, int HALF> __device__ __forceinline__ void load_v(const LdsBases& b, VFrag& f) {
  constexpr int vb = BUF * BUF_BYTES + KV_TILE_BYTES + 4096 * HALF;
  f.v00 = load_vt(b.va0, b.va1, vb);
  f.v01 = load_vt(b.va0, b.va1, vb + 512);
  f.v10 = load_vt(b.va0, b.va1, vb + 2048);
  f.v11 = load_vt(b.va0, b.va1, vb + 2048 + 512);
}

// S^T(32 keys x 32 queries) = K Q^T + c
template <int DT>
__device__ __forceinline__ f32x16_t score_mfma(const KFrag& k, const s16x8_t& q0, const s16x8_t& q1, const s16x8_t& q2,
                                               const s16x8_t& q3, f32x16_t c) {
  c = mfma32<DT>(k.k0, q0, c);
  c = mfma32<DT>(k.k1, q1, c);
  c = mfma32<DT>(k.k2, q2, c);
  c = mfma32<DT>(k.k3, q3, c);
  return c;
}

// O^T(64 dims x 32 queries) += V^T P^T
template <int DT>
__device__ __forceinline__ void out_mfma(const VFrag& v, const s16x8_t& pf0, const s16x8_t& pf1, f32x16_t& o0, f32x16_t& o1) {
  o0 = mfma32<DT>(v.v00, pf0, o0);
  o1 = mfma32<DT>(v.v01, pf0, o1);
  o0 = mfma32<DT>(v.v10, pf1, o0);
  o1 = mfma32<DT>(v.v11, pf1, o1);
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

// Slot h.  LDS: the fragments slot h+1 will need -- K(h+2) from (NKBUF, NKHALF) [PF_K], V(h) from (CBUF, CHALF) -- go
// into kn / vn.  Matrix pipe: S(h+1) = kc Q^T - M into s_next [DO_S]; O += vc^T P(h-1)^T [DO_O].  VALU: P(h-1) (kept in
// fp32 across the slot boundary, so that its 8 conversions sit in the same basic block as this slot's MFMAs) is packed
// to 16 bit; softmax of s_cur = S(h) -> p_cur (its K half is (CBUF, CHALF), read again only on the slow path).
// MASK: ragged last tile.  The sched_group_barrier sequence spreads the 8 MFMAs evenly over the VALU stream: an in-order
// wave that meets a busy matrix pipe stalls with all its VALU work behind it, so MFMAs in clusters idle both pipes.
template <int DT, int ABL, int CBUF, int CHALF, int NKBUF, int NKHALF, bool PF_K, bool DO_S, bool DO_O, bool MASK>
__device__ __forceinline__ void attn_slot(const LdsBases& b, const s16x8_t& q0, const s16x8_t& q1, const s16x8_t& q2,
                                          const s16x8_t& q3, AttnState& st, f32x16_t& s_cur, f32x16_t& s_next,
                                          const KFrag& kc, const VFrag& vc, KFrag& kn, VFrag& vn, const float (&p_prev)[16],
                                          float (&p)[16], int key0, int tokens, int h) {
  constexpr float THR = DT == VITTF_FP16 ? 8192.f : 1073741824.f;
  s16x8_t pfp0 = {}, pfp1 = {};
  if constexpr (DO_O && ABL != 1) pack_p<DT>(p_prev, pfp0, pfp1);
  // ABL (timing-only builds, wrong results): 1 no softmax VALU, 2 no MFMA, 3 no barrier / DMA wait, 4 no LDS fragment reads
  if constexpr (ABL != 4) {
    if constexpr (PF_K) load_k<NKBUF, NKHALF>(b, kn);
    load_v<CBUF, CHALF>(b, vn);
  }
  if constexpr (ABL != 2) {
    if constexpr (DO_S) s_next = score_mfma<DT>(kc, q0, q1, q2, q3, st.negm);
    if constexpr (DO_O) out_mfma<DT>(vc, pfp0, pfp1, st.o0, st.o1);
  }
  if constexpr (ABL == 1) {
    asm volatile("" : "+v"(s_cur));
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = s_cur[r];
    return;
  }
  if constexpr (MASK) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (key0 + acc_row(r, h) >= tokens) s_cur[r] = -INFINITY;
  }
  p[0] = __builtin_amdgcn_exp2f(s_cur[0]);
  p[1] = __builtin_amdgcn_exp2f(s_cur[1]);
  float psum0 = p[0], psum1 = p[1];
#pragma unroll
  for (int r = 2; r < 16; r += 2) {
    p[r] = __builtin_amdgcn_exp2f(s_cur[r]);
    p[r + 1] = __builtin_amdgcn_exp2f(s_cur[r + 1]);
    psum0 += p[r];
    psum1 += p[r + 1];
  }
  float ps = psum0 + psum1;
#if PIPE_SCHED
  if constexpr (DO_S && DO_O && !MASK && ABL == 0) {
    // 8 gaps: one MFMA, then its share of the LDS fragment reads (first four gaps), of the 16 v_exp and of the 24 other
    // VALU instructions (8 conversions first -- the O MFMAs in gaps 4..7 wait for them --, then the row-sum adds)
#define PIPE_GAP(NDS, NTR, NVA)                                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           \
    if (NDS) __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);                \
    if (NTR) __builtin_amdgcn_sched_group_barrier(0x400, NTR, 0);                \
    if (NVA) __builtin_amdgcn_sched_group_barrier(0x002, NVA, 0);
    PIPE_GAP(3, 2, 3) PIPE_GAP(3, 2, 3) PIPE_GAP(3, 2, 3) PIPE_GAP(3, 2, 3)
    PIPE_GAP(0, 2, 3) PIPE_GAP(0, 2, 3) PIPE_GAP(0, 2, 2) PIPE_GAP(0, 2, 2)
#undef PIPE_GAP
  }
#endif
  if (__builtin_expect(__any(!(ps <= THR)), 0)) {
    // ---- slow path: the half's values have outgrown the 16-bit P at the current M.  Raw scores again from LDS, the
    // true row maximum, everything accumulated so far rescaled to the new M. ----
    f32x16_t zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    KFrag kh;
    load_k<CBUF, CHALF>(b, kh);
    f32x16_t raw = score_mfma<DT>(kh, q0, q1, q2, q3, zero);
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
    const float delta = fmaxf(tmax + st.negm[0], 0.f);                          // M moves up by delta (per query column)
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    st.l_run *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st.o0[r] *= alpha;
      st.o1[r] *= alpha;
      st.negm[r] -= delta;                                                      // in place: the same registers on both paths
      if constexpr (DO_S) s_next[r] -= delta;                                   // S(h+1) was formed with the old M
    }
    psum0 = 0.f; psum1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      p[r] = __builtin_amdgcn_exp2f(raw[r] + st.negm[r]);
      p[r + 1] = __builtin_amdgcn_exp2f(raw[r + 1] + st.negm[r + 1]);
      psum0 += p[r];
      psum1 += p[r + 1];
    }
    ps = psum0 + psum1;
  }
  st.l_run += ps;
}

template <int DT, int ABL>
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

  // ---- prologue: tiles 0 and 1 land and are published together; tile 2 leaves right behind the barrier ----
  PIPE_STAGE_TILE(0, 0)
  if (nt > 1) PIPE_STAGE_TILE(1, 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nt > 2) PIPE_STAGE_TILE(2, 2)
  asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));   // Q loads retired here, not re-waited inside the loop

  // The barrier in front of tile t >= 1: every wave has fetched its last fragments of tile t-1 (they are prefetched one
  // slot ahead, and __syncthreads drains lgkmcnt), so the buffer of tile t-1 takes tile t+2; tile t+1 (requested one
  // tile ago) is published for the fragment prefetches of tile t's slots.
#define PIPE_TILE_BARRIER(BNEXT2)                                                                   \
  {                                                                                                 \
    if constexpr (ABL != 3) {                                                                       \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              \
      __syncthreads();                                                                              \
    }                                                                                               \
    if (t + 2 < nt) PIPE_STAGE_TILE(t + 2, BNEXT2)                                                  \
  }

  const bool active = __builtin_amdgcn_readfirstlane(qt * QT + wave * 32) < tokens;
  if (!active) {   // all 32 rows past the end: keep staging and synchronising, skip the arithmetic
    for (int t = 1; t < nt; ++t) {
      const int b2 = (t + 2) % NBUF;
      if (b2 == 0) PIPE_TILE_BARRIER(0) else if (b2 == 1) PIPE_TILE_BARRIER(1) else PIPE_TILE_BARRIER(2)
    }
    return;
  }

  AttnState st;
  f32x16_t sA, sB;                 // S of even / odd half steps
  float pA[16], pB[16];            // P of even / odd half steps, fp32 until the slot that multiplies it with V
#pragma unroll
  for (int r = 0; r < 16; ++r) { pA[r] = 0.f; pB[r] = 0.f; }
  KFrag kA, kB;                    // K fragments consumed in even / odd slots
  VFrag vA = {}, vB;               // V fragments consumed in even / odd slots
#pragma unroll
  for (int r = 0; r < 16; ++r) { st.o0[r] = 0.f; st.o1[r] = 0.f; st.negm[r] = 0.f; }
  st.l_run = 0.f;
  {
    // S(0) and the first maximum: M is fixed by the first 32 keys (key 0 is always valid)
    load_k<0, 0>(b, kB);
    load_k<0, 1>(b, kA);          // slot 0 forms S(1) from the second half of tile 0
    sA = score_mfma<DT>(kB, q0, q1, q2, q3, st.negm);
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

  // One tile t (ring slot B = t % 3), two slots:
  //   slot 2t  : softmax S(2t)   | S(2t+1) = kA Q^T | O += vA P(2t-1)      | fetch kB = K(t+1, half 0), vB = V(t, half 0)
  //   slot 2t+1: softmax S(2t+1) | S(2t+2) = kB Q^T | O += vB P(2t)        | fetch kA = K(t+1, half 1), vA = V(t, half 1)
#define PIPE_TILE(B, FIRST, LASTT)                                                                                     \
  {                                                                                                                    \
    constexpr int BN = ((B) + 1) % NBUF;                                                                               \
    attn_slot<DT, ABL, B, 0, BN, 0, !(LASTT), true, !(FIRST), LASTT>(b, q0, q1, q2, q3, st, sA, sB, kA, vA, kB, vB, pB, pA,   \
                                                               t * KT, tokens, h);                           \
    attn_slot<DT, ABL, B, 1, BN, 1, !(LASTT), !(LASTT), true, LASTT>(b, q0, q1, q2, q3, st, sB, sA, kB, vB, kA, vA, pA, pB,   \
                                                               t * KT + 32, tokens, h);                      \
    ++t;                                                                                                               \
  }
  int t = 0;
  unsigned long long stamp_c0 = 0, stamp_r0 = 0;
  if constexpr (ABL == 5) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
  if (nt == 1) {
    PIPE_TILE(0, true, true)
  } else {
    PIPE_TILE(0, true, false)
    while (t + 3 <= nt - 1) {
      PIPE_TILE_BARRIER(0) PIPE_TILE(1, false, false)
      PIPE_TILE_BARRIER(1) PIPE_TILE(2, false, false)
      PIPE_TILE_BARRIER(2) PIPE_TILE(0, false, false)
    }
    const int rem = (nt - 1) - t;        // 0..2 more full tiles in front of the last one; t % 3 == 1 here
    if (rem >= 1) { PIPE_TILE_BARRIER(0) PIPE_TILE(1, false, false) }
    if (rem >= 2) { PIPE_TILE_BARRIER(1) PIPE_TILE(2, false, false) }
    if (rem == 0) { PIPE_TILE_BARRIER(0) PIPE_TILE(1, false, true) }
    else if (rem == 1) { PIPE_TILE_BARRIER(1) PIPE_TILE(2, false, true) }
    else { PIPE_TILE_BARRIER(2) PIPE_TILE(0, false, true) }
  }
#undef PIPE_TILE
#undef PIPE_TILE_BARRIER
  // the output product of the very last half step: its V fragments were fetched by the last slot
  {
    s16x8_t pf0, pf1;
    pack_p<DT>(pB, pf0, pf1);
    out_mfma<DT>(vA, pf0, pf1, st.o0, st.o1);
  }
  if constexpr (ABL == 5) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const int w = blockIdx.x * 4 + wave;
    if (w < STAMP_WAVES && lane == 0) {
      g_pipe_stamps[4 * w + 0] = stamp_c0; g_pipe_stamps[4 * w + 1] = c1;
      g_pipe_stamps[4 * w + 2] = stamp_r0; g_pipe_stamps[4 * w + 3] = r1;
    }
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

// C++ linkage: called by vittf_attention (attention.hip) for q_prescaled = 1 unless VITTF_ATTN_PIPE=0.
// VITTF_ATTN_ABLATE=1..4 (fp16 only) launches a timing-only build with one component removed (wrong results; tools/).
int vittf_attention_pipe(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype,
                         hipStream_t st) {
  const int q_tiles = (tokens + QT - 1) / QT;
  const int total = batch * heads * q_tiles;
#define PIPE_LAUNCH(DTV, ABLV)                                                                                  \
  hipLaunchKernelGGL((attn_pipe_kernel<DTV, ABLV>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv, \
                     (unsigned short*)out, tokens, heads, q_tiles, total)
  const char* e = getenv("VITTF_ATTN_ABLATE");
  const int abl = e ? atoi(e) : 0;
  if (dtype == VITTF_BF16) PIPE_LAUNCH(VITTF_BF16, 0);
  else if (abl == 1) PIPE_LAUNCH(VITTF_FP16, 1);
  else if (abl == 2) PIPE_LAUNCH(VITTF_FP16, 2);
  else if (abl == 3) PIPE_LAUNCH(VITTF_FP16, 3);
  else if (abl == 4) PIPE_LAUNCH(VITTF_FP16, 4);
  else if (abl == 5) PIPE_LAUNCH(VITTF_FP16, 5);
  else PIPE_LAUNCH(VITTF_FP16, 0);
#undef PIPE_LAUNCH
  return vittf_check_launch();
}

// Diagnostic: copies the stamps of the last VITTF_ATTN_ABLATE=5 launch ([wave][shader clock start, end, 100 MHz start, end])
// to the host.  Synchronises the device.  Returns the number of waves copied.
extern "C" int vittf_debug_attention_stamps(uint64_t* out_host, int32_t max_waves) {
  if (!out_host || max_waves <= 0) return VITTF_ERR_INVALID_ARG;
  const int n = max_waves < STAMP_WAVES ? max_waves : STAMP_WAVES;
  if (hipDeviceSynchronize() != hipSuccess) return VITTF_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_pipe_stamps), (size_t)n * 4 * sizeof(uint64_t)) != hipSuccess)
    return VITTF_ERR_LAUNCH;
  return n;
}
