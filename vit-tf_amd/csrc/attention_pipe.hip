// Software-pipelined flash attention forward, head dim 64, pre-scaled q (the engine's default since round 2).
//
// Same contract as attention.hip (softmax(q k^T / 8) v per head; Attention.forward of the upstream model the
// reference calls, infer.py:177), same LDS images, same lazy running maximum -- but the three stages of a 32-key
// half step no longer run one after the other inside a wave.  At head dim 64 a 32 x 32 score block costs 8 MFMAs
// (256 matrix-pipe cycles) against 16 v_exp + 16 adds + 8 v_cvt_pk + the MFMAs' own issue slots (~290 VALU-port
// cycles): whichever pipe a wave is not using idles unless ANOTHER wave happens to be in the opposite phase (round 1:
// 62 cycles per MFMA, the two pipes co-executing 21 % of the time).  Here every wave carries independent instruction
// streams through each slot h (a slot = one 32-key half step):
//
//     matrix pipe :  S(h+1) = K(h+1) Q^T - M        and        O += V(h-1)^T P(h-1)^T
//     VALU        :  P(h) = exp2(S(h)), row sums, the overflow check; 16-bit packing of P(h-1)
//     LDS         :  the K / V^T fragments of slot h+1
//
// so the scores of the next half step and the output product of the previous one run under the softmax arithmetic of
// the current one (S and P double-buffered in registers), no MFMA waits for an LDS read issued in its own slot, and a
// sched_group_barrier sequence spaces the MFMAs evenly over the VALU stream (an in-order wave that meets a busy
// matrix pipe stalls with all its VALU work behind it: MFMAs in clusters idle both pipes).
//
// Two shapes of the same code (template parameter RB = 32-row query blocks per wave):
//   RB = 1 (default): 32 rows per wave, 2 waves per SIMD (256 VGPRs), tiles of 64 keys (3 x 16 KB, two workgroups per CU).
//   RB = 2: ONE wave per SIMD owning the whole 512-register file, 64 query rows per wave, 256 per workgroup (one
//           workgroup per CU), the output accumulators in the AGPR half (asm MFMAs, see mfma32_acc).  Both query blocks
//           of a wave share every K / V^T fragment read (half the LDS reads, DMA issues, barriers and scalar work per
//           MFMA).  K/V tiles of 128 keys, 3-deep LDS ring (96 KB), one barrier per 128 keys.  Parity-green, but slower
//           (0.998 against 0.831 ms per launch in the pipeline): a lone in-order wave overlaps its VALU and matrix
//           work even less than two do (tools/micro/mfma_ceiling: 42 against 38.8 cycles per MFMA for the bare
//           instruction mix), and the 64-row workgroups cut the grid to 12.75 rounds of one workgroup per CU.
//
// What bounds the kernel (profiles/r02_mfma_ceiling.txt): on this chip a register-only loop of v_mfma_f32_32x32x16_f16
// sustains 1.5-1.63 PFLOP/s (33 cycles per MFMA at the 1.54-1.65 GHz the chip holds under that load) = 60-65 % of the
// nominal 2.5 PFLOP/s, and the same loop with the head-dim-64 softmax mix between MFMAs (2 v_exp, 2 v_add, 1 v_cvt_pk per
// MFMA: no LDS, no memory, no barrier) 1.28-1.33 PFLOP/s = 51-53 %: v_exp_f32 costs ~9 and a plain VALU instruction ~3.8
// issue cycles per wave with two or more waves per SIMD, 29.5 cycles per MFMA slot, and an MFMA adds ~9 of its own.
//
//   * ring protocol: the barrier in front of tile t publishes tile t+1 (requested one tile earlier) and retires tile
//     t-1 (all its fragments were fetched one slot ahead, i.e. before the barrier), whose buffer then takes tile t+2.
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

constexpr int NBUF = 3;
#ifndef PIPE_SCHED
#define PIPE_SCHED 1
#endif

template <int RB> struct Shape {
  static constexpr int QT = 128 * RB;          // query rows per workgroup (4 waves x RB x 32)
  static constexpr int KT = 64 * RB;           // keys per tile
  static constexpr int SLOTS = KT / 32;        // half steps per tile
  static constexpr int KV_TILE_BYTES = KT * 128;
  static constexpr int BUF_BYTES = 2 * KV_TILE_BYTES;   // K | V
  static constexpr int PIECES = KT / 32;       // 1 KB LDS-DMA pieces per wave, operand and tile
};

// Diagnostic build only (ABL == 5): shader-clock and 100 MHz real-time stamps around the tile loop of the first waves,
// written to a buffer no other code reads (MI355X_MICROARCH.md, DVFS item 6).
constexpr int STAMP_WAVES = 4096;
__device__ unsigned long long g_pipe_stamps[STAMP_WAVES * 4];

struct LdsBases {
  const char *ka0, *ka1, *ka2, *ka3, *va0, *va1;
};

// MFMA operand fragments of one 32-key half step, fetched from LDS one slot before they are used
struct KFrag { s16x8_t k[4]; };          // K rows (A operand of S^T = K Q^T), one per 16-wide d chunk
struct VFrag { s16x8_t v[4]; };          // V^T (A operand of O^T = V^T P^T): [2 x k-step s2 + d half dvt]
struct QFrag { s16x8_t q[4]; };          // Q rows (B operand), resident for the whole kernel

template <int OFF> __device__ __forceinline__ void load_k1(const LdsBases& b, KFrag& f, int i) {
  const char* base = i == 0 ? b.ka0 : i == 1 ? b.ka1 : i == 2 ? b.ka2 : b.ka3;
  f.k[i] = *reinterpret_cast<const s16x8_t*>(base + OFF);
}
template <int OFF> __device__ __forceinline__ void load_k(const LdsBases& b, KFrag& f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) load_k1<OFF>(b, f, i);
}

__device__ __forceinline__ s16x8_t load_vt(const char* a0, const char* a1, int imm) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + imm));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a1 + imm + 1024));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int OFF, int ABL = 0> __device__ __forceinline__ void load_v1(const LdsBases& b, VFrag& f, int i) {
  if constexpr (ABL == 6) {   // timing only: what a V^T image readable with ONE ds_read_b128 per fragment would cost
    f.v[i] = *reinterpret_cast<const s16x8_t*>((i == 0 ? b.ka0 : i == 1 ? b.ka1 : i == 2 ? b.ka2 : b.ka3) + OFF);
    return;
  }
  f.v[i] = load_vt(b.va0, b.va1, OFF + 2048 * (i >> 1) + 512 * (i & 1));
}
template <int OFF> __device__ __forceinline__ void load_v(const LdsBases& b, VFrag& f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) load_v1<OFF>(b, f, i);
}

// S^T(32 keys x 32 queries) = K Q^T + c
template <int DT> __device__ __forceinline__ f32x16_t score_mfma(const KFrag& k, const QFrag& q, f32x16_t c) {
#pragma unroll
  for (int i = 0; i < 4; ++i) c = mfma32<DT>(k.k[i], q.q[i], c);
  return c;
}

// One MFMA of the output product with its accumulator (and its V^T operand) in the AGPR half of the register file.
// Written as asm because hipcc picks ONE form for every MFMA of a kernel: all accumulators in arch VGPRs (256 at most:
// the 64-rows-per-wave shape needs ~320) or all in AGPRs (then every score tile is copied to arch VGPRs for the softmax,
// 16 v_accvgpr_read per 32 x 32 block).  The output accumulators are only ever touched by MFMAs on the fast path, so they
// are the ones to live there.  What hipcc does not pad around an asm statement is padded here: nothing for the
// accumulate chain (MFMA D -> the next MFMA's C, the same registers: 0 wait states), `s_nop 1` for a B operand that a
// v_cvt_pk may have written just before; compiler code that reads or writes the accumulators (slow path, epilogue)
// sits behind mfma_acc_fence().
template <int DT> __device__ __forceinline__ void mfma32_acc(const s16x8_t& a, const s16x8_t& bb, f32x16_t& acc) {
  if constexpr (DT == VITTF_BF16)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(bb));
  else
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(bb));
}
// 8-pass MFMA result -> any non-MFMA reader / writer: 12 wait states (cdna_hip_programming.md 5.7 item 2); also covers a
// v_accvgpr_write followed by an asm MFMA
// The accumulators are operands of the fence: a register-only statement is ordered against other register-only code
// through its operands alone (a "memory" clobber orders loads and stores, nothing else).
__device__ __forceinline__ void mfma_acc_fence(f32x16_t& o0, f32x16_t& o1) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+a"(o0), "+a"(o1));
}

// step j of O^T(64 dims x 32 queries) += V^T P^T: j = 2 s2 + dvt (key step s2 feeds pf_s2, output half dvt)
template <int DT, bool ACC>
__device__ __forceinline__ void out_mfma1(const VFrag& v, const s16x8_t& pf0, const s16x8_t& pf1, f32x16_t& o0, f32x16_t& o1,
                                          int j) {
  const s16x8_t& pf = (j >> 1) ? pf1 : pf0;
  f32x16_t& o = (j & 1) ? o1 : o0;
  if constexpr (ACC) mfma32_acc<DT>(v.v[j], pf, o);
  else o = mfma32<DT>(v.v[j], pf, o);
}

// O^T(64 dims x 32 queries) += V^T P^T
template <int DT, bool ACC>
__device__ __forceinline__ void out_mfma(const VFrag& v, const s16x8_t& pf0, const s16x8_t& pf1, f32x16_t& o0, f32x16_t& o1) {
#pragma unroll
  for (int j = 0; j < 4; ++j) out_mfma1<DT, ACC>(v, pf0, pf1, o0, o1, j);
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

__device__ __forceinline__ float tile_max(const f32x16_t& s) {
  float tmax = max3_f32(s[0], s[1], s[2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) tmax = max3_f32(tmax, s[r], s[r + 1]);
  tmax = fmaxf(tmax, s[15]);
  const unsigned tb = __float_as_uint(tmax);
  const auto sw = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);
  return max3_f32(tmax, __uint_as_float(sw[0]), __uint_as_float(sw[1]));      // both lane halves agree
}

// per-wave running state of one 32-row query block
struct AttnState {
  f32x16_t o0, o1, negm;
  float l_run;
};

// Slot h of a wave with RB query blocks.
//   LDS  : the fragments slot h+1 will need -- K(h+2) at byte offset NK_OFF [PF_K], V(h) at NV_OFF -- go into kn / vn.
//   MFMA : S_a(h+1) = kc Q_a^T - M_a into s_next[a] [DO_S];  O_a += vc^T P_a(h-1)^T [DO_O].
//   VALU : softmax of s_cur[a] = S_a(h), in place, then packed to 16 bit behind the overflow check -> pfc[a] (the K half
//          of S(h), at byte offset CK_OFF, is read again only on the slow path).  The conversions land in the basic
//          block of the NEXT slot's MFMAs (no barrier in between except at a tile start), the fp32 P never outlives its
//          slot: register budget.  MASK: the half belongs to the ragged last tile.
// ABL (timing-only builds, wrong results): 1 no softmax VALU, 2 no MFMA, 3 no barrier / DMA wait, 4 no LDS fragment reads.
template <int DT, int ABL, int RB, int CK_OFF, int NK_OFF, int NV_OFF, bool PF_K, bool DO_S, bool DO_O, bool MASK>
__device__ __forceinline__ void attn_slot(const LdsBases& b, const QFrag (&q)[RB], AttnState (&st)[RB], f32x16_t (&s_cur)[RB],
                                          f32x16_t (&s_next)[RB], const KFrag& kc, const VFrag& vc, KFrag& kn, VFrag& vn,
                                          const s16x8_t (&pfp0)[RB], const s16x8_t (&pfp1)[RB], s16x8_t (&pfc0)[RB],
                                          s16x8_t (&pfc1)[RB], int key0, int tokens, int h) {
  constexpr float THR = DT == VITTF_FP16 ? 8192.f : 1073741824.f;
  constexpr bool ACC = RB == 2;
  float p[RB][16];
  float ps[RB];
  if constexpr (ABL == 1) {   // the same matrix work on unprocessed score bits, no softmax arithmetic
    if constexpr (PF_K) load_k<NK_OFF>(b, kn);
    load_v<NV_OFF>(b, vn);
#pragma unroll
    for (int g = 0; g < 8 * RB; ++g) {      // the same MFMA order as the real slot
      bool is_s;
      int a, i;
      if constexpr (RB == 2) {
        is_s = g < 8;
        if (is_s) { a = g & 1; i = g >> 1; }
        else { const int g2 = g - 8; a = (g2 >> 1) & 1; i = (g2 & 1) + 2 * (g2 >> 2); }
      } else {
        is_s = (g & 1) == 0; a = 0; i = g >> 1;
      }
      if (is_s) {
        if constexpr (DO_S) s_next[a] = mfma32<DT>(kc.k[i], q[a].q[i], i == 0 ? st[a].negm : s_next[a]);
      } else {
        if constexpr (DO_O) out_mfma1<DT, ACC>(vc, pfp0[a], pfp1[a], st[a].o0, st[a].o1, i);
      }
    }
#pragma unroll
    for (int a = 0; a < RB; ++a) {
      asm volatile("" : "+v"(s_cur[a]));
      u32x4_t u0, u1;
#pragma unroll
      for (int j = 0; j < 4; ++j) { u0[j] = __float_as_uint(s_cur[a][j]); u1[j] = __float_as_uint(s_cur[a][4 + j]); }
      pfc0[a] = __builtin_bit_cast(s16x8_t, u0);
      pfc1[a] = __builtin_bit_cast(s16x8_t, u1);
    }
    return;
  }
  if constexpr (MASK) {
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (key0 + acc_row(r, h) >= tokens) s_cur[a][r] = -INFINITY;
  }
  // 8 RB gaps, pinned by scheduling fences: gap g = one MFMA + one unit of softmax arithmetic (two v_exp, two row-sum adds,
  // one conversion: 8 units per query block) + its share of the 12 fragment reads for the next slot.  An in-order wave that
  // meets a busy matrix pipe stalls with all its VALU work behind it, so MFMAs in clusters idle both pipes: here the next
  // MFMA is issued when the previous one (32 cycles) has just left the pipe (one unit = ~28 VALU-port cycles).
  //   MFMA order: S_0 (4 k-steps), .., S_{RB-1}, then O_0 (4 steps), .., O_{RB-1}
  float psum0[RB], psum1[RB];
  u32x4_t pk0[RB], pk1[RB];
#pragma unroll
  for (int g = 0; g < 8 * RB; ++g) {
    if constexpr (ABL != 2) {
      // MFMA order: consecutive MFMAs never share an accumulator (a dependent 32x32x16 MFMA issued right behind its
      // producer waits ~64 cycles, twice the issue interval: measured, ablation build 1).
      //   RB = 2: S_0 k0, S_1 k0, S_0 k1, S_1 k1, ... then O_0 j0, O_0 j1, O_1 j0, O_1 j1, O_0 j2, ... (j0 / j1 = o0 / o1)
      //   RB = 1: S k0, O j0, S k1, O j1, S k2, O j2, S k3, O j3
      bool is_s;
      int a, i;
      if constexpr (RB == 2) {
        is_s = g < 8;
        if (is_s) { a = g & 1; i = g >> 1; }
        else { const int g2 = g - 8; a = (g2 >> 1) & 1; i = (g2 & 1) + 2 * (g2 >> 2); }
      } else {
        is_s = (g & 1) == 0; a = 0; i = g >> 1;
      }
      if (is_s) {
        if constexpr (DO_S) s_next[a] = mfma32<DT>(kc.k[i], q[a].q[i], i == 0 ? st[a].negm : s_next[a]);
      } else {
        if constexpr (DO_O) out_mfma1<DT, ACC>(vc, pfp0[a], pfp1[a], st[a].o0, st[a].o1, i);
      }
    }
    if constexpr (ABL != 4) {
      if (g < 2) {
        if constexpr (PF_K) { load_k1<NK_OFF>(b, kn, 2 * g); load_k1<NK_OFF>(b, kn, 2 * g + 1); }
      } else if (g < 4) {
        load_v1<NV_OFF, ABL>(b, vn, g - 2);
      } else if (RB == 2 ? (g == 4 || g == 5) : (g == 4)) {
        if constexpr (RB == 2) load_v1<NV_OFF, ABL>(b, vn, g - 2);
        else { load_v1<NV_OFF, ABL>(b, vn, 2); load_v1<NV_OFF, ABL>(b, vn, 3); }
      }
    }
    {
      const int a = g >> 3, i = g & 7;
      const float e0 = __builtin_amdgcn_exp2f(s_cur[a][2 * i]);
      const float e1 = __builtin_amdgcn_exp2f(s_cur[a][2 * i + 1]);
      p[a][2 * i] = e0;
      p[a][2 * i + 1] = e1;
      if (i == 0) { psum0[a] = e0; psum1[a] = e1; } else { psum0[a] += e0; psum1[a] += e1; }
      const unsigned w = pack2_h16<DT>(e0, e1);
      if (i < 4) pk0[a][i] = w; else pk1[a][i - 4] = w;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float pmax;
#pragma unroll
  for (int a = 0; a < RB; ++a) {
    ps[a] = psum0[a] + psum1[a];
    pfc0[a] = __builtin_bit_cast(s16x8_t, pk0[a]);
    pfc1[a] = __builtin_bit_cast(s16x8_t, pk1[a]);
    pmax = a == 0 ? ps[0] : fmaxf(pmax, ps[a]);
  }
  if (__builtin_expect(__any(!(pmax <= THR)), 0)) {
    // ---- slow path: a block's values have outgrown the 16-bit P at its current M.  Raw scores again from LDS, the
    // true row maximum, everything accumulated so far rescaled to the new M. ----
    f32x16_t zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    KFrag kh;
    load_k<CK_OFF>(b, kh);
    if constexpr (RB == 2) {                      // the output accumulators are about to be rescaled by VALU code
#pragma unroll
      for (int a = 0; a < RB; ++a) mfma_acc_fence(st[a].o0, st[a].o1);
    }
#pragma unroll
    for (int a = 0; a < RB; ++a) {
      f32x16_t raw = score_mfma<DT>(kh, q[a], zero);
      if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + acc_row(r, h) >= tokens) raw[r] = -INFINITY;
      }
      const float tmax = tile_max(raw);
      const float delta = fmaxf(tmax + st[a].negm[0], 0.f);                      // M moves up by delta (per query column)
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      st[a].l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[a].o0[r] *= alpha;
        st[a].o1[r] *= alpha;
        st[a].negm[r] -= delta;                                                  // in place: the same registers on both paths
        if constexpr (DO_S) s_next[a][r] -= delta;                               // S(h+1) was formed with the old M
      }
      float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        p[a][r] = __builtin_amdgcn_exp2f(raw[r] + st[a].negm[r]);
        p[a][r + 1] = __builtin_amdgcn_exp2f(raw[r + 1] + st[a].negm[r + 1]);
        psum0 += p[a][r];
        psum1 += p[a][r + 1];
      }
      ps[a] = psum0 + psum1;
      pack_p<DT>(p[a], pfc0[a], pfc1[a]);
    }
    if constexpr (RB == 2) {
#pragma unroll
      for (int a = 0; a < RB; ++a) mfma_acc_fence(st[a].o0, st[a].o1);
    }
  }
#pragma unroll
  for (int a = 0; a < RB; ++a) st[a].l_run += ps[a];
}

// The same slot with RUN-TIME ring offsets and flags, for the tiles outside the steady-state loop (the first tile, the up to
// two tiles the 3-tile loop leaves over, the ragged last tile): executed a handful of times per workgroup, so nothing is
// pinned or interleaved here -- what matters is that these tiles are ONE code path instead of seven template instances:
// with seven, hipcc spilled ~110 VGPRs around their joins (276 B of scratch per lane, ~0.4 GB of extra HBM traffic per
// launch, as much as the kernel's algorithmic traffic).
template <int DT, int RB>
__device__ __forceinline__ void attn_slot_rt(const LdsBases& b, const QFrag (&q)[RB], AttnState (&st)[RB], f32x16_t (&s_cur)[RB],
                                             f32x16_t (&s_next)[RB], const KFrag& kc, const VFrag& vc, KFrag& kn, VFrag& vn,
                                             const s16x8_t (&pfp0)[RB], const s16x8_t (&pfp1)[RB], s16x8_t (&pfc0)[RB],
                                             s16x8_t (&pfc1)[RB], int ck_off, int nk_off, int nv_off, bool pf_k, bool do_s,
                                             bool do_o, bool mask, int key0, int tokens, int h) {
  constexpr float THR = DT == VITTF_FP16 ? 8192.f : 1073741824.f;
  constexpr bool ACC = RB == 2;
  if (pf_k) {
    kn.k[0] = *reinterpret_cast<const s16x8_t*>(b.ka0 + nk_off);
    kn.k[1] = *reinterpret_cast<const s16x8_t*>(b.ka1 + nk_off);
    kn.k[2] = *reinterpret_cast<const s16x8_t*>(b.ka2 + nk_off);
    kn.k[3] = *reinterpret_cast<const s16x8_t*>(b.ka3 + nk_off);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) vn.v[i] = load_vt(b.va0, b.va1, nv_off + 2048 * (i >> 1) + 512 * (i & 1));
  if (do_s) {
#pragma unroll
    for (int a = 0; a < RB; ++a) s_next[a] = score_mfma<DT>(kc, q[a], st[a].negm);
  }
  if (do_o) {
#pragma unroll
    for (int a = 0; a < RB; ++a) out_mfma<DT, ACC>(vc, pfp0[a], pfp1[a], st[a].o0, st[a].o1);
  }
  if (mask) {
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (key0 + acc_row(r, h) >= tokens) s_cur[a][r] = -INFINITY;
  }
  float p[RB][16], ps[RB];
  float pmax = 0.f;
#pragma unroll
  for (int a = 0; a < RB; ++a) {
    float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      p[a][r] = __builtin_amdgcn_exp2f(s_cur[a][r]);
      p[a][r + 1] = __builtin_amdgcn_exp2f(s_cur[a][r + 1]);
      psum0 += p[a][r];
      psum1 += p[a][r + 1];
    }
    ps[a] = psum0 + psum1;
    pmax = a == 0 ? ps[0] : fmaxf(pmax, ps[a]);
  }
  if (__builtin_expect(__any(!(pmax <= THR)), 0)) {       // slow path, as in attn_slot
    f32x16_t zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    KFrag kh;
    kh.k[0] = *reinterpret_cast<const s16x8_t*>(b.ka0 + ck_off);
    kh.k[1] = *reinterpret_cast<const s16x8_t*>(b.ka1 + ck_off);
    kh.k[2] = *reinterpret_cast<const s16x8_t*>(b.ka2 + ck_off);
    kh.k[3] = *reinterpret_cast<const s16x8_t*>(b.ka3 + ck_off);
    if constexpr (RB == 2) {
#pragma unroll
      for (int a = 0; a < RB; ++a) mfma_acc_fence(st[a].o0, st[a].o1);
    }
#pragma unroll
    for (int a = 0; a < RB; ++a) {
      f32x16_t raw = score_mfma<DT>(kh, q[a], zero);
      if (mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + acc_row(r, h) >= tokens) raw[r] = -INFINITY;
      }
      const float tmax = tile_max(raw);
      const float delta = fmaxf(tmax + st[a].negm[0], 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      st[a].l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[a].o0[r] *= alpha;
        st[a].o1[r] *= alpha;
        st[a].negm[r] -= delta;
        if (do_s) s_next[a][r] -= delta;
      }
      float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        p[a][r] = __builtin_amdgcn_exp2f(raw[r] + st[a].negm[r]);
        p[a][r + 1] = __builtin_amdgcn_exp2f(raw[r + 1] + st[a].negm[r + 1]);
        psum0 += p[a][r];
        psum1 += p[a][r + 1];
      }
      ps[a] = psum0 + psum1;
    }
    if constexpr (RB == 2) {
#pragma unroll
      for (int a = 0; a < RB; ++a) mfma_acc_fence(st[a].o0, st[a].o1);
    }
  }
#pragma unroll
  for (int a = 0; a < RB; ++a) {
    st[a].l_run += ps[a];
    pack_p<DT>(p[a], pfc0[a], pfc1[a]);
  }
}

template <int DT, int ABL, int RB>
__global__ __launch_bounds__(256, RB == 2 ? 1 : 2) void attn_pipe_kernel(const unsigned short* __restrict__ qkv,
                                                                       unsigned short* __restrict__ out, int tokens,
                                                                       int heads, int q_tiles, int total) {
  using SH = Shape<RB>;
  constexpr int QT = SH::QT, KT = SH::KT, SLOTS = SH::SLOTS, KVB = SH::KV_TILE_BYTES, BUFB = SH::BUF_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * BUFB];  // [ring slot][K | V]
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

  // ---- Q fragments (B operand): lane holds Q[row][16 s + 8 h .. +7]; block a of the wave = rows +32 a ----
  QFrag q[RB];
  const int qrow0 = qt * QT + wave * 32 * RB + l31;
#pragma unroll
  for (int a = 0; a < RB; ++a) {
    const int qrow = qrow0 + 32 * a;
    const int qrow_c = qrow < tokens ? qrow : tokens - 1;
    const unsigned short* qp = base + (int64_t)qrow_c * ld + hd * 64 + 8 * h;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[a].q[i] = *reinterpret_cast<const s16x8_t*>(qp + 16 * i);
  }

  // ---- LDS-DMA staging: which (row, chunk) each lane fetches so that the lane-linear destination is the image.
  //      Piece i of a wave covers linear 16-byte positions [i * 256 + tid, ...) of an operand image. ----
  int voff_k[SH::PIECES], voff_v[SH::PIECES];
#pragma unroll
  for (int i = 0; i < SH::PIECES; ++i) {
    int r, cc, key, ch;
    tile_pos(i * 256 + tid, r, cc);
    voff_k[i] = (r * ld + dmodel + hd * 64 + cc * 8) * 2;
    v_pos(i * 256 + tid, key, ch);
    voff_v[i] = (key * ld + 2 * dmodel + hd * 64 + ch * 8) * 2;
  }
  const int tile_stride = KT * ld * 2;
  const int nt = (tokens + KT - 1) / KT;
  const unsigned dma_dst = (unsigned)(size_t)LDS_PTR(smem) + (__builtin_amdgcn_readfirstlane(tid & ~63) << 4);
  // the last tile carries its offset in the range-checked voffset (see attention.hip)
#define PIPE_STAGE_TILE(t_, bufi_)                                                                  \
  {                                                                                                 \
    const int so_ = (t_) * tile_stride;                                                             \
    const unsigned dst_ = dma_dst + (bufi_) * BUFB;                                                 \
    if ((t_) == nt - 1) {                                                                           \
      _Pragma("unroll") for (int i_ = 0; i_ < SH::PIECES; ++i_) {                                   \
        lds_dma16(rsrc, dst_ + i_ * 4096, voff_k[i_] + so_, 0);                                     \
        lds_dma16(rsrc, dst_ + KVB + i_ * 4096, voff_v[i_] + so_, 0);                               \
      }                                                                                             \
    } else {                                                                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < SH::PIECES; ++i_) {                                   \
        lds_dma16(rsrc, dst_ + i_ * 4096, voff_k[i_], so_);                                         \
        lds_dma16(rsrc, dst_ + KVB + i_ * 4096, voff_v[i_], so_);                                   \
      }                                                                                             \
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
#pragma unroll
  for (int a = 0; a < RB; ++a)     // Q loads retired here, not re-waited inside the loop
    asm volatile("" : "+v"(q[a].q[0]), "+v"(q[a].q[1]), "+v"(q[a].q[2]), "+v"(q[a].q[3]));

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

  const bool active = __builtin_amdgcn_readfirstlane(qt * QT + wave * 32 * RB) < tokens;
  if (!active) {   // all rows past the end: keep staging and synchronising, skip the arithmetic
    for (int t = 1; t < nt; ++t) {
      const int b2 = (t + 2) % NBUF;
      PIPE_TILE_BARRIER(b2)
    }
    return;
  }

  AttnState st[RB];
  f32x16_t sA[RB], sB[RB];         // S of even / odd half steps
  s16x8_t pA0[RB], pA1[RB], pB0[RB], pB1[RB];   // packed P of even / odd half steps
  KFrag kA, kB;                    // K fragments consumed in even / odd slots
  VFrag vA = {}, vB = {};          // V fragments consumed in even / odd slots
#pragma unroll
  for (int a = 0; a < RB; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[a].o0[r] = 0.f; st[a].o1[r] = 0.f; st[a].negm[r] = 0.f; }
    st[a].l_run = 0.f;
    pA0[a] = s16x8_t{}; pA1[a] = s16x8_t{}; pB0[a] = s16x8_t{}; pB1[a] = s16x8_t{};
  }
  {
    // S(0) and the first maximum: M is fixed by the first 32 keys (key 0 is always valid)
    load_k<0>(b, kB);
    load_k<4096>(b, kA);          // slot 0 forms S(1) from the second 32 keys of tile 0
#pragma unroll
    for (int a = 0; a < RB; ++a) {
      sA[a] = score_mfma<DT>(kB, q[a], st[a].negm);
      if (nt == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (acc_row(r, h) >= tokens) sA[a][r] = -INFINITY;
      }
      const float tmax = tile_max(sA[a]);
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[a].negm[r] = -tmax; sA[a][r] -= tmax; }
    }
  }

  // Slot j of tile t (ring slot B), h = t SLOTS + j:
  //   softmax S(h) | S(h+1) = k Q^T (k fetched by slot h-1) | O += v P(h-1) (v fetched by slot h-1)
  //   | fetch K(h+2): half j+2 of this tile, or half j+2-SLOTS of the next one, and V(h): half j of this tile
#define PIPE_SLOT(B, J, FIRST, LASTT, SC, SN, KC, VC, KN, VN, PP0, PP1, PC0, PC1)                                                   \
  {                                                                                                                    \
    constexpr int BN_ = ((B) + 1) % NBUF;                                                                              \
    constexpr int ck_ = (B) * BUFB + 4096 * (J);                                                                       \
    constexpr int nk_ = ((J) + 2 < SLOTS) ? (B) * BUFB + 4096 * ((J) + 2) : BN_ * BUFB + 4096 * ((J) + 2 - SLOTS);     \
    constexpr int nv_ = (B) * BUFB + KVB + 4096 * (J);                                                                 \
    constexpr bool pfk_ = !((LASTT) && (J) + 2 >= SLOTS);                                                              \
    constexpr bool dos_ = !((LASTT) && (J) + 1 >= SLOTS);                                                              \
    constexpr bool doo_ = !((FIRST) && (J) == 0);                                                                      \
    attn_slot<DT, ABL, RB, ck_, nk_, nv_, pfk_, dos_, doo_, LASTT>(b, q, st, SC, SN, KC, VC, KN, VN, PP0, PP1, PC0, PC1,  \
                                                                 t * KT + 32 * (J), tokens, h);                        \
  }
#define PIPE_TILE(B)                                                                                                   \
  {                                                                                                                    \
    PIPE_SLOT(B, 0, false, false, sA, sB, kA, vA, kB, vB, pB0, pB1, pA0, pA1)                                          \
    PIPE_SLOT(B, 1, false, false, sB, sA, kB, vB, kA, vA, pA0, pA1, pB0, pB1)                                          \
    if constexpr (SLOTS == 4) {                                                                                        \
      PIPE_SLOT(B, 2, false, false, sA, sB, kA, vA, kB, vB, pB0, pB1, pA0, pA1)                                        \
      PIPE_SLOT(B, 3, false, false, sB, sA, kB, vB, kA, vA, pA0, pA1, pB0, pB1)                                        \
    }                                                                                                                  \
    ++t;                                                                                                               \
  }
  // the same tile through the run-time slot: ring position, first / last handling and masking decided at run time
#define COLD_SLOT(J, SC, SN, KC, VC, KN, VN, PP0, PP1, PC0, PC1)                                                       \
  {                                                                                                                    \
    const int ck_ = rb_ * BUFB + 4096 * (J);                                                                           \
    const int nk_ = ((J) + 2 < SLOTS) ? rb_ * BUFB + 4096 * ((J) + 2) : rbn_ * BUFB + 4096 * ((J) + 2 - SLOTS);        \
    const int nv_ = rb_ * BUFB + KVB + 4096 * (J);                                                                     \
    attn_slot_rt<DT, RB>(b, q, st, SC, SN, KC, VC, KN, VN, PP0, PP1, PC0, PC1, ck_, nk_, nv_,                           \
                         !(last_ && (J) + 2 >= SLOTS), !(last_ && (J) + 1 >= SLOTS), !(t == 0 && (J) == 0), last_,     \
                         t * KT + 32 * (J), tokens, h);                                                                \
  }
#define COLD_TILE()                                                                                                    \
  {                                                                                                                    \
    const int rb_ = t % NBUF, rbn_ = (rb_ + 1) % NBUF;                                                                 \
    const bool last_ = t == nt - 1;                                                                                    \
    COLD_SLOT(0, sA, sB, kA, vA, kB, vB, pB0, pB1, pA0, pA1)                                                           \
    COLD_SLOT(1, sB, sA, kB, vB, kA, vA, pA0, pA1, pB0, pB1)                                                           \
    if constexpr (SLOTS == 4) {                                                                                        \
      COLD_SLOT(2, sA, sB, kA, vA, kB, vB, pB0, pB1, pA0, pA1)                                                         \
      COLD_SLOT(3, sB, sA, kB, vB, kA, vA, pA0, pA1, pB0, pB1)                                                         \
    }                                                                                                                  \
    ++t;                                                                                                               \
  }
  int t = 0;
  unsigned long long stamp_c0 = 0, stamp_r0 = 0;
  if constexpr (ABL == 5) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
  if constexpr (ABL == 0 || ABL == 5 || ABL == 6) {
    COLD_TILE()                                   // tile 0 (published by the prologue barrier)
    while (t + 3 <= nt - 1) {                     // steady state: t % 3 == 1 here
      PIPE_TILE_BARRIER(0) PIPE_TILE(1)
      PIPE_TILE_BARRIER(1) PIPE_TILE(2)
      PIPE_TILE_BARRIER(2) PIPE_TILE(0)
    }
    while (t < nt) {                              // up to two left-over tiles and the ragged last one
      const int b2_ = (t + 2) % NBUF;
      PIPE_TILE_BARRIER(b2_)
      COLD_TILE()
    }
  } else {                                        // timing-only builds: every middle tile through the templated slot
    COLD_TILE()
    while (t + 3 <= nt - 1) {
      PIPE_TILE_BARRIER(0) PIPE_TILE(1)
      PIPE_TILE_BARRIER(1) PIPE_TILE(2)
      PIPE_TILE_BARRIER(2) PIPE_TILE(0)
    }
    while (t < nt) {
      const int b2_ = (t + 2) % NBUF;
      PIPE_TILE_BARRIER(b2_)
      COLD_TILE()
    }
  }
#undef COLD_TILE
#undef COLD_SLOT
#undef PIPE_TILE
#undef PIPE_SLOT
#undef PIPE_TILE_BARRIER
#undef PIPE_STAGE_TILE
  // the output product of the very last half step: its V fragments were fetched by the last slot (odd parity -> vA, pB)
#pragma unroll
  for (int a = 0; a < RB; ++a) out_mfma<DT, RB == 2>(vA, pB0[a], pB1[a], st[a].o0, st[a].o1);
  if constexpr (RB == 2) {
#pragma unroll
    for (int a = 0; a < RB; ++a) mfma_acc_fence(st[a].o0, st[a].o1);
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
#pragma unroll
  for (int a = 0; a < RB; ++a) {
    float l_tot;
    {
      const unsigned lb = __float_as_uint(st[a].l_run);
      const auto sw = __builtin_amdgcn_permlane32_swap(lb, lb, false, false);
      l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l_tot;
    const int qrow = qrow0 + 32 * a;
    if (qrow < tokens) {
      unsigned short* orow = out + ((int64_t)bi * tokens + qrow) * dmodel + hd * 64 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack2_h16<DT>(st[a].o0[4 * g + 0] * inv, st[a].o0[4 * g + 1] * inv);
        pk.y = pack2_h16<DT>(st[a].o0[4 * g + 2] * inv, st[a].o0[4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + 8 * g) = pk;
        pk.x = pack2_h16<DT>(st[a].o1[4 * g + 0] * inv, st[a].o1[4 * g + 1] * inv);
        pk.y = pack2_h16<DT>(st[a].o1[4 * g + 2] * inv, st[a].o1[4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + 32 + 8 * g) = pk;
      }
    }
  }
}

}  // namespace

// C++ linkage: called by vittf_attention (attention.hip) for q_prescaled = 1 unless VITTF_ATTN_PIPE=0.
//   rows_per_wave 32 -> 2 waves per SIMD; anything else -> 64 rows per wave, one wave per SIMD (the default).
//   VITTF_ATTN_ABLATE=1..5 (fp16 only) launches a timing-only build with one component removed (wrong results; tools/).
int vittf_attention_pipe(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype,
                         int32_t rows_per_wave, hipStream_t st) {
  const int rb = rows_per_wave == 32 ? 1 : 2;
  const int q_tiles = (tokens + 128 * rb - 1) / (128 * rb);
  const int total = batch * heads * q_tiles;
#define PIPE_LAUNCH(DTV, ABLV, RBV)                                                                                  \
  hipLaunchKernelGGL((attn_pipe_kernel<DTV, ABLV, RBV>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv, \
                     (unsigned short*)out, tokens, heads, q_tiles, total)
  const char* e = getenv("VITTF_ATTN_ABLATE");
  const int abl = e ? atoi(e) : 0;
  if (rb == 1) {
    if (dtype == VITTF_BF16) PIPE_LAUNCH(VITTF_BF16, 0, 1);
    else if (abl == 5) PIPE_LAUNCH(VITTF_FP16, 5, 1);
    else if (abl == 6) PIPE_LAUNCH(VITTF_FP16, 6, 1);
    else if (abl == 4) PIPE_LAUNCH(VITTF_FP16, 4, 1);
    else if (abl == 1) PIPE_LAUNCH(VITTF_FP16, 1, 1);
    else if (abl == 2) PIPE_LAUNCH(VITTF_FP16, 2, 1);
    else PIPE_LAUNCH(VITTF_FP16, 0, 1);
  } else {
    if (dtype == VITTF_BF16) PIPE_LAUNCH(VITTF_BF16, 0, 2);
    else if (abl == 1) PIPE_LAUNCH(VITTF_FP16, 1, 2);
    else if (abl == 2) PIPE_LAUNCH(VITTF_FP16, 2, 2);
    else if (abl == 3) PIPE_LAUNCH(VITTF_FP16, 3, 2);
    else if (abl == 4) PIPE_LAUNCH(VITTF_FP16, 4, 2);
    else if (abl == 5) PIPE_LAUNCH(VITTF_FP16, 5, 2);
    else PIPE_LAUNCH(VITTF_FP16, 0, 2);
  }
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
