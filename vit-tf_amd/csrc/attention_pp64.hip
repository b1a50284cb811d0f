// Flash attention forward, head dim 64, pre-scaled q: 64 query rows per wave as TWO 32-row blocks that take turns
// ("ping-pong"), two waves per SIMD.  Same contract as attention.hip (softmax(q k^T / 8) v per head; Attention.forward of
// the upstream model the reference calls, infer.py:177), same LDS images, same lazy running maximum.
//
// What a 32-rows-per-wave kernel (round 2's software-pipelined attention_pipe.hip, 0.40 of peak) pays per MFMA besides the softmax arithmetic is operand
// delivery and synchronisation: 12 LDS fragment instructions, half a barrier and 2 LDS-DMA issues per 8 MFMAs.  Here every
// K / V^T fragment feeds the MFMAs of both blocks of the wave, and a 64-key tile serves 256 query rows of the workgroup:
// all three halve.  The software pipeline needs no second score tile either -- the two blocks ARE the two stages:
//
//   phase A(h):  VALU  softmax of S_a(h) -> P_a(h)         matrix  S_b(h)   = K(h)   Q_b^T - M_b,  O_b += V(h-1)^T P_b(h-1)^T
//   phase B(h):  VALU  softmax of S_b(h) -> P_b(h)         matrix  S_a(h+1) = K(h+1) Q_a^T - M_a,  O_a += V(h)^T   P_a(h)^T
//
// (h = 32-key half step).  The fragments K(h+1), V(h) are shared by B(h) and A(h+1); during A(h+1) each fragment register
// is refilled with K(h+2) / V(h+1) right behind the MFMA that read it last, so ONE fragment set (32 registers) suffices and
// every LDS read has most of a phase to land.  Registers: O 64 + -M tiles 32 + Q 32 + S 32 + P 16 + fragments 32 = 208 of 256.
//
//   * ring protocol (3 buffers of K | V, 48 KB, two workgroups per CU): the barrier in front of tile t publishes tile t+1
//     (its first K half is fetched by A(2t+1)) and retires tile t-1 (last read by A(2t-1)), whose buffer takes tile t+2.
//   * lazy maximum: a phase's row sum says the 16-bit P would overflow (rare, wave-uniform) -> raw scores of that block
//     again from LDS, the true maximum, O / l / -M of THAT block rescaled; the other block and the scores in flight are
//     not touched (each block has its own M).
//   * S^T = K Q^T so a lane owns one query column; the P registers feed O^T = V^T P^T directly; V^T fragments by
//     ds_read_b64_tr_b16; the ragged last tile is range-checked by the buffer descriptor and masked to -inf.
#include "attn_common.h"

#include <stdlib.h>

namespace {

// timing-only builds (tools/attn_variants.sh; never in libvittf.so, results wrong by construction): the LDS-DMA of every
// PP_DMA_EVERY-th tile only (2 = what a 512-row workgroup would issue per query row; a skipped tile's buffer still holds an
// older tile: finite scores), 0 = the first three tiles only (the ring is filled once)
#ifndef PP_DMA_EVERY
#define PP_DMA_EVERY 1
#endif
__device__ __forceinline__ constexpr bool pp_dma_on(int t) {
  return PP_DMA_EVERY == 1 || (PP_DMA_EVERY > 1 ? t % (PP_DMA_EVERY > 1 ? PP_DMA_EVERY : 1) == 0 : t < 3);
}

constexpr int NBUF = 3;
constexpr int QT = 256;                  // query rows per workgroup: 4 waves x 2 blocks x 32 rows
constexpr int KT = ATT_KT;               // keys per tile
constexpr int KVB = ATT_KV_TILE_BYTES;   // bytes per operand and tile
constexpr int BUFB = ATT_BUF_BYTES;      // K | V

struct LdsBases { const char *ka0, *ka1, *ka2, *ka3, *va0, *va1; };
struct KFrag { s16x8_t k[4]; };          // K rows (A operand of S^T = K Q^T), one per 16-wide d chunk
struct VFrag { s16x8_t v[4]; };          // V^T (A operand of O^T = V^T P^T): [2 x key step s2 + d half dvt]
struct QFrag { s16x8_t q[4]; };          // Q rows (B operand), resident
struct Blk {                             // running state of one 32-row query block
  f32x16_t o0, o1, negm;
  float l_run;
};

__device__ __forceinline__ s16x8_t ld_k(const LdsBases& b, int i, int off) {
  const char* base = i == 0 ? b.ka0 : i == 1 ? b.ka1 : i == 2 ? b.ka2 : b.ka3;
  return *reinterpret_cast<const s16x8_t*>(base + off);
}
__device__ __forceinline__ s16x8_t ld_v(const LdsBases& b, int j, int off) {
  const int imm = off + 2048 * (j >> 1) + 512 * (j & 1);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(b.va0 + imm));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(b.va1 + imm + 1024));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int DT> __device__ __forceinline__ f32x16_t score_mfma(const KFrag& k, const QFrag& q, f32x16_t c) {
#pragma unroll
  for (int i = 0; i < 4; ++i) c = mfma32<DT>(k.k[i], q.q[i], c);
  return c;
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

// executions of the overflow branch (statistics, vittf_attention_rescale_count): one relaxed atomic per execution, on a
// path that real weights take a handful of times per launch
__device__ unsigned g_rescales;

template <int DT> __device__ __forceinline__ constexpr float p_limit() { return DT == VITTF_FP16 ? 8192.f : 1073741824.f; }

// The slow path of a phase: block X's values have outgrown the 16-bit P at its current M.  Raw scores of the half step
// (K fragments at byte offset ck_off) once more, the true row maximum, everything block X has accumulated rescaled to the
// new M, P and its row sum rebuilt.
template <int DT>
__device__ __forceinline__ float rescale_block(const LdsBases& b, Blk& X, const QFrag& qX, s16x8_t (&pX)[2], int ck_off, bool mask,
                                               int key0, int tokens, int h) {
  if ((threadIdx.x & 63) == 0) atomicAdd(&g_rescales, 1u);
  f32x16_t raw;
#pragma unroll
  for (int r = 0; r < 16; ++r) raw[r] = 0.f;
  KFrag kh;
#pragma unroll
  for (int i = 0; i < 4; ++i) kh.k[i] = ld_k(b, i, ck_off);
  raw = score_mfma<DT>(kh, qX, raw);
  if (mask) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (key0 + acc_row(r, h) >= tokens) raw[r] = -INFINITY;
  }
  const float tmax = tile_max(raw);
  const float delta = fmaxf(tmax + X.negm[0], 0.f);                       // M moves up by delta (per query column)
  const float alpha = __builtin_amdgcn_exp2f(-delta);
  X.l_run *= alpha;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    X.o0[r] *= alpha;
    X.o1[r] *= alpha;
    X.negm[r] -= delta;
  }
  float psum0 = 0.f, psum1 = 0.f;
  u32x4_t u0, u1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float e0 = __builtin_amdgcn_exp2f(raw[2 * i] + X.negm[2 * i]);
    const float e1 = __builtin_amdgcn_exp2f(raw[2 * i + 1] + X.negm[2 * i + 1]);
    psum0 += e0;
    psum1 += e1;
    const unsigned w = pack2_h16<DT>(e0, e1);
    if (i < 4) u0[i] = w; else u1[i - 4] = w;
  }
  pX[0] = __builtin_bit_cast(s16x8_t, u0);
  pX[1] = __builtin_bit_cast(s16x8_t, u1);
  return psum0 + psum1;
}

// One phase of the steady state (see the header): softmax of block X's score tile sX (consumed) -> packed P in pX, beside
// 4 + 4 MFMAs for the OTHER block Y -- S_Y = kf Q_Y^T - M_Y into sY, O_Y += vf^T pY^T -- and, with LOAD, the refill of each
// fragment register right behind the MFMA that read it last (K from byte offset nk_off, V^T from nv_off).  Eight gaps,
// pinned by scheduling fences: one MFMA + one unit of softmax arithmetic (two v_exp, two row-sum adds, one conversion) +
// its share of the 12 fragment reads -- an in-order wave that meets a busy matrix pipe stalls with all its VALU work
// behind it, so MFMAs in clusters idle both pipes.  (Scores first, then the output product: 0.4 % faster than alternating
// them; a dependent 32x32x16 MFMA issued a whole gap behind its producer does not wait.)
template <int DT, bool LOAD, bool MASK>
__device__ __forceinline__ void pp_phase(const LdsBases& b, Blk& X, const QFrag& qX, f32x16_t& sX, s16x8_t (&pX)[2], Blk& Y,
                                         const QFrag& qY, f32x16_t& sY, const s16x8_t (&pY)[2], KFrag& kf, VFrag& vf, int ck_off,
                                         int nk_off, int nv_off, int key0, int tokens, int h) {
  if constexpr (MASK) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (key0 + acc_row(r, h) >= tokens) sX[r] = -INFINITY;
  }
  float psum0 = 0.f, psum1 = 0.f;
  u32x4_t pk0, pk1;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int i = g & 3;
    if (g < 4) {          // the four score MFMAs first (the K fragments are free for their refill early), then the output product
      sY = mfma32<DT>(kf.k[i], qY.q[i], i == 0 ? Y.negm : sY);
      if constexpr (LOAD) kf.k[i] = ld_k(b, i, nk_off);
    } else {
      if (i & 1) Y.o1 = mfma32<DT>(vf.v[i], pY[i >> 1], Y.o1);
      else       Y.o0 = mfma32<DT>(vf.v[i], pY[i >> 1], Y.o0);
      if constexpr (LOAD) vf.v[i] = ld_v(b, i, nv_off);
    }
    const float e0 = __builtin_amdgcn_exp2f(sX[2 * g]);
    const float e1 = __builtin_amdgcn_exp2f(sX[2 * g + 1]);
    if (g == 0) { psum0 = e0; psum1 = e1; } else { psum0 += e0; psum1 += e1; }
    const unsigned w = pack2_h16<DT>(e0, e1);
    if (g < 4) pk0[g] = w; else pk1[g - 4] = w;
    __builtin_amdgcn_sched_barrier(0);
  }
  float ps = psum0 + psum1;
  pX[0] = __builtin_bit_cast(s16x8_t, pk0);
  pX[1] = __builtin_bit_cast(s16x8_t, pk1);
  if (__builtin_expect(__any(!(ps <= p_limit<DT>())), 0)) ps = rescale_block<DT>(b, X, qX, pX, ck_off, MASK, key0, tokens, h);
  X.l_run += ps;
}

// The same phase with RUN-TIME ring offsets and flags, for the tiles outside the steady-state loop (the first tile, the up to
// two tiles the 3-tile loop leaves over, the ragged last tile): a handful of executions per workgroup, nothing pinned, ONE
// code path instead of a template instance per combination (round 2: seven instances spilled ~110 registers).
template <int DT>
__device__ __forceinline__ void pp_phase_rt(const LdsBases& b, Blk& X, const QFrag& qX, f32x16_t& sX, s16x8_t (&pX)[2], Blk& Y,
                                            const QFrag& qY, f32x16_t& sY, const s16x8_t (&pY)[2], KFrag& kf, VFrag& vf,
                                            int ck_off, int nk_off, int nv_off, bool do_s, bool do_o, bool load_k, bool load_v,
                                            bool mask, int key0, int tokens, int h) {
  if (do_s) sY = score_mfma<DT>(kf, qY, Y.negm);
  if (do_o) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j & 1) Y.o1 = mfma32<DT>(vf.v[j], pY[j >> 1], Y.o1);
      else       Y.o0 = mfma32<DT>(vf.v[j], pY[j >> 1], Y.o0);
    }
  }
  if (load_k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) kf.k[i] = ld_k(b, i, nk_off);
  }
  if (load_v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) vf.v[j] = ld_v(b, j, nv_off);
  }
  if (mask) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (key0 + acc_row(r, h) >= tokens) sX[r] = -INFINITY;
  }
  float psum0 = 0.f, psum1 = 0.f;
  u32x4_t pk0, pk1;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float e0 = __builtin_amdgcn_exp2f(sX[2 * g]);
    const float e1 = __builtin_amdgcn_exp2f(sX[2 * g + 1]);
    psum0 += e0;
    psum1 += e1;
    const unsigned w = pack2_h16<DT>(e0, e1);
    if (g < 4) pk0[g] = w; else pk1[g - 4] = w;
  }
  float ps = psum0 + psum1;
  pX[0] = __builtin_bit_cast(s16x8_t, pk0);
  pX[1] = __builtin_bit_cast(s16x8_t, pk1);
  if (__builtin_expect(__any(!(ps <= p_limit<DT>())), 0)) ps = rescale_block<DT>(b, X, qX, pX, ck_off, mask, key0, tokens, h);
  X.l_run += ps;
}

template <int DT>
__global__ __launch_bounds__(256, 2) void attn_pp64_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                           int tokens, int heads, int q_tiles, int total) {
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

  // ---- Q fragments (B operand): lane holds Q[row][16 s + 8 h .. +7]; block b of the wave = rows + 32 ----
  QFrag qa, qb;
  const int qrow_a = qt * QT + wave * 64 + l31, qrow_b = qrow_a + 32;
  {
    const unsigned short* pa_ = base + (int64_t)(qrow_a < tokens ? qrow_a : tokens - 1) * ld + hd * 64 + 8 * h;
    const unsigned short* pb_ = base + (int64_t)(qrow_b < tokens ? qrow_b : tokens - 1) * ld + hd * 64 + 8 * h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qa.q[i] = *reinterpret_cast<const s16x8_t*>(pa_ + 16 * i);
      qb.q[i] = *reinterpret_cast<const s16x8_t*>(pb_ + 16 * i);
    }
  }

  // ---- LDS-DMA staging: which (row, chunk) each lane fetches so that the lane-linear destination is the image.
  //      Piece i of a wave covers linear 16-byte positions [i * 256 + tid, ...) of an operand image. ----
  //      The second piece of an operand is the first one 32 rows further on (tile_pos / v_pos of position + 256: the same
  //      chunk of row + 32), so one voffset per operand serves both and the 32-row step rides in the scalar offset.
  int voff_k, voff_v;
  {
    int r, cc, key, ch;
    tile_pos(tid, r, cc);
    voff_k = (r * ld + dmodel + hd * 64 + cc * 8) * 2;
    v_pos(tid, key, ch);
    voff_v = (key * ld + 2 * dmodel + hd * 64 + ch * 8) * 2;
  }
  const int tile_stride = KT * ld * 2, half_stride = 32 * ld * 2;
  const int nt = (tokens + KT - 1) / KT;
  const unsigned dma_dst = (unsigned)(size_t)LDS_PTR(smem) + (__builtin_amdgcn_readfirstlane(tid & ~63) << 4);
  // the last tile carries its offset in the range-checked voffset (see attention.hip)
#define PP_STAGE_TILE(t_, bufi_)                                                                    \
  {                                                                                                 \
    const int so_ = (t_) * tile_stride;                                                             \
    const unsigned dst_ = dma_dst + (bufi_) * BUFB;                                                 \
    if (!pp_dma_on(t_)) {                                                                           \
    } else if ((t_) == nt - 1) {                                                                         \
      int vk_ = voff_k, vv_ = voff_v;   /* opaque copies: the sums below are formed here, not kept alive through the loop */ \
      asm volatile("" : "+v"(vk_), "+v"(vv_));                                                      \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                            \
        lds_dma16(rsrc, dst_ + i_ * 4096, vk_ + so_ + i_ * half_stride, 0);                         \
        lds_dma16(rsrc, dst_ + KVB + i_ * 4096, vv_ + so_ + i_ * half_stride, 0);                   \
      }                                                                                             \
    } else {                                                                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                            \
        lds_dma16(rsrc, dst_ + i_ * 4096, voff_k, so_ + i_ * half_stride);                          \
        lds_dma16(rsrc, dst_ + KVB + i_ * 4096, voff_v, so_ + i_ * half_stride);                    \
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
  PP_STAGE_TILE(0, 0)
  if (nt > 1) PP_STAGE_TILE(1, 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nt > 2) PP_STAGE_TILE(2, 2)
  // Q loads retired here, not re-waited inside the loop
  asm volatile("" : "+v"(qa.q[0]), "+v"(qa.q[1]), "+v"(qa.q[2]), "+v"(qa.q[3]), "+v"(qb.q[0]), "+v"(qb.q[1]), "+v"(qb.q[2]), "+v"(qb.q[3]));

  // The barrier in front of tile t >= 1: every wave has made its last reads of tile t-1 (the V^T fragments of its second
  // half, fetched in A(2t-1)), so that buffer takes tile t+2; tile t+1 (requested one tile ago) is published.
#define PP_TILE_BARRIER(BNEXT2)                                                                     \
  {                                                                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
    __syncthreads();                                                                                \
    if (t + 2 < nt) PP_STAGE_TILE(t + 2, BNEXT2)                                                    \
  }

  const bool active = __builtin_amdgcn_readfirstlane(qt * QT + wave * 64) < tokens;
  if (!active) {   // all rows past the end: keep staging and synchronising, skip the arithmetic
    for (int t = 1; t < nt; ++t) {
      const int b2 = (t + 2) % NBUF;
      PP_TILE_BARRIER(b2)
    }
    return;
  }

  Blk A, B;
  f32x16_t sa, sb;                 // score tiles of the two blocks
  s16x8_t pa[2], pb[2];            // their packed P (B operands of the output product, key steps 0 / 1)
  KFrag kf;
  VFrag vf = {};
#pragma unroll
  for (int r = 0; r < 16; ++r) { A.o0[r] = 0.f; A.o1[r] = 0.f; A.negm[r] = 0.f; B.o0[r] = 0.f; B.o1[r] = 0.f; B.negm[r] = 0.f; }
  A.l_run = 0.f; B.l_run = 0.f;
  pa[0] = s16x8_t{}; pa[1] = s16x8_t{}; pb[0] = s16x8_t{}; pb[1] = s16x8_t{};
  {
    // S_a(0), S_b(0) and the first maxima: M is fixed by the first 32 keys (key 0 is always valid)
#pragma unroll
    for (int i = 0; i < 4; ++i) kf.k[i] = ld_k(b, i, 0);
    sa = score_mfma<DT>(kf, qa, A.negm);
    sb = score_mfma<DT>(kf, qb, B.negm);
    if (nt == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (acc_row(r, h) >= tokens) { sa[r] = -INFINITY; sb[r] = -INFINITY; }
    }
    const float ta = tile_max(sa), tb = tile_max(sb);
#pragma unroll
    for (int r = 0; r < 16; ++r) { A.negm[r] = -ta; sa[r] -= ta; B.negm[r] = -tb; sb[r] -= tb; }
  }

  // tile t in ring slot RB_, hot: both half steps through the pinned phases
#define PP_TILE(RB_)                                                                                                        \
  {                                                                                                                         \
    constexpr int o_ = (RB_) * BUFB, on_ = (((RB_) + 1) % NBUF) * BUFB;                                                     \
    pp_phase<DT, true, false>(b, A, qa, sa, pa, B, qb, sb, pb, kf, vf, o_, o_ + 4096, o_ + KVB, 0, tokens, h);              \
    pp_phase<DT, false, false>(b, B, qb, sb, pb, A, qa, sa, pa, kf, vf, o_, 0, 0, 0, tokens, h);                            \
    pp_phase<DT, true, false>(b, A, qa, sa, pa, B, qb, sb, pb, kf, vf, o_ + 4096, on_, o_ + KVB + 4096, 0, tokens, h);      \
    pp_phase<DT, false, false>(b, B, qb, sb, pb, A, qa, sa, pa, kf, vf, o_ + 4096, 0, 0, 0, tokens, h);                     \
    ++t;                                                                                                                    \
  }
  // the same tile through the run-time phases: ring position, first / last handling and masking decided at run time
#define PP_COLD_TILE()                                                                                                      \
  {                                                                                                                         \
    const int o_ = (t % NBUF) * BUFB, on_ = ((t + 1) % NBUF) * BUFB;                                                        \
    const bool first_ = t == 0, last_ = t == nt - 1;                                                                        \
    /* a last tile of at most 32 keys (the CLS token's: N = 64 x 64 + 1) ends after its first half step */                  \
    const bool half_ = last_ && tokens - t * KT <= 32;                                                                      \
    int hc_ = h;       /* opaque: the masks' per-lane key indices are formed per tile, not carried through the loop */      \
    asm volatile("" : "+v"(hc_));                                                                                           \
    pp_phase_rt<DT>(b, A, qa, sa, pa, B, qb, sb, pb, kf, vf, o_, o_ + 4096, o_ + KVB, !first_, !first_, !half_, true, last_, \
                    t * KT, tokens, hc_);                                                                                     \
    pp_phase_rt<DT>(b, B, qb, sb, pb, A, qa, sa, pa, kf, vf, o_, 0, 0, !half_, true, false, false, last_, t * KT, tokens, hc_); \
    if (!half_) {                                                                                                           \
      pp_phase_rt<DT>(b, A, qa, sa, pa, B, qb, sb, pb, kf, vf, o_ + 4096, on_, o_ + KVB + 4096, true, true, !last_, true,    \
                      last_, t * KT + 32, tokens, hc_);                                                                       \
      pp_phase_rt<DT>(b, B, qb, sb, pb, A, qa, sa, pa, kf, vf, o_ + 4096, 0, 0, !last_, true, false, false, last_,           \
                      t * KT + 32, tokens, hc_);                                                                              \
    }                                                                                                                       \
    ++t;                                                                                                                    \
  }
  int t = 0;
  PP_COLD_TILE()                                // tile 0 (published by the prologue barrier)
  while (t + 3 <= nt - 1) {                     // steady state: t % 3 == 1 here
    PP_TILE_BARRIER(0) PP_TILE(1)
    PP_TILE_BARRIER(1) PP_TILE(2)
    PP_TILE_BARRIER(2) PP_TILE(0)
  }
  while (t < nt) {                              // up to two left-over tiles and the ragged last one
    const int b2_ = (t + 2) % NBUF;
    PP_TILE_BARRIER(b2_)
    PP_COLD_TILE()
  }
#undef PP_COLD_TILE
#undef PP_TILE
#undef PP_TILE_BARRIER
#undef PP_STAGE_TILE
  // block b's output product of the very last half step (its V^T fragments were fetched by the last A phase)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j & 1) B.o1 = mfma32<DT>(vf.v[j], pb[j >> 1], B.o1);
    else       B.o0 = mfma32<DT>(vf.v[j], pb[j >> 1], B.o0);
  }

  // ---- normalise and store: lane owns query rows qrow_a / qrow_b, columns 32 dvt + 8 g + 4 h + {0..3}
  //      (store_o_row pairs them into 16-byte runs) ----
  auto store = [&](const Blk& X, int qrow) {
    const unsigned lb = __float_as_uint(X.l_run);
    const auto sw = __builtin_amdgcn_permlane32_swap(lb, lb, false, false);
    const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
    if (qrow < tokens) store_o_row<DT>(out + ((int64_t)bi * tokens + qrow) * dmodel + hd * 64, h, X.o0, X.o1, inv);
  };
  store(A, qrow_a);
  store(B, qrow_b);
}

}  // namespace

extern "C" int64_t vittf_attention_rescale_count(int32_t reset) {
  unsigned v = 0;
  if (hipDeviceSynchronize() != hipSuccess) return VITTF_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_rescales), sizeof(v)) != hipSuccess) return VITTF_ERR_LAUNCH;
  if (reset) {
    const unsigned z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_rescales), &z, sizeof(z)) != hipSuccess) return VITTF_ERR_LAUNCH;
  }
  return (int64_t)v;
}

#ifdef PP_STANDALONE      // tools/attn_variants.sh builds this file alone
int vittf_attention_pp64(const void*, void*, int32_t, int32_t, int32_t, int32_t, hipStream_t);
extern "C" int vittf_attention_variant(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype, void* st) {
  return vittf_attention_pp64(qkv, out, batch, tokens, heads, dtype, (hipStream_t)st);
}
#endif

// C++ linkage: called by vittf_attention (attention.hip) for q_prescaled = 1
int vittf_attention_pp64(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype, hipStream_t st) {
  const int q_tiles = (tokens + QT - 1) / QT;
  const int64_t total64 = (int64_t)batch * heads * q_tiles;
  if (total64 > (1 << 30)) return VITTF_ERR_INVALID_ARG;
  const int total = (int)total64;
  if (dtype == VITTF_BF16)
    hipLaunchKernelGGL((attn_pp64_kernel<VITTF_BF16>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv,
                       (unsigned short*)out, tokens, heads, q_tiles, total);
  else if (dtype == VITTF_FP16)
    hipLaunchKernelGGL((attn_pp64_kernel<VITTF_FP16>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv,
                       (unsigned short*)out, tokens, heads, q_tiles, total);
  else
    return VITTF_ERR_INVALID_ARG;
  return vittf_check_launch();
}
