// Flash-style attention forward, head dim 64: 64 query rows per wave at TWO waves per SIMD (opt-in: VITTF_ATTN_PIPE=3).
//
// The structure of attn_kernel<DT, true> (attention.hip: S^T = K Q^T with -M as the MFMA's C input, lazy running maximum,
// the 16-bit P as the B operand of O^T = V^T P^T, LDS-DMA staging of 64-key tiles into two buffers, one barrier per tile),
// with each K fragment (ds_read_b128) and each V^T fragment (two ds_read_b64_tr_b16) feeding the MFMAs of TWO 32-row query
// blocks: 12 LDS fragment instructions per 16 MFMAs instead of per 8, half the barriers per MFMA.  Unlike the 64-row shape
// of attention_pipe.hip (388 registers, one wave per SIMD) it keeps no second score / P set and no fragment prefetch
// registers, so it fits the 256 registers of two waves per SIMD; the overlap of softmax VALU and matrix work comes from
// the other wave of the SIMD.  Workgroup = 4 waves = 256 query rows of one (slice, head).
#include "attn_common.h"

#include <stdlib.h>

namespace {

constexpr int QT64 = 256;   // query rows per workgroup
constexpr int KT = ATT_KT, KV_TILE_BYTES = ATT_KV_TILE_BYTES, BUF_BYTES = ATT_BUF_BYTES;

struct Rows64 {             // per 32-row query block
  s16x8_t q[4];
  f32x16_t o0, o1, negm;
  float l_run;
};

// softmax of one 32-key half for one query block: scores (in exp2 units, -M already added by the MFMA) -> 16-bit P
template <int DT>
__device__ __forceinline__ void softmax_half(f32x16_t& sacc, Rows64& r, bool first, s16x8_t (&pf)[2]) {
  constexpr float THR = DT == VITTF_FP16 ? 8192.f : 1073741824.f;
  float p[16];
  float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    p[i] = __builtin_amdgcn_exp2f(sacc[i]);
    p[i + 1] = __builtin_amdgcn_exp2f(sacc[i + 1]);
    psum0 += p[i];
    psum1 += p[i + 1];
  }
  float ps = psum0 + psum1;
  if (first || __any(!(ps <= THR))) {   // rare, wave-uniform: raise M, rescale what was accumulated (attention.hip)
    float tmax = max3_f32(sacc[0], sacc[1], sacc[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) tmax = max3_f32(tmax, sacc[i], sacc[i + 1]);
    tmax = max3_f32(tmax, sacc[15], sacc[15]);
    const unsigned tb = __float_as_uint(tmax);
    const auto sw = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);
    tmax = max3_f32(tmax, __uint_as_float(sw[0]), __uint_as_float(sw[1]));
    const float delta = first ? tmax : max3_f32(tmax, 0.f, 0.f);
    if (!first) {
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      r.l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { r.o0[i] *= alpha; r.o1[i] *= alpha; }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) r.negm[i] -= delta;
    psum0 = 0.f; psum1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      p[i] = __builtin_amdgcn_exp2f(sacc[i] - delta);
      p[i + 1] = __builtin_amdgcn_exp2f(sacc[i + 1] - delta);
      psum0 += p[i];
      psum1 += p[i + 1];
    }
    ps = psum0 + psum1;
  }
  r.l_run += ps;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    u32x4_t u;
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = pack2_h16<DT>(p[8 * s2 + 2 * j], p[8 * s2 + 2 * j + 1]);
    pf[s2] = __builtin_bit_cast(s16x8_t, u);
  }
}

// one 64-key tile for this wave's two query blocks
template <int DT, int BUF, bool LAST>
__device__ __forceinline__ void attn64_tile(const char* ka0, const char* ka1, const char* ka2, const char* ka3,
                                            const char* va0, const char* va1, Rows64& ra, Rows64& rb, int t, int tokens, int h) {
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    constexpr int kb = BUF * BUF_BYTES;
    __builtin_amdgcn_sched_barrier(0);   // (fences between the phases: hoisted fragment reads of the next phase spill at 256 registers)
    const s16x8_t k0 = *reinterpret_cast<const s16x8_t*>(ka0 + kb + 4096 * kt);
    const s16x8_t k1 = *reinterpret_cast<const s16x8_t*>(ka1 + kb + 4096 * kt);
    const s16x8_t k2 = *reinterpret_cast<const s16x8_t*>(ka2 + kb + 4096 * kt);
    const s16x8_t k3 = *reinterpret_cast<const s16x8_t*>(ka3 + kb + 4096 * kt);
    const bool first = (t == 0) && (kt == 0);
    s16x8_t pa[2], pb[2];
    // the two blocks one after the other (scores of one block live at a time; the K fragments stay for the second chain)
    {
      f32x16_t sa = ra.negm;                           // C input = -M: the chain returns s' - M
      sa = mfma32<DT>(k0, ra.q[0], sa);
      sa = mfma32<DT>(k1, ra.q[1], sa);
      sa = mfma32<DT>(k2, ra.q[2], sa);
      sa = mfma32<DT>(k3, ra.q[3], sa);
      if constexpr (LAST) {  // ragged last tile: keys >= tokens contribute nothing
#pragma unroll
        for (int r = 0; r < 16; ++r) if (t * KT + 32 * kt + acc_row(r, h) >= tokens) sa[r] = -INFINITY;
      }
      softmax_half<DT>(sa, ra, first, pa);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      f32x16_t sb = rb.negm;
      sb = mfma32<DT>(k0, rb.q[0], sb);
      sb = mfma32<DT>(k1, rb.q[1], sb);
      sb = mfma32<DT>(k2, rb.q[2], sb);
      sb = mfma32<DT>(k3, rb.q[3], sb);
      if constexpr (LAST) {
#pragma unroll
        for (int r = 0; r < 16; ++r) if (t * KT + 32 * kt + acc_row(r, h) >= tokens) sb[r] = -INFINITY;
      }
      softmax_half<DT>(sb, rb, first, pb);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- O^T += V^T P^T for these 32 keys: every V^T fragment feeds both query blocks ----
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int dvt = 0; dvt < 2; ++dvt) {
        constexpr int vb = BUF * BUF_BYTES + KV_TILE_BYTES;
        const int imm = vb + 4096 * kt + 2048 * s2 + 512 * dvt;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va0 + imm));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va1 + imm + 1024));
        const s16x8_t vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        if (dvt == 0) { ra.o0 = mfma32<DT>(vf, pa[s2], ra.o0); rb.o0 = mfma32<DT>(vf, pb[s2], rb.o0); }
        else          { ra.o1 = mfma32<DT>(vf, pa[s2], ra.o1); rb.o1 = mfma32<DT>(vf, pb[s2], rb.o1); }
      }
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256, 2) void attn64_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                        int tokens, int heads, int q_tiles, int total) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];  // [buffer][K | V]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;

  const int item = xcd_remap(blockIdx.x, total);
  const int qt = item % q_tiles;
  const int bh = item / q_tiles;
  const int hd = bh % heads, b = bh / heads;
  const int dmodel = heads * 64;
  const int ld = 3 * dmodel;
  const unsigned short* base = qkv + (int64_t)b * tokens * ld;
  const i32x4_t rsrc = lds_dma_rsrc(base, (unsigned)((int64_t)tokens * ld * 2));

  Rows64 ra, rb;
  const int qrow_a = qt * QT64 + wave * 64 + l31, qrow_b = qrow_a + 32;
  {
    const unsigned short* qa = base + (int64_t)(qrow_a < tokens ? qrow_a : tokens - 1) * ld + hd * 64 + 8 * h;
    const unsigned short* qb = base + (int64_t)(qrow_b < tokens ? qrow_b : tokens - 1) * ld + hd * 64 + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ra.q[s] = *reinterpret_cast<const s16x8_t*>(qa + 16 * s);
      rb.q[s] = *reinterpret_cast<const s16x8_t*>(qb + 16 * s);
    }
  }

  // ---- staging by LDS-DMA, addresses and images exactly as attention.hip ----
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
  // (the last tile carries its offset in the per-lane voffset, the operand the descriptor's range check covers: rows past
  //  the slice arrive as zeros -- attention.hip)
#define ATTN_STAGE_TILE(t, BUFI)                                                                              \
  {                                                                                                           \
    const int so_ = (t) * tile_stride;                                                                        \
    if ((t) == nt - 1) {                                                                                      \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES, voff_k0 + so_, 0);                                        \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + 4096, voff_k1 + so_, 0);                                 \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES, voff_v0 + so_, 0);                        \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES + 4096, voff_v1 + so_, 0);                 \
    } else {                                                                                                  \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES, voff_k0, so_);                                            \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + 4096, voff_k1, so_);                                     \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES, voff_v0, so_);                            \
      lds_dma16(rsrc, dma_dst + (BUFI) * BUF_BYTES + KV_TILE_BYTES + 4096, voff_v1, so_);                     \
    }                                                                                                         \
  }

  const int p_l = l31 >> 1;
  const int bslot = (((l31 & 1) << 3) | h) ^ (p_l & 15);
  const char* const ka0 = smem + (p_l << 8) + ((bslot ^ 0) << 4);
  const char* const ka1 = smem + (p_l << 8) + ((bslot ^ 2) << 4);
  const char* const ka2 = smem + (p_l << 8) + ((bslot ^ 4) << 4);
  const char* const ka3 = smem + (p_l << 8) + ((bslot ^ 6) << 4);
  const int g16 = lane >> 4;
  const int tr_q = (lane & 15) >> 2;
  const int tr_p = lane & 3;
  const int tr_ch = 2 * (g16 & 1) + (tr_p >> 1);
  const int vl0 = 64 * (4 * h + tr_q) + 16 * (tr_ch ^ h) + 8 * (tr_p & 1);
  const char* const va0 = smem + vl0;
  const char* const va1 = smem + (vl0 ^ 32);

#pragma unroll
  for (int r = 0; r < 16; ++r) { ra.o0[r] = 0.f; ra.o1[r] = 0.f; ra.negm[r] = 0.f; rb.o0[r] = 0.f; rb.o1[r] = 0.f; rb.negm[r] = 0.f; }
  ra.l_run = 0.f; rb.l_run = 0.f;

  ATTN_STAGE_TILE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  asm volatile("" : "+v"(ra.q[0]), "+v"(ra.q[1]), "+v"(ra.q[2]), "+v"(ra.q[3]), "+v"(rb.q[0]), "+v"(rb.q[1]), "+v"(rb.q[2]), "+v"(rb.q[3]));

  // waves whose 64 rows are all past the end keep staging and synchronising but skip the arithmetic
  const bool active = __builtin_amdgcn_readfirstlane(qt * QT64 + wave * 64) < tokens;
  int t = 0;
  if (!active) {
    for (; t + 1 < nt; ++t) {
      if (t & 1) ATTN_STAGE_TILE(t + 1, 0) else ATTN_STAGE_TILE(t + 1, 1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    return;
  }
#define ATTN_STEP(BUFC, BUFN)                                                                    \
  {                                                                                              \
    ATTN_STAGE_TILE(t + 1, BUFN)                                                                 \
    attn64_tile<DT, BUFC, false>(ka0, ka1, ka2, ka3, va0, va1, ra, rb, t, tokens, h);            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
    __syncthreads();                                                                             \
    ++t;                                                                                         \
  }
  while (t + 2 < nt) {
    ATTN_STEP(0, 1)
    ATTN_STEP(1, 0)
  }
  if (t + 1 < nt) ATTN_STEP(0, 1)
#undef ATTN_STEP
  if (t & 1) attn64_tile<DT, 1, true>(ka0, ka1, ka2, ka3, va0, va1, ra, rb, t, tokens, h);
  else       attn64_tile<DT, 0, true>(ka0, ka1, ka2, ka3, va0, va1, ra, rb, t, tokens, h);
#undef ATTN_STAGE_TILE

  // ---- normalise and store: lane owns query rows qrow_a / qrow_b, columns 32 dvt + 8 g + 4 h + {0..3} ----
  auto store = [&](Rows64& r, int qrow) {
    const unsigned lb = __float_as_uint(r.l_run);
    const auto sw = __builtin_amdgcn_permlane32_swap(lb, lb, false, false);
    const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
    if (qrow < tokens) {
      unsigned short* orow = out + ((int64_t)b * tokens + qrow) * dmodel + hd * 64 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack2_h16<DT>(r.o0[4 * g + 0] * inv, r.o0[4 * g + 1] * inv);
        pk.y = pack2_h16<DT>(r.o0[4 * g + 2] * inv, r.o0[4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + 8 * g) = pk;
        pk.x = pack2_h16<DT>(r.o1[4 * g + 0] * inv, r.o1[4 * g + 1] * inv);
        pk.y = pack2_h16<DT>(r.o1[4 * g + 2] * inv, r.o1[4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + 32 + 8 * g) = pk;
      }
    }
  };
  store(ra, qrow_a);
  store(rb, qrow_b);
}

}  // namespace

// C++ linkage: called by vittf_attention (attention.hip) for q_prescaled = 1 when VITTF_ATTN_PIPE=3
int vittf_attention_rows64(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype, hipStream_t st) {
  const int q_tiles = (tokens + QT64 - 1) / QT64;
  const int64_t total64 = (int64_t)batch * heads * q_tiles;
  if (total64 > (1 << 30)) return VITTF_ERR_INVALID_ARG;
  const int total = (int)total64;
  if (dtype == VITTF_BF16)
    hipLaunchKernelGGL((attn64_kernel<VITTF_BF16>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv, (unsigned short*)out,
                       tokens, heads, q_tiles, total);
  else if (dtype == VITTF_FP16)
    hipLaunchKernelGGL((attn64_kernel<VITTF_FP16>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv, (unsigned short*)out,
                       tokens, heads, q_tiles, total);
  else
    return VITTF_ERR_INVALID_ARG;
  return vittf_check_launch();
}
