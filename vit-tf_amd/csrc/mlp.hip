// The tail of a ViT-S block (D = 384) in one launch, and the MLP alone:
//     vittf_block_tail:  x' = x + a . Wp^T + bp ;  x += fc2( gelu_erf( fc1( LayerNorm(x'; g2, b2) ) + b1 ) ) + b2 ... i.e.
//                        x := x' + MLP(norm2(x')) ;  h_next = LayerNorm(x; g, b)            (a = the attention output)
//     vittf_mlp_fused:   x += fc2( gelu_erf( fc1(h) + b1 ) ) + b2 ;  h_next = LayerNorm(x; g, b)     (h = norm2's output)
//
// Replaces Attention.proj, both residual adds, norm2, Mlp.forward of the upstream DINO block and the next block's norm1
// (reached through model(...), infer.py:177).  As three GEMM launches these moved 15.4 KB per row of the residual stream
// (the fp32 rows twice in and out, the [rows][4D] hidden activation out and in, norm2's output out and in); here a row's
// 4.6 KB -- a in, x in, x out, h out -- are all that reaches HBM, and the only other stream is the weights, which every CU
// reads from L2 (2.65 MB per layer).
//
// Machine mapping (gfx950; DESIGN.md section 4 "Block tail" has the measurements behind every choice):
//   * workgroup = 4 waves = 128 rows, ONE wave per SIMD with the whole 512-entry register file, persistent; tiles are handed
//     out by a counter.  All products are computed transposed (weights = MFMA A operand from LDS, activations = B operand
//     from registers, a lane owns one row): the projection's 12 accumulator tiles (192 registers) become x', norm2 runs in
//     them and leaves fc1's 24 B operands (96 registers; k order = register order, the host packs W1 to match), the fc1
//     accumulator tile -- bias as its initial value, GELU applied, converted pairwise to 16 bit -- IS the B operand of fc2
//     (W2's hidden dim packed to match), fc2 accumulates on top of x', and a row's LayerNorm statistics never leave its wave.
//   * work is cut into UNITS of 24 MFMAs: 12 projection units (one output tile each over K = 384), then 48 fc1 units (32
//     hidden units over K = 384, one accumulator chain) interleaved with 48 fc2 units (12 output tiles x 2 k steps), fc1
//     two hidden units ahead.  The weights arrive as a STREAM of 24 KB images, one per unit, packed by the host in exactly
//     the order and LDS layout they are consumed in (weights.pack_block_tail_weights): every LDS-DMA piece is 1 KB of
//     contiguous memory.  Five ring slots, four units requested ahead, one barrier per unit, counted vmcnt.
//   * software pipeline, pinned per MFMA gap (one wave per SIMD: nothing else fills the gaps): the 24 A fragments of a
//     unit go through 8 fragment registers refilled in place right behind the MFMA that used them (the reads run 8 MFMAs
//     ahead, across unit boundaries); the GELU of the fc1 tile between the two products is cut into thirds of a value per
//     gap over both units of a pair (a one-wave SIMD hides about 24 issue cycles beside an MFMA, a whole value is 46); the
//     six DMA pieces a wave issues per unit sit behind every fourth MFMA; the residual columns of output tile ot - 1 are
//     folded in beside projection unit ot.
//   * everything that crosses HBM in rows goes THROUGH LDS (6 KB of staging per wave): a lane owns a row, so direct accesses
//     would be 32-byte runs, which a CU's memory path takes at 7 bytes per cycle; staged they are 128-byte runs.  Between two
//     tiles: drain, request the next tile's fragments and first residual chunks (ahead of the stores: vmcnt retires in order),
//     x = acc + b2 out, two-pass LayerNorm in the same registers (the other half of a row is in the lane 32 further on), h out.
#include "vittf_common.h"

#include <stdlib.h>


namespace {

constexpr int D = 384, HID = 4 * D;
constexpr int UNITS = HID / 32;            // 48 hidden units of 32
// streamed images per row tile: [Wp(0) .. Wp(11),] W1(0), W1(1), W1(2), W2(0), W1(3), W2(1), .., W1(47), W2(45), W2(46), W2(47)
constexpr int UB = 24576;                  // bytes of one image: 6 sub-images [32 rows][64 k] in the tile_off layout
constexpr int PUNITS = D / 32;             // block tail: 12 projection units in front (one image per output tile, K = 384)
constexpr int NSLOT = 5, AHEAD = NSLOT - 1;
constexpr int NF = 8;                      // fragment registers in flight
constexpr int PIECES = UB / 1024 / 4;      // LDS-DMA pieces per wave and unit
constexpr int WAIT0 = (AHEAD - 2) * PIECES;   // pieces of this wave that may be in flight when a unit starts (see mlp_unit)
constexpr int STG_OFF = NSLOT * UB;        // staging behind the ring: a quarter per wave, 32 rows of 128 bytes + 16
constexpr int STG_ROW = 144, STG_BYTES = UB / 4;
constexpr int CONST_OFF = STG_OFF + UB;    // fp32 constants behind that, in floats:
constexpr int C_B1 = 0, C_B2 = HID, C_G1 = HID + D, C_E1 = HID + 2 * D,       // b1 | b2 | gamma, beta of the LayerNorm behind the MLP
              C_BP = HID + 3 * D, C_G2 = HID + 4 * D, C_E2 = HID + 5 * D,      // block tail: proj bias | gamma, beta of norm2
              C_N = HID + 6 * D;
constexpr int NEXT_OFF = CONST_OFF + C_N * 4;     // one word: the tile the workgroup takes next
constexpr int LDS_BYTES = NEXT_OFF + 16;
constexpr int XA = 3;                      // residual tiles (32 columns) requested ahead (6: no faster, measured twice)
static_assert(32 * STG_ROW <= STG_BYTES, "staging");
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

struct TileIO {                // a row tile's activations: descriptors that end with the tile's last valid row (loads past it
  __amdgpu_buffer_rsrc_t x, h_next, h_out;    // return 0, stores are dropped)
};

struct Stream {                // where the weight stream stands (wave-uniform)
  i32x4_t rsrc;                // descriptor over one layer's NSEQ packed images
  unsigned dma_dst;            // LDS byte address of this wave's first piece in slot 0
  int nseq;                    // images per row tile (96, or 108 with the projection in front)
  int g;                       // stream position of the unit being computed (0 .. nseq - 1, wraps with the row tiles)
  int slot;                    // its ring slot
};

// fragment f (0 .. 23) of the image the bases point at: sub-image f >> 2, chunk pair f & 3
// (the bases are LDS byte addresses, not generic pointers: they rotate through the ring at run time, and hipcc turns a
// pointer it cannot prove to be LDS into flat loads, which count on vmcnt and drain the LDS-DMA queue)
typedef __attribute__((address_space(3))) const s16x8_t* lds_frag_ptr;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) const f32x4_t* lds_f4_ptr;
typedef __attribute__((address_space(3))) f32x4_t* lds_w4_ptr;
typedef __attribute__((address_space(3))) u32x2_t* lds_w2_ptr;
typedef __attribute__((address_space(3))) volatile unsigned* lds_u32_ptr;
__device__ __forceinline__ s16x8_t ld_frag(const unsigned (&base)[4], int f) {
  return *(lds_frag_ptr)(base[f & 3] + (f >> 2) * 4096);
}

// ---- timing-only variants for tools/mlp_variants.sh (never in libvittf.so: the Makefile does not define MLP_VARIANT) ----
#ifndef MLP_VARIANT
#define MLP_VARIANT 0
#endif
#if MLP_VARIANT & 16      // in-kernel stamps: s_memtime (shader cycles) at eight points of a tile, kept in SGPRs until its end
__device__ unsigned long long g_mlp_stamps[4 /*workgroups*/][4 /*tiles*/][4 /*waves*/][8][2];
#define MLP_STAMP(k)                                                                                          \
  do {                                                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_[k])::"memory");                       \
    if (k == 7 && blockIdx.x < 4 && tile_no < 4 && (threadIdx.x & 63) == 0) {                                 \
      _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_)                                                        \
        g_mlp_stamps[blockIdx.x][tile_no][threadIdx.x >> 6][q_][0] = stamp_[q_];                              \
    }                                                                                                         \
  } while (0)
#else
#define MLP_STAMP(k)
#endif
// cache policy of the ROW traffic (a / x in, x / h out: every byte is used once): MLP_VARIANT & 128 = non-temporal, so that
// the 4.8 GB a launch streams do not push the 2.65 MB of weights every CU keeps re-reading out of L2 (timing experiment)
constexpr int RT_AUX = (MLP_VARIANT & 128) ? 2 : 0;
constexpr bool V_NO_DMA = MLP_VARIANT & 1, V_M0_KEEP = MLP_VARIANT & 2 /* here: save + restore M0 */, V_NO_REFILL = MLP_VARIANT & 4, V_NO_GELU = MLP_VARIANT & 8,
               V_NO_XREQ = MLP_VARIANT & 32 /* projection units: no residual chunk requests */, V_NO_XFOLD = MLP_VARIANT & 64 /* ... no staging / adds */;

// an LDS-DMA piece that leaves M0 pointing at its destination (lds_dma16 saves and restores it: the restore waits until the
// load has left the wave's instruction buffer).  hipcc keeps nothing in M0 in this kernel: tests/test_host_cpu.py checks
// the disassembly for that.
__device__ __forceinline__ void lds_dma16_keep(i32x4_t rsrc, unsigned lds_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// the exact-erf GELU of vittf_common.h (gelu_poly: the same operations in the same order, so the same bits) cut into three
// pieces of at most 12 issue cycles, one per MFMA gap: beside an MFMA a one-wave SIMD hides about 24 cycles of other work
struct Gelu3 { float x, p; };           // (x: ONE copy of the value out of the accumulator registers, used by all three pieces)
__device__ __forceinline__ void gelu_a(Gelu3& s, float x) {
  s.x = x;
  asm volatile("" : "+v"(s.x));
  s.p = fmaf(-0.000524238159f, fabsf(s.x), 0.00741911121f);
  s.p = fmaf(s.p, fabsf(s.x), -0.0526018888f);
  s.p = fmaf(s.p, fabsf(s.x), -0.459225923f);
  asm volatile("" : "+v"(s.x), "+v"(s.p));         // pinned to this gap
}
__device__ __forceinline__ void gelu_b(Gelu3& s) {
  s.p = fmaf(s.p, fabsf(s.x), -1.15109742f);
  s.p = __builtin_amdgcn_exp2f(fmaf(s.p, fabsf(s.x), -1.0f));
  asm volatile("" : "+v"(s.p));
}
__device__ __forceinline__ float gelu_c(const Gelu3& s) {
  float v = fmaf(-fabsf(s.x), s.p, fmaxf(s.x, 0.f));
  asm volatile("" : "+v"(v));
  return v;
}

// Start of a unit.  Its image was requested AHEAD units ago; of what this wave has issued since, only the pieces of the two
// units behind the NEXT one may still be in flight: the next unit's image has landed too (its fragments are read from MFMA
// 16 on).  vmcnt counts every load and store of the wave, in order: WAITN = those 12 pieces + whatever else the wave has
// issued in its last two units.  The barrier also says that everybody is done with the slot of the unit before this one,
// which is refilled during this one.  WAITN < 0: no counted wait (the units right behind a drain, see the kernel).
template <int WAITN>
__device__ __forceinline__ void unit_wait() {
  static_assert(WAITN <= 63 && (WAITN < 0 || WAITN >= WAIT0), "vmcnt");
  if constexpr (WAITN < 0) asm volatile("s_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(V_NO_DMA ? 0 : WAITN) : "memory");
}

// what every unit does behind MFMA j besides its own work: turn the fragment bases to the next image at j == 16, refill the
// fragment register the MFMA has just used, and behind every fourth MFMA request one piece of the image AHEAD
template <bool LAST = false>
__device__ __forceinline__ void stream_gap(const Stream& st, unsigned (&base)[4], s16x8_t (&wf)[NF], int j, int g_next, int slot_free) {
  if (j == 24 - NF) {                          // from here on the refills read the next unit's image
    const int d_ = st.slot == NSLOT - 1 ? -(NSLOT - 1) * UB : UB;
#pragma unroll
    for (int i = 0; i < 4; ++i) base[i] += d_;
  }
  if (!V_NO_REFILL && !(LAST && j >= 24 - NF)) wf[j % NF] = ld_frag(base, (j + NF) % 24);   // (a tile's last unit: see the kernel)
  if ((j & 3) == 3 && !V_NO_DMA) {
    const unsigned dst = st.dma_dst + slot_free * UB + (j >> 2) * 4096;
    const int src = g_next * UB + (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 1024 + (j >> 2) * 4096;
    if (V_M0_KEEP) lds_dma16(st.rsrc, dst, (int)((threadIdx.x & 63) * 16), src);
    else lds_dma16_keep(st.rsrc, dst, (int)((threadIdx.x & 63) * 16), src);
  }
}

// chunk loads projection unit t issues (it folds chunk t - 1 in and asks for chunk t - 1 + XA), and the counted wait in front
// of unit t: the first three run behind a drain, the others count the loads of the two units before them
constexpr int proj_requests(int t) { return t >= 1 && t - 1 + XA < PUNITS ? 4 : 0; }
constexpr int proj_wait(int t) { return t < 3 ? -1 : WAIT0 + proj_requests(t - 2) + proj_requests(t - 1); }
static_assert(proj_requests(PUNITS - 2) == 0 && proj_requests(PUNITS - 1) == 0, "the MLP's first units count pieces only");

// One projection unit of the block tail = 24 MFMAs: xacc[OT] = Wp rows 32 OT .. + 31 . a^T (+ the bias tile as initial value).
// In its gaps the residual columns of the tile BEFORE it (whose product is complete) are folded in: the chunk requested
// three units ago goes through the staging rows (written eight lanes per row as it was loaded, read back a row per lane)
// and is added to xacc[OT - 1]; the freed registers then request the chunk three tiles further on (REQ).
template <int DT, int OT, int WAITN, bool REQ, bool LAST>
__device__ __forceinline__ void proj_unit(Stream& st, unsigned (&base)[4], s16x8_t (&wf)[NF], const s16x8_t (&af)[D / 16],
                                          f32x16_t (&xacc)[D / 32], const f32x16_t& bias_c, u32x4_t (&xi)[XA][4],
                                          const TileIO& io, int xo, unsigned stg_rd, unsigned stg_wr) {
  unit_wait<WAITN>();
  const int g_next = st.g + AHEAD < st.nseq ? st.g + AHEAD : st.g + AHEAD - st.nseq;
  const int slot_free = st.slot == 0 ? NSLOT - 1 : st.slot - 1;
  [[maybe_unused]] f32x4_t xv[4];
#pragma unroll
  for (int j = 0; j < 24; ++j) {
    xacc[OT] = mfma32<DT>(wf[j % NF], af[j], j == 0 ? bias_c : xacc[OT]);
    stream_gap<LAST>(st, base, wf, j, g_next, slot_free);
    if constexpr (OT > 0 && !V_NO_XFOLD) {
      if (j < 4) {
        *(lds_w4_ptr)(stg_rd + j * 8 * STG_ROW) = __builtin_bit_cast(f32x4_t, xi[(OT - 1) % XA][j]);
      } else if (j >= 6 && j < 10) {
        xv[j - 6] = *(lds_f4_ptr)(stg_wr + 32 * (j - 6));
      } else if (j >= 10 && j < 18) {
        const int r = 2 * (j - 10);
        xacc[OT - 1][r] += xv[r >> 2][r & 3];
        xacc[OT - 1][r + 1] += xv[r >> 2][(r & 3) + 1];
      }
      if constexpr (REQ && !V_NO_XREQ)
        if (j >= 18 && j < 22)
          xi[(OT - 1) % XA][j - 18] = __builtin_amdgcn_raw_buffer_load_b128(io.x, xo + (OT - 1 + XA) * 128, (j - 18) * 8 * (D * 4), RT_AUX);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  st.g = st.g + 1 == st.nseq ? 0 : st.g + 1;
  st.slot = st.slot + 1 == NSLOT ? 0 : st.slot + 1;
}

// One unit of the stream = 24 MFMAs.  FC1: gacc = W1u . h^T (+ the bias tile as initial value); FC2: xacc[ot] += W2u . gf_in.
// GH: which part of the activation of the fc1 tile gprev runs in this unit's gaps -- 0 none, 1 values 0 .. 7 -> gf_out[0],
// 2 values 8 .. 15 -> gf_out[1] (one value per three gaps, a third of it in each), 3 all sixteen (first / last units of a tile).
// WAITN < 0: no counted wait (see the kernel); ZI: a tile's first fc2 unit, its accumulators start from zero; LAST: a tile's
// last unit reads no fragments ahead.
template <int DT, bool FC1, int GH, int WAITN = WAIT0, bool ZI = false, bool LAST = false>
__device__ __forceinline__ void mlp_unit(Stream& st, unsigned (&base)[4], s16x8_t (&wf)[NF], s16x8_t (&hf)[D / 16],
                                         f32x16_t (&xacc)[D / 32], f32x16_t& gacc, const f32x16_t& bias_c,
                                         const f32x16_t& gprev, s16x8_t (&gf_out)[2], const s16x8_t (&gf_in)[2]) {
  unit_wait<WAITN>();
  const int g_next = st.g + AHEAD < st.nseq ? st.g + AHEAD : st.g + AHEAD - st.nseq;       // (the stream wraps: the next row tile)
  const int slot_free = st.slot == 0 ? NSLOT - 1 : st.slot - 1;
  u32x4_t pk0 = {}, pk1 = {};
  float vprev = 0.f;
  Gelu3 gs = {};
#pragma unroll
  for (int j = 0; j < 24; ++j) {
    if constexpr (FC1) gacc = mfma32<DT>(wf[j % NF], hf[j], j == 0 ? bias_c : gacc);
    else xacc[j >> 1] = mfma32<DT>(wf[j % NF], gf_in[j & 1], ZI && !(j & 1) ? f32x16_t{} : xacc[j >> 1]);
    stream_gap<LAST>(st, base, wf, j, g_next, slot_free);
    if constexpr (GH == 1 || GH == 2) {
      const int r = (GH == 2 ? 8 : 0) + j / 3;
      if (V_NO_GELU) {
        if (j % 3 == 2) {
          if (r & 1) { const unsigned w = pack2_h16<DT>(vprev, gprev[r]); if (GH == 1) pk0[(r & 7) >> 1] = w; else pk1[(r & 7) >> 1] = w; }
          else vprev = gprev[r];
        }
      } else if (j % 3 == 0) {
        gelu_a(gs, gprev[r]);
      } else if (j % 3 == 1) {
        gelu_b(gs);
      } else {
        const float v = gelu_c(gs);
        if (r & 1) {
          const unsigned w = pack2_h16<DT>(vprev, v);
          if (GH == 1) pk0[(r & 7) >> 1] = w; else pk1[(r & 7) >> 1] = w;
        } else {
          vprev = v;
        }
      }
    } else if constexpr (GH == 3) {
      if (j % 3 != 2) {                        // 16 whole values over 23 gaps
        const int r = 2 * (j / 3) + j % 3;
        float v = V_NO_GELU ? gprev[r] : gelu_poly(gprev[r]);
        asm volatile("" : "+v"(v));
        if (r & 1) {
          const unsigned w = pack2_h16<DT>(vprev, v);
          if (r < 8) pk0[r >> 1] = w; else pk1[(r - 8) >> 1] = w;
        } else {
          vprev = v;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (GH == 1 || GH == 3) gf_out[0] = __builtin_bit_cast(s16x8_t, pk0);
  if constexpr (GH == 2 || GH == 3) gf_out[1] = __builtin_bit_cast(s16x8_t, pk1);
  st.g = st.g + 1 == st.nseq ? 0 : st.g + 1;
  st.slot = st.slot + 1 == NSLOT ? 0 : st.slot + 1;
}

// TAIL: the block tail -- abuf holds the attention output, the projection (+ bias + residual) and norm2 run in front of the
// MLP and the fp32 residual rows are read once and written once.  Otherwise abuf holds norm2's output (the MLP alone).
template <int DT, bool TAIL>
__global__ __launch_bounds__(256, 1) void mlp_kernel(const unsigned short* __restrict__ abuf, const unsigned short* __restrict__ wpk,
                                                     const float* __restrict__ bp, const float* __restrict__ g2, const float* __restrict__ e2,
                                                     const float* __restrict__ b1, const float* __restrict__ b2,
                                                     float* __restrict__ x, int64_t rows, const float* __restrict__ ln_g,
                                                     const float* __restrict__ ln_b, float ln_eps,
                                                     unsigned short* __restrict__ hout, int ntiles,
                                                     unsigned* __restrict__ tile_ctr) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  float* const cst = reinterpret_cast<float*>(smem + CONST_OFF);
  for (int i = tid; i < C_N; i += 256) {
    float v;
    if (i < C_B2) v = b1[i];
    else if (i < C_G1) v = b2[i - C_B2];
    else if (i < C_E1) v = ln_g ? ln_g[i - C_G1] : 1.f;
    else if (i < C_BP) v = ln_b ? ln_b[i - C_E1] : 0.f;
    else if (i < C_G2) v = TAIL ? bp[i - C_BP] : 0.f;
    else if (i < C_E2) v = TAIL ? g2[i - C_G2] : 1.f;
    else v = TAIL ? e2[i - C_E2] : 0.f;
    cst[i] = v;
  }

  // this lane's view of the constants (half h reads 4 floats further on); opaque, so that every read below is this one
  // register + an immediate offset (hipcc otherwise keeps a separate address register per constant position and spills them)
  unsigned cl = (unsigned)(size_t)LDS_PTR(smem) + CONST_OFF + 16 * h;
  asm volatile("" : "+v"(cl));
  auto cst4 = [&](int i) { const f32x4_t v = *(lds_f4_ptr)(cl + 4 * i); return make_float4(v[0], v[1], v[2], v[3]); };      // floats i .. i + 3 (+ 4 h) of the constants
  // Tiles are handed out by a counter (the next one is asked for at the top of a tile and picked up in its epilogue): the
  // launch then ends when the tiles do, not when the workgroup with one tile more than the others does -- 8194 tiles on
  // 256 CUs are 32.01 rounds, not 33.  (Start offsets between the workgroups on top of it, so that they do not all reach
  // their epilogues together: measured, no effect on the launch time at 0, 1/16, 1/8 and 3/16 of a tile per phase.)
  const unsigned nxt = (unsigned)(size_t)LDS_PTR(smem) + NEXT_OFF;      // (an LDS address, not a generic pointer: no flat loads)
  if (tid == 0) *(lds_u32_ptr)nxt = atomicAdd(tile_ctr, 1u);
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane((int)*(lds_u32_ptr)nxt);
  if (tile >= ntiles) return;                  // (nothing requested yet)
  Stream st;
  st.nseq = TAIL ? PUNITS + 2 * UNITS : 2 * UNITS;
  st.rsrc = lds_dma_rsrc(wpk, (unsigned)(st.nseq * UB));
  st.dma_dst = (unsigned)(size_t)LDS_PTR(smem) + wave * 1024;
  st.g = 0;
  st.slot = 0;
  // the first AHEAD images (a wave's piece i of an image: bytes [4096 i + 1024 wave, + 1024))
#pragma unroll
  for (int u = 0; u < AHEAD; ++u)
#pragma unroll
    for (int i = 0; i < PIECES; ++i)
      lds_dma16(st.rsrc, st.dma_dst + u * UB + i * 4096, lane * 16, u * UB + wave * 1024 + i * 4096);
  const int aoff0 = tile_off(l31, h);
  unsigned base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) base[i] = (unsigned)(size_t)LDS_PTR(smem) + (aoff0 ^ (32 * i));
  // a tile's slice of a [rows][width bytes] array as a buffer descriptor: rows past the end read as zero / are not written
  auto tile_rsrc = [&](const void* p, int64_t tile, int row_bytes) {
    const int64_t first = tile * 128, left = rows - first;
    const int nrows = left <= 0 || !p ? 0 : left < 128 ? (int)left : 128;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p)) + (nrows ? first : 0) * row_bytes, 0,
                                             nrows * row_bytes, 0x00020000);
  };
  TileIO io;
  // ---- this lane's activation fragments (B operand) of the first tile: A[row][16 s + 8 h .. + 7], s = 0 .. 23 ----
  s16x8_t hf[D / 16];
  {
    const auto rs = tile_rsrc(abuf, tile, D * 2);
#pragma unroll
    for (int s = 0; s < D / 16; ++s) hf[s] = __builtin_bit_cast(s16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (wave * 32 + l31) * (D * 2) + 16 * h, 32 * s, RT_AUX));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first AHEAD images have landed (only here: the unit waits count
  __syncthreads();                                       // on a steady stream) ... everybody's pieces; the constants are written
  s16x8_t wf[NF];
  // block tail: the first residual chunks of a tile are requested before the stores of the tile in front of it (vmcnt retires
  // in order: behind them they would arrive when the last store has, a fifth of a tile later)
  [[maybe_unused]] u32x4_t xin[XA][4];
  if constexpr (TAIL) {
    const auto rx = tile_rsrc(x, tile, D * 4);
    const int xo0 = (wave * 32 + (lane >> 3)) * (D * 4) + (lane & 7) * 16;
#pragma unroll
    for (int c = 0; c < XA; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) xin[c][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo0 + c * 128, i * 8 * (D * 4), RT_AUX);
  }

  [[maybe_unused]] int tile_no = -1;
  [[maybe_unused]] unsigned long long stamp_[8];
  while (true) {
    ++tile_no;
    MLP_STAMP(0);
    unsigned next_v = 0;       // the tile after this one: asked for now, picked up in the epilogue
    if (tid == 0) next_v = atomicAdd(tile_ctr, 1u);
    // the first fragments of the tile's first image (landed: the drain in front of the previous epilogue / above).  Every
    // other unit reads its first eight fragments behind the last MFMAs of the unit before it; carried over the epilogue
    // they would be spilled
#pragma unroll
    for (int f = 0; f < NF; ++f) wf[f] = ld_frag(base, f);
    io.x = tile_rsrc(x, tile, D * 4);
    io.h_out = tile_rsrc(hout, tile, D * 2);
    int ln = lane;             // (opaque: the addresses below are recomputed per tile, a few VALU, not kept in -- spilled --
    asm volatile("" : "+v"(ln));   //  registers across the units; the constants added to them stay immediate offsets)
    const unsigned stg = (unsigned)(size_t)LDS_PTR(smem) + STG_OFF + wave * STG_BYTES;
    const unsigned stg_wr = stg + (ln & 31) * STG_ROW + 16 * (ln >> 5);     // row per lane: + 32 g
    const unsigned stg_rd = stg + (ln >> 3) * STG_ROW + (ln & 7) * 16;      // eight lanes per row: + 8 i rows
    const int xo = (wave * 32 + (ln >> 3)) * (D * 4) + (ln & 7) * 16;       // x: row 8 i + lane / 8, columns 32 ot + 4 (lane % 8) ..
    f32x16_t xacc[D / 32];     // output tile ot (32 columns) x this lane's row: the projection, then x', then fc2 on top of it
    f32x16_t bias_c;
    // bias tile of a unit: register r of lane half h = constant at + (r & 3) + 8 (r >> 2) + 4 h
    auto load_bias = [&](int at) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bv = cst4(at + 8 * q);
        bias_c[4 * q + 0] = bv.x; bias_c[4 * q + 1] = bv.y; bias_c[4 * q + 2] = bv.z; bias_c[4 * q + 3] = bv.w;
      }
    };
    if constexpr (TAIL) {
      // ---- x' = x + proj(a) + bp, one output tile per unit; the residual chunks come in beside the units (three in flight).
      //      The first three units wait for nothing but their barriers (their images landed before the drain); from then on
      //      the counted wait also counts the four chunk loads per unit of the two units before. ----
#define PROJ_UNIT(OT) load_bias(C_BP + 32 * (OT)); \
      proj_unit<DT, OT, proj_wait(OT), proj_requests(OT) != 0, (OT) == PUNITS - 1>(st, base, wf, hf, xacc, bias_c, xin, io, xo, stg_rd, stg_wr)
      PROJ_UNIT(0); PROJ_UNIT(1); PROJ_UNIT(2); PROJ_UNIT(3); PROJ_UNIT(4); PROJ_UNIT(5);
      PROJ_UNIT(6); PROJ_UNIT(7); PROJ_UNIT(8); PROJ_UNIT(9); PROJ_UNIT(10); PROJ_UNIT(11);
#undef PROJ_UNIT
      MLP_STAMP(1);
      {                       // the last tile's chunk
#pragma unroll
        for (int i = 0; i < 4; ++i) *(lds_w4_ptr)(stg_rd + i * 8 * STG_ROW) = __builtin_bit_cast(f32x4_t, xin[(PUNITS - 1) % XA][i]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t xv = *(lds_f4_ptr)(stg_wr + 32 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) xacc[PUNITS - 1][4 * g + e] += xv[e];
        }
      }
      // ---- norm2 of x' in registers -> the 24 B operands of fc1 (k step s, element e of lane half h = column
      //      32 (s >> 1) + 16 (s & 1) + 8 (e >> 2) + 4 h + (e & 3): the host packs W1's input dim in that order) ----
      float s = 0.f;
#pragma unroll
      for (int ot = 0; ot < D / 32; ++ot) {
#pragma unroll
        for (int g = 0; g < 4; ++g) s += (xacc[ot][4 * g + 0] + xacc[ot][4 * g + 1]) + (xacc[ot][4 * g + 2] + xacc[ot][4 * g + 3]);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        s = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      const float mean = s / (float)D;
      float q = 0.f;
#pragma unroll
      for (int ot = 0; ot < D / 32; ++ot) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float dv = xacc[ot][r] - mean; q += dv * dv; }
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(q), __float_as_uint(q), false, false);
        q = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      const float rstd = 1.0f / sqrtf(q / (float)D + ln_eps);
      float mean_p = mean;         // (opaque: recompute x - mean below; reusing the variance pass's 192 differences spills)
      asm volatile("" : "+v"(mean_p));
      float4 gq[2][4], bq[2][4];       // gamma / beta of an output tile's columns, read one tile ahead of their use
#pragma unroll
      for (int g = 0; g < 4; ++g) { gq[0][g] = cst4(C_G2 + 8 * g); bq[0][g] = cst4(C_E2 + 8 * g); }
#pragma unroll
      for (int ot = 0; ot < D / 32; ++ot) {
        if (ot + 1 < D / 32) {
#pragma unroll
          for (int g = 0; g < 4; ++g) { gq[(ot + 1) & 1][g] = cst4(C_G2 + 32 * (ot + 1) + 8 * g); bq[(ot + 1) & 1][g] = cst4(C_E2 + 32 * (ot + 1) + 8 * g); }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          u32x4_t pk;
#pragma unroll
          for (int g2_ = 0; g2_ < 2; ++g2_) {
            const int g = 2 * k2 + g2_;
            const float4 gg = gq[ot & 1][g], bb = bq[ot & 1][g];
            pk[2 * g2_ + 0] = pack2_h16<DT>((xacc[ot][4 * g + 0] - mean_p) * rstd * gg.x + bb.x, (xacc[ot][4 * g + 1] - mean_p) * rstd * gg.y + bb.y);
            pk[2 * g2_ + 1] = pack2_h16<DT>((xacc[ot][4 * g + 2] - mean_p) * rstd * gg.z + bb.z, (xacc[ot][4 * g + 3] - mean_p) * rstd * gg.w + bb.w);
          }
          hf[2 * ot + k2] = __builtin_bit_cast(s16x8_t, pk);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // The MLP's unit sequence (= the order of the packed images): fc1 runs two hidden units ahead of fc2, and the activation
    // of the unit in between is spread over both units of a pair.  Without the projection in front the first three units
    // wait for nothing but their barriers: their images (and the one the third reads ahead) landed before the previous tile's
    // epilogue / the loop, which gives that epilogue's stores three units to drain before a counted wait stands behind them.
    f32x16_t ga, gb = {};              // two fc1 tiles: one being accumulated, one being activated
    s16x8_t gf0[2] = {}, gf1[2] = {};  // two activated tiles (fc2's B operands): one being packed, one being consumed
    if constexpr (TAIL) {     // (the last projection unit read no fragments ahead: over norm2 they would be spilled)
#pragma unroll
      for (int f = 0; f < NF; ++f) wf[f] = ld_frag(base, f);
    }
    constexpr int WF = TAIL ? WAIT0 : -1;
    load_bias(C_B1);
    mlp_unit<DT, true, 0, WF>(st, base, wf, hf, xacc, ga, bias_c, gb, gf0, gf0);                     // fc1(0)
    if (!TAIL) MLP_STAMP(1);
    load_bias(C_B1 + 32);
    mlp_unit<DT, true, 3, WF>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf0);                     // fc1(1) | gelu(0) -> gf0
    load_bias(C_B1 + 64);
    mlp_unit<DT, true, 1, WF>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf1);                     // fc1(2) | gelu(1), values 0 .. 7
    mlp_unit<DT, false, 2, WAIT0, !TAIL>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf0);          // fc2(0) | gelu(1), values 8 .. 15
    load_bias(C_B1 + 96);
    mlp_unit<DT, true, 1>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf0);                         // fc1(3) | gelu(2)
    mlp_unit<DT, false, 2>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf1);                        // fc2(1) | gelu(2)
    MLP_STAMP(2);
    for (int u = 2; u < UNITS - 2; u += 2) {
      load_bias(C_B1 + 32 * (u + 2));
      mlp_unit<DT, true, 1>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf1);                       // fc1(u + 2) | gelu(u + 1)
      mlp_unit<DT, false, 2>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf0);                      // fc2(u)     | gelu(u + 1)
      load_bias(C_B1 + 32 * (u + 3));
      mlp_unit<DT, true, 1>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf0);                       // fc1(u + 3) | gelu(u + 2)
      mlp_unit<DT, false, 2>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf1);                      // fc2(u + 1) | gelu(u + 2)
    }
    MLP_STAMP(3);
    mlp_unit<DT, false, 3>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf0);                        // fc2(46) | gelu(47) -> gf1
    mlp_unit<DT, false, 0, WAIT0, false, true>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf1);    // fc2(47)

    // ---- epilogue: x[row][col .. col + 3] = [x +] acc + b2, lane owns row m, columns 32 ot + 8 g + 4 h + {0 .. 3}; then the
    //      LayerNorm of the new row from the same registers (passes of one output tile at a time: the compiler otherwise
    //      keeps all 192 values in flight twice and spills the next tile's fragments) ----
    MLP_STAMP(4);
    // Drain: everything this wave has requested has landed -- the four images ahead, so the first three units of the next
    // tile wait for nothing but their barriers and the stores below have until the fourth to complete (vmcnt counts in
    // order) -- and, behind the barrier, everybody's.
    // A lane owns a row: written as they stand, the accumulators would go out in 32-byte runs (two lanes per row), which a
    // CU's memory path takes at about 7 bytes per cycle.  So every output tile takes a turn through this wave's staging
    // rows: written a row per lane, read back eight lanes per row, stored in 128-byte runs; without the projection in front
    // the residual columns come in the other way first (three tiles requested ahead).  Row stride 144 bytes (36 banks).
    if (tid == 0) *(lds_u32_ptr)nxt = next_v;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int next = __builtin_amdgcn_readfirstlane((int)*(lds_u32_ptr)nxt);
    io.h_next = tile_rsrc(abuf, next, D * 2);      // (no next tile: an empty descriptor)
    int le = lane;             // (opaque again: everything the epilogue addresses is recomputed here, not carried over the units)
    asm volatile("" : "+v"(le));
    const unsigned stg_e = (unsigned)(size_t)LDS_PTR(smem) + STG_OFF + wave * STG_BYTES;
    const unsigned stg_wr_e = stg_e + (le & 31) * STG_ROW + 16 * (le >> 5);
    const unsigned stg_wh = stg_e + (le & 31) * STG_ROW + 8 * (le >> 5);      // (16-bit rows: + 64 o2 + 16 g)
    const unsigned stg_rd_e = stg_e + (le >> 3) * STG_ROW + (le & 7) * 16;
    const int xo_e = (wave * 32 + (le >> 3)) * (D * 4) + (le & 7) * 16;
    const int ho = (wave * 32 + (le >> 3)) * (D * 2) + (le & 7) * 16;       // h: rows 8 i + lane / 8, columns 64 op + 8 (lane % 8) ..
    // the next tile's activation fragments: in flight during the whole epilogue
    const int hfo = (wave * 32 + (le & 31)) * (D * 2) + 16 * (le >> 5);
    auto request_next_fragments = [&] {
#pragma unroll
      for (int s = 0; s < D / 16; ++s) hf[s] = __builtin_bit_cast(s16x8_t, __builtin_amdgcn_raw_buffer_load_b128(io.h_next, hfo, 32 * s, RT_AUX));
    };
    if constexpr (TAIL) {       // (the MLP alone: behind the residual pass, whose chunk registers they would spill)
      request_next_fragments();
      const auto rx = tile_rsrc(x, next, D * 4);
#pragma unroll
      for (int c = 0; c < XA; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) xin[c][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo_e + c * 128, i * 8 * (D * 4), RT_AUX);
    }
    constexpr int XE = 3;      // (the MLP alone: residual tiles requested ahead in the epilogue)
    [[maybe_unused]] u32x4_t xi[XE][4];
    auto x_request = [&](int ot) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xi[ot % XE][i] = __builtin_amdgcn_raw_buffer_load_b128(io.x, xo_e + ot * 128, i * 8 * (D * 4), RT_AUX);
    };
    if constexpr (!TAIL) {
#pragma unroll
      for (int ot = 0; ot < XE; ++ot) x_request(ot);
    }
    float s = 0.f;
    if constexpr (TAIL) {
      // The constants of an output tile are requested one tile ahead and the read-back of the staging rows as ONE group: left to
      // itself hipcc issues every LDS read with its own lgkmcnt(0) wait in front of the one instruction that uses it (8 exposed
      // round trips per output tile: 1.3 k cycles of the single wave a SIMD has).  The fences only delimit the groups.
      float4 bvq[2][4];
  #pragma unroll
      for (int g = 0; g < 4; ++g) bvq[0][g] = cst4(C_B2 + 8 * g);
  #pragma unroll
      for (int ot = 0; ot < D / 32; ++ot) {
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = bvq[ot & 1][g];
          f32x4_t v;
          v[0] = xacc[ot][4 * g + 0] + bv.x; v[1] = xacc[ot][4 * g + 1] + bv.y;
          v[2] = xacc[ot][4 * g + 2] + bv.z; v[3] = xacc[ot][4 * g + 3] + bv.w;
  #pragma unroll
          for (int e = 0; e < 4; ++e) xacc[ot][4 * g + e] = v[e];
          *(lds_w4_ptr)(stg_wr_e + 32 * g) = v;
          s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t rb[4];
  #pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *(lds_f4_ptr)(stg_rd_e + i * 8 * STG_ROW);
        if (ot + 1 < D / 32) {
  #pragma unroll
          for (int g = 0; g < 4; ++g) bvq[(ot + 1) & 1][g] = cst4(C_B2 + 32 * (ot + 1) + 8 * g);
        }
        __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, rb[i]), io.x, xo_e + ot * 128, i * 8 * (D * 4), RT_AUX);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {      // (the MLP alone: its residual tiles already take the registers the groups would need -- 80 spills)
  #pragma unroll
      for (int ot = 0; ot < D / 32; ++ot) {
        if constexpr (!TAIL) {
  #pragma unroll
          for (int i = 0; i < 4; ++i) *(lds_w4_ptr)(stg_rd_e + i * 8 * STG_ROW) = __builtin_bit_cast(f32x4_t, xi[ot % XE][i]);
          if (ot + XE < D / 32) x_request(ot + XE);
        }
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = cst4(C_B2 + 32 * ot + 8 * g);
          f32x4_t v;
          v[0] = xacc[ot][4 * g + 0] + bv.x; v[1] = xacc[ot][4 * g + 1] + bv.y;
          v[2] = xacc[ot][4 * g + 2] + bv.z; v[3] = xacc[ot][4 * g + 3] + bv.w;
          if constexpr (!TAIL) {
            const f32x4_t xv = *(lds_f4_ptr)(stg_wr_e + 32 * g);
  #pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = xv[e] + v[e];
          }
  #pragma unroll
          for (int e = 0; e < 4; ++e) xacc[ot][4 * g + e] = v[e];
          *(lds_w4_ptr)(stg_wr_e + 32 * g) = v;
          s += (v[0] + v[1]) + (v[2] + v[3]);
        }
  #pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4_t v = *(lds_f4_ptr)(stg_rd_e + i * 8 * STG_ROW);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), io.x, xo_e + ot * 128, i * 8 * (D * 4), RT_AUX);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    MLP_STAMP(5);
    if constexpr (!TAIL) request_next_fragments();
    if (hout) {
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        s = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      const float mean = s / (float)D;
      float q = 0.f;
#pragma unroll
      for (int ot = 0; ot < D / 32; ++ot) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float dv = xacc[ot][r] - mean; q += dv * dv; }
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(q), __float_as_uint(q), false, false);
        q = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      MLP_STAMP(6);
      const float rstd = 1.0f / sqrtf(q / (float)D + ln_eps);
      float mean_p = mean;
      asm volatile("" : "+v"(mean_p));
      if constexpr (TAIL) {
        // gamma / beta of an output tile one tile ahead, the read-back as one group (as above)
        float4 gq1[2][4], bq1[2][4];
  #pragma unroll
        for (int g = 0; g < 4; ++g) { gq1[0][g] = cst4(C_G1 + 8 * g); bq1[0][g] = cst4(C_E1 + 8 * g); }
  #pragma unroll
        for (int ot = 0; ot < D / 32; ++ot) {            // two output tiles = 64 columns = 128 bytes of a 16-bit row
          const int op = ot >> 1, o2 = ot & 1;
  #pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 gg = gq1[ot & 1][g];
            const float4 bb = bq1[ot & 1][g];
            u32x2_t pk;
            pk[0] = pack2_h16<DT>((xacc[ot][4 * g + 0] - mean_p) * rstd * gg.x + bb.x, (xacc[ot][4 * g + 1] - mean_p) * rstd * gg.y + bb.y);
            pk[1] = pack2_h16<DT>((xacc[ot][4 * g + 2] - mean_p) * rstd * gg.z + bb.z, (xacc[ot][4 * g + 3] - mean_p) * rstd * gg.w + bb.w);
            *(lds_w2_ptr)(stg_wh + 64 * o2 + 16 * g) = pk;
          }
          __builtin_amdgcn_sched_barrier(0);
          [[maybe_unused]] f32x4_t rb[4];
          if (o2) {
  #pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = *(lds_f4_ptr)(stg_rd_e + i * 8 * STG_ROW);
          }
          if (ot + 1 < D / 32) {
  #pragma unroll
            for (int g = 0; g < 4; ++g) { gq1[(ot + 1) & 1][g] = cst4(C_G1 + 32 * (ot + 1) + 8 * g); bq1[(ot + 1) & 1][g] = cst4(C_E1 + 32 * (ot + 1) + 8 * g); }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (o2) {
  #pragma unroll
            for (int i = 0; i < 4; ++i)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, rb[i]), io.h_out, ho + op * 128, i * 8 * (D * 2), RT_AUX);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
  #pragma unroll
        for (int op = 0; op < D / 64; ++op) {            // two output tiles = 64 columns = 128 bytes of a 16-bit row
  #pragma unroll
          for (int o2 = 0; o2 < 2; ++o2) {
            const int ot = 2 * op + o2;
  #pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int col = 32 * ot + 8 * g;
              const float4 gg = cst4(C_G1 + col);
              const float4 bb = cst4(C_E1 + col);
              u32x2_t pk;
              pk[0] = pack2_h16<DT>((xacc[ot][4 * g + 0] - mean_p) * rstd * gg.x + bb.x, (xacc[ot][4 * g + 1] - mean_p) * rstd * gg.y + bb.y);
              pk[1] = pack2_h16<DT>((xacc[ot][4 * g + 2] - mean_p) * rstd * gg.z + bb.z, (xacc[ot][4 * g + 3] - mean_p) * rstd * gg.w + bb.w);
              *(lds_w2_ptr)(stg_wh + 64 * o2 + 16 * g) = pk;
            }
          }
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4_t v = *(lds_f4_ptr)(stg_rd_e + i * 8 * STG_ROW);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), io.h_out, ho + op * 128, i * 8 * (D * 2), RT_AUX);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    MLP_STAMP(7);
    if (next >= ntiles) break;
    tile = next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the images requested beyond the last unit: land before the LDS goes away
}

}  // namespace

#if MLP_VARIANT & 16
extern "C" int vittf_mlp_stamps(unsigned long long* out) {      // [4][4][4][8][2], host memory
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_stamps), sizeof(g_mlp_stamps)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef MLP_STANDALONE      // tools/mlp_variants.sh builds this file alone
void vittf_note_kernel(int, const char*) {}
#endif

static int mlp_launch(bool tail, const void* a, const void* w_packed, const float* bp, const float* g2, const float* e2,
                      const float* b1, const float* b2, float* x, int64_t rows, int32_t dtype, const float* ln_g,
                      const float* ln_b, float ln_eps, void* h_out, void* tile_counter, void* stream) {
  const int64_t tiles = (rows + 127) / 128;
  if (tiles > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  // 16-byte accesses everywhere (LDS-DMA pieces, buffer loads / stores of whole 128-byte runs)
  if ((((uintptr_t)a | (uintptr_t)w_packed | (uintptr_t)x | (uintptr_t)h_out) & 15) != 0) return VITTF_ERR_INVALID_ARG;
  if (!tile_counter || ((uintptr_t)tile_counter & 3) != 0) return VITTF_ERR_INVALID_ARG;
  // one persistent workgroup per CU of the device this call runs on (asked per call: no state is kept between calls)
  const int cus = vittf_current_cus();
  if (cus <= 0) return VITTF_ERR_NO_DEVICE;
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  hipStream_t st = (hipStream_t)stream;
  // the tile counter is the caller's memory (launches on one stream are serialised; two stream lanes bring two counters)
  unsigned* ctr = (unsigned*)tile_counter;
  if (hipMemsetAsync(ctr, 0, sizeof(unsigned), st) != hipSuccess) return VITTF_ERR_LAUNCH;
#define MLP_LAUNCH(DTV, TAILV)                                                                                       \
  hipLaunchKernelGGL((mlp_kernel<DTV, TAILV>), dim3(grid), dim3(256), 0, st, (const unsigned short*)a,               \
                     (const unsigned short*)w_packed, bp, g2, e2, b1, b2, x, rows, ln_g, ln_b, ln_eps,               \
                     (unsigned short*)h_out, (int)tiles, ctr)
  if (dtype == VITTF_BF16) { if (tail) MLP_LAUNCH(VITTF_BF16, true); else MLP_LAUNCH(VITTF_BF16, false); }
  else if (dtype == VITTF_FP16) { if (tail) MLP_LAUNCH(VITTF_FP16, true); else MLP_LAUNCH(VITTF_FP16, false); }
  else return VITTF_ERR_INVALID_ARG;
#undef MLP_LAUNCH
  vittf_note_kernel(VITTF_KERNEL_MLP, tail ? "mlp_kernel<block tail>" : "mlp_kernel");
  return vittf_check_launch();
}

extern "C" int vittf_mlp_fused(const void* h, const void* w_packed, const float* b1, const float* b2, float* x, int64_t rows,
                               int32_t d, int32_t dtype, const float* ln_g, const float* ln_b, float ln_eps, void* h_out,
                               void* tile_counter, void* stream) {
  if (!h || !w_packed || !b1 || !b2 || !x || rows <= 0) return VITTF_ERR_INVALID_ARG;
  if (d != D) return VITTF_ERR_INVALID_ARG;          // the register budget is sized for ViT-S
  if ((ln_g || ln_b || h_out) && !(ln_g && ln_b && h_out)) return VITTF_ERR_INVALID_ARG;
  return mlp_launch(false, h, w_packed, nullptr, nullptr, nullptr, b1, b2, x, rows, dtype, ln_g, ln_b, ln_eps, h_out, tile_counter,
                    stream);
}

extern "C" size_t vittf_block_tail_workspace_bytes(void) { return sizeof(unsigned); }

extern "C" int vittf_block_tail(const void* attn_out, const void* w_packed, const float* proj_b, const float* ln2_g,
                                const float* ln2_b, const float* b1, const float* b2, float* x, int64_t rows, int32_t d,
                                int32_t dtype, const float* ln_g, const float* ln_b, float ln_eps, void* h_out, void* tile_counter,
                                void* stream) {
  if (!attn_out || !w_packed || !proj_b || !ln2_g || !ln2_b || !b1 || !b2 || !x || rows <= 0) return VITTF_ERR_INVALID_ARG;
  if (d != D) return VITTF_ERR_INVALID_ARG;
  if ((ln_g || ln_b || h_out) && !(ln_g && ln_b && h_out)) return VITTF_ERR_INVALID_ARG;
  return mlp_launch(true, attn_out, w_packed, proj_b, ln2_g, ln2_b, b1, b2, x, rows, dtype, ln_g, ln_b, ln_eps, h_out, tile_counter,
                    stream);
}
