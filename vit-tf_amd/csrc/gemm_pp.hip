// Persistent 256 x 256 MFMA GEMM for the long-K linears of ViT-B (D = 768: attn.qkv, mlp.fc1 + GELU, mlp.fc2, K features):
//     out = epilogue(A[rows][K] . W[N][K]^T + bias[N]),   K % 64 == 0, K >= 768, N % 256 == 0
//
// Round 4 (BASELINE configs[3]).  gemm.hip's 128 x 128 x 64 tiles (one stage, four workgroups per CU) run these shapes at
// 740-870 TFLOP/s.  Earlier forms of this file are in the history with their numbers (LAB_NOTES.md, former DESIGN section 4, "Round 4: the
// ViT-B linears": a pipelined 256 x 128 kernel with two workgroups per CU; an 8-wave ping-pong kernel with a load segment
// and an MFMA segment per K step -- hence the file's name; the same made persistent); timing-only builds of each
// (tools/pp_variants.sh) said where its time went.  This form:
//   * workgroup = 8 waves on a 256 x 256 output tile, persistent (one per CU); wave (g, c) owns rows 128 g .. + 127 and
//     columns 64 c .. + 63 = 4 x 2 accumulator tiles (128 VGPRs), computed transposed like the other GEMMs (weights = MFMA A
//     operand, activations = B operand: a lane owns one activation row and 4 consecutive columns per register quad).
//   * a K step of 32 = a stage: [256][32] activation image + [256][32] weight image = 32 KB in the "double-row" layout of
//     gemm_rows.hip (conflict-free ds_read_b128); 4-deep ring = 128 KB.  The stages of a workgroup's tiles form ONE stream:
//     the ring never drains between tiles.
//   * stage p = ONE barrier, then one pinned instruction stream (sched_barrier after every MFMA): 16 MFMAs on the fragments of
//     stage p with, behind the first twelve, the 12 fragment reads of stage p + 1 (two fragment sets in registers; a
//     ds_read_b128 in an MFMA gap is nearly free) and, behind every fourth, one of this wave's 4 LDS-DMA pieces of stage
//     p + 4 -- into stage p's slot, which is free because its fragments already sit in registers --, then the counted wait
//     for its pieces of stage p + 2.  0.75 LDS reads and 0.25 DMA pieces per MFMA.
//   * hazards by construction: stage p + 1 is read during stage p; every wave has waited for its pieces of stage p + 1 at the
//     end of stage p - 1 (they were requested during stage p - 3: two stages of slack), in front of the barrier that opens
//     stage p; the pieces of stage p + 4 overwrite the slot of stage p, whose last read (during stage p - 1, lgkmcnt(0) in
//     front of the same barrier) precedes their issue.
//   * epilogue: the tile leaves in eight parts of 32 rows through a 32 KB fp32 staging area BEHIND the ring: four waves stage
//     their accumulators, all eight apply bias / GELU / q scale (or the fp32 residual read-modify-write) and issue 16-byte
//     buffer stores (non-temporal for 16-bit outputs) that stay in flight under the next tile's K loop -- whose first
//     stages are already in the ring.  The counted waits of a tile's first two stages know how many stores lie in between.
//   * a partial last row tile reads zeros for its missing rows (descriptor bounds) and drops their outputs: a row's bits do
//     not depend on where it sits in a launch.
#include "vittf_common.h"

#include <stdlib.h>

namespace {

constexpr int PBM = 256, PBN = 256, PBK = 32;
constexpr int PIMG = 256 * PBK * 2;              // one operand image: 16 KB
constexpr int PSTAGE = 2 * PIMG;                 // 32 KB
constexpr int PSTAGES = 4;
constexpr int PRING = PSTAGES * PSTAGE;          // 128 KB
constexpr int PSTG = 32768;                      // staging behind the ring: 32 rows x 256 fp32 columns
constexpr int PLDS = PRING + PSTG;               // 160 KB
static_assert(PLDS <= 160 * 1024, "LDS");

#define PP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- timing-only variants for tools/pp_variants.sh (never in libvittf.so: the Makefile does not define PP_VARIANT) ----
#ifndef PP_VARIANT
#define PP_VARIANT 0
#endif
constexpr bool PV_NO_DMA = PP_VARIANT & 1;       // no LDS-DMA inside the K loop (wrong results)
constexpr bool PV_NO_READS = PP_VARIANT & 4;     // no fragment reads inside the K loop (wrong results)

constexpr bool PV_NO_EPI = PP_VARIANT & 32;      // no epilogue (wrong results)
// cache policy bits of the epilogue's 16-bit buffer stores: non-temporal (bit 1) by default -- the 128 KB a tile writes do not
// push the operand panels the next tiles re-read out of L2: +3.5-4 % on qkv / fc1 (profiles/r04e_gemm_pp_store_policy.txt: 871 ->
// 902, 787 -> 818 TFLOP/s); variants 512 / 1024 / 2048 = plain / sc0 nt / sc0 sc1.  The fp32 residual stores stay plain (no
// difference measured, and the LayerNorm launch behind fc2 re-reads those rows).
constexpr int PV_ST_AUX = (PP_VARIANT & 512) ? 0 : (PP_VARIANT & 1024) ? 3 : (PP_VARIANT & 2048) ? 17 : 2;
constexpr bool PV_NO_PIN = PP_VARIANT & 128;     // no scheduling fences inside the MFMA segment (the compiler places the reads)

// byte offset of 16-byte k-chunk kc (0..3) of row r inside a [R][32] operand image (two rows per 128-byte tile_off row)
__device__ __forceinline__ int p_img_off(int r, int kc) { return tile_off(r >> 1, ((r & 1) << 2) | kc); }

typedef __attribute__((ext_vector_type(4))) unsigned pu32x4_t;

// The qkv projection with q and k leaving as fp8 (e4m3) rows with MX block scales -- one power-of-two (E8M0 byte) per row and
// 32-wide block -- in the layout attention_fp8.hip reads ([slice][head][token][64] bytes in the instruction's block order,
// [slice][head][token][2] scale bytes),
// v as 16-bit values in `out` (the qkv buffer's v third) with its per-(slice, head) absolute maximum collected on the way
// (internal epilogue id; entry point vittf_gemm_qkv_fp8 below).
constexpr int PP_EPI_QKV_FP8 = 100;
struct PpFp8Out {
  unsigned char* q8; unsigned char* k8; unsigned char* qs; unsigned char* ks; unsigned* amax;
  int np, heads, batch;
};

// power-of-two scale exponent for a block with absolute maximum amax: amax * 2^-e <= 448 (e4m3 maximum), e >= -20
// (the rule of attention_fp8.hip's scale_exp)
__device__ __forceinline__ int pp_scale_exp(float amax) {
  if (!(amax > 0.f)) return 0;
  int ex;
  (void)frexpf(amax * (1.0f / 448.0f), &ex);
  return ex < -20 ? -20 : ex;
}

struct Frags { s16x8_t a[4][2], w[2][2]; };      // [32-row block][k16 step]

template <int DT, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W,
                                                         const float* __restrict__ bias, void* __restrict__ out, int64_t rows,
                                                         int n, int k, int tokens, int n_tiles, int total_tiles,
                                                         unsigned out_bytes, PpFp8Out f8) {
  __shared__ __attribute__((aligned(16))) char smem[PLDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int grp = wave >> 2, wc = wave & 3;      // waves w and w + 4 share a SIMD: the two groups are SIMD partners
  const int G = gridDim.x;
  const int nk = k / PBK;                        // even, >= 24
  // buffer stores a thread leaves in flight per tile: 8 parts x 4 (fp32 read-modify-write) or x 2 (16 bytes of 16-bit values)
  // (the fp8 qkv epilogue: 16 = its v tiles; its q / k tiles leave 32: the smaller number only waits a little earlier)
  constexpr int SPT = EPI == VITTF_EPI_BIAS_RESIDUAL ? 32 : 16;

  // ---- LDS-DMA: chunk q = i * 512 + tid of an image <- (row, k chunk) by the inverse of p_img_off; 2 + 2 pieces per wave ----
  // (chunk q + 512 is 128 image rows further on with the same swizzle: one per-lane offset, the rest is scalar)
  int voff0;
  {
    int dr, c;
    tile_pos(tid, dr, c);
    voff0 = (2 * dr + (c >> 2)) * k * 2 + (c & 3) * 16;
  }
  const int piece_rows = 128 * k * 2;
  const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(smem);
  const unsigned dma_wave = ring_lds + wave * 1024;

  // ---- the DMA stream: (tile, K step) of the next stage to request, its descriptors, its ring slot ----
  int d_vb = blockIdx.x, d_k = 0, d_slot = 0;
  bool d_live = true;
  i32x4_t d_ra, d_rw;
  auto tile_origin = [&](int vb, int64_t& m0, int& n0) {
    const int tile = xcd_remap(vb, total_tiles);
    const int mt = tile / n_tiles;
    m0 = (int64_t)mt * PBM;
    n0 = (tile - mt * n_tiles) * PBN;
  };
  auto d_open = [&](int vb) {
    int64_t m0; int n0;
    tile_origin(vb, m0, n0);
    const int rows_here = (int)(rows - m0 < PBM ? rows - m0 : PBM);
    d_ra = lds_dma_rsrc(A + m0 * k, (unsigned)((int64_t)rows_here * k * 2));
    d_rw = lds_dma_rsrc(W + (int64_t)n0 * k, (unsigned)(PBN * k * 2));
  };
  d_open(d_vb);
  // one piece (0..3) of the next stage of the stream; piece 3 advances the stream.  Returns false when the stream is over.
  auto dma_piece = [&](int i) -> bool {
    if (!d_live) return false;
    const unsigned dst = dma_wave + d_slot * PSTAGE;
    const int so = d_k * (PBK * 2);
    if (i == 0) lds_dma16(d_ra, dst, voff0, so);
    else if (i == 1) lds_dma16(d_ra, dst + 8192, voff0, so + piece_rows);
    else if (i == 2) lds_dma16(d_rw, dst + PIMG, voff0, so);
    else {
      lds_dma16(d_rw, dst + PIMG + 8192, voff0, so + piece_rows);
      d_slot = (d_slot + 1) & 3;
      if (++d_k == nk) {
        d_k = 0;
        d_vb += G;
        if (d_vb < total_tiles) d_open(d_vb); else d_live = false;
      }
    }
    return true;
  };
  auto dma_next = [&]() { dma_piece(0); dma_piece(1); dma_piece(2); dma_piece(3); };

  // ---- fragment addresses inside a stage: block mi of 32 rows = + 2048 mi with byte bit 7 flipped for odd mi, k16 step 1 =
  //      byte bit 5 flipped (the swizzle of tile_off): two base registers, everything else is recomputed per stage (as loop
  //      invariants the twelve addresses are spilled) ----
  const int a_base0 = p_img_off(grp * 128 + l31, h), w_base0 = PIMG + p_img_off(wc * 64 + l31, h);
  int r_slot = 0;                                // ring slot of the stage whose fragments are read next
  typedef __attribute__((address_space(3))) const s16x8_t* lds_frag_ptr;
  const unsigned lbase = (unsigned)(size_t)LDS_PTR(smem);
  // fragment j of 12 of the stage in slot r_slot: j < 4: weights (ni = j & 1, k16 step j >> 1), else activations
  auto read_frag = [&](Frags& f, int j, int ab, int wb) {
    if (j < 4) {
      const int ni = j & 1, st = j >> 1;
      f.w[ni][st] = *(lds_frag_ptr)(lbase + ((wb ^ (128 * (ni & 1) + 32 * st)) + 2048 * ni));
    } else {
      const int mi = (j - 4) & 3, st = (j - 4) >> 2;
      f.a[mi][st] = *(lds_frag_ptr)(lbase + ((ab ^ (128 * (mi & 1) + 32 * st)) + 2048 * mi));
    }
  };
  auto read_frags = [&](Frags& f) {
    int ab = a_base0 + r_slot * PSTAGE, wb = w_base0 + r_slot * PSTAGE;
    asm volatile("" : "+v"(ab), "+v"(wb));
#pragma unroll
    for (int j = 0; j < 12; ++j) read_frag(f, j, ab, wb);
    r_slot = (r_slot + 1) & 3;
  };

  f32x16_t acc[2][4];                            // [ni][mi]
  Frags fx, fy;

  // One stage p of the stream: the barrier that publishes stage p + 1 and retires stage p's slot, then 16 MFMAs on the
  // fragments of stage p (cur) with, pinned behind them, the 12 fragment reads of stage p + 1 (-> nxt) and this wave's 4
  // LDS-DMA pieces of stage p + 4 (into stage p's slot), then the counted wait for its pieces of stage p + 2.
  // more: the stream has a next stage; sb: buffer stores issued between the previous stage and this one (a tile's epilogue).
  auto stage = [&](const Frags& cur, Frags& nxt, bool more, bool sb) {
    PP_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
    int ab = a_base0 + r_slot * PSTAGE, wb = w_base0 + r_slot * PSTAGE;
    asm volatile("" : "+v"(ab), "+v"(wb));
    bool iss = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int st = j >> 3, ni = (j >> 2) & 1, mi = j & 3;
      acc[ni][mi] = mfma32<DT>(cur.w[ni][st], cur.a[mi][st], acc[ni][mi]);
      // the fragment order of the reads follows the order the NEXT stage's MFMAs need them in: w(0,0) a(0..3,0) w(1,0) ...
      constexpr int order[12] = {0, 4, 5, 6, 7, 1, 2, 8, 9, 10, 11, 3};
      if (j < 12 && more && !PV_NO_READS) read_frag(nxt, order[j], ab, wb);
      if ((j & 3) == 3 && !PV_NO_DMA) iss = dma_piece(j >> 2);
      if (!PV_NO_PIN) __builtin_amdgcn_sched_barrier(0);
    }
    if (more) r_slot = (r_slot + 1) & 3;
    __builtin_amdgcn_sched_barrier(0);
    // this wave's pieces of stage p + 2 have landed: all but the 8 youngest (+ the stores of an epilogue in between)
    if (iss) {
      if (sb) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(8 + SPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  // ---- prologue: four stages requested, the first two landed; the first stage's fragments ----
  dma_next(); dma_next(); dma_next(); dma_next();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  PP_BARRIER();
  read_frags(fx);

  int c_vb = blockIdx.x;
  bool first_tile = true;
  for (;;) {
    int64_t m0; int n0;
    tile_origin(c_vb, m0, n0);
    const bool last_tile = c_vb + G >= total_tiles;
    // the accumulators start from the bias of their columns: 8 consecutive floats per (ni, g) at a wave-uniform address =
    // scalar loads (lgkmcnt: no place in the in-order vector-memory queue behind the previous tile's stores), the lane
    // half picks its four
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float* bq = bias + n0 + wc * 64 + i * 32 + 8 * g4;
        const float4 lo = *reinterpret_cast<const float4*>(bq), hi = *reinterpret_cast<const float4*>(bq + 4);
        const float4 sel = h ? hi : lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j][4 * g4 + 0] = sel.x; acc[i][j][4 * g4 + 1] = sel.y; acc[i][j][4 * g4 + 2] = sel.z; acc[i][j][4 * g4 + 3] = sel.w;
        }
      }
    stage(fx, fy, true, !first_tile);
    stage(fy, fx, true, !first_tile);
    for (int t = 2; t < nk - 2; t += 2) {
      stage(fx, fy, true, false);
      stage(fy, fx, true, false);
    }
    stage(fx, fy, true, false);
    stage(fy, fx, !last_tile, false);
    first_tile = false;
    PP_BARRIER();                                // (every wave is past the tile's last MFMA and fragment read)

    // ---- epilogue: the accumulators hold C^T -- a lane owns activation row (per mi) and columns 32 ni + 8 g + 4 h + {0..3} ----
    if constexpr (PV_NO_EPI) {
      float sink = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sink += acc[i][j][r];
      if (sink == 1.2345f) reinterpret_cast<float*>(out)[tid] = sink;
    } else {
      // The tile leaves in eight parts of 32 rows x 256 columns.  Writers (the four waves of the part's group) put their raw
      // fp32 accumulators into the staging area a row per lane (16-byte chunk index XOR row: conflict-free both ways); after
      // the barrier ALL eight waves pick the part up as 1 KB row segments -- a thread owns four columns of four rows -- add
      // apply the epilogue's function (the bias is already in: the accumulators started from it) and send the result off: 512-byte runs of 16-bit values, or the fp32
      // read-modify-write of the residual stream (x of a part is requested before the part is staged).  Always four buffer
      // stores per thread and part (rows past the end fall outside the descriptor): the counted waits of the next tile's
      // first stage know the number.
      char* stg = smem + PRING;
      constexpr bool RES = EPI == VITTF_EPI_BIAS_RESIDUAL;
      // residual: a thread owns 4 columns (one staged chunk) of 4 rows; 16-bit outputs: 8 columns (two chunks -> one 16-byte
      // store: the epilogue is bound by the NUMBER of store instructions) of 2 rows
      const int ch = RES ? (tid & 63) : 2 * (tid & 31);
      [[maybe_unused]] const float qs = (n0 + 4 * ch) < n / 3 ? 0.125f * 1.44269504088896340736f : 1.0f;   // (uniform per tile when 256 | n / 3)
      [[maybe_unused]] float vmax_a = 0.f, vmax_b = 0.f;     // (fp8 qkv epilogue, v tiles) this thread's maxima: first / second slice of the tile
      [[maybe_unused]] int64_t fb0 = 0;
      if constexpr (EPI == PP_EPI_QKV_FP8) fb0 = m0 / tokens;
#pragma unroll 1
      for (int sp = 0; sp < 8; ++sp) {
        const int pg = sp >> 2, pm = sp & 3;
        const int64_t row0 = m0 + pg * 128 + pm * 32;
        const int64_t left = rows - row0;
        const int vr = left <= 0 ? 0 : left < 32 ? (int)left : 32;
        constexpr int ES = RES ? 4 : 2;
        // (K features: CLS rows dropped, the others move up -- infer.py:202 k[:, 1:] --: per-lane output row, one descriptor
        //  over the whole output, 32-bit byte offsets checked by the launcher, dropped rows get an offset outside it)
        const auto rs = EPI == VITTF_EPI_KFEAT
            ? __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out), 0, (int)out_bytes, 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out) + (vr ? (row0 * n + n0) * ES : 0), 0,
                                                vr ? ((vr - 1) * n + PBN) * ES : 0, 0x00020000);
        [[maybe_unused]] pu32x4_t xv[4];
        if constexpr (RES) {
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((i * 8 + (tid >> 6)) * n + 4 * ch) * 4, 0, 0);
        }
        if (grp == pg) {
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            if (mi != pm) continue;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int nl = wc * 64 + ni * 32 + 8 * g + 4 * h;
                const float4 v = make_float4(acc[ni][mi][4 * g + 0], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
                *reinterpret_cast<float4*>(stg + l31 * 1024 + (((nl >> 2) ^ l31) << 4)) = v;
              }
            }
          }
        }
        PP_BARRIER();
        if constexpr (EPI == PP_EPI_QKV_FP8) {
          // a thread owns 16 columns (four staged chunks) of ONE row of the part: 16 fp8 bytes = one 16-byte store.  A column
          // tile lies inside ONE third (256 | n / 3): the whole tile is q, k or v.
          const int rl = tid >> 4, c16 = tid & 15;
          const int dm = n / 3, third = n0 / dm;
          float v[16];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 dv = *reinterpret_cast<const float4*>(stg + rl * 1024 + (((4 * c16 + j) ^ rl) << 4));
            v[4 * j + 0] = dv.x * qs; v[4 * j + 1] = dv.y * qs; v[4 * j + 2] = dv.z * qs; v[4 * j + 3] = dv.w * qs;
          }
          float mx = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(v[e]));
          const int64_t m = row0 + rl;
          int64_t b = fb0;                                       // (slice of the row: at most one boundary inside a tile ...
          if (tokens < PBM) b = m / tokens;                      //  ... unless the slices are tiny)
          else if (m >= (fb0 + 1) * (int64_t)tokens) b = fb0 + 1;
          const int tok = (int)(m - b * tokens);
          if (third < 2) {
            // this row's 32-wide block = the 16 + 16 columns of two neighbouring lanes: block maximum -> E8M0 scale -> e4m3 bytes
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            const int ex = pp_scale_exp(mx);
            const float inv = ldexpf(1.0f, -ex);
            pu32x4_t pk8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j] * inv, v[4 * j + 1] * inv, 0, false);
              w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j + 2] * inv, v[4 * j + 3] * inv, w0, true);
              pk8[j] = (unsigned)w0;
            }
            const int col = (n0 - third * dm) + 16 * c16;                   // column inside the third: head 64 * hd + dim
            const int hd = col >> 6, dim = col & 63;
            // position of dim inside the 64-byte row: the matrix instruction's MX block b of a row is bytes 16 b .. 16 b + 15 of
            // BOTH lane halves (k = 32 (byte >> 4) + 16 (lane >> 5) + (byte & 15): tools/micro/mfma_f8_scale_probe2), and the
            // attention kernel's lane half hh reads bytes 32 hh .. 32 hh + 31: a row is stored as [d 0-15 | d 32-47 | d 16-31 | d 48-63]
            const int pos = ((dim >> 4) & 1) * 32 + (dim >> 5) * 16;
            const int64_t rowi = (b * f8.heads + hd) * f8.np + tok;
            const bool ok = m < rows;
            const int64_t total8 = (int64_t)f8.batch * f8.heads * f8.np;
            const auto r8 = __builtin_amdgcn_make_buffer_rsrc(third == 0 ? f8.q8 : f8.k8, 0, (int)(unsigned)(total8 * 64), 0x00020000);
            const auto rsc = __builtin_amdgcn_make_buffer_rsrc(third == 0 ? f8.qs : f8.ks, 0, (int)(unsigned)(total8 * 2), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(pk8, r8, ok ? (int)(unsigned)(rowi * 64 + pos) : (int)0x80000000u, 0, 0);      // (dropped rows: out of range without wrapping)
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(127 + ex), rsc, (ok && (c16 & 1) == 0) ? (int)(unsigned)(rowi * 2 + (dim >> 5)) : (int)0x80000000u, 0, 0);
          } else {
            // v: 16-bit values into the qkv buffer's v third; its absolute maximum per (slice, head) on the way out
            if (tokens < PBM) {          // (tiny slices: a tile holds more than two of them -- one atomic per thread and row)
              const int hd = ((n0 - 2 * dm) + 16 * c16) >> 6;
              if (m < rows && mx > 0.f) atomicMax(f8.amax + (b * f8.heads + hd) * 3 + 2, __float_as_uint(mx));
            } else if (m < rows) {
              if (b == fb0) vmax_a = fmaxf(vmax_a, mx); else vmax_b = fmaxf(vmax_b, mx);
            }
            pu32x4_t p0, p1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { p0[e] = pack2_h16<DT>(v[2 * e], v[2 * e + 1]); p1[e] = pack2_h16<DT>(v[8 + 2 * e], v[8 + 2 * e + 1]); }
            __builtin_amdgcn_raw_buffer_store_b128(p0, rs, (rl * n + 16 * c16) * 2, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p1, rs, (rl * n + 16 * c16) * 2 + 16, 0, 0);
          }
        } else if constexpr (RES) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rl = i * 8 + (tid >> 6);
            const float4 d = *reinterpret_cast<const float4*>(stg + rl * 1024 + ((ch ^ rl) << 4));
            float4 x = __builtin_bit_cast(float4, xv[i]);
            x.x += d.x; x.y += d.y; x.z += d.z; x.w += d.w;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4_t, x), rs, (rl * n + 4 * ch) * 4, 0, 0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int rl = i * 16 + (tid >> 5);
            const float4 d0 = *reinterpret_cast<const float4*>(stg + rl * 1024 + ((ch ^ rl) << 4));
            const float4 d1 = *reinterpret_cast<const float4*>(stg + rl * 1024 + (((ch + 1) ^ rl) << 4));
            float v[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if constexpr (EPI == VITTF_EPI_BIAS_GELU) v[e] = gelu_poly(v[e]);
              if constexpr (EPI == VITTF_EPI_BIAS_QKV) v[e] *= qs;   // the q third: softmax scale and exp -> exp2 base change
            }
            pu32x4_t pk;
            if constexpr (EPI == VITTF_EPI_KFEAT) {
#pragma unroll
              for (int e = 0; e < 4; ++e) pk[e] = pack2_h16<VITTF_FP16>(v[2 * e], v[2 * e + 1]);
              const int64_t m = row0 + rl;
              const int64_t b = m / tokens;
              const int tok = (int)(m - b * tokens);
              const int64_t orow = b * (tokens - 1) + tok - 1;
              const unsigned off = (m < rows && tok != 0) ? (unsigned)((orow * n + n0 + 4 * ch) * 2) : 0x80000000u;      // (dropped rows: an offset whose 16 bytes cannot wrap into the buffer)
              __builtin_amdgcn_raw_buffer_store_b128(pk, rs, (int)off, 0, PV_ST_AUX);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) pk[e] = pack2_h16<DT>(v[2 * e], v[2 * e + 1]);
              __builtin_amdgcn_raw_buffer_store_b128(pk, rs, (rl * n + 4 * ch) * 2, 0, PV_ST_AUX);
            }
          }
        }
        PP_BARRIER();
      }
      if constexpr (EPI == PP_EPI_QKV_FP8) {
        const int dm = n / 3;
        if (n0 / dm == 2) {
          // the 4 lanes that share a head (16 columns each), then the wave's four rows: one atomic per head, slice and wave
#pragma unroll
          for (int off = 1; off <= 2; off <<= 1) { vmax_a = fmaxf(vmax_a, __shfl_xor(vmax_a, off)); vmax_b = fmaxf(vmax_b, __shfl_xor(vmax_b, off)); }
#pragma unroll
          for (int off = 16; off <= 32; off <<= 1) { vmax_a = fmaxf(vmax_a, __shfl_xor(vmax_a, off)); vmax_b = fmaxf(vmax_b, __shfl_xor(vmax_b, off)); }
          if ((lane & 51) == 0) {                // lanes 0, 4, 8, 12
            const int hd = ((n0 - 2 * dm) >> 6) + (lane >> 2);
            const int64_t b0 = m0 / tokens;
            if (vmax_a > 0.f) atomicMax(f8.amax + (b0 * f8.heads + hd) * 3 + 2, __float_as_uint(vmax_a));
            if (vmax_b > 0.f && b0 + 1 < f8.batch) atomicMax(f8.amax + ((b0 + 1) * f8.heads + hd) * 3 + 2, __float_as_uint(vmax_b));
          }
        }
      }
    }
    if (last_tile) break;
    c_vb += G;
  }
}

template <int DT>
int launch_pp(const void* a, const void* w, const float* bias, void* out, int64_t rows, int n, int k, int epi, int tokens,
              hipStream_t st, PpFp8Out f8 = PpFp8Out{}) {
  const int64_t m_tiles = (rows + PBM - 1) / PBM;
  const int n_tiles = n / PBN;
  const int64_t total64 = m_tiles * n_tiles;
  if (total64 > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  const int total = (int)total64;
  const int cus = vittf_current_cus();
  if (cus <= 0) return VITTF_ERR_NO_DEVICE;
  // persistent: one workgroup per CU; the stride between a workgroup's tiles must keep it on its XCD (xcd_remap)
  const int grid = total < cus ? total : (cus & ~7) ? (cus & ~7) : cus;
  unsigned out_bytes = 0;
  if (epi == VITTF_EPI_KFEAT) {
    const int64_t ob = (rows - rows / tokens) * (int64_t)n * 2;
    if (ob > 0xfffffff0ll) return 1;             // (32-bit byte offsets in that epilogue: the caller's other kernel takes it)
    out_bytes = (unsigned)ob;
  }
  const unsigned short* A = (const unsigned short*)a;
  const unsigned short* Wp = (const unsigned short*)w;
#define VITTF_PP_CASE(E)                                                                                              \
  case E:                                                                                                             \
    hipLaunchKernelGGL((gemm_pp_kernel<DT, E>), dim3(grid), dim3(512), 0, st, A, Wp, bias, out, rows, n, k, tokens,   \
                       n_tiles, total, out_bytes, f8);                                                                \
    break;
  switch (epi) {
    VITTF_PP_CASE(VITTF_EPI_BIAS)
    VITTF_PP_CASE(VITTF_EPI_BIAS_GELU)
    VITTF_PP_CASE(VITTF_EPI_BIAS_RESIDUAL)
    VITTF_PP_CASE(VITTF_EPI_KFEAT)
    VITTF_PP_CASE(VITTF_EPI_BIAS_QKV)
    VITTF_PP_CASE(PP_EPI_QKV_FP8)
    default: return VITTF_ERR_INVALID_ARG;
  }
#undef VITTF_PP_CASE
  return vittf_check_launch();
}

}  // namespace

#ifdef PP_STANDALONE      // tools/pp_variants.sh builds this file alone
void vittf_note_kernel(int, const char*) {}
#endif

// 1 = shape not covered (the caller falls back to gemm.hip's 128 x 128 tiles)
#ifdef PP_STANDALONE
extern "C"
#endif
int vittf_gemm_pp(const void* a, const void* w, const float* bias, void* out, int64_t rows, int32_t n, int32_t k,
                  int32_t epilogue, int32_t tokens, int32_t dtype, hipStream_t st) {
  if (k < 768 || k % (2 * PBK) != 0 || n % PBN != 0) return 1;
  if ((int64_t)k * 2 * PBM > 0x7fffffff || (int64_t)n * 4 * 64 > 0x7fffffff) return 1;      // per-lane offsets are 32-bit
  if ((((uintptr_t)a | (uintptr_t)w | (uintptr_t)out) & 15) != 0) return 1;
  vittf_note_kernel(VITTF_KERNEL_GEMM, "gemm_pp_kernel");
  if (dtype == VITTF_BF16) return launch_pp<VITTF_BF16>(a, w, bias, out, rows, n, k, epilogue, tokens, st);
  if (dtype == VITTF_FP16) return launch_pp<VITTF_FP16>(a, w, bias, out, rows, n, k, epilogue, tokens, st);
  return VITTF_ERR_INVALID_ARG;
}

#ifndef PP_STANDALONE
void vittf_fp8_ws_pointers(void* ws, int32_t batch, int32_t tokens, int32_t heads, unsigned** amax, unsigned char** q8,
                           unsigned char** k8, unsigned char** qs, unsigned char** ks, int32_t* np);   // attention_fp8.hip

// Attention.qkv with fp8 outputs for vittf_attention_fp8_rows: see include/vittf.h
// The last 64-row tile of every (slice, head) of q8 / k8 and of their scale bytes, zeroed in front of the GEMM that fills its valid
// rows: the rows tokens .. np - 1 are read by the attention kernel's last key / query tile (and masked behind the product); a zero
// byte is a valid e4m3 value and a valid E8M0 scale, what an earlier call left there may be neither.
namespace {
__global__ __launch_bounds__(256) void fp8_zero_last_tile(unsigned char* __restrict__ q8, unsigned char* __restrict__ k8,
                                                          unsigned char* __restrict__ qs, unsigned char* __restrict__ ks, int np) {
  const int64_t row0 = (int64_t)blockIdx.x * np + np - 64;
  const uint4 z = {0u, 0u, 0u, 0u};
  reinterpret_cast<uint4*>(q8 + row0 * 64)[threadIdx.x] = z;
  reinterpret_cast<uint4*>(k8 + row0 * 64)[threadIdx.x] = z;
  if (threadIdx.x < 8) {
    reinterpret_cast<uint4*>(qs + row0 * 2)[threadIdx.x] = z;
    reinterpret_cast<uint4*>(ks + row0 * 2)[threadIdx.x] = z;
  }
}
}  // namespace

extern "C" int vittf_gemm_qkv_fp8(const void* a, const void* w, const float* bias, void* qkv_out, int64_t rows, int32_t n,
                                  int32_t k, int32_t tokens, int32_t heads, int32_t dtype, void* ws, size_t ws_bytes,
                                  void* stream) {
  if (!a || !w || !bias || !qkv_out || !ws || rows <= 0 || tokens <= 0 || heads <= 0) return VITTF_ERR_INVALID_ARG;
  if (n != 3 * heads * 64 || (n / 3) % PBN != 0 || k < 768 || k % (2 * PBK) != 0) return VITTF_ERR_INVALID_ARG;
  if (rows % tokens != 0 || rows / tokens > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  if ((int64_t)k * 2 * PBM > 0x7fffffff || (int64_t)n * 4 * 64 > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  if ((((uintptr_t)a | (uintptr_t)w | (uintptr_t)qkv_out) & 15) != 0 || ((uintptr_t)ws & 255) != 0) return VITTF_ERR_INVALID_ARG;
  if (dtype != VITTF_BF16 && dtype != VITTF_FP16) return VITTF_ERR_INVALID_ARG;
  const int batch = (int)(rows / tokens);
  if (ws_bytes < vittf_attention_fp8_workspace_bytes(batch, tokens, heads)) return VITTF_ERR_WORKSPACE;
  PpFp8Out f8;
  int np = 0;
  vittf_fp8_ws_pointers(ws, batch, tokens, heads, &f8.amax, &f8.q8, &f8.k8, &f8.qs, &f8.ks, &np);
  f8.np = np; f8.heads = heads; f8.batch = batch;
  if ((int64_t)batch * heads * np * 64 > 0xfffffff0ll) return VITTF_ERR_INVALID_ARG;       // 32-bit byte offsets into q8 / k8
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(f8.amax, 0, (size_t)batch * heads * 3 * 4, st) != hipSuccess) return VITTF_ERR_LAUNCH;
  if (np != tokens)
    hipLaunchKernelGGL(fp8_zero_last_tile, dim3((unsigned)(batch * heads)), dim3(256), 0, st, f8.q8, f8.k8, f8.qs, f8.ks, np);
  vittf_note_kernel(VITTF_KERNEL_GEMM_QKV, "gemm_pp_kernel<qkv -> fp8 q, k + row scales>");
  if (dtype == VITTF_BF16) return launch_pp<VITTF_BF16>(a, w, bias, qkv_out, rows, n, k, PP_EPI_QKV_FP8, tokens, st, f8);
  return launch_pp<VITTF_FP16>(a, w, bias, qkv_out, rows, n, k, PP_EPI_QKV_FP8, tokens, st, f8);
}
#endif
