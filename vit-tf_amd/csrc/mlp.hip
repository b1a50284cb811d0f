// Fused transformer MLP for D = 384 with the LayerNorm that follows it:
//     x += fc2( gelu_erf( fc1(h) + b1 ) ) + b2 ;   h_next = LayerNorm(x; g, b)
//
// Replaces Mlp.forward + the residual add of the upstream DINO block and the next block's norm1 (reached through
// model(...), infer.py:177).  The two-GEMM path writes the [rows][4D] hidden activation to HBM and reads it back -- 806 MB
// per 32-slice batch, a third of the GEMM class's traffic -- and its fc2 kernel waits for that stream;
// here the hidden activation never leaves the registers and the only streams are the weights, which every CU reads from
// L2 at the same time (2.36 MB per layer), and one pass over h / x.
//
// Machine mapping (gfx950; round 3 -- the round-1 kernel of this file had the same data flow but ran its phases one after
// the other: 0.66 ms per 32 slices against 0.44 for the two GEMMs):
//   * workgroup = 4 waves = 128 rows, ONE wave per SIMD with the whole 512-entry register file, persistent over row tiles.
//     Both products are computed transposed (weights = MFMA A operand, activations = B operand, a lane owns one row): a
//     wave keeps its 32 rows' LayerNorm output as 24 B operands (96 registers) and the full 384-wide output as 12
//     accumulator tiles (192), so the fc1 accumulator tile -- bias as its initial value, GELU applied, converted pairwise
//     to 16 bit -- IS the B operand of fc2 (k order 16 s + 8 (j >> 2) + 4 h + (j & 3): the host stores W2's hidden dim in
//     that order) and a row's LayerNorm statistics never leave its wave.
//   * the hidden dim is walked in units of 32: unit u costs 24 MFMAs for fc1 (one accumulator chain over K = 384) and 24
//     for fc2 (12 output tiles x 2 k steps).  The weights arrive as a STREAM of 24 KB images, one per unit and product,
//     packed by the host in exactly the order and LDS layout they are consumed in (weights.pack_mlp_weights): every LDS-DMA
//     piece is 1 KB of contiguous memory.  Six ring slots, five units requested ahead, one barrier per unit, counted vmcnt.
//   * software pipeline, pinned per MFMA gap (one wave per SIMD: nothing else fills the gaps): the 24 A fragments of a
//     unit are read through 8 fragment registers refilled in place right behind the MFMA that used them (the reads run 8
//     MFMAs ahead, across unit boundaries); the GELU of unit u runs in the gaps of fc1(u + 1); the six DMA pieces a wave
//     issues per unit sit behind every fourth MFMA.
//   * epilogue per tile: 16-byte read-modify-write of the fp32 residual rows, two-pass LayerNorm in the same registers
//     (the other half of a row is in the lane 32 further on), 8-byte stores of the 16-bit h.
#include "vittf_common.h"

#include <stdlib.h>

namespace {

constexpr int D = 384, HID = 4 * D;
constexpr int UNITS = HID / 32;            // 48 hidden units of 32
constexpr int NSEQ = 2 * UNITS;            // streamed images per row tile: W1(0), W1(1), W1(2), W2(0), W1(3), W2(1), .., W1(47), W2(45), W2(46), W2(47)
constexpr int UB = 24576;                  // bytes of one image: 6 sub-images [32 rows][64 k] in the tile_off layout
constexpr int NSLOT = 6, AHEAD = NSLOT - 1;
constexpr int NF = 8;                      // fragment registers in flight
constexpr int PIECES = UB / 1024 / 4;      // LDS-DMA pieces per wave and unit
constexpr int CONST_OFF = NSLOT * UB;      // b1 [1536] | b2 [384] | gamma [384] | beta [384] as fp32 behind the ring
constexpr int LDS_BYTES = CONST_OFF + (HID + 3 * D) * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

struct TileIO {                // a row tile's activations: descriptors that end with the tile's last valid row (loads past it
  __amdgpu_buffer_rsrc_t x, h_next, h_out;    // return 0, stores are dropped) and this lane's byte offsets into them
  int xoff, hoff, hooff;       // x: row * 1536 + 16 h;  h in: row * 768 + 16 h;  h out: row * 768 + 8 h
};

struct Stream {                // where the weight stream stands (wave-uniform)
  i32x4_t rsrc;                // descriptor over one layer's NSEQ packed images
  unsigned dma_dst;            // LDS byte address of this wave's first piece in slot 0
  int g;                       // stream position of the unit being computed (0 .. NSEQ - 1, wraps with the row tiles)
  int slot;                    // its ring slot
};

// fragment f (0 .. 23) of the image the bases point at: sub-image f >> 2, chunk pair f & 3
// (the bases are LDS byte addresses, not generic pointers: they rotate through the ring at run time, and hipcc turns a
// pointer it cannot prove to be LDS into flat loads, which count on vmcnt and drain the LDS-DMA queue)
typedef __attribute__((address_space(3))) const s16x8_t* lds_frag_ptr;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) const f32x4_t* lds_f4_ptr;
__device__ __forceinline__ s16x8_t ld_frag(const unsigned (&base)[4], int f) {
  return *(lds_frag_ptr)(base[f & 3] + (f >> 2) * 4096);
}

// ---- timing-only variants for tools/mlp_variants.sh (never in libvittf.so: the Makefile does not define MLP_VARIANT) ----
#ifndef MLP_VARIANT
#define MLP_VARIANT 0
#endif
#ifndef MLP_HP
#define MLP_HP 2
#endif
#if MLP_VARIANT & 16      // in-kernel stamps: s_memtime (shader cycles) | s_memrealtime (100 MHz) at eight points of a tile
__device__ unsigned long long g_mlp_stamps[4 /*workgroups*/][4 /*tiles*/][4 /*waves*/][8][2];
#define MLP_STAMP(k)                                                                                          \
  do {                                                                                                        \
    if (blockIdx.x < 4 && tile_no < 4) {                                                                      \
      unsigned long long t0_, t1_;                                                                            \
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_), "=s"(t1_)::"memory");  \
      if ((threadIdx.x & 63) == 0) {                                                                          \
        g_mlp_stamps[blockIdx.x][tile_no][threadIdx.x >> 6][k][0] = t0_;                                      \
        g_mlp_stamps[blockIdx.x][tile_no][threadIdx.x >> 6][k][1] = t1_;                                      \
      }                                                                                                       \
    }                                                                                                         \
  } while (0)
#else
#define MLP_STAMP(k)
#endif
constexpr bool V_NO_DMA = MLP_VARIANT & 1, V_M0_KEEP = MLP_VARIANT & 2 /* here: save + restore M0 */, V_NO_REFILL = MLP_VARIANT & 4, V_NO_GELU = MLP_VARIANT & 8;

// an LDS-DMA piece that leaves M0 pointing at its destination (lds_dma16 saves and restores it: the restore waits until the
// load has left the wave's instruction buffer).  hipcc keeps nothing in M0 in this kernel: tests/test_host_cpu.py checks
// the disassembly for that.
__device__ __forceinline__ void lds_dma16_keep(i32x4_t rsrc, unsigned lds_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// the exact-erf GELU of vittf_common.h (gelu_poly: the same operations in the same order, so the same bits) cut into three
// pieces of at most 16 issue cycles, one per MFMA gap: beside an MFMA a one-wave SIMD hides about 24 cycles of other work
struct Gelu3 { float z, p; };
__device__ __forceinline__ void gelu_a(Gelu3& s, float x) {
  s.z = fabsf(x) * 0.70710678118654752f;
  s.p = fmaf(-0.002965539f, s.z, 0.0296764448f);
  s.p = fmaf(s.p, s.z, -0.148780614f);
  s.p = fmaf(s.p, s.z, -0.918451846f);
  asm volatile("" : "+v"(s.z), "+v"(s.p));         // pinned to this gap
}
__device__ __forceinline__ void gelu_b(Gelu3& s) {
  s.p = fmaf(s.p, s.z, -1.6278975f);
  s.p = __builtin_amdgcn_exp2f(fmaf(s.p, s.z, -1.0f));
  asm volatile("" : "+v"(s.p));
}
__device__ __forceinline__ float gelu_c(const Gelu3& s, float x) {
  float v = fmaf(-fabsf(x), s.p, fmaxf(x, 0.f));
  asm volatile("" : "+v"(v));
  return v;
}

// One unit of the stream = 24 MFMAs.  FC1: gacc = W1u . h^T (+ the bias tile as initial value); FC2: xacc[ot] += W2u . gf_in.
// GH: which part of the activation of the fc1 tile gprev runs in this unit's gaps -- 0 none, 1 values 0 .. 7 -> gf_out[0],
// 2 values 8 .. 15 -> gf_out[1] (one value per three gaps, a third of it in each), 3 all sixteen (first / last units of a tile).
template <int DT, bool FC1, int GH, int WAITN = 3 * PIECES, int XL = -1, bool HP = false, bool LAST = false>
__device__ __forceinline__ void mlp_unit(Stream& st, unsigned (&base)[4], s16x8_t (&wf)[NF], s16x8_t (&hf)[D / 16],
                                         f32x16_t (&xacc)[D / 32], f32x16_t& gacc, const f32x16_t& bias_c,
                                         const f32x16_t& gprev, s16x8_t (&gf_out)[2], const s16x8_t (&gf_in)[2],
                                         const TileIO& io) {
  // the unit's image was requested AHEAD units ago; of what this wave issued since only the pieces of the three units behind
  // the NEXT one may still be in flight: the next unit's image has landed too (its fragments are read from MFMA 16 on).
  // WAITN = those 18 pieces + the other loads / stores the wave has issued in the last three units (vmcnt counts them all, in
  // order; at most 63).  The barrier also says that everybody is done with the slot of the unit before this one, refilled below.
  static_assert(WAITN <= 63 && WAITN >= 3 * PIECES, "vmcnt");
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(V_NO_DMA ? 0 : WAITN) : "memory");
  const int g_next = st.g + AHEAD < NSEQ ? st.g + AHEAD : st.g + AHEAD - NSEQ;       // (the stream wraps: the next row tile)
  const int slot_free = st.slot == 0 ? NSLOT - 1 : st.slot - 1;
  u32x4_t pk0 = {}, pk1 = {};
  float vprev = 0.f;
  Gelu3 gs = {};
#pragma unroll
  for (int j = 0; j < 24; ++j) {
    if constexpr (FC1) gacc = mfma32<DT>(wf[j % NF], hf[j], j == 0 ? bias_c : gacc);
    else xacc[j >> 1] = mfma32<DT>(wf[j % NF], gf_in[j & 1], xacc[j >> 1]);
    if (j == 24 - NF) {                        // from here on the refills read the next unit's image
      const int d_ = st.slot == NSLOT - 1 ? -(NSLOT - 1) * UB : UB;
#pragma unroll
      for (int i = 0; i < 4; ++i) base[i] += d_;
    }
    if (!V_NO_REFILL && !(LAST && j >= 24 - NF)) wf[j % NF] = ld_frag(base, (j + NF) % 24);   // (a tile's last unit: see the kernel)
    if ((j & 3) == 3 && !V_NO_DMA) {           // one LDS-DMA piece of the unit AHEAD behind every fourth MFMA
      const unsigned dst = st.dma_dst + slot_free * UB + (j >> 2) * 4096;
      const int src = g_next * UB + (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 1024 + (j >> 2) * 4096;
      if (V_M0_KEEP) lds_dma16(st.rsrc, dst, (int)((threadIdx.x & 63) * 16), src);
      else lds_dma16_keep(st.rsrc, dst, (int)((threadIdx.x & 63) * 16), src);
    }
    if constexpr (XL >= 0) {                   // a third of the tile's residual rows -> the fc2 accumulators' initial value
      if (j % 3 != 2) {                        // (16 x 16 bytes per lane; first needed three units on)
        const int i = 2 * (j / 3) + j % 3, ot = 4 * XL + (i >> 2), g = i & 3;
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(io.x, io.xoff, (32 * ot + 8 * g) * 4, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) xacc[ot][4 * g + e] = __uint_as_float(v[e]);
      }
    }
    if constexpr (HP)                          // fc1 is done with the tile: the next one's h
      hf[j] = __builtin_bit_cast(s16x8_t, __builtin_amdgcn_raw_buffer_load_b128(io.h_next, io.hoff, 32 * j, 0));
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
        const float v = gelu_c(gs, gprev[r]);
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
  st.g = st.g + 1 == NSEQ ? 0 : st.g + 1;
  st.slot = st.slot + 1 == NSLOT ? 0 : st.slot + 1;
}

template <int DT>
__global__ __launch_bounds__(256, 1) void mlp_kernel(const unsigned short* __restrict__ hbuf, const unsigned short* __restrict__ wpk,
                                                     const float* __restrict__ b1, const float* __restrict__ b2,
                                                     float* __restrict__ x, int64_t rows, const float* __restrict__ ln_g,
                                                     const float* __restrict__ ln_b, float ln_eps,
                                                     unsigned short* __restrict__ hout, int ntiles) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  float* const cst = reinterpret_cast<float*>(smem + CONST_OFF);
  for (int i = tid; i < HID + 3 * D; i += 256)
    cst[i] = i < HID ? b1[i] : i < HID + D ? b2[i - HID] : i < HID + 2 * D ? (ln_g ? ln_g[i - HID - D] : 1.f) : (ln_b ? ln_b[i - HID - 2 * D] : 0.f);

  // this lane's view of the constants (half h reads 4 floats further on); opaque, so that every read below is this one
  // register + an immediate offset (hipcc otherwise keeps a separate address register per constant position and spills them)
  unsigned cl = (unsigned)(size_t)LDS_PTR(smem) + CONST_OFF + 16 * h;
  asm volatile("" : "+v"(cl));
  auto cst4 = [&](int i) { const f32x4_t v = *(lds_f4_ptr)(cl + 4 * i); return make_float4(v[0], v[1], v[2], v[3]); };      // floats i .. i + 3 (+ 4 h) of the constants
  Stream st;
  st.rsrc = lds_dma_rsrc(wpk, (unsigned)(NSEQ * UB));
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
  io.xoff = (wave * 32 + l31) * (D * 4) + 16 * h;
  io.hoff = (wave * 32 + l31) * (D * 2) + 16 * h;
  io.hooff = (wave * 32 + l31) * (D * 2) + 8 * h;
  // ---- this lane's h fragments (B operand) of the first tile: H[row][16 s + 8 h .. + 7], s = 0 .. 23 ----
  s16x8_t hf[D / 16];
  {
    const auto rs = tile_rsrc(hbuf, blockIdx.x, D * 2);
#pragma unroll
    for (int s = 0; s < D / 16; ++s) hf[s] = __builtin_bit_cast(s16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, io.hoff, 32 * s, 0));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first AHEAD images have landed (only here: the unit waits count
  __syncthreads();                                       // on a steady stream) ... everybody's pieces; the constants are written
  s16x8_t wf[NF];

  [[maybe_unused]] int tile_no = -1;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    ++tile_no;
    MLP_STAMP(0);
    // the first fragments of the tile's first image (landed: the wait in front of the previous unit / above).  Every other
    // unit reads its first eight fragments behind the last MFMAs of the unit before it; carried over the epilogue they
    // would be spilled
#pragma unroll
    for (int f = 0; f < NF; ++f) wf[f] = ld_frag(base, f);
    io.x = tile_rsrc(x, tile, D * 4);
    if (MLP_HP == 0 && tile != (int)blockIdx.x) {
      const auto rs = tile_rsrc(hbuf, tile, D * 2);
#pragma unroll
      for (int s = 0; s < D / 16; ++s) hf[s] = __builtin_bit_cast(s16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, io.hoff, 32 * s, 0));
    }
    io.h_next = tile_rsrc(hbuf, (int64_t)tile + gridDim.x, D * 2);      // (no next tile: an empty descriptor)
    io.h_out = tile_rsrc(hout, tile, D * 2);
    f32x16_t xacc[D / 32];     // fc2 accumulators: output tile ot (32 columns) x this lane's row; start from the residual rows
    f32x16_t ga, gb = {}, bias_c;     // two fc1 tiles: one being accumulated, one being activated
    s16x8_t gf0[2] = {}, gf1[2] = {};  // two activated tiles (fc2's B operands): one being packed, one being consumed
    // bias tile of hidden unit u: register r of lane half h = hidden 32 u + (r & 3) + 8 (r >> 2) + 4 h
    auto load_bias = [&](int u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bv = cst4(32 * u + 8 * q);
        bias_c[4 * q + 0] = bv.x; bias_c[4 * q + 1] = bv.y; bias_c[4 * q + 2] = bv.z; bias_c[4 * q + 3] = bv.w;
      }
    };
    // The unit sequence (= the order of the packed images): fc1 runs two hidden units ahead of fc2, and the activation of the
    // unit in between is spread over both units of a pair.  Beside the first three units (fc1 only) the residual rows are
    // loaded into the fc2 accumulators, beside the last but one (fc2 only) the next tile's h: what is left between two tiles
    // is the LayerNorm arithmetic and the stores.  vmcnt per unit: 18 + what the three units before it issued on top
    // (unit 0 .. 3: the 96 stores of the previous tile's epilogue / 16 loads per unit, capped at 63).
    load_bias(0);
    mlp_unit<DT, true, 0, 63, 0>(st, base, wf, hf, xacc, ga, bias_c, gb, gf0, gf0, io);                  // fc1(0)
    MLP_STAMP(1);
    load_bias(1);
    mlp_unit<DT, true, 3, 63, 1>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf0, io);                  // fc1(1) | gelu(0) -> gf0
    load_bias(2);
    mlp_unit<DT, true, 1, 63, 2>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf1, io);                  // fc1(2) | gelu(1), values 0 .. 7
    mlp_unit<DT, false, 2, 63>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf0, io);                    // fc2(0) | gelu(1), values 8 .. 15
    load_bias(3);
    mlp_unit<DT, true, 1, 18 + 32>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf0, io);                // fc1(3) | gelu(2)
    mlp_unit<DT, false, 2, 18 + 16>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf1, io);               // fc2(1) | gelu(2)
    MLP_STAMP(2);
    for (int u = 2; u < UNITS - 2; u += 2) {
      load_bias(u + 2);
      mlp_unit<DT, true, 1>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf1, io);                       // fc1(u + 2) | gelu(u + 1)
      mlp_unit<DT, false, 2>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf0, io);                      // fc2(u)     | gelu(u + 1)
      load_bias(u + 3);
      mlp_unit<DT, true, 1>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf0, io);                       // fc1(u + 3) | gelu(u + 2)
      mlp_unit<DT, false, 2>(st, base, wf, hf, xacc, gb, bias_c, ga, gf0, gf1, io);                      // fc2(u + 1) | gelu(u + 2)
    }
    MLP_STAMP(3);
    mlp_unit<DT, false, 3, 18, -1, MLP_HP == 1>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf0, io);   // fc2(46) | gelu(47) -> gf1
    mlp_unit<DT, false, 0, MLP_HP == 1 ? 18 + 24 : 18, -1, false, true>(st, base, wf, hf, xacc, ga, bias_c, gb, gf1, gf1, io);               // fc2(47)

    // ---- epilogue: x[row][col .. col + 3] = acc + b2, lane owns row m, columns 32 ot + 8 g + 4 h + {0 .. 3}; then the
    //      LayerNorm of the new row from the same registers (three passes, one output tile at a time: the compiler otherwise
    //      keeps all 192 values in flight twice and spills the next tile's h) ----
    MLP_STAMP(4);
    if (MLP_HP == 2) {         // the next tile's h fragments: in flight while the LayerNorm below is computed
#pragma unroll
      for (int s = 0; s < D / 16; ++s) hf[s] = __builtin_bit_cast(s16x8_t, __builtin_amdgcn_raw_buffer_load_b128(io.h_next, io.hoff, 32 * s, 0));
    }
    float s = 0.f;
#pragma unroll
    for (int ot = 0; ot < D / 32; ++ot) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = 32 * ot + 8 * g;
        const float4 bv = cst4(HID + col);
        u32x4_t v;
        xacc[ot][4 * g + 0] += bv.x; xacc[ot][4 * g + 1] += bv.y; xacc[ot][4 * g + 2] += bv.z; xacc[ot][4 * g + 3] += bv.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(xacc[ot][4 * g + e]);
        __builtin_amdgcn_raw_buffer_store_b128(v, io.x, io.xoff, col * 4, 0);
        s += (xacc[ot][4 * g + 0] + xacc[ot][4 * g + 1]) + (xacc[ot][4 * g + 2] + xacc[ot][4 * g + 3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    MLP_STAMP(5);
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
#pragma unroll
      for (int ot = 0; ot < D / 32; ++ot) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = 32 * ot + 8 * g;
          const float4 gg = cst4(HID + D + col);
          const float4 bb = cst4(HID + 2 * D + col);
          u32x2_t pk;
          pk[0] = pack2_h16<DT>((xacc[ot][4 * g + 0] - mean) * rstd * gg.x + bb.x, (xacc[ot][4 * g + 1] - mean) * rstd * gg.y + bb.y);
          pk[1] = pack2_h16<DT>((xacc[ot][4 * g + 2] - mean) * rstd * gg.z + bb.z, (xacc[ot][4 * g + 3] - mean) * rstd * gg.w + bb.w);
          __builtin_amdgcn_raw_buffer_store_b64(pk, io.h_out, io.hooff, col * 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    MLP_STAMP(7);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the images requested beyond the last unit: land before the LDS goes away
}

}  // namespace

#if MLP_VARIANT & 16
extern "C" int vittf_mlp_stamps(unsigned long long* out) {      // [4][4][4][8][2], host memory
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_stamps), sizeof(g_mlp_stamps)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int vittf_mlp_fused(const void* h, const void* w_packed, const float* b1, const float* b2, float* x, int64_t rows,
                               int32_t d, int32_t dtype, const float* ln_g, const float* ln_b, float ln_eps, void* h_out,
                               void* stream) {
  if (!h || !w_packed || !b1 || !b2 || !x || rows <= 0) return VITTF_ERR_INVALID_ARG;
  if (d != D) return VITTF_ERR_INVALID_ARG;          // the register budget is sized for ViT-S
  if ((ln_g || ln_b || h_out) && !(ln_g && ln_b && h_out)) return VITTF_ERR_INVALID_ARG;
  const int64_t tiles = (rows + 127) / 128;
  if (tiles > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
  }();
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  hipStream_t st = (hipStream_t)stream;
#define MLP_LAUNCH(DTV)                                                                                              \
  hipLaunchKernelGGL((mlp_kernel<DTV>), dim3(grid), dim3(256), 0, st, (const unsigned short*)h,                      \
                     (const unsigned short*)w_packed, b1, b2, x, rows, ln_g, ln_b, ln_eps, (unsigned short*)h_out, (int)tiles)
  if (dtype == VITTF_BF16) MLP_LAUNCH(VITTF_BF16);
  else if (dtype == VITTF_FP16) MLP_LAUNCH(VITTF_FP16);
  else return VITTF_ERR_INVALID_ARG;
#undef MLP_LAUNCH
  return vittf_check_launch();
}
