// The tail of a ViT-S block (D = 384) in one launch, two waves per SIMD with split roles:
//     x' = x + a . Wp^T + bp ;  x := x' + fc2( gelu_erf( fc1( LayerNorm(x'; g2, e2) ) + b1 ) ) + b2 ;  h_next = LayerNorm(x; g, b)
// (a = the attention output).  Replaces Attention.proj, both residual adds, norm2, Mlp.forward of the upstream DINO block and
// the next block's norm1 (reached through model(...), /root/reference/infer.py:177).
//
// Why roles.  A 32-row block of this computation carries 192 registers of fp32 rows (x', then the fc2 accumulators) and 96
// registers of 16-bit rows (norm2's output, fc1's B operand): 288 + working registers.  One wave per SIMD holds that in its
// 512 registers (round 3's mlp.hip), but at its tile boundary -- projection waiting for residual rows, two LayerNorms, the
// store chains of the epilogue: a third of the tile -- the matrix pipe idles and nobody else is there to use it.  Two waves
// per SIMD have 256 registers each, so the state is SPLIT between SIMD partners:
//   * X wave (waves 0 .. 3, row block = wave): the fp32 rows.  Projection (K-major, so that no activation operand stays in
//     registers), residual, norm2, fc2 accumulating on top of x', epilogue, the next LayerNorm.  Issues ALL LDS-DMA of the
//     weight ring (its vmcnt queue holds only those pieces and its own stores).
//   * F wave (waves 4 .. 7, row block = wave - 4): the 16-bit rows.  fc1 over K = 384 with the bias as initial value and the
//     exact-erf GELU in its own MFMA gaps; having the free registers at the tile boundary, it also fetches the next tile's
//     rows (attention output, residual) in coalesced 128-byte runs long before they are needed and hands them over in the
//     layouts the X wave consumes.  Issues all row loads (no LDS-DMA: its vmcnt queue is the compiler's own).
// What they hand each other goes through 9 KB of LDS per pair, lane to lane where the MFMA operand layouts of producer and
// consumer are the same lane's registers: norm2's output X -> F once per tile, the activated fc1 tile (2 KB) F -> X once per
// hidden unit, the next tile's attention-output fragments and residual chunks F -> X at the tile boundary.
//
// The weights arrive as ONE stream of 24 KB steps through an LDS-DMA ring (packed by the host in consumption order and LDS
// layout, weights.pack_tail_fx_weights): 24 projection steps [ - | Wp k step p of all 12 output tiles ] = 12 MFMAs of X, then
// 100 main steps [ W1(u) k half | W2(u - 2) output-tile half ] = 12 MFMAs of F + 12 MFMAs of X, F two hidden units ahead.
// One bare barrier per step, counted vmcnt (X).  Measured design points: profiles/r05a_tail_fx_pricing.txt.
#include "vittf_common.h"

#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int D = 384, HID = 4 * D, UNITS = HID / 32;
constexpr int SB = 24576, HB = SB / 2;       // bytes of one ring step / of one role's half of a step
constexpr int PSTEPS = D / 32;               // projection steps in front (X: 24 MFMAs each, two k steps of all output tiles)
constexpr int LAG = 4;                       // main steps between a half of fc1(u) and the same half of fc2(u)
constexpr int MSTEPS = 2 * UNITS + LAG;      // main steps of a row tile
#ifndef FX_VARIANT
#define FX_VARIANT 0
#endif
#ifndef FX_NSLOT
#define FX_NSLOT 4
#endif
// timing-only builds (tools/fx_variants.sh; never in libvittf.so): 1 = main phase only (no tile boundary), 2 = no GELU
// arithmetic, 4 = no LDS-DMA inside the steps, 8 = no fragment refills, 16 = stamps, 32 = steps without their barrier,
// 64 = no raised priority for the F waves
constexpr bool V_MAIN_ONLY = FX_VARIANT & 1, V_NO_GELU = FX_VARIANT & 2, V_NO_DMA = FX_VARIANT & 4, V_NO_REFILL = FX_VARIANT & 8,
               V_NO_BARRIER = FX_VARIANT & 32, V_NO_PRIO = FX_VARIANT & 64;
constexpr int NSEQ = V_MAIN_ONLY ? MSTEPS : PSTEPS + MSTEPS;
constexpr int NSLOT = FX_NSLOT, AHEAD = NSLOT - 1;
constexpr int PIECES = SB / 1024 / 4;        // LDS-DMA pieces per X wave and step
constexpr int WAIT0 = (AHEAD - 2) * PIECES;  // pieces of an X wave that may be in flight when a step starts
constexpr int NF = 4;                        // weight fragments in flight per wave
constexpr int PB_OFF = NSLOT * SB;           // pair buffers behind the ring: one per row block, two halves
constexpr int PBH = 4608, PB = 2 * PBH;
constexpr int STG_ROW = 144;                 // staging rows: 128 bytes + 16 (36 banks)
constexpr int AF_SLOT = 1056, AF_H = 528;    // attention-output fragments: 4 slots per half, lane half h at + 528
constexpr int CONST_OFF = PB_OFF + 4 * PB;   // fp32 constants behind that, in floats:
constexpr int C_B1 = 0, C_B2 = HID, C_G1 = HID + D, C_E1 = HID + 2 * D,       // b1 | b2 | gamma, beta of the LayerNorm behind the MLP
              C_BP = HID + 3 * D, C_G2 = HID + 4 * D, C_E2 = HID + 5 * D,      // proj bias | gamma, beta of norm2
              C_N = HID + 6 * D;
constexpr int NEXT_OFF = CONST_OFF + C_N * 4;     // one word: the tile the workgroup takes next
constexpr int LDS_BYTES = NEXT_OFF + 16;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
static_assert(AHEAD >= 3, "ring depth");
static_assert(32 * STG_ROW <= PBH && 4 * AF_SLOT <= PBH, "pair buffer");

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((address_space(3))) const s16x8_t* lds_frag_ptr;
typedef __attribute__((address_space(3))) const f32x4_t* lds_f4_ptr;
typedef __attribute__((address_space(3))) f32x4_t* lds_w4_ptr;
typedef __attribute__((address_space(3))) u32x4_t* lds_wu4_ptr;
typedef __attribute__((address_space(3))) u32x2_t* lds_w2_ptr;
typedef __attribute__((address_space(3))) volatile unsigned* lds_u32_ptr;

struct Ring {                  // where the weight stream stands (wave-uniform)
  i32x4_t rsrc;                // descriptor over one layer's NSEQ packed steps
  unsigned dma_dst;            // LDS byte address of slot 0
  int src0;                    // byte offset inside a step of this wave's first piece
  int g;                       // stream position of the step being computed (0 .. NSEQ - 1, wraps with the row tiles)
  int slot;                    // its ring slot
};

__device__ __forceinline__ s16x8_t ld_frag(const unsigned (&base)[4], int f) {
  return *(lds_frag_ptr)(base[f & 3] + (f >> 2) * 4096);
}

#if FX_VARIANT & 16
__device__ unsigned long long g_fx_stamps[4 /*workgroups*/][4 /*tiles*/][8 /*waves*/][8];
__device__ unsigned g_fx_hwid[8];
#define FX_STAMP(k)                                                                                           \
  do {                                                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_[k])::"memory");                       \
    if (k == 7 && blockIdx.x < 4 && tile_no < 4 && (threadIdx.x & 63) == 0) {                                 \
      _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_)                                                        \
        g_fx_stamps[blockIdx.x][tile_no][threadIdx.x >> 6][q_] = stamp_[q_];                                  \
    }                                                                                                         \
  } while (0)
#else
#define FX_STAMP(k)
#endif

// an LDS-DMA piece that leaves M0 pointing at its destination (hipcc keeps nothing in M0 in this kernel:
// tests/test_host_cpu.py checks the disassembly for that)
__device__ __forceinline__ void lds_dma16_keep(i32x4_t rsrc, unsigned lds_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// the exact-erf GELU of vittf_common.h (gelu_poly: the same operations in the same order, so the same bits) in three pieces
struct Gelu3 { float x, p; };
__device__ __forceinline__ void gelu_a(Gelu3& s, float x) {
  s.x = x;
  asm volatile("" : "+v"(s.x));
  s.p = fmaf(-0.000524238159f, fabsf(s.x), 0.00741911121f);
  s.p = fmaf(s.p, fabsf(s.x), -0.0526018888f);
  s.p = fmaf(s.p, fabsf(s.x), -0.459225923f);
  asm volatile("" : "+v"(s.x), "+v"(s.p));
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

// ---- synchronisation.  Every wave of the workgroup passes the SAME sequence of barriers per row tile:
//      PSTEPS projection steps, 12 residual rounds, 12 hand-over rounds of norm2's output, MSTEPS main steps, one drain.
// X, start of a ring step: its 24 KB were requested AHEAD steps ago; of what this wave has issued since, only the pieces of the
// AHEAD - 2 youngest steps may still be in flight: the NEXT step has landed too (its first fragments are read behind this
// step's last MFMAs).  vmcnt counts every store and LDS-DMA piece of the wave, in order.  The barrier also says that
// everybody is done with the slot of the step before this one, which is refilled during this one.  WAITN < 0: no counted
// wait (the steps right behind a drain).  LGKM: this wave has LDS writes / reads that must be complete at the barrier.
template <int WAITN, bool LGKM = false>
__device__ __forceinline__ void x_wait() {
  static_assert(WAITN <= 63, "vmcnt");
  if constexpr (V_NO_BARRIER) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(V_NO_DMA || WAITN < 0 ? 0 : WAITN) : "memory");
  } else if constexpr (WAITN < 0) {
    if constexpr (LGKM) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_barrier" ::: "memory");
  } else if constexpr (LGKM) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(V_NO_DMA ? 0 : WAITN) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(V_NO_DMA ? 0 : WAITN) : "memory");
  }
}
// F (no LDS-DMA of its own: the X waves' waits + the barrier publish the ring), and both roles in the rounds without a ring step
template <bool LGKM = false>
__device__ __forceinline__ void sync_wait() {
  if constexpr (V_NO_BARRIER) return;
  if constexpr (LGKM) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_barrier" ::: "memory");
}

__device__ __forceinline__ int ring_next(const Ring& st) { return st.g + AHEAD < NSEQ ? st.g + AHEAD : st.g + AHEAD - NSEQ; }
__device__ __forceinline__ int ring_free(const Ring& st) { return st.slot == 0 ? NSLOT - 1 : st.slot - 1; }
__device__ __forceinline__ void ring_advance(Ring& st) {
  st.g = st.g + 1 == NSEQ ? 0 : st.g + 1;
  st.slot = st.slot + 1 == NSLOT ? 0 : st.slot + 1;
}
// piece i (0 .. 5) of this X wave for the step AHEAD, into the slot the barrier of this step has freed
__device__ __forceinline__ void ring_piece(const Ring& st, int i, int g_next, int slot_free) {
  if (V_NO_DMA) return;
  const int off = st.src0 + i * 1024;
  lds_dma16_keep(st.rsrc, st.dma_dst + slot_free * SB + off, (int)((threadIdx.x & 63) * 16), g_next * SB + off);
}
__device__ __forceinline__ void rotate_bases(const Ring& st, unsigned (&base)[4]) {
  const int d_ = st.slot == NSLOT - 1 ? -(NSLOT - 1) * SB : SB;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    base[i] += d_;
    asm volatile("" : "+v"(base[i]));      // (opaque: with as many slots as steps per loop trip hipcc would keep one address
  }                                        //  register per slot and fragment position, and spill them into the hot loop)
}

// what a step of NM MFMAs does behind MFMA j besides its own work: turn the fragment bases to the next step's slot NF MFMAs
// before the end, refill the fragment register the MFMA has just used, and (X) request its six pieces evenly spread
template <bool LAST, bool DMA, int NM = 12>
__device__ __forceinline__ void ring_gap(const Ring& st, unsigned (&base)[4], s16x8_t (&wf)[NF], int j, int g_next, int slot_free) {
  if (j == NM - NF && !LAST) rotate_bases(st, base);
  if (!V_NO_REFILL && !(LAST && j >= NM - NF)) wf[j % NF] = ld_frag(base, (j + NF) % NM);
  if (DMA && j % (NM / PIECES) == NM / PIECES - 1) ring_piece(st, j / (NM / PIECES), g_next, slot_free);
}

// X, a ring step without MFMAs of its own: barrier and the wave's six pieces
template <int WAITN, bool LGKM = false>
__device__ __forceinline__ void x_idle_step(Ring& st) {
  x_wait<WAITN, LGKM>();
  const int g_next = ring_next(st), slot_free = ring_free(st);
#pragma unroll
  for (int i = 0; i < PIECES; ++i) ring_piece(st, i, g_next, slot_free);
  ring_advance(st);
}

// F: one half of fc1(u) = 12 MFMAs, gacc (+)= W1(u)[k half KH] . h^T with the bias tile as initial value, and in their gaps
// the activation of values 8 KH .. 8 KH + 7 of the fc1 tile before it (gprev): four pairs, a third of both values per gap,
// packed pairwise into pk (= one B operand of fc2: the hidden order of W2 is packed to match).
template <int DT, int KH, bool GELU, bool LAST, bool LGKM>
__device__ __forceinline__ void f_step(Ring& st, unsigned (&base)[4], s16x8_t (&wf)[NF], const s16x8_t (&hf)[D / 16],
                                       f32x16_t& gacc, const f32x16_t& bias_c, const f32x16_t& gprev, u32x4_t& pk) {
  sync_wait<LGKM>();
  Gelu3 s0 = {}, s1 = {};
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    gacc = mfma32<DT>(wf[j % NF], hf[12 * KH + j], (KH == 0 && j == 0) ? bias_c : gacc);
    ring_gap<LAST, false>(st, base, wf, j, 0, 0);
    if constexpr (GELU) {
      const int r = 8 * KH + 2 * (j / 3);
      if (V_NO_GELU) {
        if (j % 3 == 2) pk[j / 3] = pack2_h16<DT>(gprev[r], gprev[r + 1]);
      } else if (j % 3 == 0) {
        gelu_a(s0, gprev[r]); gelu_a(s1, gprev[r + 1]);
      } else if (j % 3 == 1) {
        gelu_b(s0); gelu_b(s1);
      } else {
        const float v0 = gelu_c(s0), v1 = gelu_c(s1);
        pk[j / 3] = pack2_h16<DT>(v0, v1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  ring_advance(st);
}

// X: one half of fc2(u) = 12 MFMAs, xacc[6 OH + j / 2] += W2(u)[output tiles 6 OH ..][k step j & 1] . gf[j & 1]; gf = the
// activated fc1 tile of hidden unit u, read from the pair buffer at the start of the first half.
template <int DT, int OH, bool LAST>
__device__ __forceinline__ void x_step(Ring& st, unsigned (&base)[4], s16x8_t (&wf)[NF], f32x16_t (&xacc)[D / 32],
                                       s16x8_t (&gf)[2], unsigned gf_addr) {
  x_wait<WAIT0>();
  const int g_next = ring_next(st), slot_free = ring_free(st);
  if constexpr (OH == 0) {
    gf[0] = *(lds_frag_ptr)(gf_addr);
    gf[1] = *(lds_frag_ptr)(gf_addr + 1024);
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    xacc[6 * OH + (j >> 1)] = mfma32<DT>(wf[j % NF], gf[j & 1], xacc[6 * OH + (j >> 1)]);
    ring_gap<LAST, true>(st, base, wf, j, g_next, slot_free);
    __builtin_amdgcn_sched_barrier(0);
  }
  ring_advance(st);
}

// X: one projection step = 24 MFMAs, xacc[ot] += Wp[output tile ot][k steps 2 p, 2 p + 1] . a[2 p], a[2 p + 1]; the two
// fragments of the NEXT step are read from the pair buffer meanwhile (the F wave put them there at least one barrier ago).
// FIRST: the barrier has been passed and the first weight fragments are in flight (see the kernel).
template <int DT, int WAITN, bool FIRST, bool LAST>
__device__ __forceinline__ void p_step(Ring& st, unsigned (&base)[4], s16x8_t (&wf)[NF], f32x16_t (&xacc)[D / 32],
                                       const s16x8_t (&acur)[2], s16x8_t (&anext)[2], unsigned anext_addr0, unsigned anext_addr1) {
  if constexpr (!FIRST) x_wait<WAITN>();
  const int g_next = ring_next(st), slot_free = ring_free(st);
  if constexpr (!LAST) {
    anext[0] = *(lds_frag_ptr)(anext_addr0);
    anext[1] = *(lds_frag_ptr)(anext_addr1);
  }
#pragma unroll
  for (int j = 0; j < 24; ++j) {
    xacc[j % 12] = mfma32<DT>(wf[j % NF], acur[j / 12], xacc[j % 12]);
    ring_gap<LAST, true, 24>(st, base, wf, j, g_next, slot_free);
    __builtin_amdgcn_sched_barrier(0);
  }
  ring_advance(st);
}

// attention-output fragment of k step s in the pair buffer: column group s >> 2 alternates between the halves (the first one
// in the SECOND half: the first half is the X wave's staging during its epilogue), slot s & 3
__device__ __forceinline__ constexpr int af_off(int s) { return (((s >> 2) & 1) ? 0 : PBH) + (s & 3) * AF_SLOT; }

// What a role's code starts from.  The two roles are SEPARATE functions (not inlined): as two arms of one branch hipcc's register
// allocator let them interfere (the F arm, 228 registers on its own, spilled 96 beside the X arm).  Arguments arrive in
// VGPRs, so everything wave-uniform is made so again with readfirstlane (buffer descriptors must sit in SGPRs).
struct Ctx {
  const unsigned short* abuf; const unsigned short* wpk; float* x; unsigned short* hout; unsigned* tile_ctr;
  int64_t rows; float ln_eps; int ntiles; int tile; unsigned lds0;
};
__device__ __forceinline__ unsigned uni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T* uni_ptr(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  return reinterpret_cast<T*>(((uint64_t)uni((unsigned)(v >> 32)) << 32) | uni((unsigned)v));
}
#define FX_ROLE_ENV                                                                                                       \
  const unsigned short* const abuf = uni_ptr(c.abuf);                                                                     \
  float* const x = uni_ptr(c.x);                                                                                          \
  unsigned short* const hout = uni_ptr(c.hout);                                                                           \
  unsigned* const tile_ctr = uni_ptr(c.tile_ctr);                                                                         \
  const int64_t rows = (int64_t)(((uint64_t)uni((unsigned)((uint64_t)c.rows >> 32)) << 32) | uni((unsigned)c.rows));     \
  const float ln_eps = __uint_as_float(uni(__float_as_uint(c.ln_eps)));                                                   \
  const int ntiles = (int)uni((unsigned)c.ntiles);                                                                        \
  int tile = (int)uni((unsigned)c.tile);                                                                                  \
  const unsigned lds0 = uni(c.lds0);                                                                                      \
  const int tid = threadIdx.x;                                                                                            \
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                             \
  const int rb = wave & 3;                                                                                                \
  const int h = lane >> 5, l31 = lane & 31;                                                                               \
  unsigned cl = lds0 + CONST_OFF + 16 * h;                                                                                \
  asm volatile("" : "+v"(cl));                                                                                            \
  auto cst4 = [&](int i) { return *(lds_f4_ptr)(cl + 4 * i); };      /* floats i .. i + 3 (+ 4 h) of the constants */       \
  const unsigned nxt = lds0 + NEXT_OFF;                                                                                   \
  Ring st;                                                                                                                \
  st.rsrc = lds_dma_rsrc(uni_ptr(c.wpk), (unsigned)(NSEQ * SB));                                                          \
  st.src0 = rb * (PIECES * 1024);                                                                                         \
  st.dma_dst = lds0;                                                                                                      \
  st.g = 0;                                                                                                               \
  st.slot = 0;                                                                                                            \
  const int aoff0 = tile_off(l31, h);                                                                                     \
  const unsigned pbuf = lds0 + PB_OFF + rb * PB;      /* this pair's buffer: halves at + 0 and + PBH */                     \
  /* a tile's slice of a [rows][width bytes] array as a buffer descriptor: rows past the end read as zero / are not written */ \
  auto tile_rsrc = [&](const void* p, int64_t tile, int row_bytes) {                                                      \
    const int64_t first = tile * 128, left = rows - first;                                                                \
    const int nrows = tile < 0 || left <= 0 || !p ? 0 : left < 128 ? (int)left : 128;                                     \
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p)) + (nrows ? first : 0) * row_bytes, 0, \
                                             nrows * row_bytes, 0x00020000);                                              \
  };                                                                                                                      \
  [[maybe_unused]] int tile_no = -1;                                                                                      \
  [[maybe_unused]] unsigned long long stamp_[8] = {};                                                                     \
  (void)abuf; (void)x; (void)hout; (void)tile_ctr; (void)ln_eps; (void)ntiles; (void)l31; (void)aoff0; (void)nxt; (void)cst4; (void)tile_rsrc

// =============================================== F: the 16-bit rows ===============================================
template <int DT>
__device__ __attribute__((noinline)) void run_f(Ctx c) {
  FX_ROLE_ENV;
  // =============================================== F: the 16-bit rows ===============================================
  if (!V_NO_PRIO) asm volatile("s_setprio 3");     // its GELU pieces go in front of the X wave's MFMA waiting for the pipe
  unsigned pbl = pbuf + lane * 16;                 // lane-linear 16-byte slots of the pair buffer
  asm volatile("" : "+v"(pbl));
  // The next tile's rows, as loaded (eight lanes per row, 128-byte runs): fb[4 cg + i] = a[row 8 i + lane / 8][64 cg + 8 (lane % 8) ..]
  // (cg = 0 .. 5), fb[24 + 4 c + i] = x[row 8 i + lane / 8][32 c + 4 (lane % 8) ..] for the residual chunks c = 0 .. 5; chunks
  // 6 .. 11 take the registers of a's column groups as those are handed over.
  u32x4_t fb[48];
  const int q8 = lane >> 3, c8 = lane & 7;
  __amdgpu_buffer_rsrc_t ra, rx;
  const int ao = (rb * 32 + q8) * (D * 2) + c8 * 16, xo = (rb * 32 + q8) * (D * 4) + c8 * 16;
  auto load_a = [&](int cg) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[4 * cg + i] = __builtin_amdgcn_raw_buffer_load_b128(ra, ao + cg * 128, i * 8 * (D * 2), 0);
  };
  auto load_chunk = [&](int c) {
    const int r0 = c < 6 ? 24 + 4 * c : 4 * (c - 6);
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[r0 + i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo + c * 128, i * 8 * (D * 4), 0);
  };
  // column group cg of a = k steps 4 cg .. 4 cg + 3 as B fragments into the four slots of its half: this lane's 16 bytes are
  // k step 4 cg + (c8 >> 1), lane half c8 & 1, row 8 i + q8
  const unsigned af_wr = pbuf + (c8 >> 1) * AF_SLOT + (c8 & 1) * AF_H + q8 * 16;
  auto put_a = [&](int cg) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *(lds_wu4_ptr)(af_wr + af_off(4 * cg) + i * 128) = fb[4 * cg + i];
  };
  const unsigned ch_wr = pbuf + q8 * STG_ROW + c8 * 16;
  auto put_chunk = [&](int c) {      // residual chunk c into half c & 1: eight lanes per row as loaded (read back a row per lane)
    const int r0 = c < 6 ? 24 + 4 * c : 4 * (c - 6);
#pragma unroll
    for (int i = 0; i < 4; ++i) *(lds_wu4_ptr)(ch_wr + (c & 1) * PBH + i * 8 * STG_ROW) = fb[r0 + i];
  };
  auto fetch_rows = [&](int t) {     // everything the registers hold of tile t, and its first column group into the pair buffer
    ra = tile_rsrc(abuf, t, D * 2);
    rx = tile_rsrc(x, t, D * 4);
#pragma unroll
    for (int cg = 0; cg < 6; ++cg) load_a(cg);
#pragma unroll
    for (int c = 0; c < 6; ++c) load_chunk(c);
    put_a(0);
  };
  if constexpr (!V_MAIN_ONLY) fetch_rows(tile);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  while (true) {
    ++tile_no;
    FX_STAMP(0);
    unsigned next_v = 0;
    if (tid == 256) next_v = atomicAdd(tile_ctr, 1u);
    unsigned base[4];
    s16x8_t hf[D / 16];
    s16x8_t wf[NF];
    f32x16_t bias_c;
    auto load_bias = [&](int at) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4_t bv = cst4(at + 8 * q);
        bias_c[4 * q + 0] = bv[0]; bias_c[4 * q + 1] = bv[1]; bias_c[4 * q + 2] = bv[2]; bias_c[4 * q + 3] = bv[3];
      }
    };
    if constexpr (V_MAIN_ONLY) {      // (timing: any data will do)
      const auto rs = tile_rsrc(abuf, tile, D * 2);
#pragma unroll
      for (int s = 0; s < D / 16; ++s) hf[s] = __builtin_bit_cast(s16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (rb * 32 + l31) * (D * 2) + 16 * h, 32 * s, 0));
    }
    if constexpr (!V_MAIN_ONLY) {
      // ---- projection steps (k steps 2 p, 2 p + 1): the X wave computes; column group cg of a goes over during step 2 cg - 2
      //      (read from step 2 cg - 1 on; its slots were read last during step 2 cg - 4), and the registers it leaves take the
      //      next residual chunk
#pragma unroll
      for (int p = 0; p < PSTEPS; ++p) {
        sync_wait<true>();
        if (p % 2 == 0 && p / 2 + 1 < 6) put_a(p / 2 + 1);
        if (p == 0) load_chunk(6);
        if (p % 2 == 1 && p / 2 < 5) load_chunk(7 + p / 2);
        if (p == PSTEPS - 1) put_chunk(0);
        ring_advance(st);
      }
      FX_STAMP(1);
      // ---- residual rounds: chunk r is in half r & 1 (written during the round before), chunk r + 1 goes into the other half
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        sync_wait<true>();
        if (r + 1 < 12) put_chunk(r + 1);
      }
      // ---- norm2's output, two k steps per round from half r & 1
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        sync_wait<true>();
        hf[2 * r] = *(lds_frag_ptr)(pbl + (r & 1) * PBH);
        hf[2 * r + 1] = *(lds_frag_ptr)(pbl + (r & 1) * PBH + 1024);
      }
    }
    FX_STAMP(2);
    // ---- main steps
    f32x16_t ga, gb = {};
    u32x4_t pk0 = {}, pk1 = {};
#pragma unroll
    for (int i = 0; i < 4; ++i) base[i] = lds0 + st.slot * SB + (aoff0 ^ (32 * i));      // (the step the stream stands at)
    auto put_gf = [&](int half) {      // the packed tile into a half of the pair buffer (read by X behind the next barrier)
      *(lds_wu4_ptr)(pbl + half * PBH) = pk0;
      *(lds_wu4_ptr)(pbl + half * PBH + 1024) = pk1;
    };
    load_bias(C_B1);
    sync_wait<true>();                 // main step 0 (its barrier here: the fragment reads below follow it)
#pragma unroll
    for (int f = 0; f < NF; ++f) wf[f] = ld_frag(base, f);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      ga = mfma32<DT>(wf[j % NF], hf[j], j == 0 ? bias_c : ga);
      ring_gap<false, false>(st, base, wf, j, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    ring_advance(st);
    f_step<DT, 1, false, false, false>(st, base, wf, hf, ga, bias_c, gb, pk1);          // fc1(0), second half
    for (int u = 1; u < UNITS - 1; u += 2) {
      load_bias(C_B1 + 32 * u);
      f_step<DT, 0, true, false, true>(st, base, wf, hf, gb, bias_c, ga, pk0);          // fc1(u) | gelu(u - 1) -> half 0
      f_step<DT, 1, true, false, false>(st, base, wf, hf, gb, bias_c, ga, pk1);
      put_gf(0);
      load_bias(C_B1 + 32 * (u + 1));
      f_step<DT, 0, true, false, true>(st, base, wf, hf, ga, bias_c, gb, pk0);          // fc1(u + 1) | gelu(u) -> half 1
      f_step<DT, 1, true, false, false>(st, base, wf, hf, ga, bias_c, gb, pk1);
      put_gf(1);
    }
    FX_STAMP(3);
    load_bias(C_B1 + 32 * (UNITS - 1));
    f_step<DT, 0, true, false, true>(st, base, wf, hf, gb, bias_c, ga, pk0);            // fc1(47) | gelu(46) -> half 0
    f_step<DT, 1, true, true, false>(st, base, wf, hf, gb, bias_c, ga, pk1);
    put_gf(0);
    // main step 96: the last tile's activation on its own -> half 1; 97 .. 99: X's last units
    sync_wait<true>();
    ring_advance(st);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float v0 = V_NO_GELU ? gb[r] : gelu_poly(gb[r]), v1 = V_NO_GELU ? gb[r + 1] : gelu_poly(gb[r + 1]);
      const unsigned w = pack2_h16<DT>(v0, v1);
      if (r < 8) pk0[r >> 1] = w; else pk1[(r - 8) >> 1] = w;
    }
    put_gf(1);
    sync_wait<true>(); ring_advance(st);
    sync_wait(); ring_advance(st);
    sync_wait(); ring_advance(st);
    FX_STAMP(4);
    // ---- drain; the next tile's rows go on their way while the X wave runs its epilogue
    if (tid == 256) *(lds_u32_ptr)nxt = next_v;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int next = __builtin_amdgcn_readfirstlane((int)*(lds_u32_ptr)nxt);
    FX_STAMP(5); FX_STAMP(6); FX_STAMP(7);
    if (next >= ntiles) break;
    tile = next;
    if constexpr (!V_MAIN_ONLY) fetch_rows(tile);
  }
}

// =============================================== X: the fp32 rows ===============================================
template <int DT>
__device__ __attribute__((noinline)) void run_x(Ctx c) {
  FX_ROLE_ENV;
  // =============================================== X: the fp32 rows ===============================================
  unsigned base[4];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  s16x8_t wf[NF];
  f32x16_t xacc[D / 32];
  if constexpr (V_MAIN_ONLY) {
#pragma unroll
    for (int ot = 0; ot < D / 32; ++ot) xacc[ot] = f32x16_t{};
  }
  while (true) {
    ++tile_no;
    FX_STAMP(0);
    int ln = lane;             // (opaque: addresses are recomputed per phase, a few VALU, not carried -- spilled -- across the steps)
    asm volatile("" : "+v"(ln));
    if constexpr (!V_MAIN_ONLY) {
      // ---- x' = a . Wp^T + bp (+ x below): every accumulator starts from its bias tile (register r of lane half h = constant
      //      at + (r & 3) + 8 (r >> 2) + 4 h), one k step of all 12 output tiles per step
#pragma unroll
      for (int ot = 0; ot < D / 32; ++ot)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4_t bv = cst4(C_BP + 32 * ot + 8 * q);
          xacc[ot][4 * q + 0] = bv[0]; xacc[ot][4 * q + 1] = bv[1]; xacc[ot][4 * q + 2] = bv[2]; xacc[ot][4 * q + 3] = bv[3];
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) base[i] = lds0 + st.slot * SB + (aoff0 ^ (32 * i));
      const unsigned af_rd = pbuf + (ln >> 5) * AF_H + (ln & 31) * 16;
      s16x8_t aa[2], ab[2];
      // step 0: its barrier first (the F wave's first column group is visible behind it), then the fragments
      x_wait<-1, true>();
#pragma unroll
      for (int f = 0; f < NF; ++f) wf[f] = ld_frag(base, f);
      aa[0] = *(lds_frag_ptr)(af_rd + af_off(0));
      aa[1] = *(lds_frag_ptr)(af_rd + af_off(1));
      p_step<DT, -1, true, false>(st, base, wf, xacc, aa, ab, af_rd + af_off(2), af_rd + af_off(3));
      p_step<DT, -1, false, false>(st, base, wf, xacc, ab, aa, af_rd + af_off(4), af_rd + af_off(5));
#pragma unroll
      for (int p = 2; p < PSTEPS - 2; p += 2) {
        p_step<DT, WAIT0, false, false>(st, base, wf, xacc, aa, ab, af_rd + af_off(2 * p + 2), af_rd + af_off(2 * p + 3));
        p_step<DT, WAIT0, false, false>(st, base, wf, xacc, ab, aa, af_rd + af_off(2 * p + 4), af_rd + af_off(2 * p + 5));
      }
      p_step<DT, WAIT0, false, false>(st, base, wf, xacc, aa, ab, af_rd + af_off(2 * PSTEPS - 2), af_rd + af_off(2 * PSTEPS - 1));
      p_step<DT, WAIT0, false, true>(st, base, wf, xacc, ab, aa, 0, 0);
      FX_STAMP(1);
      // ---- + x: residual chunk r (32 columns) from half r & 1 of the pair buffer, a row per lane
      const unsigned ch_rd = pbuf + (ln & 31) * STG_ROW + 16 * (ln >> 5);
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        sync_wait<true>();
        f32x4_t xv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xv[g] = *(lds_f4_ptr)(ch_rd + (r & 1) * PBH + 32 * g);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) xacc[r][4 * g + e] += xv[g][e];
      }
      // ---- norm2 of x' in registers -> the 24 B operands of fc1 (k step s, element e of lane half h = column
      //      32 (s >> 1) + 16 (s & 1) + 8 (e >> 2) + 4 h + (e & 3): the host packs W1's input dim in that order), handed to the
      //      F wave two k steps per round through half r & 1
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
      float mean_p = mean;
      asm volatile("" : "+v"(mean_p));
      const unsigned hw_ = pbuf + ln * 16;
      f32x4_t gq[2][4], bq[2][4];       // gamma / beta of an output tile's columns, read one tile ahead of their use
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
            const f32x4_t gg = gq[ot & 1][g], bb = bq[ot & 1][g];
            pk[2 * g2_ + 0] = pack2_h16<DT>((xacc[ot][4 * g + 0] - mean_p) * rstd * gg[0] + bb[0], (xacc[ot][4 * g + 1] - mean_p) * rstd * gg[1] + bb[1]);
            pk[2 * g2_ + 1] = pack2_h16<DT>((xacc[ot][4 * g + 2] - mean_p) * rstd * gg[2] + bb[2], (xacc[ot][4 * g + 3] - mean_p) * rstd * gg[3] + bb[3]);
          }
          *(lds_wu4_ptr)(hw_ + (ot & 1) * PBH + k2 * 1024) = pk;
        }
        sync_wait<true>();
      }
    }
    FX_STAMP(2);
    // ---- main steps: 0 .. 3 idle (the F wave is two hidden units ahead)
    s16x8_t gf[2];
    x_idle_step<WAIT0>(st); x_idle_step<WAIT0>(st); x_idle_step<WAIT0>(st);
    x_wait<WAIT0>();          // (main step 3: the fragments of step 4 are read behind its barrier)
    {
      const int g_next = ring_next(st), slot_free = ring_free(st);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) ring_piece(st, i, g_next, slot_free);
      ring_advance(st);
#pragma unroll
      for (int i = 0; i < 4; ++i) base[i] = lds0 + st.slot * SB + HB + (aoff0 ^ (32 * i));
#pragma unroll
      for (int f = 0; f < NF; ++f) wf[f] = ld_frag(base, f);
    }
    const unsigned gfa = pbuf + ln * 16;
    for (int u = 0; u < UNITS - 2; u += 2) {
      x_step<DT, 0, false>(st, base, wf, xacc, gf, gfa);               // fc2(u): half 0 of the pair buffer
      x_step<DT, 1, false>(st, base, wf, xacc, gf, gfa);
      x_step<DT, 0, false>(st, base, wf, xacc, gf, gfa + PBH);         // fc2(u + 1)
      x_step<DT, 1, false>(st, base, wf, xacc, gf, gfa + PBH);
    }
    FX_STAMP(3);
    x_step<DT, 0, false>(st, base, wf, xacc, gf, gfa);                 // fc2(46)
    x_step<DT, 1, false>(st, base, wf, xacc, gf, gfa);
    x_step<DT, 0, false>(st, base, wf, xacc, gf, gfa + PBH);           // fc2(47)
    x_step<DT, 1, true>(st, base, wf, xacc, gf, gfa + PBH);
    FX_STAMP(4);
    // ---- drain: everything this wave has requested has landed (the first AHEAD steps of the next tile: its first steps wait
    //      for nothing but their barriers, which gives the stores below until then), and the next tile is known
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int next = __builtin_amdgcn_readfirstlane((int)*(lds_u32_ptr)nxt);
    if constexpr (!V_MAIN_ONLY) {
      // ---- epilogue: x[row][col .. col + 3] = acc + b2, lane owns row m, columns 32 ot + 8 g + 4 h + {0 .. 3}, through the
      //      first half of the pair buffer (written a row per lane, read back eight lanes per row, stored in 128-byte runs);
      //      then the LayerNorm of the new row from the same registers, h out the same way
      const auto rxo = tile_rsrc(x, tile, D * 4);
      const auto rho = tile_rsrc(hout, tile, D * 2);
      int le = lane;
      asm volatile("" : "+v"(le));
      const unsigned stg_wr_e = pbuf + (le & 31) * STG_ROW + 16 * (le >> 5);
      const unsigned stg_wh = pbuf + (le & 31) * STG_ROW + 8 * (le >> 5);      // (16-bit rows: + 64 o2 + 16 g)
      const unsigned stg_rd_e = pbuf + (le >> 3) * STG_ROW + (le & 7) * 16;
      const int xo_e = (rb * 32 + (le >> 3)) * (D * 4) + (le & 7) * 16;
      const int ho = (rb * 32 + (le >> 3)) * (D * 2) + (le & 7) * 16;         // h: rows 8 i + lane / 8, columns 64 op + 8 (lane % 8) ..
      float s = 0.f;
      {
        // The constants of an output tile are requested one tile ahead and the read-back of the staging rows as ONE group: left to
        // itself hipcc issues every LDS read with its own lgkmcnt(0) wait in front of the one instruction that uses it (8 exposed
        // round trips per output tile).  The fences only delimit the groups.
        f32x4_t bvq[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bvq[0][g] = cst4(C_B2 + 8 * g);
#pragma unroll
        for (int ot = 0; ot < D / 32; ++ot) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4_t bv = bvq[ot & 1][g];
            f32x4_t v;
            v[0] = xacc[ot][4 * g + 0] + bv[0]; v[1] = xacc[ot][4 * g + 1] + bv[1];
            v[2] = xacc[ot][4 * g + 2] + bv[2]; v[3] = xacc[ot][4 * g + 3] + bv[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) xacc[ot][4 * g + e] = v[e];
            *(lds_w4_ptr)(stg_wr_e + 32 * g) = v;
            s += (v[0] + v[1]) + (v[2] + v[3]);
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x4_t rbk[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) rbk[i] = *(lds_f4_ptr)(stg_rd_e + i * 8 * STG_ROW);
          if (ot + 1 < D / 32) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bvq[(ot + 1) & 1][g] = cst4(C_B2 + 32 * (ot + 1) + 8 * g);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, rbk[i]), rxo, xo_e + ot * 128, i * 8 * (D * 4), 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      FX_STAMP(5);
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
        FX_STAMP(6);
        const float rstd = 1.0f / sqrtf(q / (float)D + ln_eps);
        float mean_p = mean;
        asm volatile("" : "+v"(mean_p));
        // gamma / beta of an output tile one tile ahead, the read-back as one group (as above)
        f32x4_t gq1[2][4], bq1[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { gq1[0][g] = cst4(C_G1 + 8 * g); bq1[0][g] = cst4(C_E1 + 8 * g); }
#pragma unroll
        for (int ot = 0; ot < D / 32; ++ot) {            // two output tiles = 64 columns = 128 bytes of a 16-bit row
          const int op = ot >> 1, o2 = ot & 1;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4_t gg = gq1[ot & 1][g], bb = bq1[ot & 1][g];
            u32x2_t pk;
            pk[0] = pack2_h16<DT>((xacc[ot][4 * g + 0] - mean_p) * rstd * gg[0] + bb[0], (xacc[ot][4 * g + 1] - mean_p) * rstd * gg[1] + bb[1]);
            pk[1] = pack2_h16<DT>((xacc[ot][4 * g + 2] - mean_p) * rstd * gg[2] + bb[2], (xacc[ot][4 * g + 3] - mean_p) * rstd * gg[3] + bb[3]);
            *(lds_w2_ptr)(stg_wh + 64 * o2 + 16 * g) = pk;
          }
          __builtin_amdgcn_sched_barrier(0);
          [[maybe_unused]] f32x4_t rbk[4];
          if (o2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rbk[i] = *(lds_f4_ptr)(stg_rd_e + i * 8 * STG_ROW);
          }
          if (ot + 1 < D / 32) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { gq1[(ot + 1) & 1][g] = cst4(C_G1 + 32 * (ot + 1) + 8 * g); bq1[(ot + 1) & 1][g] = cst4(C_E1 + 32 * (ot + 1) + 8 * g); }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (o2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, rbk[i]), rho, ho + op * 128, i * 8 * (D * 2), 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    FX_STAMP(7);
    if (next >= ntiles) break;
    tile = next;
  }
  if (V_MAIN_ONLY && x) {      // (timing build: keep the accumulators alive)
    float s = 0.f;
#pragma unroll
    for (int ot = 0; ot < D / 32; ++ot)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += xacc[ot][r];
    if (s == 12345.678f) x[tid] = s;
  }
}

template <int DT>
__global__ __launch_bounds__(512, 1) void tail_fx_kernel(const unsigned short* __restrict__ abuf, const unsigned short* __restrict__ wpk,
                                                         const float* __restrict__ bp, const float* __restrict__ g2, const float* __restrict__ e2,
                                                         const float* __restrict__ b1, const float* __restrict__ b2,
                                                         float* __restrict__ x, int64_t rows, const float* __restrict__ ln_g,
                                                         const float* __restrict__ ln_b, float ln_eps,
                                                         unsigned short* __restrict__ hout, int ntiles,
                                                         unsigned* __restrict__ tile_ctr) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const cst = reinterpret_cast<float*>(smem + CONST_OFF);
  for (int i = tid; i < C_N; i += 512) {
    float v;
    if (i < C_B2) v = b1[i];
    else if (i < C_G1) v = b2[i - C_B2];
    else if (i < C_E1) v = ln_g ? ln_g[i - C_G1] : 1.f;
    else if (i < C_BP) v = ln_b ? ln_b[i - C_E1] : 0.f;
    else if (i < C_G2) v = bp[i - C_BP];
    else if (i < C_E2) v = g2[i - C_G2];
    else v = e2[i - C_E2];
    cst[i] = v;
  }
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
  const unsigned nxt = lds0 + NEXT_OFF;
  if (tid == 0) *(lds_u32_ptr)nxt = atomicAdd(tile_ctr, 1u);
  __syncthreads();
  const int tile = __builtin_amdgcn_readfirstlane((int)*(lds_u32_ptr)nxt);
  if (tile >= ntiles) return;                  // (nothing requested yet)
  if (wave < 4) {                              // the first AHEAD steps of the weight stream (X waves: six pieces per step each)
    const i32x4_t rsrc = lds_dma_rsrc(wpk, (unsigned)(NSEQ * SB));
#pragma unroll
    for (int u = 0; u < AHEAD; ++u)
#pragma unroll
      for (int i = 0; i < PIECES; ++i)
        lds_dma16(rsrc, lds0 + wave * (PIECES * 1024) + u * SB + i * 1024, lane * 16, u * SB + wave * (PIECES * 1024) + i * 1024);
  }
#if FX_VARIANT & 16
  if (blockIdx.x == 0 && lane == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    g_fx_hwid[wave] = hw;
  }
#endif
  const Ctx c = {abuf, wpk, x, hout, tile_ctr, rows, ln_eps, ntiles, tile, lds0};
  if (wave >= 4) run_f<DT>(c);
  else run_x<DT>(c);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the steps requested beyond the last one: land before the LDS goes away
}

}  // namespace

#if FX_VARIANT & 16
extern "C" int vittf_fx_stamps(unsigned long long* out, unsigned* hwid) {
  if (hipMemcpyFromSymbol(hwid, HIP_SYMBOL(g_fx_hwid), sizeof(g_fx_hwid)) != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fx_stamps), sizeof(g_fx_stamps)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef FX_STANDALONE      // tools/fx_variants.sh builds this file alone
void vittf_note_kernel(int, const char*) {}
#endif

extern "C" size_t vittf_block_tail_workspace_bytes(void) { return sizeof(unsigned); }

extern "C" int vittf_block_tail(const void* attn_out, const void* w_packed, const float* proj_b, const float* ln2_g,
                                const float* ln2_b, const float* b1, const float* b2, float* x, int64_t rows, int32_t d,
                                int32_t dtype, const float* ln_g, const float* ln_b, float ln_eps, void* h_out, void* tile_counter,
                                void* stream) {
  if (!attn_out || !w_packed || !proj_b || !ln2_g || !ln2_b || !b1 || !b2 || !x || rows <= 0) return VITTF_ERR_INVALID_ARG;
  if (d != D) return VITTF_ERR_INVALID_ARG;          // the register budget is sized for ViT-S
  if ((ln_g || ln_b || h_out) && !(ln_g && ln_b && h_out)) return VITTF_ERR_INVALID_ARG;
  const int64_t tiles = (rows + 127) / 128;
  if (tiles > 0x7fffffff) return VITTF_ERR_INVALID_ARG;
  // 16-byte accesses everywhere (LDS-DMA pieces, buffer loads / stores of whole 128-byte runs)
  if ((((uintptr_t)attn_out | (uintptr_t)w_packed | (uintptr_t)x | (uintptr_t)h_out) & 15) != 0) return VITTF_ERR_INVALID_ARG;
  if (!tile_counter || ((uintptr_t)tile_counter & 3) != 0) return VITTF_ERR_INVALID_ARG;
  // one persistent workgroup per CU of the device this call runs on (asked per call: no state is kept between calls)
  const int cus = vittf_current_cus();
  if (cus <= 0) return VITTF_ERR_NO_DEVICE;
  unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
#ifdef FX_STANDALONE      // (timing builds: the same tiles per workgroup on fewer CUs tell a bandwidth bound from a latency chain)
  if (const char* e = getenv("VITTF_FX_GRID")) grid = (unsigned)atoi(e) < grid ? (unsigned)atoi(e) : grid;
#endif
  hipStream_t st = (hipStream_t)stream;
  // the tile counter is the caller's memory (launches on one stream are serialised; two streams bring two counters)
  unsigned* ctr = (unsigned*)tile_counter;
  if (hipMemsetAsync(ctr, 0, sizeof(unsigned), st) != hipSuccess) return VITTF_ERR_LAUNCH;
#define FX_LAUNCH(DTV)                                                                                               \
  hipLaunchKernelGGL((tail_fx_kernel<DTV>), dim3(grid), dim3(512), 0, st, (const unsigned short*)attn_out,           \
                     (const unsigned short*)w_packed, proj_b, ln2_g, ln2_b, b1, b2, x, rows, ln_g, ln_b, ln_eps,     \
                     (unsigned short*)h_out, (int)tiles, ctr)
  if (dtype == VITTF_BF16) FX_LAUNCH(VITTF_BF16);
  else if (dtype == VITTF_FP16) FX_LAUNCH(VITTF_FP16);
  else return VITTF_ERR_INVALID_ARG;
#undef FX_LAUNCH
  vittf_note_kernel(VITTF_KERNEL_MLP, "tail_fx_kernel");
  return vittf_check_launch();
}
