// Shared device helpers for the gfx950 (CDNA4) kernels of libvittf.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vittf.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;    // one MFMA A/B fragment: 8 x 16-bit = 4 VGPRs
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- 16-bit conversions (round to nearest even; plain casts lower to v_cvt_pk_bf16_f32 / v_cvt_f16_f32) ----
template <int DT> __device__ __forceinline__ unsigned short f32_to_h16(float x) {
  if constexpr (DT == VITTF_BF16) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(unsigned short, b);
  } else {
    _Float16 h = (_Float16)x;
    return __builtin_bit_cast(unsigned short, h);
  }
}
template <int DT> __device__ __forceinline__ float h16_to_f32(unsigned short u) {
  if constexpr (DT == VITTF_BF16) {
    return __uint_as_float(((unsigned)u) << 16);
  } else {
    return (float)__builtin_bit_cast(_Float16, u);
  }
}
__device__ __forceinline__ unsigned short f32_to_f16bits(float x) { return f32_to_h16<VITTF_FP16>(x); }
__device__ __forceinline__ float f16bits_to_f32(unsigned short u) { return h16_to_f32<VITTF_FP16>(u); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// two fp32 -> one dword of two 16-bit values, ONE v_cvt_pk_{bf16,f16}_f32 (round to nearest even)
template <int DT> __device__ __forceinline__ unsigned pack2_h16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  if constexpr (DT == VITTF_BF16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}

// ---- MFMA 32x32x16, fp32 accumulate.  Lane l: A[row l&31][k 8(l>>5)+j], B[k 8(l>>5)+j][col l&31];
//      C/D: col = l&31, row = (r&3) + 8(r>>2) + 4(l>>5) for register r of 16. ----
template <int DT> __device__ __forceinline__ f32x16_t mfma32(s16x8_t a, s16x8_t b, f32x16_t c) {
  if constexpr (DT == VITTF_BF16) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
}
// row of accumulator register r (0..15) inside the 32x32 tile, for lane half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- LDS image of a [rows][64 x 16-bit] operand tile (128-byte rows) read with ds_read_b128 ----
// Two rows share one 256-byte bank row; the 16 chunk slots of a row pair are XORed with the pair index so
// that the 16-lane groups of ds_read_b128 (rows 0-3,12-15,20-27 / ...) hit 16 distinct 16-byte slots.
// Chunk c (0..7, 8 elements each) of row r lives at byte offset tile_off(r, c).
__device__ __forceinline__ int tile_off(int r, int c) {
  const int p = r >> 1;
  const int slot = (((r & 1) << 3) | c) ^ (p & 15);
  return ((p << 4) | slot) << 4;
}
// inverse: linear 16-byte position q of the image -> (row, chunk); used to pre-swizzle the SOURCE address
// of global_load_lds, whose LDS destination is always lane-linear.
__device__ __forceinline__ void tile_pos(int q, int& r, int& c) {
  const int p = q >> 4;
  const int slot = (q & 15) ^ (p & 15);
  r = (p << 1) | (slot >> 3);
  c = slot & 7;
}

// XCD-aware, bijective block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each of
// the 8 residue classes a contiguous range of logical work items (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// exact-erf GELU  x Phi(x) = x - 0.5 x erfc(|x| / sqrt 2)  (x >= 0),  0.5 x erfc(|x| / sqrt 2)  (x < 0), with
// erfc(z) = 2^-Q(z), Q a degree-5 polynomial without constant term (weighted minimax fit on [0, 4.2], monotone
// beyond): |gelu error| <= 1.4e-6 absolute over all x (fit + check: tools/fit_gelu.py), far below the 16-bit
// rounding of the result.  One v_exp_f32 and no reciprocal: 40 issue cycles per wave against 68 for the
// Abramowitz-Stegun 7.1.26 form (the epilogue of fc1 is VALU-bound: 201 M activations per launch).
__device__ __forceinline__ float gelu_poly(float x) {
  // Q in a = |x| itself: the 1 / sqrt 2 of z = a / sqrt 2 is folded into the coefficients (c_k 2^(-k/2), rounded once from the
  // z form's: -1.6278975, -0.918451846, -0.148780614, 0.0296764448, -0.002965539) -- one multiply less per value, the same
  // 1.26e-6 maximum error, within 2.4e-7 of the z form everywhere (7e-5 of all inputs round to another fp16 value)
  float p = fmaf(-0.000524238159f, fabsf(x), 0.00741911121f);
  p = fmaf(p, fabsf(x), -0.0526018888f);
  p = fmaf(p, fabsf(x), -0.459225923f);
  p = fmaf(p, fabsf(x), -1.15109742f);
  const float e = __builtin_amdgcn_exp2f(fmaf(p, fabsf(x), -1.0f));   // 0.5 erfc(|x| / sqrt 2)
  return fmaf(-fabsf(x), e, fmaxf(x, 0.f));                    // max(x, 0) - |x| 0.5 erfc(|x| / sqrt 2): both signs, 8 VALU
}

// sqrtf for NORMAL positive x, correctly rounded: v_sqrt_f32 (1 ulp) + the library's own correction step (try the two
// neighbours, keep the one whose square brackets x) without its scaling of denormal inputs -- the same bits as sqrtf for
// every x the threshold lets through, a third of the instructions and none of the per-element condition masks that made
// 64 inlined sqrtf calls spill 140-190 registers at this kernel's 128-VGPR budget.
__device__ __forceinline__ float sqrt_cr_normal(float x) {
  float y = __builtin_amdgcn_sqrtf(x);
  const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
  const float rm = fmaf(-ym, y, x), rp = fmaf(-yp, y, x);
  y = rm <= 0.f ? ym : y;
  y = rp > 0.f ? yp : y;
  return y;
}

// Running maximum of non-negative floats kept as their bit pattern.  Workgroups that finish together would queue their
// atomics on the one address (the L2 serialises them at ~9 ns each: 1024-2048 of them were 7-18 us of a 50 us similarity
// launch); most of them carry a value the slot already exceeds, and a relaxed L2 read tells without queueing -- a stale
// read only costs the atomic it would have saved.
__device__ __forceinline__ void atomic_max_nonneg(unsigned* slot, float m) {
  const unsigned bits = __float_as_uint(m);
  if (m > 0.f && bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
}

// ---- LDS-DMA hidden from hipcc -------------------------------------------------------------------------------
// One piece: 64 lanes x 16 B -> 1 KB at LDS byte address lds_addr (lane-linear), fetched through a buffer descriptor
// (bytes past num_records read as zeros).  hipcc orders every later ds_read / ds_write against an outstanding
// LDS-DMA it knows of (the builtins) with s_waitcnt vmcnt(0) -- it cannot tell which LDS bytes the DMA writes --
// which drains the prefetch (and, vmcnt retiring in issue order, every store before it).  Written as asm, the
// piece is invisible to that bookkeeping: the kernel waits for it itself (counted vmcnt, then a barrier).
// rsrc, lds_addr and soff must be wave-uniform and SALU-computed (no VALU->SGPR hazard handling inside asm).
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
__device__ __forceinline__ i32x4_t lds_dma_rsrc(const void* base_ptr, unsigned bytes) {
  const uint64_t base = reinterpret_cast<uint64_t>(base_ptr);
  i32x4_t r;
  r[0] = (int)(unsigned)base;
  r[1] = (int)((unsigned)(base >> 32) & 0xffffu);
  r[2] = (int)bytes;
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void lds_dma16(i32x4_t rsrc, unsigned lds_addr, int voff, int soff) {
  unsigned keep;   // M0 is saved and restored: hipcc does not accept it as a clobber
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// the same with a per-lane 64-bit global address instead of a buffer descriptor
__device__ __forceinline__ void lds_dma16_flat(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// ---- optional per-kernel-class timing with HIP events on the launch stream (engine.hip; bench.py's roofline legs) ----
// Not part of the C ABI (C++ linkage).  begin returns a token (or null when the class is not being recorded).
void* vittf_prof_begin(int cls, void* stream);
void vittf_prof_end(void* token, void* stream);
struct ProfScope {
  void* tok; void* st;
  ProfScope(int cls, void* stream) : tok(vittf_prof_begin(cls, stream)), st(stream) {}
  ~ProfScope() { if (tok) vittf_prof_end(tok, st); }
};

// which kernel a dispatcher launched for a profiler class (bench.py asks instead of re-deriving the dispatch rules)
void vittf_note_kernel(int cls, const char* name);

// Compute units of the device the CALLING thread has current -- asked on every call, never latched: a process that drives
// two GPUs (HipViT takes a device) must size each persistent grid for the device it launches on.  <= 0: no device.
static inline int vittf_current_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  return hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess ? n : 0;
}

static inline int vittf_check_launch() {
  return hipGetLastError() == hipSuccess ? VITTF_OK : VITTF_ERR_LAUNCH;
}
