// Ceiling micro-benchmark for the attention kernel's inner loop on gfx950 (tools only, not part of libvittf).
//
// A wave issues v_mfma_f32_32x32x16_f16 from registers (random operands: zeros inflate the clock), optionally with the
// softmax's VALU mix between consecutive MFMAs (two v_exp_f32, two v_add_f32, one v_cvt_pk per MFMA at head dim 64), and
// stamps the shader clock (s_memtime) and the 100 MHz counter (s_memrealtime) around the loop.  Reported per variant:
// cycles per MFMA and wave, the clock the chip holds, and the resulting TFLOP/s of the whole chip -- i.e. what
// "100 % matrix-pipe busy" and "the VALU-bound limit" mean on this device under this power state.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_ceiling tools/micro/mfma_ceiling.hip && tools/micro/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// NACC independent accumulators, FILL = 0: bare MFMAs, 1: + the softmax VALU mix per MFMA (on values that do not
// depend on the MFMA results of this iteration), DEP = 1: a dependent chain on one accumulator
// LDSOP = 1: the A operand of every MFMA comes from LDS (ds_read_b128 of a conflict-free image, issued one MFMA ahead into
// the other of two register sets) instead of staying in registers; LDSOP = 2: additionally two ds_read_b64_tr_b16 per second
// MFMA (the V^T fragments of the attention kernel: 12 LDS instructions per 8 MFMAs in all)
template <int NACC, int FILL, int MFMA = 1, int LDSOP = 0>
__global__ __launch_bounds__(256) void mfma_loop(const _Float16* __restrict__ src, float* __restrict__ out,
                                                 unsigned long long* __restrict__ stamps, int iters) {
  const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
  f16x8_t a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const f16x8_t*>(src + ((gid * 8 + i) % 4096) * 8);
    b[i] = *reinterpret_cast<const f16x8_t*>(src + ((gid * 8 + 4 + i) % 4096) * 8);
  }
  f32x16_t acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float x[16], s0 = 0.f, s1 = 0.f, eA0 = 0.f, eA1 = 0.f, eB0 = 0.f, eB1 = 0.f;
  unsigned pk = 0, pk2 = 0, pkprev = 0xb800b400u;
  __attribute__((ext_vector_type(4))) float lsum0 = {0.f, 0.f, 0.f, 0.f}, lsum1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = -0.01f * (float)((tid + r) & 63);
  __shared__ __attribute__((aligned(16))) char lds[32768];
  typedef __attribute__((ext_vector_type(4))) short s4;
  s4 trsink = {0, 0, 0, 0};
  if constexpr (LDSOP) {
    for (int i = tid; i < 2048; i += 256) reinterpret_cast<f16x8_t*>(lds)[i] = a[i & 3];
    __syncthreads();
  }
  const char* lbase = lds + (tid & 63) * 16 + (tid >> 6) * 4096;     // lane-linear 16-byte reads: conflict-free
  f16x8_t la[2];
  if constexpr (LDSOP) { la[0] = *reinterpret_cast<const f16x8_t*>(lbase); la[1] = la[0]; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if constexpr (LDSOP) {
        if (LDSOP != 3 || (m & 1))   // LDSOP = 3: half the fragment reads (one per two MFMAs + two transposing per four): 64 query rows per wave
        la[(m + 1) & 1] = *reinterpret_cast<const f16x8_t*>(lbase + 1024 * ((m + 1) & 3));      // operand of the NEXT MFMA
        if constexpr (LDSOP >= 2) {      // transposing reads at an 8-byte lane stride: conflict-free (a 16-byte stride is 2-way)
          const char* tbase = lds + (tid & 63) * 8 + (tid >> 6) * 4096;
          if ((LDSOP == 2 && (m & 1)) || (LDSOP == 3 && (m & 3) == 3)) {
            const s4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(tbase + 2048));
            const s4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(tbase + 2560));
            trsink ^= t0 ^ t1;
          }
        }
        if constexpr (MFMA) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(la[m & 1], b[m & 3], acc[m % NACC], 0, 0, 0);
      } else if constexpr (MFMA) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[m & 3], acc[m % NACC], 0, 0, 0);
      // exact instruction mixes in asm (nothing for the compiler to hoist, fuse or chain): the adds and the conversion
      // consume the exps of the PREVIOUS gap (no VALU result is used right after it is produced)
      float& ea0 = (m & 1) ? eB0 : eA0; float& ea1 = (m & 1) ? eB1 : eA1;       // written in this gap
      float& eb0 = (m & 1) ? eA0 : eB0; float& eb1 = (m & 1) ? eA1 : eB1;       // written in the previous gap
      if constexpr (FILL == 1) {        // the head-dim-64 softmax mix: 2 v_exp, 2 v_add, 1 v_cvt_pk per MFMA
        asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\tv_cvt_pk_f16_f32 %4, %7, %8"
                     : "=&v"(ea0), "=&v"(ea1), "+v"(s0), "+v"(s1), "=&v"(pk) : "v"(x[0]), "v"(x[1]), "v"(eb0), "v"(eb1));
      } else if constexpr (FILL == 2) { // two v_exp_f32
        asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=&v"(ea0), "=&v"(ea1) : "v"(x[0]), "v"(x[1]));
      } else if constexpr (FILL == 3) { // one v_exp_f32
        asm volatile("v_exp_f32 %0, %1" : "=&v"(ea0) : "v"(x[0]));
      } else if constexpr (FILL == 4) { // five plain VALU
        asm volatile("v_add_f32 %0, %5, %6\n\tv_add_f32 %1, %5, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\tv_cvt_pk_f16_f32 %4, %7, %8"
                     : "=&v"(ea0), "=&v"(ea1), "+v"(s0), "+v"(s1), "=&v"(pk) : "v"(x[0]), "v"(x[1]), "v"(eb0), "v"(eb1));
      } else if constexpr (FILL == 6) { // f16 exponentials on packed halves: 1 v_cvt_pk + 2 v_exp_f16 (lo, hi by op_sel) per MFMA
        asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_exp_f16 %1, %4\n\tv_exp_f16_sdwa %1, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"
                     : "=&v"(pk), "+v"(pk2) : "v"(x[0]), "v"(x[1]), "v"(pkprev));
        pkprev = pk;
      } else if constexpr (FILL == 8) { // the mix with its row sums on the matrix pipe: 2 v_exp + 1 v_cvt_pk per MFMA, and per
                                        // 2 MFMAs one v_mfma_f32_4x4x4_16B_f16 with A = ones (sums the 4 packed halves a lane holds)
        asm volatile("v_exp_f32 %0, %3\n\tv_exp_f32 %1, %4\n\tv_cvt_pk_f16_f32 %2, %5, %6"
                     : "=&v"(ea0), "=&v"(ea1), "=&v"(pk) : "v"(x[0]), "v"(x[1]), "v"(eb0), "v"(eb1));
        if (m & 1) {
          typedef __attribute__((ext_vector_type(4))) _Float16 h4;
          typedef __attribute__((ext_vector_type(4))) float f4;
          union { unsigned u[2]; h4 v; } pb; pb.u[0] = pk; pb.u[1] = pkprev;
          const h4 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
          f4& la4 = (m & 2) ? lsum1 : lsum0;
          la4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, pb.v, la4, 0, 0, 0);
        }
        pkprev = pk;
      } else if constexpr (FILL == 7) { // two v_exp_f16 only
        asm volatile("v_exp_f16 %0, %1\n\tv_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(pk2) : "v"(pkprev));
      } else if constexpr (FILL == 9) { // the mix with its row sums as ONE v_dot2_f32_f16 of the packed pair: 2 v_exp + 1 v_cvt_pk + 1 v_dot2
        asm volatile("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_cvt_pk_f16_f32 %2, %6, %7\n\tv_dot2_f32_f16 %3, %8, %9, %3"
                     : "=&v"(ea0), "=&v"(ea1), "=&v"(pk), "+v"(s0) : "v"(x[0]), "v"(x[1]), "v"(eb0), "v"(eb1), "v"(pkprev), "v"(0x3c003c00u));
        pkprev = pk;
      } else if constexpr (FILL == 10) { // the same with the sum by v_dot2c_f32_f16 (VOP2 form, accumulates in place)
        asm volatile("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_cvt_pk_f16_f32 %2, %6, %7\n\tv_dot2c_f32_f16 %3, %8, %9"
                     : "=&v"(ea0), "=&v"(ea1), "=&v"(pk), "+v"(s0) : "v"(x[0]), "v"(x[1]), "v"(eb0), "v"(eb1), "v"(pkprev), "v"(0x3c003c00u));
        pkprev = pk;
      } else if constexpr (FILL == 11 || FILL == 12) {
       if (FILL == 11 || (m & 1)) {
        // the exponentials of a gap's TWO scores WITHOUT the transcendental unit -- Cody-Waite on packed
        // fp16: t = cvt_pk(s0, s1); clamp (far keys must flush to zero, not wrap the exponent field); u = t + 1536 (ulp 1:
        // the integer part lands in the low mantissa bits); r = u - 1536; g = t - r in [-1/2, 1/2]; degree-3 polynomial for
        // 2^g (3 v_pk_fma_f16); exponent insert by integer arithmetic (v_pk_mad_u16: u * 1024 + bits, mod 2^16); max(., 0)
        // (an underflowed exponent field wraps into the sign bit: negative / NaN -> 0); row sum of the packed pair by
        // v_dot2_f32_f16.  11 VALU for two values where the v_exp mix has 5 (2 v_exp, 2 v_add, 1 v_cvt_pk).
        // FILL == 12: every second gap this way, the others with v_exp ("half of a tile's exponentials off the unit").
        unsigned t_, u_, r_, g_, p_;
        asm volatile("v_cvt_pk_f16_f32 %0, %6, %7\n\tv_pk_max_f16 %0, %0, %8\n\tv_pk_add_f16 %1, %0, %9\n\t"
                     "v_pk_add_f16 %2, %1, %9 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f16 %3, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_fma_f16 %4, %3, %10, %11\n\tv_pk_fma_f16 %4, %4, %3, %12\n\tv_pk_fma_f16 %4, %4, %3, %13\n\t"
                     "v_pk_mad_u16 %4, %1, %14, %4\n\tv_pk_max_f16 %4, %4, %15\n\tv_dot2_f32_f16 %5, %4, %16, %5"
                     : "=&v"(t_), "=&v"(u_), "=&v"(r_), "=&v"(g_), "=&v"(p_), "+v"(s0)
                     : "v"(x[0]), "v"(x[1]), "v"(0xcb00cb00u), "v"(0x66006600u), "v"(0x2b1b2b1bu), "v"(0x33b033b0u), "v"(0x398c398cu),
                       "v"(0x3c003c00u), "v"(0x04000400u), "v"(0u), "v"(0x3c003c00u));
        pk = p_;
       } else        // (the even gaps of FILL == 12: the v_exp mix)
        asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\tv_cvt_pk_f16_f32 %4, %7, %8"
                     : "=&v"(ea0), "=&v"(ea1), "+v"(s0), "+v"(s1), "=&v"(pk) : "v"(x[0]), "v"(x[1]), "v"(eb0), "v"(eb1));
      } else if constexpr (FILL == 13) { // the packed-fp16 route with its row sums left out (its floor): 10 packed VALU per two values
        unsigned t_, u_, r_, g_, p_;
        asm volatile("v_cvt_pk_f16_f32 %0, %5, %6\n\tv_pk_max_f16 %0, %0, %7\n\tv_pk_add_f16 %1, %0, %8\n\t"
                     "v_pk_add_f16 %2, %1, %8 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f16 %3, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_fma_f16 %4, %3, %9, %10\n\tv_pk_fma_f16 %4, %4, %3, %11\n\tv_pk_fma_f16 %4, %4, %3, %12\n\t"
                     "v_pk_mad_u16 %4, %1, %13, %4\n\tv_pk_max_f16 %4, %4, %14"
                     : "=&v"(t_), "=&v"(u_), "=&v"(r_), "=&v"(g_), "=&v"(p_)
                     : "v"(x[0]), "v"(x[1]), "v"(0xcb00cb00u), "v"(0x66006600u), "v"(0x2b1b2b1bu), "v"(0x33b033b0u), "v"(0x398c398cu),
                       "v"(0x3c003c00u), "v"(0x04000400u), "v"(0u));
        pk = p_;
      } else if constexpr (FILL == 14) { // row sums on the PACKED pair: 2 v_exp + 1 v_cvt_pk + 1 v_pk_add_f16 (7 of 8 gaps; fp32 once per phase)
        asm volatile("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_cvt_pk_f16_f32 %2, %6, %7\n\tv_pk_add_f16 %3, %3, %8"
                     : "=&v"(ea0), "=&v"(ea1), "=&v"(pk), "+v"(pk2) : "v"(x[0]), "v"(x[1]), "v"(eb0), "v"(eb1), "v"(pkprev));
        pkprev = pk;
      } else if constexpr (FILL == 15) { // no row sums at all (the floor of any scheme that moves them): 2 v_exp + 1 v_cvt_pk
        asm volatile("v_exp_f32 %0, %3\n\tv_exp_f32 %1, %4\n\tv_cvt_pk_f16_f32 %2, %5, %6"
                     : "=&v"(ea0), "=&v"(ea1), "=&v"(pk) : "v"(x[0]), "v"(x[1]), "v"(eb0), "v"(eb1));
      } else if constexpr (FILL == 16) { // 2 v_exp + 1 v_cvt_pk + 1 v_add_f32 (e.g. one fp32 add of a pre-added pair)
        asm volatile("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_cvt_pk_f16_f32 %2, %6, %7\n\tv_add_f32 %3, %3, %6"
                     : "=&v"(ea0), "=&v"(ea1), "=&v"(pk), "+v"(s0) : "v"(x[0]), "v"(x[1]), "v"(eb0), "v"(eb1));
      } else if constexpr (FILL == 5) { // three plain VALU (the mix without its exps)
        asm volatile("v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\tv_cvt_pk_f16_f32 %2, %3, %4"
                     : "+v"(s0), "+v"(s1), "=&v"(pk) : "v"(eb0), "v"(eb1));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = s0 + s1 + (float)(pk & 1) + (float)(pk2 & 3) + lsum0[0] + lsum1[1] + eA0 + eA1 + eB0 + eB1 + (float)trsink[0];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[n][r];
  out[gid] = sum;
  if ((tid & 63) == 0) {
    const int w = gid >> 6;
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

template <int NACC, int FILL, int MFMA = 1, int LDSOP = 0> void run(const char* name, int waves_per_simd, const _Float16* src, float* out, unsigned long long* st) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int blocks = cus * waves_per_simd;          // 256 threads = 4 waves = one per SIMD
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {               // repeat: the clock settles under sustained load
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<NACC, FILL, MFMA, LDSOP>), dim3(blocks), dim3(256), 0, 0, src, out, st, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const int nw = blocks * 4;
  std::vector<unsigned long long> h(2 * nw);
  (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc(nw), clk(nw);
  for (int w = 0; w < nw; ++w) { cyc[w] = (double)h[2 * w]; clk[w] = (double)h[2 * w] / ((double)h[2 * w + 1] / 100e6); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double mfmas = 8.0 * iters;
  const double flops = (double)nw * mfmas * 32 * 32 * 16 * 2;
  printf("%-44s %d wave(s)/SIMD: %6.1f cycles per MFMA and wave, %5.1f per MFMA and SIMD, clock %.3f GHz, %7.1f TFLOP/s (%.1f %% of 2.5 PF)\n",
         name, waves_per_simd, cyc[nw / 2] / mfmas, cyc[nw / 2] / mfmas / waves_per_simd, clk[nw / 2] / 1e9, flops / (ms * 1e-3) / 1e12,
         flops / (ms * 1e-3) / 1e12 / 25.0);
}

int main() {
  _Float16* src; float* out; unsigned long long* st;
  std::vector<_Float16> h(4096 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  (void)hipMalloc(&src, h.size() * 2);
  (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 256 * 8 * 256 * 4);
  (void)hipMalloc(&st, 256 * 8 * 4 * 16);
  for (int w = 1; w <= 2; ++w) {
    run<4, 0>("bare MFMA, 4 independent accumulators", w, src, out, st);
    run<1, 0>("bare MFMA, one accumulator chain", w, src, out, st);
    run<4, 1>("MFMA + 2 v_exp + 2 v_add + 1 v_cvt_pk", w, src, out, st);
    run<4, 0, 1, 1>("bare MFMA, A operand by ds_read_b128", w, src, out, st);
    run<4, 1, 1, 1>("MFMA + softmax mix, A operand by ds_read_b128", w, src, out, st);
    run<4, 1, 1, 2>("MFMA + softmax mix, 12 LDS reads per 8 MFMAs", w, src, out, st);
    run<4, 1, 1, 3>("MFMA + softmax mix, 6 LDS reads per 8 MFMAs", w, src, out, st);
    run<4, 0, 1, 2>("bare MFMA, 12 LDS reads per 8 MFMAs", w, src, out, st);
    run<4, 0, 1, 3>("bare MFMA, 6 LDS reads per 8 MFMAs", w, src, out, st);
    run<4, 2>("MFMA + 2 v_exp", w, src, out, st);
    run<4, 3>("MFMA + 1 v_exp", w, src, out, st);
    run<4, 4>("MFMA + 4 v_add + 1 v_cvt_pk", w, src, out, st);
    run<4, 5>("MFMA + 2 v_add + 1 v_cvt_pk", w, src, out, st);
    run<4, 8>("MFMA + 2 v_exp + 1 v_cvt_pk + 1/2 mfma_4x4x4 (row sums)", w, src, out, st);
    run<4, 9>("MFMA + 2 v_exp + 1 v_cvt_pk + 1 v_dot2_f32_f16", w, src, out, st);
    run<4, 10>("MFMA + 2 v_exp + 1 v_cvt_pk + 1 v_dot2c_f32_f16", w, src, out, st);
    run<4, 14>("MFMA + 2 v_exp + 1 v_cvt_pk + 1 v_pk_add_f16", w, src, out, st);
    run<4, 15>("MFMA + 2 v_exp + 1 v_cvt_pk (no row sums)", w, src, out, st);
    run<4, 16>("MFMA + 2 v_exp + 1 v_cvt_pk + 1 v_add_f32", w, src, out, st);
    run<4, 14, 0>("no MFMA: 2 v_exp + 1 v_cvt_pk + 1 v_pk_add_f16", w, src, out, st);
    run<4, 11>("MFMA + packed-fp16 polynomial exp2 x 2 (11 VALU, no v_exp)", w, src, out, st);
    run<4, 12>("MFMA + v_exp mix / packed polynomial in alternate gaps", w, src, out, st);
    run<4, 13>("MFMA + packed polynomial exp2 x 2 without row sums (10)", w, src, out, st);
    run<4, 11, 0>("no MFMA: packed-fp16 polynomial exp2 x 2 (11 VALU)", w, src, out, st);
    run<4, 9, 0>("no MFMA: 2 v_exp + 1 v_cvt_pk + 1 v_dot2_f32_f16", w, src, out, st);
    run<4, 10, 0>("no MFMA: 2 v_exp + 1 v_cvt_pk + 1 v_dot2c_f32_f16", w, src, out, st);
    run<4, 6>("MFMA + 1 v_cvt_pk + 2 v_exp_f16", w, src, out, st);
    run<4, 7>("MFMA + 2 v_exp_f16", w, src, out, st);
    run<4, 6, 0>("no MFMA: 1 v_cvt_pk + 2 v_exp_f16", w, src, out, st);
    run<4, 7, 0>("no MFMA: 2 v_exp_f16", w, src, out, st);
    run<4, 1, 0>("no MFMA: 2 v_exp + 2 v_add + 1 v_cvt_pk", w, src, out, st);
    run<4, 2, 0>("no MFMA: 2 v_exp", w, src, out, st);
    run<4, 4, 0>("no MFMA: 4 v_add + 1 v_cvt_pk", w, src, out, st);
  }
  return 0;
}
