// Helpers of the attention kernel (LDS images, v_max3).
#pragma once
#include "vittf_common.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

constexpr int ATT_KT = 64;                       // keys per LDS tile
constexpr int ATT_KV_TILE_BYTES = ATT_KT * 64 * 2;   // 8 KB
constexpr int ATT_BUF_BYTES = 2 * ATT_KV_TILE_BYTES; // K | V

// V image: [8 key groups][2 column halves] subtiles of 8 keys x 32 columns (512 B), chunk XOR by (key>>2)&3;
// read with ds_read_b64_tr_b16, conflict free
__device__ __forceinline__ int v_off(int key, int ch) {
  return 1024 * (key >> 3) + 512 * (ch >> 2) + 64 * (key & 7) + 16 * ((ch & 3) ^ ((key >> 2) & 3));
}
// inverse of v_off for the LDS-DMA source side: linear 16-byte position q -> (key, chunk)
__device__ __forceinline__ void v_pos(int q, int& key, int& ch) {
  const int kg = q >> 6, half = (q >> 5) & 1, k7 = (q >> 2) & 7, x = q & 3;
  key = 8 * kg + k7;
  ch = 4 * half + (x ^ ((key >> 2) & 3));
}

// max of three scores.  Plain fmaxf: attention.hip is built with -fno-honor-nans (Makefile), which drops the
// canonicalising v_max hipcc otherwise inserts per operand and lets it form v_max3_f32 itself.  (An inline-asm
// v_max3 is NOT an option: asm consumers of an MFMA result get none of the MFMA -> VALU wait states the compiler
// inserts for its own instructions, and read the accumulator before the matrix pipe has written it.)
__device__ __forceinline__ float max3_f32(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
