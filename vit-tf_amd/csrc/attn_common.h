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

// Output rows of the transposed product O^T = V^T P^T: a lane holds, of ITS query row, the columns
// 32 dvt + 8 g + 4 h + {0..3} (h = lane >> 5) in o0 (dvt = 0) / o1 (dvt = 1).  Written as they lie that is 16 8-byte
// stores per row; the end of a workgroup is bound by store ISSUE, so the column groups (g, g + 1) are paired across the
// lane halves first (v_permlane32_swap: the lower half ends up with [own g | upper's g] = columns 8 g .. 8 g + 7, the
// upper half with [lower's g + 1 | own g + 1]) and a row goes out as 8 16-byte stores.  `row` points at column 0 of
// this head.  +1.0 % on the 16-bit kernel (`tools/attn_ab.py`, profiles/r04j_attn_wide_store.txt).
template <int DT, typename ACC>
__device__ __forceinline__ void store_o_row(unsigned short* row, int h, const ACC& o0, const ACC& o1, float inv) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  unsigned short* p = row + 8 * h;
#pragma unroll
  for (int dv = 0; dv < 2; ++dv) {
    const ACC& o = dv ? o1 : o0;
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      const unsigned ax = pack2_h16<DT>(o[4 * g + 0] * inv, o[4 * g + 1] * inv);
      const unsigned ay = pack2_h16<DT>(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
      const unsigned bx = pack2_h16<DT>(o[4 * g + 4] * inv, o[4 * g + 5] * inv);
      const unsigned by = pack2_h16<DT>(o[4 * g + 6] * inv, o[4 * g + 7] * inv);
      const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
      u32x4_t pk;
      pk.x = sx[0]; pk.y = sy[0]; pk.z = sx[1]; pk.w = sy[1];
      *reinterpret_cast<u32x4_t*>(p + 32 * dv + 8 * g) = pk;
    }
  }
}
