// Flash-style multi-head self-attention forward, head dim 64, for the DINO ViT blocks.
//
// Replaces Attention.forward of the upstream model the reference calls (infer.py:177):
//   softmax(q k^T / sqrt(64)) v  per head, without materialising the N x N score matrix.
//
// Machine mapping (gfx950, wave64):
//   * one 256-thread workgroup = 4 waves = 128 query rows of one (slice, head); each wave owns 32 query rows
//   * K/V tiles of 64 keys are register-staged (global_load_dwordx4 issued before the tile's MFMAs, ds_write
//     after them) into two LDS buffers, one barrier per tile
//   * scores are computed TRANSPOSED, S^T = K Q^T with v_mfma_f32_32x32x16 (K rows as the A operand, Q rows as
//     the B operand, Q fragments live in registers for the whole kernel), so a lane owns one query column:
//     the row maximum / row sum of the online softmax are in-lane reductions plus ONE exchange with lane^32
//   * the S^T accumulator registers, converted pairwise to 16 bit, are directly the B operand of the second
//     product O^T = V^T P^T (k order 16s + 8(j>>2) + 4h + (j&3)); the matching V^T A-fragments come from
//     ds_read_b64_tr_b16 transposed reads of a row-major V image (8-row x 32-col subtiles, conflict free)
//   * the K image uses the tile_off() swizzle shared with the GEMM and is read with ds_read_b128
//   * exp2 with the softmax scale folded into one FMA: p = exp2(s*c - m*c), c = log2(e)/8
//   * token count need not be tile aligned (N = f0*f1 + 1): query/key rows past the end are clamped on load,
//     the last key tile is masked to -inf, stores are guarded
//   * workgroups are remapped so that the q-tiles of one (slice, head) share an XCD's L2 (K/V re-reads)
#include "vittf_common.h"

namespace {

constexpr int QT = 128;   // query rows per workgroup
constexpr int KT = 64;    // keys per tile
constexpr int KV_TILE_BYTES = KT * 64 * 2;  // 8 KB

// V image: [8 key groups][2 column halves] subtiles of 8 keys x 32 columns (512 B), chunk XOR by (key>>2)&3
__device__ __forceinline__ int v_off(int key, int ch) {
  return 1024 * (key >> 3) + 512 * (ch >> 2) + 64 * (key & 7) + 16 * ((ch & 3) ^ ((key >> 2) & 3));
}

template <int DT>
__global__ __launch_bounds__(256, 2) void attn_kernel(const unsigned short* __restrict__ qkv,
                                                      unsigned short* __restrict__ out, int tokens, int heads,
                                                      int q_tiles, int total, float c) {
  __shared__ __attribute__((aligned(16))) char smem[2][2][KV_TILE_BYTES];  // [buffer][K | V]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;

  const int item = xcd_remap(blockIdx.x, total);
  const int qt = item % q_tiles;
  const int bh = item / q_tiles;
  const int hd = bh % heads, b = bh / heads;
  const int dmodel = heads * 64;
  const int64_t ld = 3 * (int64_t)dmodel;
  const unsigned short* base = qkv + (int64_t)b * tokens * ld;
  const unsigned short* kbase = base + dmodel + hd * 64;
  const unsigned short* vbase = base + 2 * dmodel + hd * 64;

  // ---- Q fragments (B operand): lane holds Q[row l31][16 s + 8 h .. +7] ----
  const int qrow = qt * QT + wave * 32 + l31;
  const int qrow_c = qrow < tokens ? qrow : tokens - 1;
  s16x8_t qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *reinterpret_cast<const s16x8_t*>(base + (int64_t)qrow_c * ld + hd * 64 + 16 * s + 8 * h);

  // ---- staging assignment: 512 chunks of 16 B per tile, two per thread (rows r, r + 32; chunk tid & 7) ----
  const int st_row = tid >> 3, st_ch = tid & 7;
  uint4 kreg[2], vreg[2];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = t * KT + st_row + 32 * i;
      key = key < tokens ? key : tokens - 1;
      kreg[i] = *reinterpret_cast<const uint4*>(kbase + (int64_t)key * ld + st_ch * 8);
      vreg[i] = *reinterpret_cast<const uint4*>(vbase + (int64_t)key * ld + st_ch * 8);
    }
  };
  auto write_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = st_row + 32 * i;
      *reinterpret_cast<uint4*>(&smem[buf][0][tile_off(r, st_ch)]) = kreg[i];
      *reinterpret_cast<uint4*>(&smem[buf][1][v_off(r, st_ch)]) = vreg[i];
    }
  };

  f32x16_t o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // transposed-read lane constants
  const int g16 = lane >> 4;                 // 16-lane group 0..3
  const int tr_q = (lane & 15) >> 2;         // row inside the 4-row block
  const int tr_p = lane & 3;
  const int tr_ch = 2 * (g16 & 1) + (tr_p >> 1);
  const int tr_b8 = 8 * (tr_p & 1);

  const int nt = (tokens + KT - 1) / KT;
  load_tile(0);
  write_tile(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) load_tile(t + 1);
    const char* k_t = smem[buf][0];
    const char* v_t = smem[buf][1];

    // ---- S^T = K Q^T : two 32-key subtiles ----
    f32x16_t sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const s16x8_t kf = *reinterpret_cast<const s16x8_t*>(k_t + tile_off(32 * kt + l31, 2 * s + h));
        sacc[kt] = mfma32<DT>(kf, qf[s], sacc[kt]);
      }
    }
    if (t == nt - 1) {  // ragged last tile: keys >= tokens contribute nothing
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KT + 32 * kt + acc_row(r, h);
          if (key >= tokens) sacc[kt][r] = -INFINITY;
        }
    }

    // ---- online softmax (lane = one query column; 32 of the tile's 64 keys are in this lane) ----
    float tmax = sacc[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sacc[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[1][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    float psum = 0.f;
    s16x8_t pf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], c, -mc));
        psum += p[r];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        union { s16x8_t v; unsigned u[4]; } cv;
#pragma unroll
        for (int j = 0; j < 4; ++j) cv.u[j] = pack2_h16<DT>(p[8 * s2 + 2 * j], p[8 * s2 + 2 * j + 1]);
        pf[kt][s2] = cv.v;
      }
    }
    l_run = l_run * alpha + psum;
    if (!__all(m_new == m_run)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    m_run = m_new;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int dvt = 0; dvt < 2; ++dvt) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int key0 = 32 * kt + 16 * s2 + 4 * h + tr_q;
          union { s16x8_t v; s16x4_t hlf[2]; } vf;
          vf.hlf[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(v_t + v_off(key0, 4 * dvt + tr_ch) + tr_b8));
          vf.hlf[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(v_t + v_off(key0 + 8, 4 * dvt + tr_ch) + tr_b8));
          o[dvt] = mfma32<DT>(vf.v, pf[kt][s2], o[dvt]);
        }
      }
    }

    if (t + 1 < nt) write_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane owns query row `qrow`, columns 32 dvt + 8 g + 4 h + {0..3} ----
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (qrow < tokens) {
    unsigned short* orow = out + ((int64_t)b * tokens + qrow) * dmodel + hd * 64;
#pragma unroll
    for (int dvt = 0; dvt < 2; ++dvt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack2_h16<DT>(o[dvt][4 * g + 0] * inv, o[dvt][4 * g + 1] * inv);
        pk.y = pack2_h16<DT>(o[dvt][4 * g + 2] * inv, o[dvt][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + 32 * dvt + 8 * g + 4 * h) = pk;
      }
  }
}

}  // namespace

extern "C" int vittf_attention(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads,
                               int32_t dtype, void* stream) {
  if (!qkv || !out || batch <= 0 || tokens <= 0 || heads <= 0) return VITTF_ERR_INVALID_ARG;
  const int q_tiles = (tokens + QT - 1) / QT;
  const int64_t total64 = (int64_t)batch * heads * q_tiles;
  if (total64 > (1 << 30)) return VITTF_ERR_INVALID_ARG;
  const int total = (int)total64;
  const float c = 0.125f * 1.44269504088896340736f;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VITTF_BF16) {
    hipLaunchKernelGGL((attn_kernel<VITTF_BF16>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv,
                       (unsigned short*)out, tokens, heads, q_tiles, total, c);
  } else if (dtype == VITTF_FP16) {
    hipLaunchKernelGGL((attn_kernel<VITTF_FP16>), dim3(total), dim3(256), 0, st, (const unsigned short*)qkv,
                       (unsigned short*)out, tokens, heads, q_tiles, total, c);
  } else {
    return VITTF_ERR_INVALID_ARG;
  }
  return vittf_check_launch();
}
