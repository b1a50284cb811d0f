// fp8 (OCP e4m3) multi-head self-attention on the block-scaled matrix instruction of gfx950 -- the attention path of
// BASELINE.json configs[3] (ViT-B/8 features, "fp8 MFMA attention path").  Opt-in (vittf_vit_config.attention_fp8,
// infer.py --attention fp8): its error is that of 3-mantissa-bit operands and is stated where it is tested
// (tests/test_gpu_kernels.py::test_attention_fp8, 6e-2 relative Frobenius on the attention output), far outside the 1e-3
// of the default path.
//
// Replaces, like attention.hip, Attention.forward of the upstream model the reference calls (infer.py:177):
// softmax(q k^T / 8) v per head, q arriving pre-scaled by log2(e)/8 (VITTF_EPI_BIAS_QKV).
//
//   v_mfma_scale_f32_32x32x64_f8f6f4: 32 x 32 x 64 per instruction, A and B = 32 fp8 bytes per lane (lane l: row / column
//   l & 31, k = 32 (l >> 5) + byte index: probed with exact integers, profiles/r02b_mfma_f8_probe.txt), one E8M0 scale
//   byte per operand (0x7F = 1.0).  At head dim 64 the whole q . k of a 32 x 32 score block is ONE instruction (four in
//   16-bit), and a 64-key step of the output product two (eight): 4 MFMAs per 64 keys and 32 queries instead of 16, at
//   twice the cycles each -- half the matrix-pipe time and a quarter of the MFMA issue slots of the 16-bit kernels.
//
// Three launches per call:
//   1. absmax per (slice, head) of q, k and v                      -> power-of-two scales 2^e (exact: applied by the
//      matrix instruction's own scale operands, so the scores still arrive in exp2 units with -M as the C input)
//   2. quantise + re-lay: Q8 / K8 [slice][head][token][64] bytes; V8T [slice][head][64 dims][keys] with the keys of every
//      64-key tile stored in the order the P operand will have them (below), zero padding to a multiple of 64 keys
//   3. flash attention: 4 waves x 32 query rows; per 64-key tile S^T_b = K_b Q^T (b = 0, 1: two MFMAs), the lazy-maximum
//      softmax of attention.hip in fp32, P packed to fp8 with a fixed factor 2 (v_cvt_pk_fp8_f32; un-done by
//      the scale operand of the next instruction), O^T_dh += V^T_dh P^T (two MFMAs).
//      The S^T accumulators of lane half h hold keys (r & 3) + 8 (r >> 2) + 4 h of each 32-key block; packed in register
//      order they ARE the B operand of the output product if V^T's k slots are laid out to match:
//          slot 32 h + 16 b + r  <->  key 32 b + (r & 3) + 8 (r >> 2) + 4 h        (done once, by launch 2)
//      so every operand fragment of the kernel is 32 contiguous bytes: two ds_read_b128, no transposing reads.
//
// Round 4: a second operand preparation, for q / k that the qkv GEMM itself writes as fp8 (gemm_pp.hip, vittf_gemm_qkv_fp8):
// the instruction's scale operands are per LANE, i.e. per row and 32-wide k block -- the MX block format -- so q and k rows
// carry their own power-of-two scales ([slice][head][token][2] E8M0 bytes; the bytes of a row stored in the instruction's
// block order [d 0-15 | d 32-47 | d 16-31 | d 48-63]: block b = bytes 16 b .. 16 b + 15 of both lane halves), computed where
// the row is produced: no absmax
// pass over q and k, no quantise pass, half the bytes out of the GEMM.  v keeps its per-(slice, head) scale (its 32-element
// blocks run along the KEYS: they cross the rows a GEMM tile produces); its absmax comes from that GEMM's epilogue and only
// the v third goes through the quantise + re-lay kernel (vittf_attention_fp8_rows).
#include "attn_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

constexpr int QT = 128;      // query rows per workgroup
constexpr int KT = 64;       // keys per tile
constexpr int TILE_B = KT * 64;            // one fp8 operand tile: 4 KB
constexpr int BUF_B = 2 * TILE_B;          // K | V^T
// P is stored as fp8(2 p), p = exp2(s - M) with M such that the row maximum of the tile that set it is 1.  The overflow
// check is the lane's row SUM over the tile's 32 values (no per-score maximum on the fast path): <= 256 keeps every value
// <= 256 < 448 (e4m3 maximum).  The factor 2 is the hysteresis: a flat row sums to 64, so M is only moved again when the
// scores have risen by about two binades -- a target at the bound itself (first version: 64, sum of a flat row 2048) took
// the slow path on every tile of a diffuse-attention workload (2.54 against 1.66 ms per ViT-B launch).
constexpr float P_HEADROOM = 2.f;
constexpr int P_HEADROOM_LOG2 = 1;
constexpr float P_SUM_BOUND = 256.f;

// power-of-two scale exponent for a tensor with absolute maximum amax: amax * 2^-e <= 448 (e4m3 maximum), e >= -20
__device__ __forceinline__ int scale_exp(float amax) {
  if (!(amax > 0.f)) return 0;
  int ex;
  (void)frexpf(amax * (1.0f / 448.0f), &ex);        // amax / 448 = m 2^ex, m in [0.5, 1)
  return ex < -20 ? -20 : ex;
}

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

// slot of key `kin` (0..63) inside its tile of V8T: inverse of key = 32 b + (r & 3) + 8 (r >> 2) + 4 h, slot = 32 h + 16 b + r
__device__ __forceinline__ int vt_slot(int kin) {
  const int b = kin >> 5, w = kin & 31;
  const int h = (w >> 2) & 1, r = (w & 3) + 4 * (w >> 3);
  return 32 * h + 16 * b + r;
}

// ---------------------------------------------------------------- 1. absmax per (slice, head, q|k|v)
template <int DT>
__global__ __launch_bounds__(256) void absmax_kernel(const unsigned short* __restrict__ qkv, int tokens, int heads,
                                                     unsigned* __restrict__ amax_bits) {
  __shared__ unsigned smax[3];
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int tid = threadIdx.x;
  if (tid < 3) smax[tid] = 0u;
  __syncthreads();
  const int dmodel = heads * 64, ld = 3 * dmodel;
  const int tok = blockIdx.x * 64 + (tid >> 2), quarter = tid & 3;
  float m[3] = {0.f, 0.f, 0.f};
  if (tok < tokens) {
    const unsigned short* row = qkv + ((int64_t)b * tokens + tok) * ld + hd * 64 + 16 * quarter;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const s16x8_t v0 = *reinterpret_cast<const s16x8_t*>(row + p * dmodel);
      const s16x8_t v1 = *reinterpret_cast<const s16x8_t*>(row + p * dmodel + 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        m[p] = fmaxf(m[p], fabsf(h16_to_f32<DT>((unsigned short)v0[j])));
        m[p] = fmaxf(m[p], fabsf(h16_to_f32<DT>((unsigned short)v1[j])));
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m[p] = fmaxf(m[p], __shfl_xor(m[p], off));
    if ((tid & 63) == 0) atomicMax(&smax[p], __float_as_uint(m[p]));      // non-negative floats order like their bits
  }
  __syncthreads();
  if (tid < 3 && smax[tid]) atomicMax(amax_bits + bh * 3 + tid, smax[tid]);
}

// ---------------------------------------------------------------- 2. quantise + re-lay (one 64-token tile of one head)
template <int DT>
__global__ __launch_bounds__(256) void quant_kernel(const unsigned short* __restrict__ qkv, int tokens, int heads, int np,
                                                    const unsigned* __restrict__ amax_bits, unsigned char* __restrict__ q8,
                                                    unsigned char* __restrict__ k8, unsigned char* __restrict__ v8t) {
  __shared__ __attribute__((aligned(16))) unsigned char vt[64][64 + 16];   // [dim][slot], padded rows
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int tid = threadIdx.x;
  const int dmodel = heads * 64, ld = 3 * dmodel;
  const int tile = blockIdx.x, kin = tid >> 2, quarter = tid & 3;
  const int tok = tile * 64 + kin;
  float inv[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) inv[p] = ldexpf(1.0f, -scale_exp(__uint_as_float(amax_bits[bh * 3 + p])));
  unsigned w[3][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  if (tok < tokens) {
    const unsigned short* row = qkv + ((int64_t)b * tokens + tok) * ld + hd * 64 + 16 * quarter;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const s16x8_t v0 = *reinterpret_cast<const s16x8_t*>(row + p * dmodel);
      const s16x8_t v1 = *reinterpret_cast<const s16x8_t*>(row + p * dmodel + 8);
      float f[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f[j] = h16_to_f32<DT>((unsigned short)v0[j]) * inv[p];
        f[8 + j] = h16_to_f32<DT>((unsigned short)v1[j]) * inv[p];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) w[p][j] = pack4_fp8(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3]);
    }
  }
  // q, k: natural [token][64] rows (rows >= tokens are written as zeros: the key padding of the last tile)
  const int64_t rowoff = ((int64_t)bh * np + tok) * 64 + 16 * quarter;
  *reinterpret_cast<uint4*>(q8 + rowoff) = make_uint4(w[0][0], w[0][1], w[0][2], w[0][3]);
  *reinterpret_cast<uint4*>(k8 + rowoff) = make_uint4(w[1][0], w[1][1], w[1][2], w[1][3]);
  // v: through LDS into [dim][slot]
  const int slot = vt_slot(kin);
#pragma unroll
  for (int j = 0; j < 16; ++j) vt[16 * quarter + j][slot] = (unsigned char)(w[2][j >> 2] >> (8 * (j & 3)));
  __syncthreads();
  const int d = tid >> 2;
  const uint4 o = *reinterpret_cast<const uint4*>(&vt[d][16 * quarter]);
  *reinterpret_cast<uint4*>(v8t + ((int64_t)bh * 64 + d) * np + tile * 64 + 16 * quarter) = o;
}

// ---------------------------------------------------------------- 2b. the v third alone (q / k arrive as fp8 from the GEMM)
template <int DT>
__global__ __launch_bounds__(256) void quant_v_kernel(const unsigned short* __restrict__ qkv, int tokens, int heads, int np,
                                                      const unsigned* __restrict__ amax_bits, unsigned char* __restrict__ v8t) {
  __shared__ __attribute__((aligned(16))) unsigned char vt[64][64 + 16];   // [dim][slot], padded rows
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int tid = threadIdx.x;
  const int dmodel = heads * 64, ld = 3 * dmodel;
  const int tile = blockIdx.x, kin = tid >> 2, quarter = tid & 3;
  const int tok = tile * 64 + kin;
  const float inv = ldexpf(1.0f, -scale_exp(__uint_as_float(amax_bits[bh * 3 + 2])));
  unsigned w[4] = {0u, 0u, 0u, 0u};
  if (tok < tokens) {
    const unsigned short* row = qkv + ((int64_t)b * tokens + tok) * ld + 2 * dmodel + hd * 64 + 16 * quarter;
    const s16x8_t v0 = *reinterpret_cast<const s16x8_t*>(row);
    const s16x8_t v1 = *reinterpret_cast<const s16x8_t*>(row + 8);
    float f[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = h16_to_f32<DT>((unsigned short)v0[j]) * inv;
      f[8 + j] = h16_to_f32<DT>((unsigned short)v1[j]) * inv;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = pack4_fp8(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3]);
  }
  const int slot = vt_slot(kin);
#pragma unroll
  for (int j = 0; j < 16; ++j) vt[16 * quarter + j][slot] = (unsigned char)(w[j >> 2] >> (8 * (j & 3)));
  __syncthreads();
  const int d = tid >> 2;
  const uint4 o = *reinterpret_cast<const uint4*>(&vt[d][16 * quarter]);
  *reinterpret_cast<uint4*>(v8t + ((int64_t)bh * 64 + d) * np + tile * 64 + 16 * quarter) = o;
}

// ---------------------------------------------------------------- 3. attention
// LDS image of an operand tile: 64 rows (keys for K, dims for V^T) x 64 bytes, the four 16-byte chunks of row r stored at
// chunk position c ^ ((r >> 2) & 3): the 16-lane groups of ds_read_b128 then hit distinct banks
__device__ __forceinline__ int img_off(int r, int c) { return r * 64 + 16 * (c ^ ((r >> 2) & 3)); }

__device__ __forceinline__ i32x8_t read_frag(const char* tile, int row, int hh) {
  const uint4 lo = *reinterpret_cast<const uint4*>(tile + img_off(row, 2 * hh));
  const uint4 hi = *reinterpret_cast<const uint4*>(tile + img_off(row, 2 * hh + 1));
  i32x8_t f;
  f[0] = (int)lo.x; f[1] = (int)lo.y; f[2] = (int)lo.z; f[3] = (int)lo.w;
  f[4] = (int)hi.x; f[5] = (int)hi.y; f[6] = (int)hi.z; f[7] = (int)hi.w;
  return f;
}

// ROWSC: q and k carry one E8M0 scale byte per row and 32-wide k block (qs / ks: [slice][head][token][2]) instead of one
// exponent per (slice, head): a lane's scale operand is then the byte of ITS row and block.
// A per-lane scale operand, finished well ahead of the matrix instruction that reads it: hipcc folds the byte's zero extension
// into a v_and_b32 right in front of the v_mfma_scale and leaves no wait states between the two -- with a uniform scale every
// stale value is the right one, with per-row scales it is the previous tile's (found with host-made MX operands: 0.6
// relative error from 64 tokens on, none with uniform bytes).  The mask and its wait states are one asm statement here.
__device__ __forceinline__ int scale_operand(int raw) {
  int r;
  asm volatile("v_and_b32 %0, 0xff, %1\n\ts_nop 7" : "=v"(r) : "v"(raw));
  return r;
}

template <int DT, bool ROWSC>
__global__ __launch_bounds__(256, 3) void attn_fp8_kernel(const unsigned char* __restrict__ q8, const unsigned char* __restrict__ k8,
                                                          const unsigned char* __restrict__ v8t,
                                                          const unsigned* __restrict__ amax_bits, unsigned short* __restrict__ out,
                                                          int tokens, int heads, int np, int q_tiles, int total,
                                                          const unsigned char* __restrict__ qs, const unsigned char* __restrict__ ks) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const int item = xcd_remap(blockIdx.x, total);
  const int qt = item % q_tiles, bh = item / q_tiles;
  const int hd = bh % heads, b = bh / heads;
  const int dmodel = heads * 64;

  const int eq = ROWSC ? 0 : scale_exp(__uint_as_float(amax_bits[bh * 3 + 0]));
  const int ek = ROWSC ? 0 : scale_exp(__uint_as_float(amax_bits[bh * 3 + 1]));
  const int ev = scale_exp(__uint_as_float(amax_bits[bh * 3 + 2]));
  int sc_q = (127 + eq) * 0x01010101;
  const int sc_k = (127 + ek) * 0x01010101, sc_v = (127 + ev) * 0x01010101;
  const int sc_p = (127 - P_HEADROOM_LOG2) * 0x01010101;

  // Q fragment (B operand): lane holds Q[row l31][32 hh .. +31]
  const int qrow = qt * QT + wave * 32 + l31;
  const int qrow_c = qrow < tokens ? qrow : tokens - 1;
  if constexpr (ROWSC) sc_q = scale_operand((int)qs[((int64_t)bh * np + qrow_c) * 2 + hh]);      // this lane's row and block
  // K scales: lane (l31, hh) of key block bk needs the byte of key 64 t + 32 bk + l31, block hh
  const unsigned char* ksl = ROWSC ? ks + ((int64_t)bh * np + l31) * 2 + hh : nullptr;
  int ksc[2] = {sc_k, sc_k};
  if constexpr (ROWSC) { ksc[0] = scale_operand((int)ksl[0]); ksc[1] = scale_operand((int)ksl[64]); }
  i32x8_t qf;
  {
    const unsigned char* qp = q8 + ((int64_t)bh * np + qrow_c) * 64 + 32 * hh;
    const uint4 lo = *reinterpret_cast<const uint4*>(qp), hi = *reinterpret_cast<const uint4*>(qp + 16);
    qf[0] = (int)lo.x; qf[1] = (int)lo.y; qf[2] = (int)lo.z; qf[3] = (int)lo.w;
    qf[4] = (int)hi.x; qf[5] = (int)hi.y; qf[6] = (int)hi.z; qf[7] = (int)hi.w;
  }

  // staging: LDS position p = tid (16-byte chunk index 0..255) of each operand image <- row p >> 2, source chunk
  const int srow = tid >> 2, schunk = (tid & 3) ^ ((srow >> 2) & 3);
  const unsigned char* ksrc = k8 + (int64_t)bh * np * 64 + srow * 64 + 16 * schunk;             // + t * 4096
  const unsigned char* vsrc = v8t + ((int64_t)bh * 64 + srow) * np + 16 * schunk;               // + t * 64
  const unsigned dma_dst = (unsigned)(size_t)LDS_PTR(smem) + (__builtin_amdgcn_readfirstlane(tid & ~63) << 4);
  auto stage = [&](int t, int buf) {
    lds_dma16_flat(ksrc + (int64_t)t * TILE_B, dma_dst + buf * BUF_B);
    lds_dma16_flat(vsrc + t * KT, dma_dst + buf * BUF_B + TILE_B);
  };

  f32x16_t o0, o1, negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; negm[r] = 0.f; }
  float l_run = 0.f;
  const int nt = np / KT;
  // Two LDS buffers, the DMA of tile t+1 under the arithmetic of tile t, one barrier per tile; 139 VGPRs = three waves per
  // SIMD, which is where the overlap of matrix and softmax work comes from here.  (A software-pipelined variant -- the
  // scores of tile t+1 issued before the softmax of tile t, 3-deep ring -- needs 192 VGPRs, i.e. two waves per SIMD, and
  // ran 25 % slower: 3.17 against 2.54 ms per ViT-B launch with the two preparation kernels.)
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  asm volatile("" : "+v"(qf));

  const bool active = __builtin_amdgcn_readfirstlane(qt * QT + wave * 32) < tokens;
  constexpr float THR = P_SUM_BOUND;
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
    int ksn[2] = {sc_k, sc_k};                  // the next tile's key scales (waited for with the tile's DMA below)
    if constexpr (ROWSC) {
      if (t + 1 < nt) { ksn[0] = (int)ksl[(t + 1) * 128]; ksn[1] = (int)ksl[(t + 1) * 128 + 64]; }
    }
    if (active) {
      const char* kt = smem + buf * BUF_B;
      const char* vtile = kt + TILE_B;
      f32x16_t s[2];
#pragma unroll
      for (int bk = 0; bk < 2; ++bk) {
        const i32x8_t kf = read_frag(kt, 32 * bk + l31, hh);
        s[bk] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf, negm, 0, 0, 0, ksc[bk], 0, sc_q);
      }
      if (t == nt - 1) {
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t * KT + 32 * bk + acc_row(r, hh) >= tokens) s[bk][r] = -INFINITY;
      }
      float p[2][16];
      float ps = 0.f;
#pragma unroll
      for (int bk = 0; bk < 2; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[bk][r] = __builtin_amdgcn_exp2f(s[bk][r]); ps += p[bk][r]; }
      if (t == 0 || __any(!(ps <= THR))) {
        // first tile / the values have outgrown the fp8 range: move M so that the row maximum becomes P_HEADROOM
        float tmax = -INFINITY;
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
          for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[bk][r]);
        const unsigned tb = __float_as_uint(tmax);
        const auto sw = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);
        tmax = fmaxf(tmax, fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
        // s already contains -M_old; shift by delta so that the largest exponent of the tile is log2(P_HEADROOM)
        const float delta = (t == 0) ? tmax - (float)P_HEADROOM_LOG2 : fmaxf(tmax - (float)P_HEADROOM_LOG2, 0.f);
        if (t != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] -= delta;
        ps = 0.f;
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
          for (int r = 0; r < 16; ++r) { p[bk][r] = __builtin_amdgcn_exp2f(s[bk][r] - delta); ps += p[bk][r]; }
      }
      l_run += ps;
      i32x8_t pf;
#pragma unroll
      for (int bk = 0; bk < 2; ++bk)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pf[4 * bk + j] = (int)pack4_fp8(p[bk][4 * j], p[bk][4 * j + 1], p[bk][4 * j + 2], p[bk][4 * j + 3]);
      const i32x8_t vf0 = read_frag(vtile, l31, hh);
      const i32x8_t vf1 = read_frag(vtile, 32 + l31, hh);
      o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf0, pf, o0, 0, 0, 0, sc_v, 0, sc_p);
      o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf1, pf, o1, 0, 0, 0, sc_v, 0, sc_p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (ROWSC) { ksc[0] = scale_operand(ksn[0]); ksc[1] = scale_operand(ksn[1]); }
  }
  if (!active) return;

  // normalise (l carries the head-room factor, O does not) and store: lane owns query row qrow,
  // columns 32 dvt + 8 g + 4 hh + {0..3}
  float l_tot;
  {
    const unsigned lb = __float_as_uint(l_run);
    const auto sw = __builtin_amdgcn_permlane32_swap(lb, lb, false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = P_HEADROOM / l_tot;
  if (qrow < tokens) store_o_row<DT>(out + ((int64_t)b * tokens + qrow) * dmodel + hd * 64, hh, o0, o1, inv);
}

struct Fp8Ws { size_t amax, q8, k8, v8t, qs, ks, total; int np; };
Fp8Ws fp8_ws(int batch, int tokens, int heads) {
  Fp8Ws w;
  w.np = (tokens + KT - 1) / KT * KT;
  const size_t per = (size_t)batch * heads * w.np * 64;
  const size_t sc = ((size_t)batch * heads * w.np * 2 + 255) & ~(size_t)255;     // row scales (vittf_gemm_qkv_fp8)
  w.amax = 0;
  w.q8 = ((size_t)batch * heads * 3 * 4 + 255) & ~(size_t)255;
  w.k8 = w.q8 + per;
  w.v8t = w.k8 + per;
  w.qs = w.v8t + per;
  w.ks = w.qs + sc;
  w.total = w.ks + sc;
  return w;
}

}  // namespace

extern "C" size_t vittf_attention_fp8_workspace_bytes(int32_t batch, int32_t tokens, int32_t heads) {
  if (batch <= 0 || tokens <= 0 || heads <= 0) return 0;
  return fp8_ws(batch, tokens, heads).total;
}

extern "C" int vittf_attention_fp8(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype,
                                   void* ws, size_t ws_bytes, void* stream) {
  vittf_note_kernel(VITTF_KERNEL_ATTENTION, "attn_fp8_kernel (+ absmax + quantise)");
  if (!qkv || !out || !ws || batch <= 0 || tokens <= 0 || heads <= 0) return VITTF_ERR_INVALID_ARG;
  if (dtype != VITTF_BF16 && dtype != VITTF_FP16) return VITTF_ERR_INVALID_ARG;
  if (((uintptr_t)ws & 255) != 0) return VITTF_ERR_INVALID_ARG;
  const Fp8Ws w = fp8_ws(batch, tokens, heads);
  if (ws_bytes < w.total) return VITTF_ERR_WORKSPACE;
  if ((int64_t)batch * heads > 65535) return VITTF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)ws;
  unsigned* amax = (unsigned*)(base + w.amax);
  unsigned char* q8 = (unsigned char*)(base + w.q8);
  unsigned char* k8 = (unsigned char*)(base + w.k8);
  unsigned char* v8t = (unsigned char*)(base + w.v8t);
  if (hipMemsetAsync(amax, 0, (size_t)batch * heads * 3 * 4, st) != hipSuccess) return VITTF_ERR_LAUNCH;
  const int tiles = w.np / KT;
  const dim3 grid(tiles, batch * heads);
  const int q_tiles = (tokens + QT - 1) / QT;
  const int total = batch * heads * q_tiles;
#define FP8_LAUNCH(DTV)                                                                                                 \
  {                                                                                                                     \
    hipLaunchKernelGGL((absmax_kernel<DTV>), grid, dim3(256), 0, st, (const unsigned short*)qkv, tokens, heads, amax);  \
    hipLaunchKernelGGL((quant_kernel<DTV>), grid, dim3(256), 0, st, (const unsigned short*)qkv, tokens, heads, w.np,    \
                       amax, q8, k8, v8t);                                                                              \
    hipLaunchKernelGGL((attn_fp8_kernel<DTV, false>), dim3(total), dim3(256), 0, st, q8, k8, v8t, amax,                 \
                       (unsigned short*)out, tokens, heads, w.np, q_tiles, total, (const unsigned char*)nullptr,        \
                       (const unsigned char*)nullptr);                                                                  \
  }
  if (dtype == VITTF_BF16) FP8_LAUNCH(VITTF_BF16) else FP8_LAUNCH(VITTF_FP16)
#undef FP8_LAUNCH
  return vittf_check_launch();
}

// Where vittf_gemm_qkv_fp8 (gemm_pp.hip) puts its outputs inside the workspace of this file (C++ linkage, not part of the ABI).
void vittf_fp8_ws_pointers(void* ws, int32_t batch, int32_t tokens, int32_t heads, unsigned** amax, unsigned char** q8,
                           unsigned char** k8, unsigned char** qs, unsigned char** ks, int32_t* np) {
  const Fp8Ws w = fp8_ws(batch, tokens, heads);
  char* base = (char*)ws;
  *amax = (unsigned*)(base + w.amax); *q8 = (unsigned char*)(base + w.q8); *k8 = (unsigned char*)(base + w.k8);
  *qs = (unsigned char*)(base + w.qs); *ks = (unsigned char*)(base + w.ks); *np = w.np;
}

extern "C" int vittf_attention_fp8_rows(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype,
                                        void* ws, size_t ws_bytes, void* stream) {
  vittf_note_kernel(VITTF_KERNEL_ATTENTION, "attn_fp8_kernel<row scales> (+ quantise v)");
  if (!qkv || !out || !ws || batch <= 0 || tokens <= 0 || heads <= 0) return VITTF_ERR_INVALID_ARG;
  if (dtype != VITTF_BF16 && dtype != VITTF_FP16) return VITTF_ERR_INVALID_ARG;
  if (((uintptr_t)ws & 255) != 0) return VITTF_ERR_INVALID_ARG;
  const Fp8Ws w = fp8_ws(batch, tokens, heads);
  if (ws_bytes < w.total) return VITTF_ERR_WORKSPACE;
  if ((int64_t)batch * heads > 65535) return VITTF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)ws;
  const unsigned* amax = (const unsigned*)(base + w.amax);
  const unsigned char* q8 = (const unsigned char*)(base + w.q8);
  const unsigned char* k8 = (const unsigned char*)(base + w.k8);
  unsigned char* v8t = (unsigned char*)(base + w.v8t);
  const unsigned char* qs = (const unsigned char*)(base + w.qs);
  const unsigned char* ks = (const unsigned char*)(base + w.ks);
  const int tiles = w.np / KT;
  const dim3 grid(tiles, batch * heads);
  const int q_tiles = (tokens + QT - 1) / QT;
  const int total = batch * heads * q_tiles;
#define FP8R_LAUNCH(DTV)                                                                                                \
  {                                                                                                                     \
    hipLaunchKernelGGL((quant_v_kernel<DTV>), grid, dim3(256), 0, st, (const unsigned short*)qkv, tokens, heads, w.np,  \
                       amax, v8t);                                                                                      \
    hipLaunchKernelGGL((attn_fp8_kernel<DTV, true>), dim3(total), dim3(256), 0, st, q8, k8, v8t, amax,                  \
                       (unsigned short*)out, tokens, heads, w.np, q_tiles, total, qs, ks);                              \
  }
  if (dtype == VITTF_BF16) FP8R_LAUNCH(VITTF_BF16) else FP8R_LAUNCH(VITTF_FP16)
#undef FP8R_LAUNCH
  return vittf_check_launch();
}
