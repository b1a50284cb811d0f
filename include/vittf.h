/* vittf.h -- C ABI of the MI355X-native vit-tf hot path (libvittf.so, gfx950 only).
 *
 * The reference (xeTaiz/vit-tf) has no FFI: its hot path is Python calling stock PyTorch ops.
 * This header is the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md): every
 * entry point names the reference lines whose arithmetic it replaces.  Conventions:
 *
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked [host]
 *   - the caller owns all memory; nothing is allocated inside (workspace passed in, size from the
 *     matching *_workspace_bytes query); no host synchronisation inside any call
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work of a call is
 *     enqueued on it in order, so calls compose without further synchronisation
 *   - return value: 0 (VITTF_OK) or a negative vittf_status; no exceptions cross the ABI
 *   - "h16" = the 16-bit arithmetic type selected by vittf_dtype (bf16 or fp16); feature volumes
 *     and K features are always IEEE fp16 like the reference's files (infer.py:134, 337-340)
 *   - thread-compatible: no global mutable state
 */
#ifndef VITTF_H
#define VITTF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITTF_ABI_VERSION 6

typedef enum vittf_status {
  VITTF_OK = 0,
  VITTF_ERR_INVALID_ARG = -1,   /* bad shape / null pointer / unsupported configuration */
  VITTF_ERR_WORKSPACE = -2,     /* workspace too small */
  VITTF_ERR_LAUNCH = -3,        /* HIP launch error (hipGetLastError != hipSuccess) */
  VITTF_ERR_NO_DEVICE = -4      /* no gfx950 device visible */
} vittf_status;

typedef enum vittf_dtype { VITTF_BF16 = 0, VITTF_FP16 = 1 } vittf_dtype;

/* slice axis, naming of infer.py:138-152 ('z' slices along the last volume dim) */
typedef enum vittf_axis { VITTF_AXIS_X = 0, VITTF_AXIS_Y = 1, VITTF_AXIS_Z = 2 } vittf_axis;

typedef enum vittf_sample_mode { VITTF_SAMPLE_NEAREST = 0, VITTF_SAMPLE_TRILINEAR = 1 } vittf_sample_mode;

int vittf_abi_version(void);
const char* vittf_status_string(int status);
/* number of visible gfx950 devices (<= 0: none).  Does not create a context on failure. */
int vittf_device_count(void);

/* ------------------------------------------------------------------------------------------
 * ViT description.  Replaces the module tree of the upstream DINO VisionTransformer that the
 * reference fetches with torch.hub (infer.py:42-43, 323).  Head dim must be 64 (all DINO ViTs),
 * embed_dim a multiple of 128, patch 8 or 16.
 * ---------------------------------------------------------------------------------------- */
typedef struct vittf_vit_config {
  int32_t embed_dim;   /* D: 384 (ViT-S) / 768 (ViT-B) */
  int32_t depth;       /* L: 12 */
  int32_t heads;       /* D / 64 */
  int32_t patch;       /* P: 8 */
  int32_t dtype;       /* vittf_dtype of the MFMA operands */
  float   ln_eps;      /* 1e-6 */
  int32_t attention_fp8; /* 0: 16-bit attention (default).  1: the fp8 (e4m3) block-scaled MFMA attention path of BASELINE
                            configs[3] (vittf_attention_fp8): 3-mantissa-bit operands, ~3e-2 on the features -- opt-in */
  int32_t flags;       /* 0 = the measured path.  Bits select the slower alternatives the parity tests also run (ABI 6: they
                            were environment variables latched inside the library before -- the library reads no environment
                            on this path any more): */
} vittf_vit_config;
enum {
  VITTF_CFG_SEPARATE_LN = 1,       /* every LayerNorm as its own launch instead of riding on the kernel in front of it */
  VITTF_CFG_UNSCALED_Q = 2,        /* q as the model produces it + the online-maximum attention kernel (vittf_attention(.., 0)) */
  VITTF_CFG_FP8_HEAD_SCALES = 4    /* fp8 attention with one scale per (slice, head) (absmax + quantise launches) instead of row scales */
};
#define VITTF_TAIL_STEPS 112       /* 24 KB steps of one block in vittf_vit_weights.tail_packed */

/* All weights live in HBM for the lifetime of the engine (about 43 MB for ViT-S).  Per-layer tensors
 * are stacked along a leading L dimension.  "h16" matrices are row-major [out][in] exactly like the
 * nn.Linear weights of the DINO state dict, converted once to the engine dtype. */
typedef struct vittf_vit_weights {
  const float* pe_w_t;   /* [P*P][D]  patch-embed conv folded to ONE input channel, transposed (k-major);
                            folding of the 3 identical grey channels + ImageNet mean/std: infer.py:39-40,154-155 */
  const float* pe_b;     /* [D]       folded bias */
  const void*  qkv_w;    /* h16 [L][3D][D]   blocks.i.attn.qkv.weight */
  const float* qkv_b;    /* [L][3D] */
  const void*  proj_w;   /* h16 [L][D][D]    blocks.i.attn.proj.weight */
  const float* proj_b;   /* [L][D] */
  const void*  fc1_w;    /* h16 [L][4D][D]   blocks.i.mlp.fc1.weight */
  const float* fc1_b;    /* [L][4D] */
  const void*  fc2_w;    /* h16 [L][D][4D]   blocks.i.mlp.fc2.weight */
  const float* fc2_b;    /* [L][D] */
  const void*  tail_packed; /* h16 [L][112][12288] or NULL: proj_w, fc1_w and fc2_w of every block as the weight stream of the
                            block-tail kernel (vittf_block_tail; packing: vit-tf_amd/weights.py pack_block_tail_weights); when
                            non-NULL and D == 384 the engine runs everything behind the attention of a block -- proj, residual,
                            norm2, MLP, residual, the next block's norm1 -- as one launch */
  const void*  qkv_packed; /* h16 [L][3D/32][12288] or NULL: qkv_w as the 24 KB LDS images of vittf_gemm_as (weights.py
                            pack_row_images); when non-NULL and D == 384 the qkv projection runs on the activation-stationary kernel */
  const float* ln1_g;    /* [L][D] */
  const float* ln1_b;    /* [L][D] */
  const float* ln2_g;    /* [L][D] */
  const float* ln2_b;    /* [L][D] */
} vittf_vit_weights;

/* Position embedding for ONE image size, already interpolated on the host in the upstream
 * scale-factor/bicubic form: row 0 = cls_token + pos[0], rows 1.. = patch part. */
typedef struct vittf_pos_embed {
  const float* cls_plus_pos0;  /* [D] */
  const float* patch_pos;      /* [f0*f1][D] */
} vittf_pos_embed;

/* Where the slices of one axis sit inside the resident fp32 volume (W, H, D), C-contiguous.
 * Replaces make_4d(vol).permute(permute_in) (infer.py:138-142, 154) without materialising it. */
typedef struct vittf_slice_view {
  const float* vol;        /* fp32 volume in HBM */
  int64_t stride_slice;    /* element strides of the (slice, row, col) view */
  int64_t stride_row;
  int64_t stride_col;
  int32_t in_rows;         /* slice size in volume voxels */
  int32_t in_cols;
  int32_t out_rows;        /* network input size (im_sz of the axis, multiple of P), nearest-resized: */
  int32_t out_cols;        /*   src = min(floor(dst * in/out), in-1), F.interpolate 'nearest' infer.py:177 */
  const float* minmax;     /* [2] device: global min, max of the volume (norm_minmax infer.py:32-34) */
} vittf_slice_view;

/* min / max of n floats -> out[2] (device).  ws: >= vittf_minmax_workspace_bytes() bytes.
 * Replaces t.min(), t.max() of norm_minmax (infer.py:33). */
size_t vittf_minmax_workspace_bytes(void);
int vittf_volume_minmax(const float* vol, int64_t n, float* out_minmax, void* ws, size_t ws_bytes, void* stream);

/* Bytes of workspace for a forward over `batch` slices of `tokens` = f0*f1 + 1 tokens. */
size_t vittf_vit_workspace_bytes(const vittf_vit_config* cfg, int32_t batch, int32_t tokens);

/* The ViT leg of compute_qkv (infer.py:173-177 + the hooked K third, infer.py:133-135, 189-203):
 * for slices [slice0, slice0 + batch) of `view`: normalise, nearest-resize, patch-embed, (L-1) full
 * blocks, then ONLY LayerNorm1 + the K projection of block L; the K features of the patch tokens
 * (CLS dropped, infer.py:202) are rounded to fp16 (infer.py:134) and written token-major:
 *     k_out[(s - slice0) * f0*f1 + token][D]      token = row-major over the (rows/P, cols/P) grid
 * qkv_part selects the third of the hooked qkv tensor: 0 = q, 1 = k (what infer.py's __main__ asks for,
 * infer.py:326,331), 2 = v  (compute_qkv's return_keys, infer.py:195-209).
 * [host] cfg, w, pos, view are host structs holding device pointers. */
int vittf_vit_k_features(const vittf_vit_config* cfg, const vittf_vit_weights* w, const vittf_pos_embed* pos,
                         const vittf_slice_view* view, int32_t slice0, int32_t batch, int32_t qkv_part,
                         uint16_t* k_out, void* ws, size_t ws_bytes, void* stream);

/* Optional timing of the launches inside vittf_vit_k_features and vittf_similarity, by kernel class, with HIP events recorded on
 * the caller's stream (what bench.py's roofline leg reads).  Process-global, off by default, not thread-safe:
 * the one exception to "no global mutable state".  enable(mask) clears earlier records and starts recording the
 * classes whose bit (1 << vittf_kernel_class) is set (-1: all; two event records per launch cost ~2-3 % of the
 * throughput when every launch is bracketed, so bench.py times with the dominant class only), enable(0) stops; collect() waits for the recorded events and returns, per class, the summed launch
 * durations in ms and the launch counts (arrays of VITTF_KERNEL_CLASSES entries, [host]). */
typedef enum vittf_kernel_class {
  VITTF_KERNEL_PATCH_EMBED = 0, VITTF_KERNEL_LAYERNORM = 1,
  VITTF_KERNEL_GEMM = 2,        /* linears without a class of their own: the K-feature projection */
  VITTF_KERNEL_ATTENTION = 3, VITTF_KERNEL_MLP = 4,
  VITTF_KERNEL_GEMM_QKV = 5,    /* attn.qkv (+ q pre-scale) */
  VITTF_KERNEL_GEMM_PROJ = 6,   /* attn.proj + residual (+ norm2) */
  VITTF_KERNEL_GEMM_FC1 = 7,    /* mlp.fc1 + GELU */
  VITTF_KERNEL_GEMM_FC2 = 8,    /* mlp.fc2 + residual (+ the next norm1) */
  VITTF_KERNEL_SIMILARITY = 9,  /* the voxel x query accumulation kernel(s) inside vittf_similarity / _maps_f32 */
  VITTF_KERNEL_CLASSES = 10
} vittf_kernel_class;
int vittf_profiler_enable(int32_t class_mask);
int vittf_profiler_collect(double* ms_per_class, int64_t* launches_per_class);
/* Name of the kernel the library's dispatcher launched LAST for a class that has several candidates (attention, similarity):
 * "" if none was launched yet.  Static storage; never NULL. */
const char* vittf_profiler_kernel_name(int32_t kernel_class);

/* ------------------------------------------------------------------------------------------
 * Individual kernels, exported so that each one is parity-tested on its own (tests/).
 * Row counts: activations are [rows][cols] row-major; `rows` need not be tile aligned.
 * ---------------------------------------------------------------------------------------- */

/* tokens[b][0] = cls+pos0 ; tokens[b][1+p] = patch_embed(slice slice0+b)[p] + pos[p]   -> fp32 [batch][tokens][D] */
int vittf_patch_embed(const vittf_vit_config* cfg, const vittf_vit_weights* w, const vittf_pos_embed* pos,
                      const vittf_slice_view* view, int32_t slice0, int32_t batch, float* tokens_out, void* stream);

/* y = LayerNorm(x) * g + b, x fp32 [rows][D] -> y h16 [rows][D]  (nn.LayerNorm, biased variance) */
int vittf_layernorm(const float* x, const float* g, const float* b, void* y, int64_t rows, int32_t d,
                    float eps, int32_t dtype, void* stream);

typedef enum vittf_epilogue {
  VITTF_EPI_BIAS = 0,        /* out h16 [rows][n] = a.w^T + bias */
  VITTF_EPI_BIAS_GELU = 1,   /* out h16 [rows][n] = gelu_erf(a.w^T + bias)        (Mlp.fc1 + nn.GELU) */
  VITTF_EPI_BIAS_RESIDUAL = 2,/* out fp32 [rows][n] += a.w^T + bias                (x = x + proj/fc2(...)) */
  VITTF_EPI_KFEAT = 3,       /* out fp16: rows whose (row % tokens) == 0 (CLS) are dropped, the others are
                                written densely: out[(row/tokens)*(tokens-1) + row%tokens - 1][n] */
  VITTF_EPI_BIAS_QKV = 4     /* as VITTF_EPI_BIAS, but columns [0, n/3) (the q third of Attention.qkv) are multiplied by
                                log2(e)/8 before the single rounding to h16: the softmax scale and the exp -> exp2 base
                                change folded into q, for vittf_attention(..., q_prescaled = 1) */
} vittf_epilogue;

/* out = epilogue(a[rows][k] . w[n][k]^T + bias[n]);  a, w: h16; k % 64 == 0, n % 128 == 0.
 * `tokens` is only used by VITTF_EPI_KFEAT. */
int vittf_gemm(const void* a, const void* w, const float* bias, void* out, int64_t rows, int32_t n, int32_t k,
               int32_t epilogue, int32_t tokens, int32_t dtype, void* stream);

/* Residual linear + the LayerNorm that follows it:  x[rows][n] (fp32) += a[rows][k] . w[n][k]^T + bias;
 * h[rows][n] (h16) = LayerNorm(x; ln_g, ln_b, ln_eps).  Replaces attn.proj + residual + norm2 and mlp.fc2 + residual +
 * the next block's norm1.  For n = 384 (ViT-S) the LayerNorm is computed in the epilogue of a whole-row GEMM; other
 * shapes run vittf_gemm(BIAS_RESIDUAL) + vittf_layernorm. */
int vittf_gemm_residual_ln(const void* a, const void* w, const float* bias, float* x, int64_t rows, int32_t n, int32_t k,
                           int32_t dtype, const float* ln_g, const float* ln_b, float ln_eps, void* h, void* stream);

/* The K = 384 linears with a wide 16-bit output (Attention.qkv of ViT-S) with the ACTIVATIONS stationary: a wave keeps its 32 rows
 * as 24 MFMA operands and the weights stream through LDS, so every activation byte is read once.  out = epilogue(a . w^T + bias),
 * epilogue VITTF_EPI_BIAS or VITTF_EPI_BIAS_QKV; k == 384; n a multiple of 64, at most 1536 (a multiple of 96 for _QKV).
 * w_packed: h16 [n / 32][12288], the 24 KB images of weights.py pack_row_images (image u = rows 32 u .. + 31 of w in the LDS
 * tile layout).  Same bits as vittf_gemm on the same operands.  tile_counter: vittf_gemm_as_workspace_bytes() bytes of
 * caller-owned device memory, 4-byte aligned, private to this call until it has finished on `stream` (256-row tiles are handed
 * out through it; the call zeroes it on `stream` itself); concurrent calls need one each. */
size_t vittf_gemm_as_workspace_bytes(void);
int vittf_gemm_as(const void* a, const void* w_packed, const float* bias, void* out, int64_t rows, int32_t n, int32_t k,
                  int32_t epilogue, int32_t dtype, void* tile_counter, void* stream);

/* Everything behind the attention of one block for D == 384, in one launch:
 *   x' = x + attn_out . Wp^T + proj_b ;  x_new = x' + fc2(gelu_erf(fc1(LayerNorm(x'; ln2)) + b1)) + b2 ;  x := x_new ;
 *   h_out = LayerNorm(x_new; ln_g, ln_b)  (the next block's norm1; optional: ln_g, ln_b, h_out all NULL leaves it out).
 * Replaces Attention.proj + the two residual adds + norm2 + Mlp.forward of the upstream DINO block (reached through
 * model(...), infer.py:177).  attn_out: h16 [rows][D] (vittf_attention's output); w_packed: h16 [VITTF_TAIL_STEPS][12288], one
 * block of vittf_vit_weights.tail_packed.  The fp32 residual rows are read once and written once, and neither x' nor norm2's
 * output nor the hidden activation reaches HBM.  Same result as vittf_gemm_residual_ln (proj) + vittf_gemm(BIAS_GELU) +
 * vittf_gemm_residual_ln (fc2) up to fp32 summation order.  attn_out, w_packed, x and h_out must be 16-byte aligned
 * (VITTF_ERR_INVALID_ARG otherwise); D != 384: VITTF_ERR_INVALID_ARG (use the GEMM entries).  tile_counter:
 * vittf_block_tail_workspace_bytes() bytes of caller-owned device memory, 4-byte aligned, private to this call until it has
 * finished on `stream` (the persistent workgroups hand out 128-row tiles through it; the call zeroes it on `stream` itself;
 * concurrent calls -- two streams -- need one each; inside vittf_vit_k_features it lives in the engine workspace). */
size_t vittf_block_tail_workspace_bytes(void);
int vittf_block_tail(const void* attn_out, const void* w_packed, const float* proj_b, const float* ln2_g, const float* ln2_b,
                     const float* b1, const float* b2, float* x, int64_t rows, int32_t d, int32_t dtype, const float* ln_g,
                     const float* ln_b, float ln_eps, void* h_out, void* tile_counter, void* stream);

/* Multi-head self-attention over `batch` independent sequences of `tokens` rows.
 * qkv h16 [batch*tokens][3D] with columns [q | k | v], heads of 64 concatenated inside each third
 * (layout of Attention.qkv's output); out h16 [batch*tokens][D] = softmax(q k^T / 8) v per head.
 * q_prescaled = 0: q as the model produces it.  q_prescaled = 1: q already multiplied by log2(e)/8
 * (VITTF_EPI_BIAS_QKV) -- same result, computed by the faster kernel the engine uses (two 32-row query blocks per wave taking
 * turns, lazy running maximum). */
int vittf_attention(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype,
                    int32_t q_prescaled, void* stream);

/* Statistics: how many times (per 32-row query block and 32-key half step) the pre-scaled kernel's lazy running maximum
 * took its overflow branch (a row sum said the 16-bit P would overflow: true maximum, everything accumulated rescaled)
 * since the counter was last reset.  Rare on trained weights; the tests use it to prove that a case drives the branch.
 * Synchronises the device.  reset != 0 zeroes the counter after reading it. */
int64_t vittf_attention_rescale_count(int32_t reset);

/* fp8 attention (BASELINE configs[3]: "ViT-B/8 features, fp8 MFMA attention path").  Same contract as vittf_attention with
 * q_prescaled = 1, computed with OCP e4m3 operands on v_mfma_scale_f32_32x32x64_f8f6f4: per (slice, head) power-of-two
 * scales for q, k and v (absmax / 448, applied by the instruction's scale operands), fp32 softmax statistics and
 * accumulators, P as fp8.  Error of the attention output ~3e-2 relative (3-bit mantissas):
 * an opt-in, never the default.  ws: vittf_attention_fp8_workspace_bytes(batch, tokens, heads) bytes, 256-byte aligned. */
size_t vittf_attention_fp8_workspace_bytes(int32_t batch, int32_t tokens, int32_t heads);
int vittf_attention_fp8(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype, void* ws,
                        size_t ws_bytes, void* stream);

/* The same path with q and k quantised where they are produced (round 4; what the engine runs for embed_dim >= 768):
 *   vittf_gemm_qkv_fp8: Attention.qkv (+ the q scale of VITTF_EPI_BIAS_QKV) for rows = batch * tokens rows, n = 3 * heads * 64,
 *     n / 3 a multiple of 256, k >= 768 and a multiple of 64.  q and k leave as e4m3 bytes [slice][head][token][64] with one
 *     power-of-two scale (E8M0 byte) per row and 32-wide block -- the MX block format: the matrix instruction's scale operand
 *     is per lane, i.e. per row and block -- into `ws`; v is written as h16 into the v third of qkv_out [rows][n] (the q and k
 *     thirds of qkv_out are NOT written) and its per-(slice, head) absolute maximum is collected into `ws`.
 *   vittf_attention_fp8_rows: quantises + re-lays the v third (per-(slice, head) scale), then the attention kernel with the
 *     row scales.  Same output contract and error class as vittf_attention_fp8 (finer scales for q and k).
 * ws: the same vittf_attention_fp8_workspace_bytes(batch, tokens, heads) workspace, passed to both calls. */
int vittf_gemm_qkv_fp8(const void* a, const void* w, const float* bias, void* qkv_out, int64_t rows, int32_t n, int32_t k,
                       int32_t tokens, int32_t heads, int32_t dtype, void* ws, size_t ws_bytes, void* stream);
int vittf_attention_fp8_rows(const void* qkv, void* out, int32_t batch, int32_t tokens, int32_t heads, int32_t dtype, void* ws,
                             size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Feature-volume epilogue (infer.py:201-203 permute_out, :329 AdaptiveAvgPool3d, :330-332 axis sum).
 * ---------------------------------------------------------------------------------------- */

/* Average-pool the slice axis of the token-major K features of ONE axis and scatter them, feature-major,
 * into a pooled slab:  for window i in [win0, win0 + nwin):
 *     slices [floor(i*S/n_out), ceil((i+1)*S/n_out))  (adaptive rule); running sum in slice order kept in
 *     fp16 (one rounding per add, as the CPU AdaptiveAvgPool3d does for half tensors), then fp16(sum / count);  written to dst[d*dst_stride_d + (i - win0)*dst_stride_win + r*dst_stride_row + c*dst_stride_col]
 * k_slices holds slices [k_slice0, k_slice0 + k_nslices) of the axis as [slice][f0][f1][D] fp16; every
 * slice a requested window touches must be inside that range.  n_out == S gives the un-pooled layout. */
int vittf_pool_slices(const uint16_t* k_slices, int32_t k_slice0, int32_t k_nslices, int32_t total_slices,
                      int32_t n_out, int32_t win0, int32_t nwin, int32_t f0, int32_t f1, int32_t d,
                      uint16_t* dst, int64_t dst_stride_d, int64_t dst_stride_win, int64_t dst_stride_row,
                      int64_t dst_stride_col, void* stream);

/* out = fp16(fp16(z + y) + x) element-wise: the reference's running fp16 sum in z, y, x order
 * (infer.py:330-332).  Inputs are the per-axis pooled volumes as gathered from `nranks` ranks:
 * axis a is stored [nranks][D][...] where each rank block holds `chunk[a]` windows of the axis' slice
 * dimension (slab layout written by vittf_pool_slices); out is (D, n0, n1, n2) fp16, C-contiguous. */
int vittf_assemble_sum(const uint16_t* gz, const uint16_t* gy, const uint16_t* gx, int32_t nranks,
                       const int32_t chunk[3], int32_t d, int32_t n0, int32_t n1, int32_t n2, uint16_t* out,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Similarity query (predict_ntf.py:52-72, 95-100) and label assignment (predict_ntf.py:203-215).
 * ---------------------------------------------------------------------------------------- */

/* F.grid_sample of the feature volume (infer.py:48-72): feat fp16 or fp32 (F, n0, n1, n2); rel fp32 [A][3]
 * relative coordinates in [-1, 1] in VOLUME dim order (the flip to grid_sample's x,y,z order of infer.py:67
 * happens inside); align_corners=False, zero padding; mode nearest or trilinear ('bilinear').  out fp32 [A][F]. */
int vittf_sample_features(const void* feat, int32_t feat_is_fp16, int32_t f, int32_t n0, int32_t n1, int32_t n2,
                          const float* rel, int32_t a, int32_t mode, const float* voxel_norm, float* out, void* stream);

/* Per-voxel L2 norm of the F-major fp16 feature volume, clamped like F.normalize's denominator:
 * out[v] = max(|feat[:, v]|_2, 1e-12), fp32 [nvox].  Passing it as `voxel_norm` to vittf_sample_features /
 * vittf_similarity makes them operate on F.normalize(feat, dim=0) -- the cosine similarity of
 * compare_feat_sampling.py:45 / tests/test_vishum.py:12 -- without materialising a normalised copy. */
int vittf_voxel_norm(const uint16_t* feat, int32_t f, int64_t nvox, float* out, void* stream);

size_t vittf_similarity_workspace_bytes(int32_t classes, int64_t nvox, int32_t annotations);

/* Fused similarity: for every voxel v and class c with annotations [class_start[c], class_start[c+1]):
 *     dot[a] = sum_f feat[f][v] * qf[a][f]                       (einsum predict_ntf.py:65)
 *     sim_c  = mean_a( dot[a] >= 0.25 ? dot[a]^2.5 : 0 )         (predict_ntf.py:71-72)
 *     (big_a_mean != 0: the single-class A > 1024 variant, mean of raw dots first, predict_ntf.py:62-63)
 *     u8     = trunc(255 / (0.99 * max_v sim_c) * sim_c) mod 256 (predict_ntf.py:98-99, x86 wrap-around)
 * then nearest-resized to (o0, o1, o2) (predict_ntf.py:100).  feat fp16 (F, n0, n1, n2) F-major;
 * qf fp32 [A][F]; class_start int32 [classes + 1] on the HOST; out uint8 [classes][o0][o1][o2].
 * voxel_norm: NULL (predict_ntf.py semantics: raw dot products) or the vittf_voxel_norm array (every dot divided by
 * the voxel's norm: cosine similarity against queries sampled with the same array). */
int vittf_similarity(const uint16_t* feat, int32_t f, int32_t n0, int32_t n1, int32_t n2, const float* qf,
                     const int32_t* class_start_host, int32_t classes, int32_t big_a_mean, const float* voxel_norm,
                     int32_t o0, int32_t o1, int32_t o2, uint8_t* out, void* ws, size_t ws_bytes, void* stream);

/* The interactive query in ONE call (predict_ntf.py:56-72, 95-100): samples the annotations' features (trilinear, as
 * vittf_sample_features) at rel_host -- fp32 [A][3] relative coordinates in HOST memory, formed as predict_ntf.py:56 does; they
 * travel as a kernel argument, so A = class_start_host[classes] is at most VITTF_QUERY_MAX_A --, accumulates the class maps and
 * quantises + resizes them: the same bytes as vittf_sample_features + vittf_similarity, with three launches and no copy or
 * memset between the call and the first kernel.  ws: vittf_similarity_query_workspace_bytes(classes, nvox, A, f) bytes. */
#define VITTF_QUERY_MAX_A 64
size_t vittf_similarity_query_workspace_bytes(int32_t classes, int64_t nvox, int32_t annotations, int32_t f);
int vittf_similarity_query(const uint16_t* feat, int32_t f, int32_t n0, int32_t n1, int32_t n2, const float* rel_host,
                           const int32_t* class_start_host, int32_t classes, int32_t big_a_mean, const float* voxel_norm,
                           int32_t o0, int32_t o1, int32_t o2, uint8_t* out, void* ws, size_t ws_bytes, void* stream);

/* fp32 per-group maps without quantisation or resizing: maps_out fp32 [classes][n0*n1*n2].
 *   mode 0: predict_ntf.py:65-72 (the input of the bilateral-solver branch :73-96); mode 1: its A > 1024 variant (:62-63);
 *   mode 2: the second similarity of resample_topk (infer.py:104-106): clamp(dot, 0, 1) ** exponent, mean over each group
 *           of queries; feat may then be fp32 (feat_is_fp16 = 0), e.g. an already normalised volume.
 * ws: vittf_similarity_workspace_bytes(classes, 0, annotations) bytes (no fp32 map region needed). */
int vittf_similarity_maps_f32(const void* feat, int32_t feat_is_fp16, int32_t f, int32_t n0, int32_t n1, int32_t n2,
                              const float* qf, const int32_t* class_start_host, int32_t classes, int32_t mode,
                              float exponent, const float* voxel_norm, float* maps_out, void* ws, size_t ws_bytes,
                              void* stream);

/* Per map (nmaps maps of nvox fp32 values): the first k voxel indices, in index order, whose value is >= the k-th largest
 * value of the map -- torch.topk(s.flatten(), K).values[-1]; (s >= that).nonzero()[:K]  (infer.py:95-96).
 * idx_out int32 [nmaps][k] (device). */
int vittf_topk_voxels(const float* maps, int32_t nmaps, int64_t nvox, int32_t k, int32_t* idx_out, void* stream);

/* take_most_dissimilar's distance (infer.py:118-121): dist[i] = 1 - mean_j cos(x_i, x_j) (measure 0, F.cosine_similarity
 * with eps 1e-8) or mean_j |x_i - x_j|_2 (measure 1, torch.cdist); x fp32 [n][f], dist fp32 [n]. */
int vittf_mean_pairwise_distance(const float* x, int32_t n, int32_t f, int32_t measure, float* dist, void* stream);

/* ---- label-volume helpers: sampler candidate masks and scores --------------------------- */
/* dst = binary_erosion(set, generate_binary_structure(3, connectivity)) with scipy.ndimage's defaults (one iteration,
 * border_value 0): set = {src == class_id} (class_id 0..255) or {src != 0} (class_id < 0); uint8 volumes (n0, n1, n2),
 * dst 0/1, src != dst.  connectivity 1 = 6 face neighbours, 2 = 18, >= 3 = the full 3x3x3 cube. */
int vittf_erode_mask(const uint8_t* src, int32_t n0, int32_t n1, int32_t n2, int32_t class_id, int32_t connectivity,
                     uint8_t* dst, void* stream);

/* sample_surface's candidate set (compare_feat_sampling.py:19-24): outer = erosion of {labels == class_id} by
 * generate_binary_structure(3, connectivity = dist_from_surface), shell = outer XOR erosion of outer by the 6-neighbour
 * element; shell uint8 0/1 (n0, n1, n2); ws: vittf_surface_shell_workspace_bytes(n0, n1, n2). */
size_t vittf_surface_shell_workspace_bytes(int32_t n0, int32_t n1, int32_t n2);
int vittf_surface_shell(const uint8_t* labels, int32_t n0, int32_t n1, int32_t n2, int32_t class_id, int32_t connectivity,
                        uint8_t* shell, void* ws, size_t ws_bytes, void* stream);

/* counts[t * classes + p] = number of voxels with target t and prediction p (sklearn confusion_matrix with labels
 * 0..classes-1; predict_ntf.py:228-246, evaluate_similarities.py:63-68); counts[classes * classes] = voxels holding a
 * value >= classes in either volume.  counts: int64 [classes * classes + 1] (device, overwritten); classes <= 16. */
int vittf_confusion_matrix(const uint8_t* target, const uint8_t* pred, int64_t n, int32_t classes, int64_t* counts,
                           void* stream);

/* F.interpolate(mode='nearest') of a uint8 volume (n0, n1, n2) -> (o0, o1, o2): src index = min(floor(dst * (float)in /
 * out), in - 1) per dim.  equals < 0: plain resize -- the label up-sample of predict_ntf.py:217-218; equals = c in 0..255:
 * the resized class mask (src == c) as 0/1 -- evaluate_similarities.py:63 without materialising the full-size mask. */
int vittf_resize_nearest_u8(const uint8_t* src, int32_t n0, int32_t n1, int32_t n2, uint8_t* dst, int32_t o0, int32_t o1,
                            int32_t o2, int32_t equals, void* stream);

/* dst fp32 [n] = src fp16 [n] (exact): volumes are stored as fp16 (create_synthetic_volumes.py:44-46) and uploaded as
 * such; the vol.float() of infer.py:137 / the astype(np.float32) of infer.py:230-232 happens in HBM.  16-byte aligned. */
int vittf_widen_f16(const uint16_t* src, int64_t n, float* dst, void* stream);

/* ---- 3-D bilateral solver post-process (bilateral_solver3d.py, predict_ntf.py:73-96) ---------- */
typedef struct vittf_bilateral_params {
  double sigma_spatial;        /* grid_params['sigma_spatial'] (predict_ntf.py:75-79: 7) */
  double lam;                  /* bs_params_default (bilateral_solver3d.py:162-167): 256 */
  double a_diag_min;           /* 1e-5 */
  double cg_tol;               /* 1e-5, relative to |b| (scipy cg rtol) */
  int32_t cg_maxiter;          /* 25 */
  int32_t bistochastize_iters; /* 10 (bilateral_solver3d.py:107) */
  int32_t pad;                 /* crop_pad(..., pad=2) (predict_ntf.py:90) */
  float crop_threshold;        /* crop_pad(..., thresh=0.1) */
} vittf_bilateral_params;

/* luma_bins = 1 + the largest luma bin; 0 = unsupported size. */
size_t vittf_bilateral_workspace_bytes(int32_t o0, int32_t o1, int32_t o2, double sigma_spatial, int32_t luma_bins);

/* One class of predict_ntf.py:80-94: volume (v0, v1, v2) fp32 and the class map sim_in (n0, n1, n2) fp32 are resized
 * (trilinear) to (o0, o1, o2); the volume becomes the uint8 grey reference (min-max); the box where sim > crop_threshold
 * (+ pad) is refined by the bilateral solver with the Sobel confidence and written back.  sim_out: fp32 (o0, o1, o2).
 * [host] luma_bin_host[256]: bilateral-space luma bin of each grey level, (rgb2yuv(v, v, v)[0] / sigma_luma).astype(int)
 * (bilateral_solver3d.py:19-20, 47); the chroma bins are constant for a grey reference and drop out.
 * [host] info_host (nullable): {vertices, voxels in the crop box}; {0, 0} when nothing exceeds the threshold (the map is
 * then returned unrefined; the reference raises in that case).  Synchronises the stream twice (box, vertex count). */
int vittf_bilateral_refine(const float* sim_in, int32_t n0, int32_t n1, int32_t n2, const float* volume, int32_t v0,
                           int32_t v1, int32_t v2, int32_t o0, int32_t o1, int32_t o2, const int32_t* luma_bin_host,
                           int32_t luma_bins, const vittf_bilateral_params* prm, float* sim_out, int32_t* info_host,
                           void* ws, size_t ws_bytes, void* stream);

/* (255 / (0.99 * max(sim)) * sim).to(uint8) with the x86 wrap-around (predict_ntf.py:95-96).  max_scratch: 4 device bytes. */
int vittf_quantize_wrap_u8(const float* sim, int64_t n, uint8_t* out, float* max_scratch, void* stream);

/* pred = 0; best = 0; for i: mask = sims[i] > thr[i] && sims[i] > best; pred[mask] = i+1; best[mask] = sims[i]
 * sims uint8 [classes][n]; thr int32 [classes] on the HOST (= int(t*255), predict_ntf.py:212); labels uint8 [n]. */
int vittf_assign_labels(const uint8_t* sims, int32_t classes, int64_t n, const int32_t* thr_host, uint8_t* labels,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITTF_H */
