// Orchestration of one batch of slices through the ViT: the C-ABI entry the Python host calls once per batch.
//
// Replaces the body of the reference's hot loop `model(F.interpolate(im_in[batch], ...).to(dev))` plus the
// forward hook on blocks[-1].attn.qkv (infer.py:133-135, 173-177): patch embed, (L-1) full pre-norm blocks and
// then only LayerNorm1 + the K third of the last block's qkv projection -- the rest of block L and the final
// norm never influence the hooked tensor, so they are not executed.
//
// Workspace layout for `batch` slices of N tokens (rows = batch * N, padded to the GEMM tile):
//   X    fp32 [rows][D]      residual stream
//   H    h16  [rows][D]      LayerNorm output / attention output (MFMA operand)
//   QKV  h16  [rows][3D]     attention input; QKV|O is re-used as the [rows][4D] MLP hidden buffer
//   O    h16  [rows][D]
//   (fp8 attention operands, opt-in)   tile counter of the block-tail / qkv kernels (4 bytes; one workspace = one stream)
// Everything is enqueued on the caller's stream; nothing synchronises with the host.
#include "vittf_common.h"

#include <stdlib.h>
#include <vector>

namespace {
// ---- optional per-kernel-class timing with HIP events on the caller's stream (bench.py's roofline leg) ----
// Process-global and off by default; the only mutable state in the library.  Not thread-safe.
struct ProfRec { hipEvent_t a, b; int cls; };
unsigned g_prof_mask = 0;   // bit per kernel class
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;

hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct WsLayout {
  size_t x, h, qkv, o, fp8, fp8_bytes, tile_ctr, total;
};

WsLayout ws_layout(int d, int64_t rows, size_t fp8_bytes = 0) {
  const int64_t rp = (rows + 127) / 128 * 128;
  WsLayout l;
  size_t off = 0;
  l.x = off;   off += align256((size_t)rp * d * 4);
  l.h = off;   off += align256((size_t)rp * d * 2);
  l.qkv = off; off += (size_t)rp * 3 * d * 2;   // QKV and O contiguous: together the [rows][4D] hidden buffer
  l.o = off;   off += align256((size_t)rp * d * 2);
  l.fp8 = off; l.fp8_bytes = fp8_bytes; off += align256(fp8_bytes);     // operands of the fp8 attention path (opt-in)
  l.tile_ctr = off; off += align256(vittf_block_tail_workspace_bytes()); // tile counter of the block-tail / fused-MLP launches
  l.total = off;
  return l;
}

bool config_ok(const vittf_vit_config* c) {
  return c && c->embed_dim > 0 && c->embed_dim % 128 == 0 && c->embed_dim <= 1024 && c->depth >= 1 &&
         c->heads * 64 == c->embed_dim && (c->patch == 8 || c->patch == 16) &&
         (c->dtype == VITTF_BF16 || c->dtype == VITTF_FP16) && c->ln_eps > 0.f;
}
}  // namespace

void* vittf_prof_begin(int cls, void* stream) {
  if (!((g_prof_mask >> cls) & 1u)) return nullptr;
  ProfRec* r = new ProfRec;
  r->cls = cls; r->a = prof_event(); r->b = prof_event();
  (void)hipEventRecord(r->a, (hipStream_t)stream);
  return r;
}
void vittf_prof_end(void* token, void* stream) {
  ProfRec* r = (ProfRec*)token;
  (void)hipEventRecord(r->b, (hipStream_t)stream);
  g_prof_recs.push_back(*r);
  delete r;
}

extern "C" int vittf_abi_version(void) { return VITTF_ABI_VERSION; }

extern "C" const char* vittf_status_string(int status) {
  switch (status) {
    case VITTF_OK: return "ok";
    case VITTF_ERR_INVALID_ARG: return "invalid argument";
    case VITTF_ERR_WORKSPACE: return "workspace too small";
    case VITTF_ERR_LAUNCH: return "HIP launch error";
    case VITTF_ERR_NO_DEVICE: return "no gfx950 device";
    default: return "unknown status";
  }
}

extern "C" int vittf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && __builtin_strstr(p.gcnArchName, "gfx950")) ++ok;
  }
  return ok;
}

extern "C" size_t vittf_vit_workspace_bytes(const vittf_vit_config* cfg, int32_t batch, int32_t tokens) {
  if (!config_ok(cfg) || batch <= 0 || tokens <= 1) return 0;
  return ws_layout(cfg->embed_dim, (int64_t)batch * tokens,
                   cfg->attention_fp8 ? vittf_attention_fp8_workspace_bytes(batch, tokens, cfg->heads) : 0).total;
}

extern "C" int vittf_vit_k_features(const vittf_vit_config* cfg, const vittf_vit_weights* w, const vittf_pos_embed* pos,
                                    const vittf_slice_view* view, int32_t slice0, int32_t batch, int32_t qkv_part,
                                    uint16_t* k_out, void* ws, size_t ws_bytes, void* stream) {
  if (!config_ok(cfg) || !w || !pos || !view || !k_out || !ws || batch <= 0 || slice0 < 0) return VITTF_ERR_INVALID_ARG;
  if (qkv_part < 0 || qkv_part > 2) return VITTF_ERR_INVALID_ARG;
  if (!w->qkv_w || !w->qkv_b || !w->proj_w || !w->proj_b || !w->fc1_w || !w->fc1_b || !w->fc2_w || !w->fc2_b ||
      !w->ln1_g || !w->ln1_b || !w->ln2_g || !w->ln2_b)
    return VITTF_ERR_INVALID_ARG;
  const int d = cfg->embed_dim, p = cfg->patch, L = cfg->depth, dt = cfg->dtype;
  if (view->out_rows % p || view->out_cols % p) return VITTF_ERR_INVALID_ARG;
  const int tokens = (view->out_rows / p) * (view->out_cols / p) + 1;
  const int64_t rows = (int64_t)batch * tokens;
  // the GEMM kernels index a call's widest buffer -- [rows][4 D] 16-bit hidden values, [rows][3 D] on the block-tail path -- with
  // 32-bit element offsets: a batch beyond that is refused, not computed wrong (ViT-B/8 at N = 4097: 349 slices)
  if (rows * (int64_t)((d == 384 && w->tail_packed) ? 3 * d : 4 * d) > 0xffffffffll) return VITTF_ERR_INVALID_ARG;
  const WsLayout lay = ws_layout(d, rows, cfg->attention_fp8 ? vittf_attention_fp8_workspace_bytes(batch, tokens, cfg->heads) : 0);
  if (ws_bytes < lay.total) return VITTF_ERR_WORKSPACE;
  if (((uintptr_t)ws & 255) != 0) return VITTF_ERR_INVALID_ARG;
  char* base = (char*)ws;
  float* X = (float*)(base + lay.x);
  void* H = base + lay.h;
  void* QKV = base + lay.qkv;
  void* O = base + lay.o;
  void* G = QKV;  // [rows][4D] hidden, aliases QKV|O (both dead while the MLP runs)

  int rc;
  { ProfScope ps(VITTF_KERNEL_PATCH_EMBED, stream); rc = vittf_patch_embed(cfg, w, pos, view, slice0, batch, X, stream); }
  if (rc) return rc;
  const size_t esz = 2;
  // Every LayerNorm but the first rides on the kernel in front of it (D = 384: the block tail; D = 768: the epilogue of the
  // residual GEMM, vittf_gemm_residual_ln: proj -> norm2, fc2 -> the next block's norm1).  VITTF_CFG_SEPARATE_LN: own launches.
  const bool res_ln = !(cfg->flags & VITTF_CFG_SEPARATE_LN) && (d == 384 || d == 768);
  // q pre-scaled by log2(e) / 8 in the qkv epilogue + the two-blocks-per-wave attention kernel; VITTF_CFG_UNSCALED_Q: q as the
  // model produces it and the online-maximum kernel.
  const int pre = (cfg->flags & VITTF_CFG_UNSCALED_Q) ? 0 : 1;
  // fp8 attention at D = 768 (BASELINE configs[3]): q and k leave the qkv GEMM as fp8 rows with their own block scales, v
  // with its absolute maxima collected on the way (vittf_gemm_qkv_fp8 + vittf_attention_fp8_rows) -- no absmax pass, a
  // quantise pass over the v third only.  VITTF_CFG_FP8_HEAD_SCALES: the per-(slice, head) scales of round 2 (three launches).
  const bool fp8_rows = !(cfg->flags & VITTF_CFG_FP8_HEAD_SCALES) && cfg->attention_fp8 && pre && d >= 768 && d % 256 == 0;
  // D = 384 with packed weights: the block tail (tail_fx.hip) and the activation-stationary qkv GEMM (gemm_as.hip)
  const bool tail = d == 384 && w->tail_packed && res_ln;
  const bool qkv_as = d == 384 && w->qkv_packed && !cfg->attention_fp8;
  for (int l = 0; l < L; ++l) {
    const char* qkv_w = (const char*)w->qkv_w + (size_t)l * 3 * d * d * esz;
    if (res_ln ? l == 0 : true) {
      ProfScope ps(VITTF_KERNEL_LAYERNORM, stream);
      rc = vittf_layernorm(X, w->ln1_g + (size_t)l * d, w->ln1_b + (size_t)l * d, H, rows, d, cfg->ln_eps, dt, stream);
      if (rc) return rc;
    }
    if (l == L - 1) {
      // hooked tensor, one third only: rows [part*D, (part+1)*D) of qkv.weight / qkv.bias  (infer.py:189-201)
      ProfScope ps(VITTF_KERNEL_GEMM, stream);   // the K-feature projection (tiled kernel)
      return vittf_gemm(H, qkv_w + (size_t)qkv_part * d * d * esz, w->qkv_b + (size_t)l * 3 * d + (size_t)qkv_part * d,
                        k_out, rows, d, d,
                        VITTF_EPI_KFEAT, tokens, dt, stream);
    }
    { ProfScope ps(VITTF_KERNEL_GEMM_QKV, stream);
      if (fp8_rows)
        rc = vittf_gemm_qkv_fp8(H, qkv_w, w->qkv_b + (size_t)l * 3 * d, QKV, rows, 3 * d, d, tokens, cfg->heads, dt,
                                base + lay.fp8, lay.fp8_bytes, stream);
      else if (qkv_as)
        rc = vittf_gemm_as(H, (const char*)w->qkv_packed + (size_t)l * 3 * d * d * esz, w->qkv_b + (size_t)l * 3 * d, QKV, rows, 3 * d, d,
                           pre ? VITTF_EPI_BIAS_QKV : VITTF_EPI_BIAS, dt, base + lay.tile_ctr, stream);
      else
        rc = vittf_gemm(H, qkv_w, w->qkv_b + (size_t)l * 3 * d, QKV, rows, 3 * d, d,
                        pre ? VITTF_EPI_BIAS_QKV : VITTF_EPI_BIAS, 0, dt, stream); }
    if (rc) return rc;
    { ProfScope ps(VITTF_KERNEL_ATTENTION, stream);
      if (fp8_rows)
        rc = vittf_attention_fp8_rows(QKV, O, batch, tokens, cfg->heads, dt, base + lay.fp8, lay.fp8_bytes, stream);
      else if (cfg->attention_fp8 && pre)
        rc = vittf_attention_fp8(QKV, O, batch, tokens, cfg->heads, dt, base + lay.fp8, lay.fp8_bytes, stream);
      else
        rc = vittf_attention(QKV, O, batch, tokens, cfg->heads, dt, pre, stream); }
    if (rc) return rc;
    if (tail) {
      // everything behind the attention -- proj, residual, norm2, fc1, GELU, fc2, residual, the next block's norm1 -- in one
      // launch: the residual rows are read once and written once, nothing else of it reaches HBM
      ProfScope ps(VITTF_KERNEL_MLP, stream);
      rc = vittf_block_tail(O, (const char*)w->tail_packed + (size_t)l * VITTF_TAIL_STEPS * 12288 * esz, w->proj_b + (size_t)l * d,
                            w->ln2_g + (size_t)l * d, w->ln2_b + (size_t)l * d, w->fc1_b + (size_t)l * 4 * d,
                            w->fc2_b + (size_t)l * d, X, rows, d, dt, w->ln1_g + (size_t)(l + 1) * d,
                            w->ln1_b + (size_t)(l + 1) * d, cfg->ln_eps, H, base + lay.tile_ctr, stream);
      if (rc) return rc;
      continue;
    }
    { ProfScope ps(VITTF_KERNEL_GEMM_PROJ, stream);
      if (res_ln)
        rc = vittf_gemm_residual_ln(O, (const char*)w->proj_w + (size_t)l * d * d * esz, w->proj_b + (size_t)l * d, X, rows, d,
                                    d, dt, w->ln2_g + (size_t)l * d, w->ln2_b + (size_t)l * d, cfg->ln_eps, H, stream);
      else
        rc = vittf_gemm(O, (const char*)w->proj_w + (size_t)l * d * d * esz, w->proj_b + (size_t)l * d, X, rows, d, d,
                        VITTF_EPI_BIAS_RESIDUAL, 0, dt, stream); }
    if (rc) return rc;
    if (!res_ln) {
      ProfScope ps(VITTF_KERNEL_LAYERNORM, stream);
      rc = vittf_layernorm(X, w->ln2_g + (size_t)l * d, w->ln2_b + (size_t)l * d, H, rows, d, cfg->ln_eps, dt, stream);
      if (rc) return rc;
    }
    { ProfScope ps(VITTF_KERNEL_GEMM_FC1, stream);
      rc = vittf_gemm(H, (const char*)w->fc1_w + (size_t)l * 4 * d * d * esz, w->fc1_b + (size_t)l * 4 * d, G, rows,
                      4 * d, d, VITTF_EPI_BIAS_GELU, 0, dt, stream); }
    if (rc) return rc;
    { ProfScope ps(VITTF_KERNEL_GEMM_FC2, stream);
      if (res_ln)   // (l + 1 < L always holds here: the last block returns above)
        rc = vittf_gemm_residual_ln(G, (const char*)w->fc2_w + (size_t)l * 4 * d * d * esz, w->fc2_b + (size_t)l * d, X, rows,
                                    d, 4 * d, dt, w->ln1_g + (size_t)(l + 1) * d, w->ln1_b + (size_t)(l + 1) * d,
                                    cfg->ln_eps, H, stream);
      else
        rc = vittf_gemm(G, (const char*)w->fc2_w + (size_t)l * 4 * d * d * esz, w->fc2_b + (size_t)l * d, X, rows, d,
                        4 * d, VITTF_EPI_BIAS_RESIDUAL, 0, dt, stream); }
    if (rc) return rc;
  }
  return VITTF_OK;
}

static const char* g_kernel_names[VITTF_KERNEL_CLASSES] = {};
void vittf_note_kernel(int cls, const char* name) {
  if (cls >= 0 && cls < VITTF_KERNEL_CLASSES) g_kernel_names[cls] = name;
}
extern "C" const char* vittf_profiler_kernel_name(int32_t cls) {
  return (cls >= 0 && cls < VITTF_KERNEL_CLASSES && g_kernel_names[cls]) ? g_kernel_names[cls] : "";
}

extern "C" int vittf_profiler_enable(int32_t on) {
  for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
  g_prof_recs.clear();
  g_prof_mask = (unsigned)on;
  return VITTF_OK;
}

extern "C" int vittf_profiler_collect(double* ms_per_class, int64_t* launches_per_class) {
  if (!ms_per_class || !launches_per_class) return VITTF_ERR_INVALID_ARG;
  for (int i = 0; i < VITTF_KERNEL_CLASSES; ++i) { ms_per_class[i] = 0.0; launches_per_class[i] = 0; }
  for (auto& r : g_prof_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return VITTF_ERR_LAUNCH;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return VITTF_ERR_LAUNCH;
    ms_per_class[r.cls] += ms;
    launches_per_class[r.cls] += 1;
  }
  return VITTF_OK;
}
