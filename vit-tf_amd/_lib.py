"""ctypes binding of libvittf.so (include/vittf.h).

The product has no CPU fallback: if the library is missing or no gfx950 device is visible the
calls below raise, they never route around the HIP path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libvittf.so')

BF16, FP16 = 0, 1
DTYPES = {'bf16': BF16, 'bfloat16': BF16, 'fp16': FP16, 'float16': FP16, 'half': FP16}
SAMPLE_MODES = {'nearest': 0, 'bilinear': 1, 'trilinear': 1}
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, EPI_KFEAT, EPI_BIAS_QKV = 0, 1, 2, 3, 4


class VitConfig(C.Structure):
    _fields_ = [('embed_dim', C.c_int32), ('depth', C.c_int32), ('heads', C.c_int32), ('patch', C.c_int32),
                ('dtype', C.c_int32), ('ln_eps', C.c_float), ('attention_fp8', C.c_int32), ('flags', C.c_int32)]


CFG_SEPARATE_LN, CFG_UNSCALED_Q, CFG_FP8_HEAD_SCALES = 1, 2, 4


class VitWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'pe_w_t', 'pe_b', 'qkv_w', 'qkv_b', 'proj_w', 'proj_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'tail_packed', 'qkv_packed',
        'ln1_g', 'ln1_b', 'ln2_g', 'ln2_b')]


class PosEmbed(C.Structure):
    _fields_ = [('cls_plus_pos0', C.c_void_p), ('patch_pos', C.c_void_p)]


class SliceView(C.Structure):
    _fields_ = [('vol', C.c_void_p), ('stride_slice', C.c_int64), ('stride_row', C.c_int64),
                ('stride_col', C.c_int64), ('in_rows', C.c_int32), ('in_cols', C.c_int32),
                ('out_rows', C.c_int32), ('out_cols', C.c_int32), ('minmax', C.c_void_p)]


class VittfError(RuntimeError):
    pass


class BilateralParams(C.Structure):
    _fields_ = [('sigma_spatial', C.c_double), ('lam', C.c_double), ('a_diag_min', C.c_double), ('cg_tol', C.c_double),
                ('cg_maxiter', C.c_int32), ('bistochastize_iters', C.c_int32), ('pad', C.c_int32),
                ('crop_threshold', C.c_float)]


_vp, _i32, _i64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
_P = C.POINTER

# name -> (restype, argtypes); must list every function include/vittf.h declares (tests check this)
SIGNATURES = {
    'vittf_abi_version': (C.c_int, []),
    'vittf_status_string': (C.c_char_p, [C.c_int]),
    'vittf_device_count': (C.c_int, []),
    'vittf_minmax_workspace_bytes': (_sz, []),
    'vittf_volume_minmax': (C.c_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    'vittf_vit_workspace_bytes': (_sz, [_P(VitConfig), _i32, _i32]),
    'vittf_vit_k_features': (C.c_int, [_P(VitConfig), _P(VitWeights), _P(PosEmbed), _P(SliceView), _i32, _i32, _i32,
                                       _vp, _vp, _sz, _vp]),
    'vittf_profiler_enable': (C.c_int, [_i32]),
    'vittf_profiler_collect': (C.c_int, [_P(C.c_double), _P(_i64)]),
    'vittf_profiler_kernel_name': (C.c_char_p, [_i32]),
    'vittf_patch_embed': (C.c_int, [_P(VitConfig), _P(VitWeights), _P(PosEmbed), _P(SliceView), _i32, _i32, _vp, _vp]),
    'vittf_layernorm': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, C.c_float, _i32, _vp]),
    'vittf_gemm': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    'vittf_gemm_residual_ln': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, C.c_float, _vp, _vp]),
    'vittf_block_tail_workspace_bytes': (_sz, []),
    'vittf_block_tail': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, C.c_float, _vp, _vp, _vp]),
    'vittf_gemm_as_workspace_bytes': (_sz, []),
    'vittf_gemm_as': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    'vittf_attention': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    'vittf_attention_rescale_count': (_i64, [_i32]),
    'vittf_attention_fp8_workspace_bytes': (_sz, [_i32, _i32, _i32]),
    'vittf_attention_fp8': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    'vittf_gemm_qkv_fp8': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    'vittf_attention_fp8_rows': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    'vittf_pool_slices': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _i64,
                                    _i64, _vp]),
    'vittf_assemble_sum': (C.c_int, [_vp, _vp, _vp, _i32, _P(_i32), _i32, _i32, _i32, _i32, _vp, _vp]),
    'vittf_sample_features': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    'vittf_voxel_norm': (C.c_int, [_vp, _i32, _i64, _vp, _vp]),
    'vittf_similarity_workspace_bytes': (_sz, [_i32, _i64, _i32]),
    'vittf_similarity': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _P(_i32), _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp,
                                   _sz, _vp]),
    'vittf_similarity_query_workspace_bytes': (_sz, [_i32, _i64, _i32, _i32]),
    'vittf_similarity_query': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _P(C.c_float), _P(_i32), _i32, _i32, _vp, _i32, _i32, _i32,
                                         _vp, _vp, _sz, _vp]),
    'vittf_similarity_maps_f32': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _P(_i32), _i32, _i32, C.c_float, _vp, _vp,
                                            _vp, _sz, _vp]),
    'vittf_topk_voxels': (C.c_int, [_vp, _i32, _i64, _i32, _vp, _vp]),
    'vittf_mean_pairwise_distance': (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    'vittf_erode_mask': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'vittf_surface_shell_workspace_bytes': (_sz, [_i32, _i32, _i32]),
    'vittf_surface_shell': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    'vittf_confusion_matrix': (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    'vittf_resize_nearest_u8': (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    'vittf_widen_f16': (C.c_int, [_vp, _i64, _vp, _vp]),
    'vittf_bilateral_workspace_bytes': (_sz, [_i32, _i32, _i32, C.c_double, _i32]),
    'vittf_bilateral_refine': (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _P(_i32), _i32,
                                         _P(BilateralParams), _vp, _P(_i32), _vp, _sz, _vp]),
    'vittf_quantize_wrap_u8': (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    'vittf_assign_labels': (C.c_int, [_vp, _i32, _i64, _P(_i32), _vp, _vp]),
}

_lib = None


def load():
    """Load libvittf.so (once).  Raises VittfError with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VittfError(
            f'{LIB_PATH} not found: the HIP library has not been built. Run '
            f'`python -c "import __graft_entry__ as g; g.build()"` (or `make -C vit-tf_amd/csrc`) first. '
            'There is no CPU fallback for this path.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.vittf_abi_version() != ABI_VERSION:
        raise VittfError('libvittf.so ABI version mismatch; rebuild it')
    _lib = lib
    return lib


KERNEL_CLASSES = ('patch_embed', 'layernorm', 'gemm', 'attention', 'mlp', 'gemm_qkv', 'gemm_proj', 'gemm_fc1', 'gemm_fc2',
                  'similarity')
ABI_VERSION = 6
QUERY_MAX_A = 64          # VITTF_QUERY_MAX_A: annotations of a vittf_similarity_query call


def profiler_enable(on=True, classes=None):
    """Record HIP events around the engine's launches: all kernel classes, or only `classes` (names of KERNEL_CLASSES)."""
    mask = 0
    if on:
        mask = -1 if classes is None else sum(1 << KERNEL_CLASSES.index(c) for c in classes)
    check(load().vittf_profiler_enable(mask), 'vittf_profiler_enable')


def kernel_name(kernel_class):
    """Name of the kernel the library launched last for 'attention' / 'similarity' ('' if none yet)."""
    return load().vittf_profiler_kernel_name(KERNEL_CLASSES.index(kernel_class)).decode()


def profiler_collect():
    """{class: (total ms, launches)} of the vittf_vit_k_features launches recorded since profiler_enable(True)."""
    ms = (C.c_double * len(KERNEL_CLASSES))()
    n = (C.c_int64 * len(KERNEL_CLASSES))()
    check(load().vittf_profiler_collect(ms, n), 'vittf_profiler_collect')
    return {k: (ms[i], n[i]) for i, k in enumerate(KERNEL_CLASSES)}


def check(rc, what=''):
    if rc != 0:
        msg = load().vittf_status_string(rc).decode()
        raise VittfError(f'{what or "vittf call"} failed: {msg} ({rc})')


def require_device():
    """Torch-visible GPU + library present; used by every product entry before touching the device."""
    import torch
    lib = load()
    if not torch.cuda.is_available():
        raise VittfError('no GPU visible to PyTorch-ROCm: the vit-tf hot path runs only on MI355X (gfx950); '
                         'there is no CPU fallback')
    return lib


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device (or host) address of a tensor as c_void_p; None stays NULL."""
    return None if t is None else C.c_void_p(t.data_ptr())
