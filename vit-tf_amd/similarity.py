"""Similarity queries against a feature volume, on the GPU through libvittf.

Host-side mirror of sample_features3d (infer.py:48-72), compute_similarities (predict_ntf.py:24-101,
without the bilateral solver) and the label assignment of predict_ntf.py:203-215.  Same argument
meaning and return types as the reference functions; the arithmetic is in similarity.hip.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

CT_ORG_THRESHOLDS = (0.486, 0.264, 0.236, 0.68, 0.291)   # predict_ntf.py:208


def _device_features(features, device):
    """Feature volume as a contiguous fp16 (F, n0, n1, n2) device tensor.  Feature files hold fp16
    (infer.py:134, 337-340), so fp32 inputs that came from such a file convert back exactly."""
    t = torch.as_tensor(features).squeeze()
    if t.ndim != 4:
        raise ValueError(f'features must be (F, W, H, D), got {tuple(t.shape)}')
    return t.to(device=device, dtype=torch.float16).contiguous()


def voxel_norms(feat):
    """max(|feat[:, v]|, 1e-12) per voxel of a contiguous fp16 (F, n0, n1, n2) device tensor: fp32 (n0, n1, n2)."""
    lib = _lib.require_device()
    out = torch.empty(feat.shape[1:], dtype=torch.float32, device=feat.device)
    _lib.check(lib.vittf_voxel_norm(_lib.ptr(feat), feat.shape[0], out.numel(), _lib.ptr(out), _lib.stream_ptr()),
               'vittf_voxel_norm')
    return out


def sample_features3d(feat_vol, rel_coords, mode='nearest'):
    """infer.py:48-72.  feat_vol ([M,] F, W, H, D); rel_coords ([M,] C, A, 3) in [-1, 1], volume dim
    order.  Returns ([M,] C, A, F) on the device, fp32.  Unlike the reference it does not mutate `rel_coords`."""
    lib = _lib.require_device()
    dev = feat_vol.device if isinstance(feat_vol, torch.Tensor) and feat_vol.is_cuda else torch.device('cuda', torch.cuda.current_device())
    fv = torch.as_tensor(feat_vol)
    if fv.ndim == 4:
        fv = fv[None]
    rc = torch.as_tensor(rel_coords)
    while rc.ndim < 4:
        rc = rc[None]
    if rc.shape[0] != fv.shape[0]:
        rc = rc.expand(fv.shape[0], -1, -1, -1)
    m, c, a, _ = rc.shape
    out = torch.empty((m, c, a, fv.shape[1]), dtype=torch.float32, device=dev)
    for i in range(m):
        f = fv[i]
        is_half = f.dtype == torch.float16
        f = f.to(dev, torch.float16 if is_half else torch.float32).contiguous()
        rel = rc[i].reshape(-1, 3).to(dev, torch.float32).contiguous()
        _lib.check(lib.vittf_sample_features(_lib.ptr(f), int(is_half), f.shape[0], f.shape[1], f.shape[2], f.shape[3],
                                             _lib.ptr(rel), rel.shape[0], _lib.SAMPLE_MODES[mode], None, _lib.ptr(out[i]),
                                             _lib.stream_ptr()), 'vittf_sample_features')
    return out


def compute_similarities(volume, features, annotations, bilateral_solver=False, device=None, normalize=False,
                         keep_on_device=False, voxel_norm=None):
    """predict_ntf.py:24-101.  volume: (W, H, D) array/tensor (only its shape is used unless bilateral_solver);
    features: (F, W', H', D'); annotations: {name: (n, 3) voxel coords}.
    bilateral_solver=True: every class map is refined by the 3-D bilateral solver against the volume
    (predict_ntf.py:73-96, bilateral.py) before quantisation, instead of the nearest resize.
    Returns {name: uint8 CPU tensor (W//2, H//2, D//2)} with one entry per key of `annotations`, in its order; None
    when there is nothing to query (predict_ntf.py:51-55).  A class with zero annotations keeps its key with an all-zero
    map, as in the reference (the mean over an empty slice is NaN, which its uint8 conversion turns into 0,
    predict_ntf.py:46-49, 70-72, 99), so the label ids of the later classes do not shift (predict_ntf.py:203-215).
    normalize=True: cosine similarity -- the volume is L2-normalised per voxel first, as
    compare_feat_sampling.py:45 and tests/test_vishum.py:12 do (predict_ntf.py itself does not).
    voxel_norm: voxel_norms(features) computed earlier (an interactive session queries one volume many times);
    implies normalize.
    keep_on_device=True: the maps stay on the GPU (for assign_labels / further kernels) instead of the reference's
    CPU tensors: at 512^3 the (256, 256, 256) uint8 maps are 17 MB per class, 2 ms of copies for a 0.3 ms query."""
    if len(annotations) == 0:
        return None
    names = [k for k, v in annotations.items() if torch.as_tensor(v).shape[0] > 0]
    if not names:
        return None
    lib = _lib.require_device()
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    feat = _device_features(features, dev)
    f, n0, n1, n2 = feat.shape
    in_dims = tuple(int(s) for s in tuple(volume.shape)[-3:])
    sim_shape = tuple(d // 2 for d in in_dims)

    coords = torch.cat([torch.as_tensor(annotations[k]).reshape(-1, 3) for k in names])
    a_total = coords.shape[0]
    if voxel_norm is not None:
        vnorm = torch.as_tensor(voxel_norm).to(dev, torch.float32).contiguous()
        if vnorm.numel() != n0 * n1 * n2:
            raise ValueError(f'voxel_norm has {vnorm.numel()} entries for {n0 * n1 * n2} voxels')
    else:
        vnorm = voxel_norms(feat) if normalize else None
    counts = [int(torch.as_tensor(annotations[k]).reshape(-1, 3).shape[0]) for k in names]
    starts = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
    big = int(len(annotations) == 1 and counts[0] > 1024)      # predict_ntf.py:62
    nclass = len(names)
    # rel = (abs + 0.5) / extent * 2 - 1 in fp32, exactly the reference expression (predict_ntf.py:56)
    if not bilateral_solver and a_total <= _lib.QUERY_MAX_A and nclass <= 256:
        # the interactive query: ONE library call -- the coordinates travel as a kernel argument (numpy float32 here: the same
        # correctly rounded operations in the same order as the torch expression below), three launches, no copy, no memset
        rel_h = np.ascontiguousarray(((coords.numpy().astype(np.float32) + np.float32(0.5)) / np.asarray([in_dims], np.float32)
                                      * np.float32(2.0) - np.float32(1.0)), dtype=np.float32)
        ws_bytes = lib.vittf_similarity_query_workspace_bytes(nclass, n0 * n1 * n2, a_total, f)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        out = torch.empty((nclass, *sim_shape), dtype=torch.uint8, device=dev)
        _lib.check(lib.vittf_similarity_query(_lib.ptr(feat), f, n0, n1, n2, rel_h.ctypes.data_as(C.POINTER(C.c_float)),
                                              starts.ctypes.data_as(C.POINTER(C.c_int32)), nclass, big, _lib.ptr(vnorm),
                                              sim_shape[0], sim_shape[1], sim_shape[2], _lib.ptr(out), _lib.ptr(ws), ws_bytes,
                                              _lib.stream_ptr()), 'vittf_similarity_query')
        host = out if keep_on_device else _to_host(out)
        return _with_empty_classes({k: host[i] for i, k in enumerate(names)}, annotations, sim_shape, dev, keep_on_device)
    ext = torch.tensor([list(in_dims)], dtype=torch.float32)
    rel = ((coords.float() + 0.5) / ext * 2.0 - 1.0).to(dev).contiguous()
    qf = torch.empty((a_total, f), dtype=torch.float32, device=dev)
    _lib.check(lib.vittf_sample_features(_lib.ptr(feat), 1, f, n0, n1, n2, _lib.ptr(rel), a_total,
                                         _lib.SAMPLE_MODES['bilinear'], _lib.ptr(vnorm), _lib.ptr(qf), _lib.stream_ptr()),
               'vittf_sample_features')

    ws_bytes = lib.vittf_similarity_workspace_bytes(nclass, n0 * n1 * n2, a_total)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    if bilateral_solver:
        from . import bilateral
        maps = torch.empty((nclass, n0, n1, n2), dtype=torch.float32, device=dev)
        _lib.check(lib.vittf_similarity_maps_f32(_lib.ptr(feat), 1, f, n0, n1, n2, _lib.ptr(qf),
                                                 starts.ctypes.data_as(C.POINTER(C.c_int32)), nclass, big, 0.0,
                                                 _lib.ptr(vnorm), _lib.ptr(maps), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()),
                   'vittf_similarity_maps_f32')
        vol = torch.as_tensor(np.asarray(volume, dtype=np.float32) if not isinstance(volume, torch.Tensor) else volume)
        vol = vol.squeeze().to(device=dev, dtype=torch.float32).contiguous()
        res = {}
        for i, k in enumerate(names):
            refined = bilateral.refine_similarity(maps[i], vol, sim_shape)
            q = bilateral.quantize_u8(refined)
            res[k] = q if keep_on_device else _to_host(q)
        return _with_empty_classes(res, annotations, sim_shape, dev, keep_on_device)
    out = torch.empty((nclass, *sim_shape), dtype=torch.uint8, device=dev)
    _lib.check(lib.vittf_similarity(_lib.ptr(feat), f, n0, n1, n2, _lib.ptr(qf),
                                    starts.ctypes.data_as(C.POINTER(C.c_int32)), nclass, big, _lib.ptr(vnorm),
                                    sim_shape[0], sim_shape[1], sim_shape[2], _lib.ptr(out), _lib.ptr(ws), ws_bytes,
                                    _lib.stream_ptr()), 'vittf_similarity')
    host = out if keep_on_device else _to_host(out)
    return _with_empty_classes({k: host[i] for i, k in enumerate(names)}, annotations, sim_shape, dev, keep_on_device)


def _to_host(t):
    """Device tensor -> CPU tensor through PINNED host memory from torch's caching host allocator: one asynchronous copy at
    the PCIe rate and one stream synchronisation.  (`t.cpu()` lands in freshly mapped pageable memory: the 16.7 MB map of a
    512^3 volume then costs 1.9 ms of page faults and staging against 0.35 ms for the copy itself.)  The result is an
    ordinary CPU tensor owned by the caller; its block returns to the allocator's cache when the caller drops it."""
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host


def _with_empty_classes(res, annotations, sim_shape, dev, keep_on_device):
    """One entry per annotation key, in annotation order: classes without annotations get an all-zero uint8 map."""
    if len(res) == len(annotations):
        return res
    zero = torch.zeros(sim_shape, dtype=torch.uint8, device=dev if keep_on_device else 'cpu')
    return {k: res[k] if k in res else zero.clone() for k in annotations}


def assign_labels(similarities, thresholds=CT_ORG_THRESHOLDS, device=None):
    """predict_ntf.py:203-215: list/dict of uint8 class maps (annotation order; CPU or GPU tensors / arrays) -> uint8
    label volume (numpy)."""
    lib = _lib.require_device()
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    maps = list(similarities.values()) if isinstance(similarities, dict) else list(similarities)
    maps = maps[:len(thresholds)]                       # zip() with the 5 CT-ORG names, predict_ntf.py:211
    sims = torch.stack([torch.as_tensor(m).to(dev, torch.uint8) for m in maps]).contiguous()
    n = sims[0].numel()
    thr = (C.c_int32 * len(maps))(*[int(t * 255) for t in thresholds[:len(maps)]])
    labels = torch.empty(sims.shape[1:], dtype=torch.uint8, device=dev)
    _lib.check(lib.vittf_assign_labels(_lib.ptr(sims), len(maps), n, thr, _lib.ptr(labels), _lib.stream_ptr()),
               'vittf_assign_labels')
    return _to_host(labels).numpy()


def resample_topk(feat_vol, sims, K=8, similarity_exponent=2.0, feature_sampling_mode='nearest'):
    """infer.py:75-106: re-sample the feature volume at the K most similar voxels of every (class, annotation) map and
    average the K new similarity maps clamp(feat . q, 0, 1) ** exponent.
    feat_vol ([M,] F, W, H, D) fp16 / fp32 (normalised, as the caller of the reference passes it); sims ([M,] C, A, W, H, D).
    Returns ([M,] C, A, W, H, D) in feat_vol's dtype, on the GPU."""
    lib = _lib.require_device()
    dev = torch.device('cuda', torch.cuda.current_device())
    fv = torch.as_tensor(feat_vol)
    sv = torch.as_tensor(sims)
    if fv.ndim == 4:
        fv = fv[None]
    if sv.ndim == 5:
        sv = sv[None]
    m, c, a = sv.shape[:3]
    dims = tuple(int(d) for d in sv.shape[-3:])
    if tuple(fv.shape[-3:]) != dims or fv.shape[0] != m:
        raise ValueError(f'feature volume {tuple(fv.shape)} and similarity volume {tuple(sv.shape)} do not match')
    nvox = dims[0] * dims[1] * dims[2]
    f = fv.shape[1]
    is_half = fv.dtype == torch.float16
    fv = fv.to(dev, torch.float16 if is_half else torch.float32).contiguous()
    sv = sv.to(dev, torch.float32).contiguous()
    idx = torch.empty((m * c * a, K), dtype=torch.int32, device=dev)
    _lib.check(lib.vittf_topk_voxels(_lib.ptr(sv), m * c * a, nvox, K, _lib.ptr(idx), _lib.stream_ptr()), 'vittf_topk_voxels')
    flat = idx.to(torch.int64)
    coords = torch.stack((flat // (dims[1] * dims[2]), (flat // dims[2]) % dims[1], flat % dims[2]), -1).float()
    rel = ((coords + 0.5) / torch.tensor(dims, dtype=torch.float32, device=dev) * 2.0 - 1.0).reshape(m, c * a * K, 3)
    out = torch.empty((m, c * a, nvox), dtype=torch.float32, device=dev)
    groups = c * a
    starts = (np.arange(groups + 1) * K).astype(np.int32)
    ws_bytes = lib.vittf_similarity_workspace_bytes(groups, 0, groups * K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    for i in range(m):
        qf = torch.empty((groups * K, f), dtype=torch.float32, device=dev)
        reli = rel[i].contiguous()
        _lib.check(lib.vittf_sample_features(_lib.ptr(fv[i]), int(is_half), f, dims[0], dims[1], dims[2], _lib.ptr(reli),
                                             groups * K, _lib.SAMPLE_MODES[feature_sampling_mode], None, _lib.ptr(qf),
                                             _lib.stream_ptr()), 'vittf_sample_features')
        _lib.check(lib.vittf_similarity_maps_f32(_lib.ptr(fv[i]), int(is_half), f, dims[0], dims[1], dims[2], _lib.ptr(qf),
                                                 starts.ctypes.data_as(C.POINTER(C.c_int32)), groups, 2,
                                                 float(similarity_exponent), None, _lib.ptr(out[i]), _lib.ptr(ws), ws_bytes,
                                                 _lib.stream_ptr()), 'vittf_similarity_maps_f32')
    out = out.reshape(m, c, a, *dims).to(torch.float16 if is_half else torch.float32)
    return out          # always with the M dimension, like the reference (it unsqueezes 5-D sims in place)


def take_most_dissimilar(features, num_prototypes=35, measure='cosine'):
    """infer.py:108-126: the num_prototypes rows of features (N, F) with the largest mean distance to all rows."""
    feats = torch.as_tensor(features)
    if feats.shape[0] <= num_prototypes:
        return feats
    if measure not in ('cosine', 'euclidean'):
        raise ValueError(f'Unknown measure: {measure}')
    lib = _lib.require_device()
    dev = torch.device('cuda', torch.cuda.current_device())
    x = feats.to(dev, torch.float32).contiguous()
    n, f = x.shape
    dist = torch.empty((1, n), dtype=torch.float32, device=dev)
    _lib.check(lib.vittf_mean_pairwise_distance(_lib.ptr(x), n, f, 0 if measure == 'cosine' else 1, _lib.ptr(dist),
                                                _lib.stream_ptr()), 'vittf_mean_pairwise_distance')
    sel = torch.empty((1, num_prototypes), dtype=torch.int32, device=dev)
    _lib.check(lib.vittf_topk_voxels(_lib.ptr(dist), 1, n, num_prototypes, _lib.ptr(sel), _lib.stream_ptr()), 'vittf_topk_voxels')
    return feats[sel[0].to(torch.int64).to(feats.device)]
