"""Oracle: query sampling, similarity maps and label assignment (hot path B), CPU fp32.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

  sample_features          infer.py:48-72         (F.grid_sample, align_corners=False, zeros)
  similarity_maps          predict_ntf.py:24-72, 95-100  (no bilateral solver)
  assign_labels            predict_ntf.py:203-215
  evaluate_predictions     evaluate_similarities.py:57-81 (scoring only)

``similarity_maps`` computes the mathematically intended per-class map also when
there is a single annotation in total; the reference's ``.squeeze(1)``
(predict_ntf.py:65) drops the annotation axis in that case and produces a
constant volume.  Golden vectors use >= 2 annotations.

The reference's uint8 quantiser ``(255 / (0.99 * max) * sim).to(uint8)``
(predict_ntf.py:98-99) produces values up to 257.57; on the x86 CPU the cast
wraps (257 -> 1).  ``quantize_u8`` reproduces that: truncate toward zero, then
mod 256.
"""
import numpy as np
import torch
import torch.nn.functional as F

CT_ORG_THRESHOLDS = (0.486, 0.264, 0.236, 0.68, 0.291)   # predict_ntf.py:208


def rel_coords(abs_coords, vol_extent):
    """predict_ntf.py:56 -- voxel index -> [-1, 1] with voxel centres at (i + 0.5) / extent."""
    ext = torch.as_tensor(vol_extent, dtype=torch.float32).view(1, 3)
    return (abs_coords.float() + 0.5) / ext * 2.0 - 1.0


def sample_features(feat_vol, rel, mode='bilinear'):
    """infer.py:48-72 for a single modality: feat_vol (F, W, H, D), rel (A, 3) -> (A, F).

    grid_sample's last grid dim is (x, y, z) = (D, H, W) order, hence the flip (infer.py:67).
    Does not mutate ``rel`` (the reference does, infer.py:66).
    """
    grid = rel.flip(-1).to(feat_vol.dtype).view(1, 1, -1, 1, 3)
    out = F.grid_sample(feat_vol[None], grid, mode=mode, align_corners=False)   # (1, F, 1, A, 1)
    return out[0, :, 0, :, 0].t().contiguous()


def quantize_u8(sim):
    """(255 / (0.99 * max) * sim).to(uint8) with x86 wrap-around (predict_ntf.py:98-99)."""
    quant = 0.99 * sim.max()
    scaled = (255.0 / quant * sim)
    return (scaled.to(torch.int64) % 256).to(torch.uint8), scaled


def class_similarity(features, qf_class, big_a_mean=False):
    """One class: features (F, W, H, D) fp32, qf_class (a, F) -> fp32 (W, H, D) before quantising.

    predict_ntf.py:65, 71-72: dot products, keep >= 0.25, ** 2.5, mean over the class's
    annotations.  ``big_a_mean`` is the single-class A > 1024 variant (predict_ntf.py:62-63):
    mean of the raw dots first, then threshold / power on that mean.
    """
    fdim = features.shape[0]
    flat = features.reshape(fdim, -1)
    dots = qf_class @ flat                                                    # (a, Nvox)
    if big_a_mean:
        dots = dots.sum(0, keepdim=True) / qf_class.shape[0]
    s = torch.where(dots >= 0.25, dots, torch.zeros(1)) ** 2.5
    return s.mean(0).reshape(features.shape[1:])


def class_maps_fp32(volume_shape, features, annotations, normalize=False):
    """Per-class fp32 maps at the feature-grid resolution, before quantisation (predict_ntf.py:52-72)."""
    coords = torch.cat([torch.as_tensor(v) for v in annotations.values()])
    in_dims = tuple(volume_shape[-3:])
    if normalize:
        features = F.normalize(features, dim=0)
    qf = sample_features(features, rel_coords(coords, in_dims), 'bilinear')   # (A, F)
    big = len(annotations) == 1 and coords.shape[0] > 1024
    out, start = {}, 0
    for name, v in annotations.items():
        n = torch.as_tensor(v).shape[0]
        out[name] = class_similarity(features, qf[start:start + n], big_a_mean=big)
        start += n
    return out


def similarity_maps(volume_shape, features, annotations, normalize=False, volume=None):
    """predict_ntf.compute_similarities.
    normalize=True applies F.normalize(features, dim=0) first (compare_feat_sampling.py:45, tests/test_vishum.py:12).
    volume: the (W, H, D) volume itself -> the bilateral-solver branch (predict_ntf.py:73-96, oracle/bilateral.py);
    None -> the plain branch (quantise, then nearest resize; predict_ntf.py:97-100).

    volume_shape: (W, H, D) of the full volume; features: (F, W', H', D') fp32;
    annotations: {name: (n, 3) integer voxel coords}.  Returns {name: uint8 (W//2, H//2, D//2)}.
    """
    if len(annotations) == 0:
        return None
    coords = torch.cat([torch.as_tensor(v) for v in annotations.values()])
    if coords.numel() == 0:
        return None
    sim_shape = tuple(d // 2 for d in tuple(volume_shape[-3:]))
    out = {}
    for name, sim in class_maps_fp32(volume_shape, features, annotations, normalize).items():
        if volume is not None:
            from . import bilateral
            q, _ = quantize_u8(bilateral.refine_similarity(sim, volume, sim_shape))
            out[name] = q
        else:
            q, _ = quantize_u8(sim)
            out[name] = F.interpolate(q[None, None], sim_shape, mode='nearest')[0, 0]
    return out


def resample_topk(feat_vol, sims, K=8, similarity_exponent=2.0, mode='nearest'):
    """infer.py:75-106.  feat_vol (M, F, W, H, D) fp32; sims (M, C, A, W, H, D).  Returns (M, C, A, W, H, D)."""
    m, c, a = sims.shape[:3]
    dims = sims.shape[-3:]
    out = torch.empty((m, c, a, *dims), dtype=torch.float32)
    ext = torch.tensor(list(dims), dtype=torch.float32)
    for mi in range(m):
        flat = feat_vol[mi].reshape(feat_vol.shape[1], -1).float()
        for ci in range(c):
            for ai in range(a):
                s = sims[mi, ci, ai]
                kth = torch.topk(s.flatten(), K).values[-1]
                top = (s >= kth).nonzero()[:K]                                   # first K in index order
                rel = (top.float() + 0.5) / ext * 2.0 - 1.0
                qf = sample_features(feat_vol[mi].float(), rel, mode)             # (K, F)
                out[mi, ci, ai] = ((qf @ flat).clamp(0, 1) ** similarity_exponent).mean(0).reshape(dims)
    return out


def mean_pairwise_distance(features, measure='cosine'):
    """infer.py:118-121: the per-row distance take_most_dissimilar ranks by."""
    if measure == 'cosine':
        return 1 - F.cosine_similarity(features[None], features[:, None], dim=-1).mean(0)
    return torch.cdist(features[None], features[None])[0].mean(0)


def take_most_dissimilar(features, num_prototypes=35, measure='cosine'):
    """infer.py:108-126; the order of the returned rows is unspecified (topk(sorted=False)): compare as a set."""
    if features.shape[0] <= num_prototypes:
        return features
    dist = mean_pairwise_distance(features, measure)
    return features[torch.topk(dist, num_prototypes, largest=True, sorted=False).indices]


def assign_labels(sims, thresholds=CT_ORG_THRESHOLDS):
    """predict_ntf.py:203-215: per-class threshold + running maximum -> uint8 labels.

    sims: list of uint8 (or float-valued) volumes in annotation order.  At most
    len(thresholds) classes are used (the reference zips with the 5 CT-ORG names).
    """
    sims = [s.float() for s in sims]
    pred = torch.zeros_like(sims[0])
    best = torch.zeros_like(sims[0])
    for i, (thr, sim) in enumerate(zip(thresholds, sims)):
        mask = (sim > int(thr * 255)) & (sim > best)
        pred[mask] = i + 1
        best[mask] = sim[mask]
    return pred.numpy().astype(np.uint8)


def evaluate_predictions(pred, target):
    """Binary-mask scores in the shape evaluate_similarities.py:65-68 reports (no sklearn needed).

    pred, target: uint8/bool arrays of equal size, values {0, 1}.  Returns per-class
    [background, foreground] precision / recall / f1 / iou, the 2x2 confusion matrix and accuracy.
    """
    p = np.asarray(pred).reshape(-1).astype(np.int64)
    t = np.asarray(target).reshape(-1).astype(np.int64)
    cm = np.zeros((2, 2), dtype=np.int64)
    np.add.at(cm, (t, p), 1)
    tp = np.diag(cm).astype(np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        prec = np.nan_to_num(tp / cm.sum(0))
        rec = np.nan_to_num(tp / cm.sum(1))
        f1 = np.nan_to_num(2 * prec * rec / (prec + rec))
        iou = np.nan_to_num(tp / (cm.sum(0) + cm.sum(1) - tp))
    return {'accuracy': float(tp.sum() / cm.sum()), 'precision': prec.tolist(), 'recall': rec.tolist(),
            'f1': f1.tolist(), 'iou': iou.tolist(), 'confusion_matrix': cm.tolist()}
