"""Oracle: slice-wise K-feature volume extraction (hot path A), CPU fp32.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates, function by function, what /root/reference/infer.py does between
loading a volume and saving ``{'k': fp16 (F, W', H', D')}``:

  sizing rule            infer.py:317-319
  per-axis tables        infer.py:138-152
  min-max + ImageNet     infer.py:32-34, 39-40, 154-155
  nearest resize + ViT   infer.py:173-177
  hook -> fp16           infer.py:133-135      (rounding point #1)
  K columns, CLS drop    infer.py:184-203
  adaptive avg-pool      infer.py:329, 203     (rounding point #2)
  fp16 z -> y -> x sum   infer.py:330-332      (rounding points #3, #4)

It is structured differently from the reference (no forward hook, no 3-channel
expand, the K third is computed directly) but is checked value-for-value
against the imported reference harness by tests/golden/make_golden.py.
"""
import torch
import torch.nn.functional as F

IN_MEAN = (0.485, 0.456, 0.406)   # infer.py:39
IN_STD = (0.229, 0.224, 0.225)    # infer.py:40

# axis -> (volume dim that is sliced, the two in-plane volume dims in image (row, col) order)
# z: image[s] = vol[:, :, s]  (rows = W, cols = H)      infer.py:139
# y: image[s] = vol[:, s, :]  (rows = W, cols = D)      infer.py:140
# x: image[s] = vol[s, :, :]  (rows = H, cols = D)      infer.py:141
AXIS_DIMS = {'z': (2, (0, 1)), 'y': (1, (0, 2)), 'x': (0, (1, 2))}


def sizing(vol_shape, feature_output_size, patch_size):
    """infer.py:317-319 -> (im_sz, feat_out_sz), both 3-tuples of int."""
    ref_fact = sorted(vol_shape[-3:])[1] / feature_output_size
    im_sz = tuple(int(patch_size * (d // ref_fact)) for d in vol_shape[-3:])
    feat_out_sz = tuple(d // patch_size for d in im_sz)
    return im_sz, feat_out_sz


def axis_image_size(im_sz, axis):
    """infer.py:143-147."""
    _, (a, b) = AXIS_DIMS[axis]
    return im_sz[a], im_sz[b]


def normalized_slices(vol, axis, minmax=None):
    """(S, 3, rows, cols) fp32: global min-max to [0, 1], then per-channel ImageNet mean/std.

    infer.py:137, 154-155 (``normalize`` is torchvision's (x - mean[c]) / std[c]).
    ``minmax``: (lo, hi) of the WHOLE volume when ``vol`` is only a slab of it (a few slices of a 512^3 volume).
    """
    vol = vol.float().squeeze()
    sl, (a, b) = AXIS_DIMS[axis]
    img = vol.permute(sl, a, b)                       # (S, rows, cols)
    lo, hi = (vol.min(), vol.max()) if minmax is None else minmax
    img = (img - lo) / (hi - lo)
    mean = torch.tensor(IN_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IN_STD).view(1, 3, 1, 1)
    return (img[:, None] - mean) / std


def k_tokens(model, images):
    """Output of blocks[-1].attn.qkv restricted to its K third: (B, N, D) fp32.

    Equivalent to the hooked tensor's columns [D, 2D) (infer.py:133-135, 189-201).
    """
    return model.last_block_k(images)


def k_features_axis(vol, model, patch_size, im_sizes, axis, batch_size=1):
    """Un-pooled K features of one axis in the reference's output layout.

    Returns fp16 (D, *spatial) where spatial is the feature grid with the slice axis at its
    volume position and at FULL slice resolution -- e.g. z: (D, W', H', S)  (infer.py:201-203).
    """
    imgs = normalized_slices(vol, axis)
    rows, cols = axis_image_size(im_sizes, axis)
    f0, f1 = rows // patch_size, cols // patch_size
    out = []
    with torch.no_grad():
        for idx in torch.arange(imgs.shape[0]).split(batch_size):
            x = F.interpolate(imgs[idx], size=(rows, cols), mode='nearest')
            k = k_tokens(model, x).half()             # rounding point #1 (hook: .cpu().half())
            out.append(k[:, 1:])                      # drop CLS
    k = torch.cat(out)                                # (S, f0*f1, D)
    k = k.view(k.shape[0], f0, f1, -1)                # token order: row-major over (rows, cols)
    sl, (a, b) = AXIS_DIMS[axis]
    # place (S, f0, f1) at volume dims (sl, a, b), features first
    order = [None, None, None]
    order[sl], order[a], order[b] = 0, 1, 2
    return k.permute(3, *order).contiguous()


def adaptive_pool(feat, feat_out_sz):
    """AdaptiveAvgPool3d on an unbatched (C, D, H, W) fp16 tensor (infer.py:329, 203)."""
    return F.adaptive_avg_pool3d(feat, feat_out_sz)


def feature_volume(vol, model, patch_size, feature_output_size=64, slice_along='all', batch_size=1):
    """What infer.py's __main__ saves under key 'k' (infer.py:325-333)."""
    im_sz, feat_out_sz = sizing(tuple(vol.shape), feature_output_size, patch_size)
    if slice_along in ('x', 'y', 'z'):
        return k_features_axis(vol, model, patch_size, im_sz, slice_along, batch_size)
    acc = 0.0
    for ax in ('z', 'y', 'x'):
        pooled = adaptive_pool(k_features_axis(vol, model, patch_size, im_sz, ax, batch_size), feat_out_sz)
        acc = (torch.as_tensor(acc) + pooled.squeeze().half())   # fp16 add, rounding #3/#4
    return acc


def pool_windows(n_in, n_out):
    """Adaptive-pool window i = [floor(i*in/out), ceil((i+1)*in/out)) -- for tests."""
    return [((i * n_in) // n_out, -((-(i + 1) * n_in) // n_out)) for i in range(n_out)]
