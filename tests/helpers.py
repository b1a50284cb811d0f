"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

import vit_tf_amd as vt
from oracle import dino_vit, feature_volume as ofv

TINY_ARCH = (128, 3, 2, 8)


def load_golden(golden_dir, name):
    with np.load(os.path.join(golden_dir, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def tiny_model(seed):
    sd = vt.synthetic_state_dict(TINY_ARCH, seed)
    return dino_vit.build_vit(TINY_ARCH, sd), sd


def rel_fro(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def oracle_window(oracle, vol, axis, lo, hi, minmax, im_sz):
    """fp16 pooled features (D, f0, f1) of the window that averages slices [lo, hi) of `axis`: the oracle on those slices
    only, AdaptiveAvgPool3d's own fp16 arithmetic for the mean (infer.py:329)."""
    sl, (a, b) = ofv.AXIS_DIMS[axis]
    sub = vol.narrow(sl, lo, hi - lo).float()
    imgs = ofv.normalized_slices(sub, axis, minmax=minmax)
    rows, cols = ofv.axis_image_size(im_sz, axis)
    ks = []
    with torch.no_grad():
        for i in range(imgs.shape[0]):
            x = F.interpolate(imgs[i:i + 1], size=(rows, cols), mode='nearest')
            ks.append(ofv.k_tokens(oracle, x).half()[0, 1:])          # hook -> fp16, CLS dropped
    k = torch.stack(ks).view(len(ks), rows // 8, cols // 8, -1)      # (n, f0, f1, D)
    return ofv.adaptive_pool(k.permute(3, 1, 2, 0).contiguous(), (rows // 8, cols // 8, 1))[..., 0]


# The pooling windows of the 512^3 configurations that the full-size GPU tests compare with the oracle: (arch, weight seed, axis,
# window).  The oracle's pooled window (8 slices of 512 x 512 through the fp32 CPU ViT: 20-60 s each on the GPU box's host cores)
# is committed as a fixture, every 4th feature row / column of it (tests/golden/windows512.npz, made by
# tests/golden/make_window_goldens.py); the 256^3 configuration and the smaller pipeline tests keep running the oracle live.
WINDOWS512 = (('vits8', 0, 'z', 37), ('vits8', 0, 'y', 11), ('vits8', 0, 'x', 50), ('vitb8', 2, 'y', 29))
WINDOW_STRIDE = 4


def window_key(arch, seed, axis, w):
    return f'{arch}_seed{seed}_{axis}{w}'
