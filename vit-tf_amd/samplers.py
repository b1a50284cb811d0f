"""Annotation samplers feeding ``predict_ntf --num-samples`` (compare_feat_sampling.py:13-33).
Input selection on the host (torch + scipy.ndimage); not part of the GPU path."""
import numpy as np
import torch
from scipy.ndimage import binary_erosion, generate_binary_structure


def _pick(idxs, n):
    """n rows without replacement, uniform (torch.multinomial over equal weights, like the reference)."""
    w = torch.ones(idxs.shape[0])
    return idxs[torch.multinomial(w, n)]


def sample_uniform(vol, n_samples, thin_to_reasonable=False):
    idxs = torch.as_tensor(vol).nonzero()
    while thin_to_reasonable and idxs.shape[0] > 2 ** 24:      # multinomial's category limit
        idxs = idxs[::2]
    return _pick(idxs, n_samples)


def sample_surface(vol, n_samples, dist_from_surface=4):
    """Voxels of the one-voxel shell `dist_from_surface` erosions inside the mask."""
    outer = binary_erosion(np.asarray(vol), generate_binary_structure(rank=3, connectivity=dist_from_surface))
    inner = binary_erosion(outer, generate_binary_structure(rank=3, connectivity=1))
    shell = torch.as_tensor(np.logical_xor(inner, outer)).nonzero()
    if shell.shape[0] > n_samples:
        return _pick(shell, n_samples)
    print(f'Full surface only has {shell.shape[0]} voxels (< n_samples={n_samples}).')
    return shell


def sample_both(vol, n_samples, dist_from_surface=4, thin_to_reasonable=False):
    half = n_samples // 2
    return torch.cat([sample_uniform(vol, half, thin_to_reasonable=thin_to_reasonable),
                      sample_surface(vol, half, dist_from_surface=dist_from_surface)])
