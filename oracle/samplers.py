"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the annotation samplers feeding ``predict_ntf --num-samples``
(compare_feat_sampling.py:13-33): torch + scipy.ndimage, the reference's own dependencies.  The product
(vit-tf_amd/samplers.py) builds the candidate sets on the GPU; tests compare its masks with these bit for bit."""
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


def surface_shell(vol, dist_from_surface=4):
    """(:19-24) the mask eroded by generate_binary_structure(3, dist_from_surface), minus its 6-neighbour erosion."""
    outer = binary_erosion(np.asarray(vol), generate_binary_structure(rank=3, connectivity=dist_from_surface))
    inner = binary_erosion(outer, generate_binary_structure(rank=3, connectivity=1))
    return np.logical_xor(inner, outer)


def sample_surface(vol, n_samples, dist_from_surface=4):
    """Voxels of the one-voxel shell just inside the mask."""
    shell = torch.as_tensor(surface_shell(vol, dist_from_surface)).nonzero()
    if shell.shape[0] > n_samples:
        return _pick(shell, n_samples)
    print(f'Full surface only has {shell.shape[0]} voxels (< n_samples={n_samples}).')
    return shell


def sample_both(vol, n_samples, dist_from_surface=4, thin_to_reasonable=False):
    half = n_samples // 2
    return torch.cat([sample_uniform(vol, half, thin_to_reasonable=thin_to_reasonable),
                      sample_surface(vol, half, dist_from_surface=dist_from_surface)])
