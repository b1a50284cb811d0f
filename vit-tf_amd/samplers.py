"""Annotation samplers feeding ``predict_ntf --num-samples`` (compare_feat_sampling.py:13-33), candidate sets built on
the GPU: the class mask, its eroded one-voxel shell (vittf_surface_shell replaces two scipy binary_erosion passes that
take seconds per class at 512^3) and the voxel lists; the draw itself is torch.multinomial over equal weights, as in
the reference, on the device (its random stream differs from the CPU generator's -- as it would for the reference).

`vol` is a boolean / 0-1 mask like in the reference, or -- with `class_id` -- a uint8 label volume that may already
live on the GPU (`device_labels`), so that all classes are drawn from one upload."""
import numpy as np
import torch

from . import _lib


def device_labels(labels):
    """uint8 copy of a label volume (numpy / tensor, values 0..255) on the current GPU."""
    _lib.require_device()
    t = torch.as_tensor(np.ascontiguousarray(labels) if isinstance(labels, np.ndarray) else labels)
    if t.ndim != 3:
        raise ValueError(f'expected a 3-D volume, got {tuple(t.shape)}')
    return t.to(torch.device('cuda', torch.cuda.current_device()), torch.uint8).contiguous()


def _as_labels(vol, class_id):
    if class_id is None:          # a mask: any non-zero voxel belongs to the set
        t = torch.as_tensor(np.ascontiguousarray(vol) if isinstance(vol, np.ndarray) else vol)
        return device_labels(t != 0), 1
    t = vol if (torch.is_tensor(vol) and vol.is_cuda and vol.dtype == torch.uint8 and vol.is_contiguous()) else device_labels(vol)
    return t, int(class_id)


def _pick(idxs, n):
    """n rows without replacement, uniform (torch.multinomial over equal weights, like the reference)."""
    w = torch.ones(idxs.shape[0], device=idxs.device)
    return idxs[torch.multinomial(w, n)]


def surface_shell(vol, dist_from_surface=4, class_id=None):
    """uint8 0/1 device volume: the set eroded by generate_binary_structure(3, dist_from_surface) XOR that eroded once
    more by the 6-neighbour element (compare_feat_sampling.py:20-24)."""
    lib = _lib.require_device()
    lab, cls = _as_labels(vol, class_id)
    n0, n1, n2 = (int(d) for d in lab.shape)
    shell = torch.empty_like(lab)
    ws_bytes = lib.vittf_surface_shell_workspace_bytes(n0, n1, n2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=lab.device)
    _lib.check(lib.vittf_surface_shell(_lib.ptr(lab), n0, n1, n2, cls, int(dist_from_surface), _lib.ptr(shell), _lib.ptr(ws),
                                       ws_bytes, _lib.stream_ptr()), 'vittf_surface_shell')
    return shell


def sample_uniform(vol, n_samples, thin_to_reasonable=False, class_id=None):
    lab, cls = _as_labels(vol, class_id)
    idxs = (lab == cls).nonzero()
    while thin_to_reasonable and idxs.shape[0] > 2 ** 24:      # multinomial's category limit
        idxs = idxs[::2]
    return _pick(idxs, n_samples).cpu()


def sample_surface(vol, n_samples, dist_from_surface=4, class_id=None):
    """Voxels of the one-voxel shell just inside the mask."""
    shell = surface_shell(vol, dist_from_surface, class_id).nonzero()
    if shell.shape[0] > n_samples:
        return _pick(shell, n_samples).cpu()
    print(f'Full surface only has {shell.shape[0]} voxels (< n_samples={n_samples}).')
    return shell.cpu()


def sample_both(vol, n_samples, dist_from_surface=4, thin_to_reasonable=False, class_id=None):
    half = n_samples // 2
    lab, cls = _as_labels(vol, class_id)
    return torch.cat([sample_uniform(lab, half, thin_to_reasonable=thin_to_reasonable, class_id=cls),
                      sample_surface(lab, half, dist_from_surface=dist_from_surface, class_id=cls)])
