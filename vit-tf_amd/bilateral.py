"""3-D bilateral-solver refinement of class similarity maps on the GPU (libvittf, bilateral.hip).

Host-side mirror of the optional post-process in predict_ntf.compute_similarities (predict_ntf.py:73-96) and of
apply_bilateral_solver3d (bilateral_solver3d.py:211-245): same parameters and defaults, device tensors in and out.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

GRID_PARAMS = {'sigma_spatial': 7, 'sigma_chroma': 5, 'sigma_luma': 5}                 # predict_ntf.py:75-79
GRID_PARAMS_DEFAULT = {'sigma_luma': 4, 'sigma_chroma': 4, 'sigma_spatial': 24}       # bilateral_solver3d.py:156-160
BS_PARAMS_DEFAULT = {'lam': 256, 'A_diag_min': 1e-5, 'cg_tol': 1e-5, 'cg_maxiter': 25}   # bilateral_solver3d.py:162-167
_RGB_TO_YUV = np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]])   # bilateral_solver3d.py:11-12


def luma_bins(sigma_luma):
    """Bilateral-space luma bin of the 256 grey levels: (rgb2yuv(v, v, v)[0] / sigma_luma).astype(int), float64
    (bilateral_solver3d.py:19-20, 47).  Host-side table, 256 entries."""
    grey = np.repeat(np.arange(256, dtype=np.float64)[:, None], 3, axis=1)
    # the full 3 x 3 product, like rgb2yuv: the float64 rounding of the luma row depends on the BLAS kernel shape, and
    # grey levels that are multiples of sigma_luma sit exactly on a bin edge
    luma = (np.tensordot(grey, _RGB_TO_YUV, ([1], [1])) + np.array([0.0, 128.0, 128.0]))[:, 0]
    return (luma / sigma_luma).astype(int).astype(np.int32)


def refine_similarity(sim, volume, sim_shape, grid_params=None, bs_params=None, crop_threshold=0.1, pad=2, info=None):
    """One class of predict_ntf.py:80-94.  sim: fp32 device tensor (n0, n1, n2) (class map after threshold / power /
    mean); volume: fp32 device tensor (W, H, D).  Returns the refined fp32 device tensor of shape sim_shape."""
    lib = _lib.require_device()
    gp = {**GRID_PARAMS, **(grid_params or {})}
    bs = {**BS_PARAMS_DEFAULT, **(bs_params or {})}
    sim = sim.to(torch.float32).contiguous()
    volume = volume.to(device=sim.device, dtype=torch.float32).contiguous()
    lut = luma_bins(gp['sigma_luma'])
    nbins = int(lut.max()) + 1
    o0, o1, o2 = (int(s) for s in sim_shape)
    need = lib.vittf_bilateral_workspace_bytes(o0, o1, o2, float(gp['sigma_spatial']), nbins)
    if need == 0:
        raise _lib.VittfError(f'bilateral solver: unsupported size {tuple(sim_shape)} / sigma {gp["sigma_spatial"]}')
    ws = torch.empty(need, dtype=torch.uint8, device=sim.device)
    out = torch.empty((o0, o1, o2), dtype=torch.float32, device=sim.device)
    prm = _lib.BilateralParams(float(gp['sigma_spatial']), float(bs['lam']), float(bs['A_diag_min']), float(bs['cg_tol']),
                               int(bs['cg_maxiter']), 10, int(pad), float(crop_threshold))
    info_host = (C.c_int32 * 2)()
    _lib.check(lib.vittf_bilateral_refine(_lib.ptr(sim), sim.shape[0], sim.shape[1], sim.shape[2], _lib.ptr(volume),
                                          volume.shape[0], volume.shape[1], volume.shape[2], o0, o1, o2,
                                          lut.ctypes.data_as(C.POINTER(C.c_int32)), nbins, C.byref(prm), _lib.ptr(out),
                                          info_host, _lib.ptr(ws), need, _lib.stream_ptr()), 'vittf_bilateral_refine')
    if info is not None:
        info['vertices'], info['voxels'] = int(info_host[0]), int(info_host[1])
    return out


def quantize_u8(sim):
    """(255 / (0.99 * sim.max()) * sim).to(uint8) with the x86 wrap-around (predict_ntf.py:95-96); device in / out."""
    lib = _lib.require_device()
    sim = sim.to(torch.float32).contiguous()
    out = torch.empty(sim.shape, dtype=torch.uint8, device=sim.device)
    scratch = torch.empty(1, dtype=torch.float32, device=sim.device)
    _lib.check(lib.vittf_quantize_wrap_u8(_lib.ptr(sim), sim.numel(), _lib.ptr(out), _lib.ptr(scratch), _lib.stream_ptr()),
               'vittf_quantize_wrap_u8')
    return out
