"""Oracle: synthetic SDF test volumes (input side of the hot path), CPU.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates create_synthetic_volumes.py:8-69: a sphere shell, a filled sphere, a torus shell and a
filled torus on a [-1, 1]^3 grid (``meshgrid(..., indexing='xy')``), optional uniform noise
clamped to [0, 1], volumes as fp16 and labels (> 0.5) as uint8.  The reference draws its noise
from the unseeded global generator (create_synthetic_volumes.py:40); here a seed is explicit.
"""
import torch


def _grid(size):
    ls = torch.linspace(-1, 1, size)
    # indexing='xy' swaps the first two output dims relative to 'ij'
    gx, gy, gz = torch.meshgrid(ls, ls, ls, indexing='xy')
    return torch.stack((gx, gy, gz), dim=-1)


def _sdf_sphere(pos, r):
    return pos.norm(dim=-1) - r


def _sdf_torus(pos, r_major, r_minor):
    ring = pos[..., :2].norm(dim=-1) - r_major
    return torch.stack((ring, pos[..., 2]), dim=-1).norm(dim=-1) - r_minor


def synthetic_volumes(size=128, noise=0.0, seed=None):
    """Returns {name: (volume fp16 (size,)*3, label uint8)} for the four reference shapes."""
    pos = _grid(size)
    clean = {
        'sphere_thick': (_sdf_sphere(pos, 0.5).abs() < 0.05).float(),
        'sphere_filled': (_sdf_sphere(pos, 0.5) <= 0).float(),
        'torus_thick': (_sdf_torus(pos, 0.5, 0.2).abs() < 0.05).float(),
        'torus_filled': (_sdf_torus(pos, 0.5, 0.2) <= 0).float(),
    }
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    out = {}
    for name, v in clean.items():
        noisy = v
        if noise != 0.0:
            noisy = v + torch.rand(v.shape, generator=gen) * noise
        out[name] = (noisy.clamp(0, 1).half(), (v > 0.5).to(torch.uint8))
    return out
