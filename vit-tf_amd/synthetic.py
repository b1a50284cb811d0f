"""Seeded synthetic input volumes: the four SDF shapes of create_synthetic_volumes.py:8-69 and the
512^3 "CT-ORG-style" benchmark volume of SURVEY.md section 8d.  Input generation only (CPU torch)."""
import torch


def _positions(size):
    ls = torch.linspace(-1, 1, size)
    return torch.stack(torch.meshgrid(ls, ls, ls, indexing='xy'), dim=-1)


def sdf_sphere(pos, r):
    return torch.linalg.vector_norm(pos, dim=-1) - r


def sdf_torus(pos, r1, r2):
    q = torch.linalg.vector_norm(pos[..., :2], dim=-1) - r1
    return torch.sqrt(q * q + pos[..., 2] * pos[..., 2]) - r2


def shapes(size):
    """{name: fp32 0/1 volume} for sphere_thick, sphere_filled, torus_thick, torus_filled."""
    pos = _positions(size)
    sph, tor = sdf_sphere(pos, 0.5), sdf_torus(pos, 0.5, 0.2)
    return {
        'sphere_thick': (sph.abs() < 0.05).float(),
        'sphere_filled': (sph <= 0).float(),
        'torus_thick': (tor.abs() < 0.05).float(),
        'torus_filled': (tor <= 0).float(),
    }


def synthetic_volume(name='torus_filled', size=128, noise=0.0, seed=0):
    """One reference shape with seeded uniform noise (the reference's is unseeded,
    create_synthetic_volumes.py:40): (fp16 volume, uint8 label)."""
    clean = shapes(size)[name]
    vol = clean
    if noise != 0.0:
        g = torch.Generator().manual_seed(seed)
        vol = clean + torch.rand(clean.shape, generator=g) * noise
    return vol.clamp(0, 1).half(), (clean > 0.5).to(torch.uint8)


def ct_like_volume(size=512, seed=0):
    """HU-like air / soft tissue / bone-shell volume with Gaussian noise + labels 0..3
    (SURVEY.md 8d): -1000 + 1400*sphere_filled + 400*torus_filled + 2000*sphere_thick + 30*N(0,1).
    Built by broadcasting 1-D coordinate arrays (a 512^3 position grid would cost 1.6 GB per rank)."""
    ls = torch.linspace(-1, 1, size)
    # meshgrid(..., indexing='xy') of shapes(): component 0 varies along dim 1, component 1 along dim 0
    px, py, pz = ls.view(1, size, 1), ls.view(size, 1, 1), ls.view(1, 1, size)
    rho2 = px * px + py * py                                  # (size, size, 1)
    sph = torch.sqrt(rho2 + pz * pz) - 0.5
    q = torch.sqrt(rho2) - 0.5
    tor = torch.sqrt(q * q + pz * pz) - 0.2
    sphere_filled, torus_filled, sphere_thick = sph <= 0, tor <= 0, sph.abs() < 0.05
    g = torch.Generator().manual_seed(seed)
    vol = torch.full((size, size, size), -1000.0)
    vol += 1400.0 * sphere_filled + 400.0 * torus_filled + 2000.0 * sphere_thick
    vol += 30.0 * torch.randn(vol.shape, generator=g)
    label = torch.zeros(vol.shape, dtype=torch.uint8)
    label[sphere_filled] = 1
    label[torus_filled] = 2
    label[sphere_thick] = 3
    return vol.half(), label
