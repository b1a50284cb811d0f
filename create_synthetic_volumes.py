"""Drop-in for the reference's ``create_synthetic_volumes.py`` (:28-69): writes the four SDF test volumes
(+ labels) as fp16 ``.npy`` / ``.pt``.  Adds ``--seed`` (the reference's noise is unseeded, :40)."""
from argparse import ArgumentParser
from pathlib import Path

import numpy as np
import torch

import vit_tf_amd as vt

NAMES = ('sphere_thick', 'sphere_filled', 'torus_thick', 'torus_filled')


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('outdir', type=Path, help='directory the .npy / .pt files are written to')
    parser.add_argument('--size', type=int, default=128, help='edge length N of the N^3 volumes')
    parser.add_argument('--noise', type=float, default=0.0, help='sigma of the additive Gaussian noise')
    parser.add_argument('--torch', action='store_true', help='write torch .pt files instead of .npy')
    parser.add_argument('--seed', type=int, default=0, help='Noise seed')
    args = parser.parse_args(argv)
    outdir = Path(args.outdir)
    outdir.mkdir(exist_ok=True)
    for i, name in enumerate(NAMES):
        vol, label = vt.synthetic_volume(name, args.size, args.noise, args.seed + i)
        if args.torch:
            torch.save(vol, outdir / f'{name}.pt')
            torch.save(label, outdir / f'{name}_label.pt')
        else:
            np.save(outdir / f'{name}.npy', vol.numpy())
            np.save(outdir / f'{name}_label.npy', label.numpy())


if __name__ == '__main__':
    main()
