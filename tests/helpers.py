"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch

import vit_tf_amd as vt
from oracle import dino_vit

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
