#!/usr/bin/env python3
"""In-kernel clock and cycles per 64-key tile of the pipelined attention kernel (diagnostic build, VITTF_ATTN_ABLATE=5):
runs the headline shape back to back for ~2 s, then reads the stamps of the last launch."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['VITTF_ATTN_ABLATE'] = '5'
from vit_tf_amd import _lib   # noqa: E402

lib = _lib.load()
dev = torch.device('cuda', 0)
batch, tokens, heads, d = 32, 4097, 6, 384
g = torch.Generator().manual_seed(0)
qkv = torch.randn(batch * tokens, 3 * d, generator=g)
qkv[:, :2 * d] *= 1.5
qkv[:, :d] *= 0.125 * 1.4426950408889634
qkv = qkv.half().to(dev)
out = torch.empty(batch * tokens, d, dtype=torch.float16, device=dev)
t0 = time.time()
n = 0
while time.time() - t0 < 2.0:
    for _ in range(50):
        _lib.check(lib.vittf_attention(_lib.ptr(qkv), _lib.ptr(out), batch, tokens, heads, 1, 1, _lib.stream_ptr()))
    torch.cuda.synchronize()
    n += 50
buf = (C.c_uint64 * (4096 * 4))()
got = lib.vittf_debug_attention_stamps(buf, 4096)
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4)[:got].astype(np.int64)
a = a[(a[:, 1] > a[:, 0]) & (a[:, 3] > a[:, 2])]
cyc = a[:, 1] - a[:, 0]
rt = (a[:, 3] - a[:, 2]) / 100e6
tiles = (tokens + 63) // 64
print(f'{len(a)} waves: loop {np.median(cyc):.0f} shader cycles (median), {np.median(cyc) / tiles:.1f} per 64-key tile, '
      f'{np.median(cyc) / tiles / 16:.1f} per MFMA and wave; in-kernel clock {np.median(cyc / rt) / 1e9:.3f} GHz '
      f'(p10 {np.percentile(cyc / rt, 10) / 1e9:.3f}, p90 {np.percentile(cyc / rt, 90) / 1e9:.3f}); {n} launches')
