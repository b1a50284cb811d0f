"""Times the builds made by tools/attn_variants.sh (csrc/attention_pp64.hip with the LDS-DMA of every n-th key tile only)
against the library's kernel, interleaved in one process: HIP events around repeated launches, BATCH slices of TOKENS tokens,
6 heads.  Prices what halving a workgroup's K / V traffic per query row (512-row workgroups) could buy at most."""
import ctypes
import glob
import os
import re
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from vit_tf_amd import _lib  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    batch, tokens, heads = int(os.environ.get('BATCH', '256')), int(os.environ.get('TOKENS', '4097')), 6
    d = heads * 64
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(batch * tokens, 3 * d, generator=g)
    qkv[:, :2 * d] *= 1.5
    qkv[:, :d] *= 0.125 * 1.4426950408889634
    qkv = qkv.half().to(dev)
    out = torch.empty(batch * tokens, d, dtype=torch.float16, device=dev)
    lib = _lib.load()
    fns = [('lib', lambda: lib.vittf_attention(_lib.ptr(qkv), _lib.ptr(out), batch, tokens, heads, _lib.DTYPES['fp16'], 1, _lib.stream_ptr()))]
    keep = []
    for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'micro', 'build', 'libattn_v*.so'))):
        so = ctypes.CDLL(p)
        f = so.vittf_attention_variant
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        keep.append(so)
        v = re.search(r'_v(\w+)\.so', p).group(1)
        fns.append((f'dma every {v}' if v != '0' else 'dma: first 3 tiles only',
                    lambda f=f: f(qkv.data_ptr(), out.data_ptr(), batch, tokens, heads, _lib.DTYPES['fp16'], _lib.stream_ptr())))
    fl = batch * 4 * tokens * tokens * d
    for _, fn in fns:
        assert fn() == 0
    torch.cuda.synchronize()
    for rnd in range(int(os.environ.get('ROUNDS', '3'))):
        for name, fn in fns:
            for _ in range(2):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(6):
                fn()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 6
            print(f'round {rnd} {name:>26s}: {ms:.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s = {fl / ms / 1e9 / 2500:.3f}', flush=True)


if __name__ == '__main__':
    main()
