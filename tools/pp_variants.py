#!/usr/bin/env python3
"""Times the builds of tools/pp_variants.sh interleaved in one process (clock drift shows as a spread between rounds, not
between variants): the ViT-B linears at D = 768, BATCH slices of 4097 tokens, fp16, random data."""
import ctypes
import glob
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vit_tf_amd import _lib   # noqa: E402


def main():
    batch = int(os.environ.get('BATCH', '64'))
    rows, d = batch * 4097, 768
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    libs = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'micro', 'build', 'libpp_v*.so')),
                  key=lambda p: int(re.search(r'_v(\d+)', p).group(1)))
    fns = []
    for p in libs:
        lib = ctypes.CDLL(p)
        f = lib.vittf_gemm_pp
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int32] * 5 + [ctypes.c_void_p]
        fns.append((re.search(r'_v(\d+)\.so', p).group(1), f))
    for name, n, k, epi in (('qkv', 3 * d, d, 4), ('fc1+gelu', 4 * d, d, 1), ('fc2+res', d, 4 * d, 2)):
        a = torch.randn(rows, k, generator=g).half().to(dev)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).half().to(dev)
        bias = torch.randn(n, generator=g).to(dev)
        o = torch.zeros(rows, n, dtype=torch.float32 if epi == 2 else torch.float16, device=dev)

        def run(f):
            rc = f(a.data_ptr(), w.data_ptr(), bias.data_ptr(), o.data_ptr(), rows, n, k, epi, 4097, _lib.DTYPES['fp16'],
                   _lib.stream_ptr())
            assert rc == 0, rc
        todo = fns
        for _, f in todo:
            for _ in range(3):
                run(f)
        torch.cuda.synchronize()
        fl = 2 * rows * n * k
        for rnd in range(3):
            for v, f in todo:
                for _ in range(3):
                    run(f)
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record()
                for _ in range(10):
                    run(f)
                eb.record(); torch.cuda.synchronize()
                ms = ea.elapsed_time(eb) / 10
                print(f'{name:9s} round {rnd} variant {v:>5s}: {ms:.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
