#!/usr/bin/env python3
"""Interleaved A/B of environment switches of the residual GEMMs (proj + LN: K = 384, fc2 + LN: K = 1536) in ONE process:
   python tools/gemm_ab.py "VITTF_ROWS_SPREAD=0,VITTF_ROWS_SPREAD=1" [batch] [rounds]
Every setting also has its x / h outputs compared with the first one's (same arithmetic -> same bits expected)."""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vit_tf_amd import _lib   # noqa: E402


def main():
    settings = (sys.argv[1] if len(sys.argv) > 1 else 'VITTF_ROWS_SPREAD=0,VITTF_ROWS_SPREAD=1').split(',')
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    d = int(os.environ.get('DIM', '384'))
    tokens = 4097
    rows = batch * tokens
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    keys = sorted({s.split('=')[0] for s in settings})
    for k in (d, 4 * d):
        act = (torch.randn(rows, k, generator=g)).half().to(dev)
        w = (torch.randn(d, k, generator=g) / k ** 0.5).half().to(dev)
        bias = torch.randn(d, generator=g).to(dev)
        x0 = torch.randn(rows, d, generator=g).to(dev)
        lg = (1 + 0.1 * torch.randn(d, generator=g)).to(dev); lb = (0.1 * torch.randn(d, generator=g)).to(dev)
        x = x0.clone(); h = torch.empty(rows, d, dtype=torch.half, device=dev)

        def call():
            _lib.check(lib.vittf_gemm_residual_ln(_lib.ptr(act), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(x), rows, d, k, _lib.DTYPES['fp16'],
                                                  _lib.ptr(lg), _lib.ptr(lb), 1e-6, _lib.ptr(h), _lib.stream_ptr()))

        def setenv(s):
            for kk in keys:
                os.environ.pop(kk, None)
            kk, v = s.split('=')
            os.environ[kk] = v

        ref = None
        for s in settings:                     # parity first
            setenv(s)
            x.copy_(x0)
            call()
            torch.cuda.synchronize()
            cur = (x.clone(), h.clone())
            if ref is None:
                ref = cur
            else:
                print(f'K = {k}: {s}: x bit-equal {torch.equal(cur[0], ref[0])}, h bit-equal {torch.equal(cur[1], ref[1])}')
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.7:
            for _ in range(5):
                call()
            torch.cuda.synchronize()
        res = {s: [] for s in settings}
        reps = max(3, 320 // batch)
        for _ in range(rounds):
            for s in settings:
                setenv(s)
                call()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    call()
                b.record()
                torch.cuda.synchronize()
                res[s].append(a.elapsed_time(b) / reps)
        fl = 2 * rows * d * k
        for s in settings:
            med = statistics.median(res[s])
            print(f'K = {k:4d} batch {batch}: {s:24s} median {med:.4f} ms  min {min(res[s]):.4f}  {fl / med / 1e9:7.1f} TFLOP/s  ({med * 32 / batch:.4f} ms per 32 slices)')
        del act, x, x0, h


if __name__ == '__main__':
    main()
