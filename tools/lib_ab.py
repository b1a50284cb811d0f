#!/usr/bin/env python3
"""A/B of TWO builds of libvittf.so in one process (the tree's against a copy built from another commit), on the qkv / fc1 GEMM
shapes of ViT-S (K = 384): python tools/lib_ab.py tools/micro/build/libvittf_prev.so [batch] [rounds].  Outputs compared bit for bit."""
import ctypes as C
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vit_tf_amd import _lib   # noqa: E402


def main():
    other = sys.argv[1]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    libs = {'tree': _lib.load(), 'other': C.CDLL(other)}
    res_t, arg_t = _lib.SIGNATURES['vittf_gemm']
    libs['other'].vittf_gemm.restype, libs['other'].vittf_gemm.argtypes = res_t, arg_t
    d, tokens = 384, 4097
    rows = batch * tokens
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    for name, n, epi in (('qkv', 3 * d, _lib.EPI_BIAS_QKV), ('fc1 + GELU', 4 * d, _lib.EPI_BIAS_GELU), ('plain', 3 * d, _lib.EPI_BIAS)):
        acts = [torch.randn(rows, d, generator=g).half().to(dev) for _ in range(3)]
        w = (torch.randn(n, d, generator=g) / d ** 0.5).half().to(dev)
        bias = torch.randn(n, generator=g).to(dev)
        out = torch.empty(rows, n, dtype=torch.half, device=dev)
        turn = [0]

        def call(lib):
            a = acts[turn[0] % 3]
            turn[0] += 1
            _lib.check(lib.vittf_gemm(_lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out), rows, n, d, epi, tokens,
                                      _lib.DTYPES['fp16'], _lib.stream_ptr()))
        ref = None
        for k, lib in libs.items():
            turn[0] = 0
            call(lib)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                print(f'{name}: {k} bit-equal to tree: {torch.equal(out, ref)}')
        del ref
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.7:
            for _ in range(5):
                call(libs['tree'])
            torch.cuda.synchronize()
        res = {k: [] for k in libs}
        reps = max(3, 1536 // batch)
        for _ in range(rounds):
            for k, lib in libs.items():
                call(lib)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    call(lib)
                b.record()
                torch.cuda.synchronize()
                res[k].append(a.elapsed_time(b) / reps)
        fl = 2.0 * rows * n * d
        for k in libs:
            med = statistics.median(res[k])
            print(f'{name} [{rows} x {n}] {k:5s}: median {med:.4f} ms  min {min(res[k]):.4f}  {fl / med / 1e9:7.1f} TFLOP/s')
        del acts, out


if __name__ == '__main__':
    main()
