#!/usr/bin/env python3
"""Interleaved A/B of attention kernel variants in ONE process on one device (the boxes of the pool differ by ~5 %, and
separate invocations add more): rounds x variants, each timing `reps` back-to-back launches with events on the launch
stream, on gaussian data.  Usage: python tools/attn_ab.py "1:0,0:0" [batch] [rounds]
   variant = VITTF_ATTN_PIPE (0: the round-1 lazy-maximum kernel, anything else: the default) : VITTF_PP_VARIANT (a
   build-time experiment switch, if the library under test reads one)"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vit_tf_amd import _lib   # noqa: E402


def main():
    variants = [tuple(v.split(':')) for v in (sys.argv[1] if len(sys.argv) > 1 else '1:0,0:0').split(',')]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    tokens, heads, d = int(os.environ.get('TOKENS', '4097')), int(os.environ.get('HEADS', '6')), 64 * int(os.environ.get('HEADS', '6'))
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(0)
    rows = batch * tokens
    qkv = torch.randn(rows, 3 * d, generator=g)
    qkv[:, :2 * d] *= 1.5
    qkv[:, :d] *= 0.125 * 1.4426950408889634
    qkv = qkv.to(torch.float16).to(dev)
    out = torch.empty(rows, d, dtype=torch.float16, device=dev)
    reps = max(3, 320 // batch)

    def launch():
        _lib.check(lib.vittf_attention(_lib.ptr(qkv), _lib.ptr(out), batch, tokens, heads, _lib.DTYPES['fp16'], 1, _lib.stream_ptr()))

    def setenv(v):
        os.environ['VITTF_ATTN_PIPE'] = v[0]
        os.environ['VITTF_PP_VARIANT'] = v[1]

    # settle the clock under load
    setenv(variants[0])
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for _ in range(reps):
            launch()
        torch.cuda.synchronize()
    res = {v: [] for v in variants}
    for _ in range(rounds):
        for v in variants:
            setenv(v)
            launch()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                launch()
            b.record()
            torch.cuda.synchronize()
            res[v].append(a.elapsed_time(b) / reps)
    fl = batch * 4 * tokens * tokens * d
    for v in variants:
        med, mn = statistics.median(res[v]), min(res[v])
        print(f'pipe {v[0]} var {v[1]}: median {med:.4f} ms  min {mn:.4f} ms  {fl / med / 1e9:7.1f} TFLOP/s ({fl / med / 1e9 / 25:.1f} % of 2.5 PF)  batch {batch}')


if __name__ == '__main__':
    main()
