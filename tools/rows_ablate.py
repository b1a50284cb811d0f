#!/usr/bin/env python3
"""Where the whole-row GEMM's time goes: the complete kernel against timing-only builds without its MFMAs, without its
epilogue, and with neither (tools/rows_ablate.sh builds them; their results are wrong by construction).
proj (K = 384) and fc2 (K = 1536) shapes of a 32-slice batch, with the LayerNorm epilogue."""
import ctypes as C
import glob
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dev = torch.device('cuda', 0)
    batch, tokens, d = int(os.environ.get('BATCH', '32')), 4097, 384
    rows = batch * tokens
    g = torch.Generator().manual_seed(0)
    libs = sorted(glob.glob(os.path.join(ROOT, 'tools', 'micro', 'build', 'libvittf_rows_*.so')))
    st = torch.cuda.current_stream().cuda_stream
    # DVFS: settle the clocks first
    t0 = time.time()
    a = torch.randn(4096, 4096, device=dev, dtype=torch.half)
    while time.time() - t0 < 0.5:
        (a @ a).sum().item()
    for k in [int(v) for v in os.environ.get('KS', '384,1536').split(',')]:
        act = torch.randn(rows, k, generator=g).half().to(dev)
        w = (torch.randn(d, k, generator=g) / k ** 0.5).half().to(dev)
        bias = torch.randn(d, generator=g).to(dev)
        x = torch.zeros(rows, d, device=dev)
        lg = torch.ones(d, device=dev); lb = torch.zeros(d, device=dev)
        h = torch.empty(rows, d, dtype=torch.half, device=dev)
        for path in libs:
            lib = C.CDLL(path)
            fn = lib.vittf_gemm_residual_ln
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                           C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
            call = lambda: fn(act.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(), rows, d, k, 1, lg.data_ptr(),
                              lb.data_ptr(), 1e-6, h.data_ptr(), st)
            for _ in range(5):
                assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 30
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print(f'K = {k:4d}  {os.path.basename(path)[len("libvittf_rows_"):-3]:12s} {ms:.4f} ms', flush=True)


if __name__ == '__main__':
    main()
