"""Times the builds made by tools/fx_variants.sh (csrc/tail_fx.hip, the block tail) against each other and against the library's
own build, interleaved in one process: HIP events around repeated launches, BATCH slices
of 4097 tokens.  Builds with bit 1 set run the main phase only (no tile boundary) and builds with bits 2 / 4 / 8 drop one
ingredient each: their results are wrong by construction.  Builds with bit 16 also print in-kernel stamps."""
import ctypes
import glob
import os
import re
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import vit_tf_amd as vt  # noqa: E402
from vit_tf_amd import _lib  # noqa: E402

ARGS = ([ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])


def main():
    dev = torch.device('cuda', 0)
    batch = int(os.environ.get('BATCH', '256'))
    rows, d = batch * 4097, 384
    g = torch.Generator().manual_seed(0)
    w1 = (torch.randn(4 * d, d, generator=g) / d ** 0.5).half().to(dev)
    w2 = (torch.randn(d, 4 * d, generator=g) / (4 * d) ** 0.5).half().to(dev)
    wp = (torch.randn(d, d, generator=g) / d ** 0.5).half().to(dev)
    wfx = vt.weights.pack_block_tail_weights(wp[None], w1[None], w2[None])[0].contiguous()
    b1 = torch.randn(4 * d, generator=g).to(dev); b2 = torch.randn(d, generator=g).to(dev)
    hh = torch.randn(rows, d, generator=g).half().to(dev)
    lg = torch.ones(d, device=dev); lb = torch.zeros(d, device=dev)
    hn = torch.empty(rows, d, dtype=torch.float16, device=dev)
    x = torch.zeros(rows, d, device=dev)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    fns = []
    lib = _lib.load()
    fns.append(('lib:tail_fx', lib.vittf_block_tail, wfx, None))
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'micro', 'build', 'libfx_v*.so')),
                   key=lambda p: (int(re.search(r'_v(\d+)', p).group(1)), p))
    for p in paths:
        so = ctypes.CDLL(p)
        f = so.vittf_block_tail
        f.restype = ctypes.c_int
        f.argtypes = ARGS
        v = re.search(r'_v(\w+)\.so', p).group(1)
        # main-phase-only builds stream 100 steps per tile: the main steps of the packed stream
        w = wfx[12:].contiguous() if int(re.match(r'\d+', v).group(0)) & 1 else wfx
        fns.append((v, f, w, so))

    def run(f, w):
        rc = f(hh.data_ptr(), w.data_ptr(), b2.data_ptr(), lg.data_ptr(), lb.data_ptr(), b1.data_ptr(), b2.data_ptr(), x.data_ptr(),
               rows, d, _lib.DTYPES['fp16'], lg.data_ptr(), lb.data_ptr(), 1e-6, hn.data_ptr(), ctr.data_ptr(), _lib.stream_ptr())
        assert rc == 0, rc
    for _, f, w, _so in fns:
        run(f, w)
    torch.cuda.synchronize()
    fl = 18 * rows * d * d
    for rnd in range(int(os.environ.get('ROUNDS', '3'))):
        for v, f, w, so in fns:
            x.zero_()
            for _ in range(3):
                run(f, w)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                run(f, w)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 10
            print(f'round {rnd} {v:>12s}: {ms:.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s (of the whole tail)', flush=True)
            if so is not None and int(re.match(r'\d+', v).group(0)) & 16 and rnd == 0:
                stamps(so, v)      # (the stamps of the launch that ran last: this entry's)


def stamps(so, v):
    buf = np.zeros((4, 4, 8, 8), np.uint64)
    hw = np.zeros(8, np.uint32)
    assert so.vittf_fx_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(hw.ctypes.data)) == 0
    print('  HW_ID of workgroup 0 waves (simd = bits 5:4): ' + ' '.join(f'w{w}:simd{(int(hw[w]) >> 4) & 3}' for w in range(8)))
    cyc = buf.astype(np.int64)
    for wg in range(2):
        for t in (1, 2):
            for w in (0, 4):
                d = [int(cyc[wg, t, w, k + 1] - cyc[wg, t, w, k]) for k in range(7)]
                tot = int(cyc[wg, t + 1, w, 0] - cyc[wg, t, w, 0])
                print(f'  {v} wg {wg} tile {t} wave {w}: stamps {d} | tile {tot} cycles = {tot / 100:.0f} per main step')


if __name__ == '__main__':
    main()
