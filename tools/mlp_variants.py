"""Times the fused-MLP timing variants built by tools/mlp_variants.sh (HIP events around repeated launches, BATCH slices
of 4097 tokens).  Variant 0 = the kernel as shipped; the others drop one ingredient each (their results are wrong)."""
import ctypes, glob, os, re, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import vit_tf_amd as vt  # noqa: E402
from vit_tf_amd import _lib  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    batch = int(os.environ.get('BATCH', '256'))
    rows, d = batch * 4097, 384
    g = torch.Generator().manual_seed(0)
    w1 = (torch.randn(4 * d, d, generator=g) / d ** 0.5).half().to(dev)
    w2 = (torch.randn(d, 4 * d, generator=g) / (4 * d) ** 0.5).half().to(dev)
    tail = os.environ.get('TAIL', '0') == '1'          # the block-tail kernel (projection + norm2 in front of the MLP)
    wp = (torch.randn(d, d, generator=g) / d ** 0.5).half().to(dev)
    wpk = (vt.weights.pack_block_tail_weights(wp[None], w1[None], w2[None]) if tail else vt.weights.pack_mlp_weights(w1[None], w2[None]))[0].contiguous()
    b1 = torch.randn(4 * d, generator=g).to(dev); b2 = torch.randn(d, generator=g).to(dev)
    hh = torch.randn(rows, d, generator=g).half().to(dev)
    lg = torch.ones(d, device=dev); lb = torch.zeros(d, device=dev)
    hn = torch.empty(rows, d, dtype=torch.float16, device=dev)
    x = torch.zeros(rows, d, device=dev)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    libs = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'micro', 'build', 'libmlp_v*.so')),
                  key=lambda p: (int(re.search(r'_v(\d+)', p).group(1)), p))
    fns = []
    for p in libs:
        lib = ctypes.CDLL(p)
        f = lib.vittf_block_tail if tail else lib.vittf_mlp_fused
        f.restype = ctypes.c_int
        f.argtypes = ([ctypes.c_void_p] * (8 if tail else 5) + [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                                ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])
        fns.append((re.search(r'_v(\w+)\.so', p).group(1), f))

    def run(f):
        pre = (hh.data_ptr(), wpk.data_ptr(), b2.data_ptr(), lg.data_ptr(), lb.data_ptr()) if tail else (hh.data_ptr(), wpk.data_ptr())
        rc = f(*pre, b1.data_ptr(), b2.data_ptr(), x.data_ptr(), rows, d, _lib.DTYPES['fp16'],
               lg.data_ptr(), lb.data_ptr(), 1e-6, hn.data_ptr(), ctr.data_ptr(), _lib.stream_ptr())
        assert rc == 0, rc
    for _, f in fns:
        run(f)
    torch.cuda.synchronize()
    fl = (18 if tail else 16) * rows * d * d
    for rnd in range(3):               # interleaved rounds: clock drift shows as a spread between rounds, not between variants
        for v, f in fns:
            x.zero_()
            for _ in range(3):
                run(f)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                run(f)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 10
            if int(re.match(r'\d+', v).group(0)) & 16 and rnd == 0:
                stamps(libs, v)
            print(f'round {rnd} variant {v:>5s}: {ms:.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  {ms * 32 / batch:.4f} ms per 32 slices', flush=True)


def stamps(libs, v='16'):
    """libmlp_v16.so: where a tile's time goes (cycles between the stamp points, per wave, tiles 1 .. 3 of workgroups 0 .. 3)."""
    import numpy as np
    p = ([q for q in libs if q.endswith('_v' + v + '.so')] + [''])[0]
    if not os.path.exists(p):
        return
    lib = ctypes.CDLL(p)
    buf = np.zeros((4, 4, 4, 8, 2), np.uint64)
    assert lib.vittf_mlp_stamps(ctypes.c_void_p(buf.ctypes.data)) == 0
    names = ['top -> unit 0 done (TAIL=1: projection units)', 'units 1 .. 5 (TAIL=1: norm2 + units 0 .. 5)', 'units 6 .. 93', 'units 94, 95', 'h loads issued + x = acc + b2, stores', 'LayerNorm statistics',
             'normalise + h stores', 'to the next tile top']
    cyc = buf[..., 0].astype(np.int64)
    for wg in range(2):
        for t in (1, 2):
            for w in (0, 3):
                d = [cyc[wg, t, w, k + 1] - cyc[wg, t, w, k] for k in range(7)] + [cyc[wg, t + 1, w, 0] - cyc[wg, t, w, 7]]
                tot_c = cyc[wg, t + 1, w, 0] - cyc[wg, t, w, 0]
                print(f'wg {wg} tile {t} wave {w}: ' + ', '.join(f'{n} {int(v)}' for n, v in zip(names, d)) +
                      f' | tile {int(tot_c)} cycles')


if __name__ == '__main__':
    main()
