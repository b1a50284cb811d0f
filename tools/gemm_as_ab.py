"""qkv projection of ViT-S (rows x 384 -> 1152, q pre-scaled) on the weight-stationary kernel (vittf_gemm -> gemm_ws.hip) and on the
activation-stationary one (vittf_gemm_as, gemm_as.hip), interleaved in one process; BATCH slices of 4097 tokens, fp16."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import vit_tf_amd as vt  # noqa: E402
from vit_tf_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    batch = int(os.environ.get('BATCH', '256'))
    rows, k, n = batch * 4097, 384, 1152
    g = torch.Generator().manual_seed(0)
    a = torch.randn(rows, k, generator=g).half().to(dev)
    w = (1.3 * torch.randn(n, k, generator=g) / k ** 0.5).half().to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    wpk = vt.weights.pack_row_images(w[None])[0]
    o1 = torch.empty(rows, n, dtype=torch.float16, device=dev)
    o2 = torch.empty(rows, n, dtype=torch.float16, device=dev)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    fns = {
        'gemm_ws': lambda: lib.vittf_gemm(_lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(o1), rows, n, k, _lib.EPI_BIAS_QKV, 0, _lib.DTYPES['fp16'], _lib.stream_ptr()),
        'gemm_as': lambda: lib.vittf_gemm_as(_lib.ptr(a), _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(o2), rows, n, k, _lib.EPI_BIAS_QKV, _lib.DTYPES['fp16'], _lib.ptr(ctr), _lib.stream_ptr()),
    }
    import ctypes, glob, re
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'micro', 'build', 'libas_v*.so'))):
        so = ctypes.CDLL(path)
        f = so.vittf_gemm_as
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        fns['as_v' + re.search(r'_v(\w+)\.so', path).group(1)] = (
            lambda f=f: f(a.data_ptr(), wpk.data_ptr(), bias.data_ptr(), o2.data_ptr(), rows, n, k, _lib.EPI_BIAS_QKV, _lib.DTYPES['fp16'], ctr.data_ptr(), _lib.stream_ptr()))
    for f in fns.values():
        assert f() == 0
    torch.cuda.synchronize()
    assert fns['gemm_as']() == 0; torch.cuda.synchronize()
    print('bit-equal:', bool(torch.equal(o1, o2)))
    fl = 2 * rows * n * k
    for rnd in range(3):
        for name, f in fns.items():
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f'round {rnd} {name}: {ms:.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  {(rows * k * 2 + rows * n * 2) / ms / 1e6:7.0f} GB/s algorithmic', flush=True)


if __name__ == '__main__':
    main()
