import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_tf_amd as vt
from vit_tf_amd import _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
rows, d = 32 * 4097, 384
epi = int(os.environ.get('EPI', '1')); n = int(os.environ.get('N', '1536'))
g = torch.Generator().manual_seed(0)
a = torch.randn(rows, d, generator=g).bfloat16().to(dev)
w = (torch.randn(n, d, generator=g) / d ** 0.5).bfloat16().to(dev)
b = torch.randn(n, generator=g).to(dev)
o = torch.zeros(rows, n, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    _lib.check(lib.vittf_gemm(_lib.ptr(a), _lib.ptr(w), _lib.ptr(b), _lib.ptr(o), rows, n, d, epi, 4097, 0, _lib.stream_ptr()))
torch.cuda.synchronize()
buf = (C.c_ulonglong * (12 * 64 * 8))()
print('rc', lib.vittf_debug_ws_trace(buf))
import numpy as np
t = np.array(buf[:], dtype=np.int64).reshape(12, 64, 8)
base = t[:, :, 0].min(axis=0)      # per iteration: earliest top
for i in (10, 11, 20, 21):
    print('iteration', i, ' (ticks of 10 ns relative to the earliest wave top of this iteration)')
    print(' wave: top  arriveA exitA  mfma_done pack_done(arriveS) exitS stores_done')
    for w in range(12):
        r = t[w, i] - base[i]
        print(f'  {w:2d}: ' + ' '.join(f'{int(v):5d}' for v in r[:7]))
