#!/usr/bin/env python3
"""Where the time of a 5 x 1024-query similarity call goes (BASELINE configs[4]): whole call (maps to the host / left on the
device) against host-side pieces.  tools only."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vit_tf_amd as vt

dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
feat = torch.randn(384, 64, 64, 64, generator=g).half().to(dev)
ann = {f'c{i}': torch.randint(0, 256, (1024, 3), generator=g) for i in range(5)}
vol = np.zeros((256, 256, 256), np.float32)

def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def kernel_ms(fv, reps=10):
    """events around the accumulation kernel alone (the library's profiler class 'similarity')"""
    vt.compute_similarities(vol, fv, ann, keep_on_device=True)
    vt._lib.profiler_enable(True, classes=['similarity'])
    for _ in range(reps):
        vt.compute_similarities(vol, fv, ann, keep_on_device=True)
    torch.cuda.synchronize()
    ms, n = vt._lib.profiler_collect()['similarity']
    vt._lib.profiler_enable(False)
    return ms / max(1, n), vt._lib.kernel_name('similarity')


# one / two voxel blocks per wave, interleaved in this process (5 x 1024 queries, 64^3 x 384), and the 768-feature volume
feat_n = torch.nn.functional.normalize(feat.float(), dim=0).half()
for rnd in range(3):
    for vb in ('1', '2'):
        os.environ['VITTF_SIM_MFMA_VB'] = vb
        ms, name = kernel_ms(feat_n)
        print(f'round {rnd}: accumulation kernel {name:34s} {ms:.3f} ms  ({2 * 2 * 5120 * 262144 * 384 / ms / 1e9:.0f} TFLOP/s incl. the hi + lo split)')
os.environ.pop('VITTF_SIM_MFMA_VB')
feat768 = torch.nn.functional.normalize(torch.randn(768, 64, 64, 64, generator=g), dim=0).half().to(dev)
ms, name = kernel_ms(feat768)
print(f'768 features: accumulation kernel {name:34s} {ms:.3f} ms  ({2 * 2 * 5120 * 262144 * 768 / ms / 1e9:.0f} TFLOP/s incl. the split)')
del feat768

# the same call on volumes whose (query, voxel) similarities look like a segmentation task's: unit-norm features around 5
# cluster centres (a query passes the 0.25 threshold on its own cluster's voxels only), and unit-norm noise (never)
centres = torch.nn.functional.normalize(torch.randn(5, 384, generator=g), dim=1)
which = torch.randint(0, 5, (64, 64, 64), generator=g)
clustered = torch.nn.functional.normalize(centres[which].permute(3, 0, 1, 2) + 0.04 * torch.randn(384, 64, 64, 64, generator=g), dim=0).half().to(dev)
noise = torch.nn.functional.normalize(torch.randn(384, 64, 64, 64, generator=g), dim=0).half().to(dev)
for name, fv in (('clustered unit-norm features', clustered), ('unit-norm noise', noise)):
    print(f'{name:30s}: {t(lambda: vt.compute_similarities(vol, fv, ann, keep_on_device=True)):.2f} ms (maps left on the GPU)')
print('unnormalised Gaussian features (half of all dot products pass the threshold):')
print(f'maps to the host      : {t(lambda: vt.compute_similarities(vol, feat, ann)):.2f} ms')
print(f'maps left on the GPU  : {t(lambda: vt.compute_similarities(vol, feat, ann, keep_on_device=True)):.2f} ms')
os.environ['VITTF_SIM_MFMA'] = '0'
print(f'VALU kernel (MFMA off): {t(lambda: vt.compute_similarities(vol, feat, ann, keep_on_device=True), reps=3):.2f} ms')
