"""Makes tests/golden/windows512.npz: the fp32 CPU oracle's pooled K-feature windows of the 512^3 CT-like volume that
tests/test_gpu_fullsize.py compares the GPU path with (helpers.WINDOWS512), every 4th feature row / column, fp16 as pooled.

    python tests/golden/make_window_goldens.py        (CPU only; ~10 minutes on 8 cores)

The inputs are reproducible by construction (vt.ct_like_volume(512, 0), vt.synthetic_state_dict(arch, seed)); the oracle is
oracle/dino_vit.py + oracle/feature_volume.py, the same functions the live comparisons call (helpers.oracle_window)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
import vit_tf_amd as vt                                        # noqa: E402
from oracle import dino_vit, feature_volume as ofv              # noqa: E402
import helpers                                                  # noqa: E402


def main():
    vol, _ = vt.ct_like_volume(512, 0)
    lo_hi = (float(vol.float().min()), float(vol.float().max()))
    im_sz, feat_out = vt.sizing(tuple(vol.shape), 64, 8)
    out, models = {}, {}
    for arch, seed, axis, w in helpers.WINDOWS512:
        if (arch, seed) not in models:
            models[(arch, seed)] = dino_vit.build_vit(arch, vt.synthetic_state_dict(arch, seed))
        sl = ofv.AXIS_DIMS[axis][0]
        s_lo, s_hi = vt.extract.window_bounds(w, vol.shape[sl], feat_out[sl])
        t0 = time.time()
        ref = helpers.oracle_window(models[(arch, seed)], vol, axis, s_lo, s_hi, lo_hi, im_sz)
        st = helpers.WINDOW_STRIDE
        out[helpers.window_key(arch, seed, axis, w)] = ref[:, ::st, ::st].contiguous().numpy()
        print(f'{arch} seed {seed} axis {axis} window {w}: slices [{s_lo}, {s_hi}) {tuple(ref.shape)} {ref.dtype} in {time.time() - t0:.0f} s', flush=True)
    np.savez_compressed(os.path.join(HERE, 'windows512.npz'), **out)


if __name__ == '__main__':
    main()
