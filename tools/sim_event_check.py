#!/usr/bin/env python3
"""Is the HIP-event duration of the 16-query similarity accumulation launch right?  The same launch once per profiler
scope (what bench.py's roofline_similarity reads) against VITTF_SIM_REPEAT=50 launches inside one scope (duration / 50:
any per-scope error of the event pair is amortised), and against the wall clock of the repeated launches.
Run each setting in its own process:  VITTF_SIM_REPEAT=1|50 python tools/sim_event_check.py [fos]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_tf_amd as vt  # noqa: E402


def main():
    fos = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    rep = max(1, int(os.environ.get('VITTF_SIM_REPEAT', '1')))
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    kind = os.environ.get('FEAT', 'unit')      # unit: unit-norm noise (no dot product reaches the 0.25 threshold); gauss: N(0, 1)
    feat = torch.randn(384, fos, fos, fos, generator=g)            # features (every second one does); big: N(0, 1) x 8
    feat = (torch.nn.functional.normalize(feat, dim=0) * (16 if kind == 'unit16' else 1) if kind.startswith('unit') else feat * (8 if kind == 'big' else 1)).half().to(dev)   # unit16: the bits of 'unit' with the exponent shifted by 4 (dot products x 256: many pass)
    vol = torch.zeros(8 * fos, 8 * fos, 8 * fos, dtype=torch.float16)
    n_q = int(os.environ.get('QUERIES', '16'))
    ann = {'q': torch.randint(0, 8 * fos, (n_q, 3), generator=g)}
    for _ in range(3):
        vt.compute_similarities(vol, feat, ann, keep_on_device=True)
    torch.cuda.synchronize()
    vt._lib.profiler_enable(True, classes=['similarity'])
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        vt.compute_similarities(vol, feat, ann, keep_on_device=True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    ms, scopes = vt._lib.profiler_collect()['similarity']
    vt._lib.profiler_enable(False)
    nbytes = feat.numel() * 2 + feat[0].numel() * 4
    per = ms / scopes / rep
    print(f'{kind} features, fos {fos}, {n_q} queries, VITTF_SIM_MFMA_MIN={os.environ.get("VITTF_SIM_MFMA_MIN", "default")}: {rep} launch(es) per scope: {per * 1e3:.1f} us per launch by events ({nbytes / per / 1e6:.0f} GB/s algorithmic); '
          f'whole call {wall * 1e3:.3f} ms wall', flush=True)


if __name__ == '__main__':
    main()
