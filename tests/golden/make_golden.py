#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own functions in the build container.

    python tests/golden/make_golden.py            # needs /root/reference (never present on the GPU box)

The reference modules are imported from /root/reference as-is.  Three things the reference expects from
its environment are provided here, none of them part of the algorithm:
  * ``infer.normalize`` -- the name is only bound under ``__main__`` (infer.py:293, torchvision is absent):
    torchvision's definition ``(t - mean[:, None, None]) / std[:, None, None]`` is injected
  * ``icecream`` -- absent; predict_ntf.py:14 only uses it for printing, a no-op stub is registered
  * the DINO model -- fetched from the network by the reference (infer.py:42-43); the oracle's restatement
    (oracle/dino_vit.py) with seeded synthetic weights stands in, driven through the reference's own
    ``compute_qkv`` harness (hook on blocks[-1].attn.qkv, K slicing, permutes, pooling, fp16 sum)
  * bilateral_solver3d.py is broken at the reference's HEAD in this environment: it uses
    ``F.`` without importing it and passes ``tol=`` to SciPy's cg, which SciPy >= 1.14 calls ``rtol``.  Both names are
    bound in that module's namespace before it is called (``F`` = torch.nn.functional, ``cg`` = a wrapper mapping tol
    to rtol); nothing else is touched
Only inputs and expected outputs are stored -- no reference source.
"""
import contextlib
import importlib.util
import io
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import bilateral as obil, dino_vit, feature_volume as ofv, similarity as osim, synthetic as osyn   # noqa: E402
import vit_tf_amd as vt   # noqa: E402  (weights recipe only; no GPU involved)

TINY_ARCH = (128, 3, 2, 8)      # embed_dim, depth, heads, patch -- head dim 64 like every DINO ViT


def load_reference():
    """Import the reference's infer / predict_ntf under private names, with `infer` visible to their imports."""
    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    saved = {k: sys.modules.get(k) for k in ('infer', 'icecream', 'compare_feat_sampling', 'bilateral_solver3d')}
    ice = types.ModuleType('icecream')
    ice.ic = lambda *a, **k: None
    ice.argumentToString = types.SimpleNamespace(register=lambda *_a: (lambda f: f))
    sys.modules['icecream'] = ice
    ref_infer = load('infer', os.path.join(REF, 'infer.py'))
    mean_std = lambda t, mean, std: (t - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1)
    ref_infer.normalize = mean_std
    sys.modules['infer'] = ref_infer
    sys.path.insert(0, REF)
    try:
        ref_ntf = load('ref_predict_ntf', os.path.join(REF, 'predict_ntf.py'))
        ref_syn = load('ref_create_synthetic_volumes', os.path.join(REF, 'create_synthetic_volumes.py'))
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    from scipy.sparse.linalg import cg as scipy_cg
    bs_globals = ref_ntf.apply_bilateral_solver3d.__globals__
    bs_globals['F'] = torch.nn.functional
    bs_globals['cg'] = lambda A, b, x0=None, M=None, maxiter=None, tol=1e-5: scipy_cg(A, b, x0=x0, M=M, maxiter=maxiter, rtol=tol)
    return ref_infer, ref_ntf, ref_syn


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def fresh_model(seed):
    sd = vt.synthetic_state_dict(TINY_ARCH, seed)
    return dino_vit.build_vit(TINY_ARCH, sd), sd


def feature_cases(ref_infer):
    cases = {
        # name: (volume shape, feature_output_size)
        'even': ((24, 16, 32), 3),      # windows of 8 / 8 / 8 slices, integer resize
        'resize': ((20, 12, 28), 2),    # non-integer nearest down-scaling 20->16, 12->8, 28->16
        'overlap': ((10, 10, 10), 4),   # 10 slices -> 4 overlapping windows, 3.2x nearest up-scaling
    }
    out = {}
    for ci, (name, (shape, fos)) in enumerate(cases.items()):
        g = torch.Generator().manual_seed(100 + ci)
        vol = (torch.rand(shape, generator=g) * 3.0 - 1.0).half().float()       # fp16-exact values
        im_sz, feat_out = ofv.sizing(shape, fos, 8)
        rec = {'vol': vol.numpy(), 'fos': fos, 'seed': 7 + ci, 'im_sz': np.array(im_sz), 'feat_out': np.array(feat_out)}
        # --- reference harness, single axes (fresh model per call: the reference never removes its hook) ---
        for ax in 'zyx':
            model, sd = fresh_model(7 + ci)
            res = quiet(ref_infer.compute_qkv, vol, model, 8, im_sz, batch_size=3, slice_along=ax, return_keys='k')
            rec[f'k_{ax}'] = res['k'].numpy()
            got = ofv.k_features_axis(vol, model, 8, im_sz, ax, batch_size=3)
            check_close(f'{name}/k_{ax}', got, res['k'])
        rec['weights_checksum'] = vt.weights.state_dict_checksum(sd)
        # --- reference 'all' mode loop (infer.py:328-333) ---
        acc = defaultdict(float)
        pool = torch.nn.AdaptiveAvgPool3d(output_size=feat_out)
        for ax in ['z', 'y', 'x']:
            model, _ = fresh_model(7 + ci)
            for k, v in quiet(ref_infer.compute_qkv, vol, model, 8, im_sz, pool_fn=pool, batch_size=2, return_keys='k',
                              slice_along=ax).items():
                acc[k] = (torch.as_tensor(acc[k]) + v.squeeze().half())
        rec['k_all'] = acc['k'].numpy()
        model, _ = fresh_model(7 + ci)
        check_close(f'{name}/k_all', ofv.feature_volume(vol, model, 8, fos, 'all', batch_size=2), acc['k'])
        out[name] = rec
    return out


def check_close(what, got, ref, max_ulp_frac=2e-3):
    """Oracle restatement vs reference output: identical up to rare 1-ulp fp16 flips (different GEMM shapes)."""
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    diff = (got - ref).abs()
    frac = float((diff > 0).float().mean())
    rel = float(diff.max() / ref.abs().max())
    print(f'  oracle vs reference {what}: mismatching elements {frac:.2e}, max rel diff {rel:.2e}')
    assert frac <= max_ulp_frac and rel < 2e-3, what


def sampling_case(ref_infer):
    g = torch.Generator().manual_seed(5)
    feat = torch.randn((32, 6, 5, 7), generator=g).half().float()
    rel = torch.rand((16, 3), generator=g) * 2.2 - 1.1          # some coordinates outside [-1, 1]
    rel[0] = torch.tensor([-1.0, 1.0, 0.0])
    rec = {'feat': feat.numpy(), 'rel': rel.numpy()}
    for mode in ('nearest', 'bilinear'):
        ref = ref_infer.sample_features3d(feat, rel.clone(), mode=mode)        # (1, 1, A, F)
        rec[mode] = ref[0, 0].numpy()
        got = osim.sample_features(feat, rel, mode)
        assert torch.equal(got, ref[0, 0]), mode
    # tests/test_vishum.py:18-23 property: nearest sampling at voxel centres == integer indexing
    ext = torch.tensor([48., 40., 56.])
    coord = torch.tensor([[17., 33., 50.], [0., 0., 0.], [47., 39., 55.]])
    relc = (coord + 0.5) / ext * 2.0 - 1.0
    near = ref_infer.sample_features3d(feat, relc.clone(), mode='nearest')[0, 0]
    idx = (coord // 8).long()
    assert torch.equal(near, feat[:, idx[:, 0], idx[:, 1], idx[:, 2]].t())
    return rec


def similarity_case(ref_ntf):
    g = torch.Generator().manual_seed(11)
    feat = torch.nn.functional.normalize(torch.randn((64, 8, 8, 8), generator=g), dim=0)
    # make a few voxels strongly similar to the queries so that the 0.25 threshold and the wrap-around fire
    feat = (feat + 0.8 * feat[:, 2:3, 3:4, 4:5]).half().float()
    feat = torch.nn.functional.normalize(feat, dim=0).half().float()
    volume = torch.zeros((24, 16, 20))            # sim maps (12, 8, 10): non-trivial nearest resize from the 8^3 grid
    ann = {'ntf1': torch.tensor([[4, 6, 8], [5, 7, 9]]), 'ntf2': torch.tensor([[12, 3, 1], [2, 13, 14], [8, 8, 8]])}
    ref = quiet(ref_ntf.compute_similarities, volume.numpy(), feat, {k: v.clone() for k, v in ann.items()})
    rec = {'feat': feat.numpy(), 'vol_shape': np.array(volume.shape)}
    for k, v in ann.items():
        rec[f'ann_{k}'] = v.numpy()
        rec[f'sim_{k}'] = ref[k].numpy()
    got = osim.similarity_maps(tuple(volume.shape), feat, ann)
    for k in ann:
        d = (got[k].int() - ref[k].int()).abs()
        print(f'  oracle vs reference sim_{k}: differing voxels {int((d > 0).sum())} / {d.numel()}, '
              f'max {int(ref[k].max())}, wrapped present: {bool((ref[k] < 3).any() and (ref[k] > 250).any())}')
        assert int((d > 0).sum()) == 0, k
    # label assignment exactly as predict_ntf.py:203-215 (thresholds lowered so both classes appear)
    sims = torch.stack([v.float() for v in ref.values()])
    pred = torch.zeros_like(sims[0]); pred_vals = torch.zeros_like(sims[0])
    for i, sim in enumerate(sims):
        mask = (sim > int(ref_ntf_thresholds()[i] * 255)) & (sim > pred_vals)
        pred[mask] = i + 1
        pred_vals[mask] = sim[mask]
    rec['labels'] = pred.numpy().astype(np.uint8)
    assert np.array_equal(osim.assign_labels(list(ref.values())), rec['labels'])
    # the big-A single-class variant (predict_ntf.py:62-63): > 1024 annotations of one class
    g2 = torch.Generator().manual_seed(12)
    big = {'ntf1': torch.randint(0, 16, (1030, 3), generator=g2)}
    refb = quiet(ref_ntf.compute_similarities, volume.numpy(), feat, {k: v.clone() for k, v in big.items()})
    rec['ann_big'] = big['ntf1'].numpy()
    rec['sim_big'] = refb['ntf1'].numpy()
    gotb = osim.similarity_maps(tuple(volume.shape), feat, big)
    nb = int((gotb['ntf1'].int() - refb['ntf1'].int()).abs().gt(0).sum())
    print(f'  oracle vs reference sim_big: differing voxels {nb}')
    assert nb <= 2
    return rec


def bilateral_case(ref_ntf):
    """(1) apply_bilateral_solver3d on a 20x18x22 target / grey reference; (2) compute_similarities(...,
    bilateral_solver=True) end to end on an 8^3 x 64 feature volume under a 24x16x20 volume."""
    g = torch.Generator().manual_seed(21)
    shape = (20, 18, 22)
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, n) for n in shape], indexing='ij')
    blob = torch.exp(-4 * ((zz - 0.2) ** 2 + yy ** 2 + (xx + 0.1) ** 2))
    ref = (255 * (0.15 + 0.7 * (blob > 0.35).float() + 0.1 * torch.rand(shape, generator=g))).clamp(0, 255).to(torch.uint8)
    target = (blob + 0.3 * torch.randn(shape, generator=g)).clamp(0, 1)
    out = quiet(ref_ntf.apply_bilateral_solver3d, target[None], ref.expand(3, -1, -1, -1),
                grid_params={'sigma_spatial': 7, 'sigma_chroma': 5, 'sigma_luma': 5})
    rec = {'solver_target': target.numpy(), 'solver_ref': ref.numpy(), 'solver_out': out.numpy()}
    mine = obil.solve(target, ref)
    err = float((mine - out).abs().max())
    print(f'  oracle vs reference bilateral solve: max abs diff {err:.3e} (output range {float(out.min()):.3f}..{float(out.max()):.3f})')
    assert err < 1e-6
    conf_ref = ref_ntf.apply_bilateral_solver3d.__globals__['filter_sobel_separated'](ref[None, None].float() / 255.0)[0, 0]
    assert torch.equal(obil.sobel_confidence(ref), conf_ref.max() - conf_ref)

    # end to end through the reference's compute_similarities
    g = torch.Generator().manual_seed(22)
    feat = torch.nn.functional.normalize(torch.randn((64, 8, 8, 8), generator=g), dim=0)
    feat = (feat + 0.9 * feat[:, 2:3, 3:4, 4:5]).half().float()
    feat = torch.nn.functional.normalize(feat, dim=0).half().float()
    vshape = (24, 16, 20)
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, n) for n in vshape], indexing='ij')
    volume = (torch.exp(-3 * (zz ** 2 + yy ** 2 + xx ** 2)) * 900 - 400 + 20 * torch.randn(vshape, generator=g)).float()
    ann = {'ntf1': torch.tensor([[7, 6, 10], [8, 7, 11]]), 'ntf2': torch.tensor([[12, 3, 1], [2, 13, 14], [8, 8, 8]])}
    refs = quiet(ref_ntf.compute_similarities, volume.numpy(), feat, {k: v.clone() for k, v in ann.items()}, True)
    rec.update({'e2e_feat': feat.numpy(), 'e2e_volume': volume.numpy()})
    sims = osim.class_maps_fp32(tuple(vshape), feat, ann)
    for k, v in ann.items():
        rec[f'e2e_ann_{k}'] = v.numpy()
        rec[f'e2e_sim_{k}'] = refs[k].numpy()
        refined = obil.refine_similarity(sims[k], volume, tuple(d // 2 for d in vshape))
        q, _ = osim.quantize_u8(refined)
        d = (q.int() - refs[k].int()).abs()
        d = torch.minimum(d, 256 - d)
        print(f'  oracle vs reference bilateral e2e {k}: differing voxels {int((d > 0).sum())} / {d.numel()} (max {int(d.max())})')
        assert int(d.max()) <= 1 and int((d > 0).sum()) <= 2, k
    return rec


def refinement_case(ref_infer):
    """resample_topk and take_most_dissimilar (infer.py:75-126), outputs of the reference's own functions."""
    g = torch.Generator().manual_seed(31)
    feat = torch.nn.functional.normalize(torch.randn((1, 48, 6, 7, 5), generator=g), dim=1)
    sims = torch.rand((1, 2, 3, 6, 7, 5), generator=g)
    sims[0, 0, 1, 2, 3, 1] = sims[0, 0, 1, 4, 0, 2] = 0.9999                  # a tie inside the top K
    rec = {'rt_feat': feat.numpy(), 'rt_sims': sims.numpy()}
    for K, expo, mode in ((3, 2.0, 'nearest'), (8, 1.5, 'bilinear')):
        ref = quiet(ref_infer.resample_topk, feat.clone(), sims.clone(), K, expo, mode)
        rec[f'rt_out_K{K}'] = ref.numpy()
        got = osim.resample_topk(feat, sims, K, expo, mode)
        err = float((got - ref).abs().max())
        print(f'  oracle vs reference resample_topk K={K}: max abs diff {err:.2e}')
        assert err < 1e-6
    x = torch.randn((60, 24), generator=g)
    x[7] = x[3] * 2.0                                                            # cosine-identical rows
    rec['md_x'] = x.numpy()
    for measure in ('cosine', 'euclidean'):
        ref = quiet(ref_infer.take_most_dissimilar, x.clone(), 9, measure)
        rec[f'md_{measure}'] = ref.numpy()
        mine = osim.take_most_dissimilar(x, 9, measure)
        assert sorted(map(tuple, mine.tolist())) == sorted(map(tuple, ref.tolist())), measure
    return rec


def samplers_case(ref_ntf):
    """compare_feat_sampling.py:19-30 through predict_ntf's import: asked for more samples than the shell holds,
    sample_surface returns the whole shell (index order), i.e. its deterministic candidate set; all structuring elements."""
    g = torch.Generator().manual_seed(11)
    shape = (20, 18, 24)
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, n) for n in shape], indexing='ij')
    blob = (zz ** 2 + 0.8 * yy ** 2 + 0.6 * xx ** 2) < 0.9                      # touches the volume border
    labels = torch.zeros(shape, dtype=torch.uint8)
    labels[blob] = 2
    labels[(xx > 0.3) & blob] = 5
    labels[torch.rand(shape, generator=g) < 0.04] = 0                              # pin holes
    rec = {'labels': labels.numpy()}
    for cls in (2, 5):
        for dist in (1, 2, 3, 4):
            mask = (labels == cls).numpy()
            shell = quiet(ref_ntf.sample_surface, mask, 10 ** 6, dist_from_surface=dist)
            rec[f'shell_c{cls}_d{dist}'] = shell.numpy().astype(np.int16)
    return rec


def ref_ntf_thresholds():
    return [0.486, 0.264, 0.236, 0.68, 0.291]      # predict_ntf.py:208 (a local of its __main__, restated)


def synthetic_case(ref_syn, tmp):
    argv = sys.argv
    sys.argv = ['create_synthetic_volumes.py', tmp, '--size', '16']
    try:
        quiet(ref_syn.main)
    finally:
        sys.argv = argv
    rec = {}
    mine = osyn.synthetic_volumes(16, 0.0)
    for name in ('sphere_thick', 'sphere_filled', 'torus_thick', 'torus_filled'):
        v = np.load(os.path.join(tmp, f'{name}.npy'))
        l = np.load(os.path.join(tmp, f'{name}_label.npy'))
        rec[f'{name}'] = v
        rec[f'{name}_label'] = l
        assert np.array_equal(mine[name][0].numpy(), v) and np.array_equal(mine[name][1].numpy(), l), name
    return rec


def main():
    if not os.path.isdir(REF):
        raise SystemExit('make_golden.py needs the reference checkout at /root/reference')
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref_infer, ref_ntf, ref_syn = load_reference()
    print('sample_surface candidate sets')
    np.savez_compressed(os.path.join(HERE, 'samplers.npz'), **samplers_case(ref_ntf))
    if '--only-samplers' in sys.argv:
        return
    print('feature-volume cases (reference compute_qkv harness + oracle ViT)')
    for name, rec in feature_cases(ref_infer).items():
        np.savez_compressed(os.path.join(HERE, f'featvol_{name}.npz'), **rec)
    print('sample_features3d')
    np.savez_compressed(os.path.join(HERE, 'sampling.npz'), **sampling_case(ref_infer))
    print('compute_similarities + labels')
    np.savez_compressed(os.path.join(HERE, 'similarity.npz'), **similarity_case(ref_ntf))
    print('bilateral solver (bilateral_solver3d + compute_similarities(bilateral_solver=True))')
    np.savez_compressed(os.path.join(HERE, 'bilateral.npz'), **bilateral_case(ref_ntf))
    print('resample_topk / take_most_dissimilar')
    np.savez_compressed(os.path.join(HERE, 'refinement.npz'), **refinement_case(ref_infer))
    print('create_synthetic_volumes --size 16')
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        np.savez_compressed(os.path.join(HERE, 'synthetic16.npz'), **synthetic_case(ref_syn, tmp))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f'  {f}: {os.path.getsize(os.path.join(HERE, f)) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
