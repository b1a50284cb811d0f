"""GPU: BASELINE.json's configurations at FULL size through the product's entry points.

  configs[1]  256^3 volume -> 768 slices of 512 x 512 (2x nearest up-sampling), ViT-S/8
  configs[2]  512^3 CT-like volume -> 1536 slices of 512 x 512, ViT-S/8          (the metric's configuration)
  configs[4]  512^3 volume, 5 x 1024 annotations (matrix-core similarity) + 3-D bilateral solver -> 256^3 maps

The CPU oracle cannot run a whole volume (a 512^3 extraction is > 1 h on the host cores), so whole volumes are held to
size-independent properties -- every value finite, bits independent of the engine batch, the z -> y -> x fp16 sum of the
three pooled axes reproducing the volume bit for bit -- and ONE pooling window per axis is compared with the oracle run
on just the 4 / 8 slices that window averages (same global min / max), at the operand type whose bound is the 1e-3 of
BASELINE.json's north_star (fp16: the engine's default and the reference's own GPU autocast type, infer.py:309).
The oracle's windows of the 512^3 volume are a committed fixture (tests/golden/windows512.npz: every 4th feature row / column,
tests/golden/make_window_goldens.py -- two minutes of host time per window otherwise); the 256^3 one runs the oracle live.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import vit_tf_amd as vt
from oracle import dino_vit, feature_volume as ofv, similarity as osim
from helpers import rel_fro, oracle_window as _oracle_window, load_golden, window_key, WINDOW_STRIDE

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

pytestmark = pytest.mark.gpu
TOL = 1e-3                      # relative Frobenius error of fp16-operand features against the fp32 CPU path


@pytest.fixture(scope='module')
def vits8(gpu):
    sd = vt.synthetic_state_dict('vits8', 0)
    return sd, vt.HipViT(sd, 'vits8', 'fp16', device=gpu)


@pytest.fixture(scope='module')
def oracle_vits8(vits8):
    return dino_vit.build_vit('vits8', vits8[0])


def _window_ref(oracle, vol, ax, w, s_lo, s_hi, lo_hi, im_sz):
    """The oracle's pooled window and the stride it is known at: live (oracle = a model), or from the fixture (oracle = the
    (arch, seed) the fixture was made with)."""
    if isinstance(oracle, tuple):
        return torch.from_numpy(load_golden(GOLDEN, 'windows512.npz')[window_key(*oracle, ax, w)]), WINDOW_STRIDE
    return _oracle_window(oracle, vol, ax, s_lo, s_hi, lo_hi, im_sz), 1


def _whole_volume_checks(gpu, vits8, oracle, vol, windows, tol=TOL):
    sd, model = vits8
    D = model.embed_dim
    dvol = vt.DeviceVolume(vol, gpu)
    im_sz, feat_out = vt.sizing(dvol.shape, 64, 8)
    assert im_sz == (512, 512, 512) and feat_out == (64, 64, 64)
    feats = vt.feature_volume(None, model, 64, 'all', 32, dvol=dvol)
    assert feats.shape == (D, 64, 64, 64) and feats.dtype == torch.float16
    assert bool(torch.isfinite(feats).all())
    for eb in (31, vt.extract.engine_batch_for(4097, D)):                   # other batchings of the same slices (the default: 512 / 256)
        again = vt.feature_volume(None, model, 64, 'all', eb, dvol=dvol)
        assert torch.equal(feats, again), f'bits depend on the engine batch ({eb} vs 32)'
        del again
    pooled = {ax: vt.pooled_axis(None, model, ax, im_sz, feat_out, 32, dvol=dvol) for ax in 'zyx'}
    total = (pooled['z'] + pooled['y']) + pooled['x']                        # fp16 adds, one rounding each (infer.py:330-332)
    assert torch.equal(total, feats), 'z -> y -> x fp16 sum of the pooled axes does not reproduce the volume'
    lo_hi = (float(dvol.minmax[0]), float(dvol.minmax[1]))
    assert lo_hi == (float(vol.float().min()), float(vol.float().max()))
    errs = {}
    for ax, w in windows.items():
        sl = ofv.AXIS_DIMS[ax][0]
        s_lo, s_hi = vt.extract.window_bounds(w, dvol.shape[sl], feat_out[sl])
        ref, st = _window_ref(oracle, vol, ax, w, s_lo, s_hi, lo_hi, im_sz)
        got = pooled[ax].select(1 + sl, w).cpu()
        assert got.shape == (D, 64, 64)
        got = got[:, ::st, ::st]
        assert got.shape == ref.shape and ref.dtype == torch.float16
        errs[ax] = rel_fro(got, ref)
        print(f'{tuple(dvol.shape)} axis {ax} window {w} = slices [{s_lo}, {s_hi}): rel fro {errs[ax]:.2e} vs the CPU oracle')
    assert max(errs.values()) <= tol, errs
    return feats


def test_config1_256_whole_volume(gpu, vits8, oracle_vits8):
    """BASELINE configs[1]: every slice of the 256^3 torus volume (2x nearest up-sampling to 512 x 512)."""
    vol, _ = vt.synthetic_volume('torus_filled', 256, 0.1, 0)
    _whole_volume_checks(gpu, vits8, oracle_vits8, vol, {'y': 40})       # one window against the oracle (z / x: configs[2] below)


@pytest.fixture(scope='module')
def ct512():
    return vt.ct_like_volume(512, 0)


@pytest.fixture(scope='module')
def feats512(gpu, vits8, ct512):
    """BASELINE configs[2] on one GPU (the metric's configuration): the 512^3 CT-like volume, 1536 slices, no up-sampling
    (537 MB resident volume, 1.6 GB of K features per axis).  Checked once, reused by the configs[4] tests below."""
    return _whole_volume_checks(gpu, vits8, ('vits8', 0), ct512[0], {'z': 37, 'y': 11, 'x': 50})


def test_config2_512_whole_volume(feats512):
    assert feats512.shape == (384, 64, 64, 64)


def test_config3_vitb8_fp8_attention_512_whole_volume(gpu, ct512):
    """BASELINE configs[3]: the 512^3 volume through ViT-B/8 (D = 768, 12 heads) with the fp8 (e4m3, block-scaled MFMA)
    attention path.  Same whole-volume properties as configs[2] (the per-(slice, head) quantisation scales make the bits
    independent of the batching as well); one pooled window against the fp32 CPU oracle at the path's own stated bound
    (5e-2 relative Frobenius: 3-bit mantissas on q, k, v and P in 11 attention layers; measured 1.5e-2), and the same
    window with 16-bit attention at the contract's 1e-3."""
    sd = vt.synthetic_state_dict('vitb8', 2)
    model8 = vt.HipViT(sd, 'vitb8', 'fp16', device=gpu, attention='fp8')
    feats = _whole_volume_checks(gpu, (sd, model8), ('vitb8', 2), ct512[0], {'y': 29}, tol=5e-2)
    assert feats.shape == (768, 64, 64, 64)
    del model8, feats
    torch.cuda.empty_cache()
    model16 = vt.HipViT(sd, 'vitb8', 'fp16', device=gpu)
    dvol = vt.DeviceVolume(ct512[0], gpu)
    im_sz, feat_out = vt.sizing(dvol.shape, 64, 8)
    pooled = vt.pooled_axis(None, model16, 'y', im_sz, feat_out, 32, dvol=dvol)
    sl = ofv.AXIS_DIMS['y'][0]
    ref, st = _window_ref(('vitb8', 2), None, 'y', 29, None, None, None, None)
    e = rel_fro(pooled.select(1 + sl, 29).cpu()[:, ::st, ::st], ref)
    print(f'ViT-B/8, 16-bit attention, axis y window 29: rel fro {e:.2e} vs the CPU oracle')
    assert e <= TOL


def _annotations_5x1024(label):
    """configs[4]: 1024 annotations for each of 5 classes (the evaluate_similarities 1024 preset: sample_both = 512
    uniform + 512 surface per class; the three label classes of the CT-like volume + two sub-regions as classes 4, 5)."""
    torch.manual_seed(0)
    dev_lab = vt.samplers.device_labels(label)
    ann = {}
    for i in (1, 2, 3):
        ann[f'ntf{i}'] = vt.samplers.sample_both(dev_lab, 1024, thin_to_reasonable=True, class_id=i).cpu()
    g = torch.Generator().manual_seed(1)
    for i, (cls, lo) in enumerate(((1, 0), (2, 256)), start=4):               # one half of classes 1 / 2 along dim 0
        idx = (label[lo:lo + 256:2, ::2, ::2] == cls).nonzero() * 2
        idx[:, 0] += lo
        ann[f'ntf{i}'] = idx[torch.randperm(idx.shape[0], generator=g)[:1024]]
    assert all(v.shape == (1024, 3) for v in ann.values())
    return ann


@pytest.fixture(scope='module')
def sims512(gpu, feats512, ct512):
    """5 x 1024 queries against the normalised 64^3 x 384 feature volume of the 512^3 input: GPU maps (matrix-core
    similarity kernel) and oracle maps, 256^3 uint8 each."""
    vol, label = ct512
    ann = _annotations_5x1024(label)
    feat = F.normalize(feats512.float(), dim=0).half()                          # compare_feat_sampling.py:45
    got = vt.compute_similarities(vol, feat, ann)
    ref = osim.similarity_maps(tuple(vol.shape), feat.float().cpu(), ann)
    return ann, feat, got, ref


def test_config4_1024_queries_similarity(sims512):
    ann, _, got, ref = sims512
    for k in ann:
        assert got[k].shape == (256, 256, 256) and got[k].dtype == torch.uint8
        assert int(ref[k].max()) == 255 or int(ref[k].max()) >= 250
        d = (got[k].int() - ref[k].int()).abs()
        d = torch.minimum(d, 256 - d)
        frac = float((d > 0).float().mean())
        print(f'{k}: {int((d > 0).sum())} of {d.numel()} voxels differ by 1 LSB ({frac:.2e}), max |diff| {int(d.max())}')
        assert int(d.max()) <= 1 and frac <= 1e-3, k


def test_config4_labels_from_gpu_maps_vs_oracle_maps(sims512):
    """Label volume (predict_ntf.py:203-215) assigned from the GPU maps against the one assigned from the ORACLE's maps.
    The label kernel itself is bit-exact on equal maps (test_labels_bit_exact_random); the maps differ by +-1 LSB on
    isolated voxels (fp32 summation order of 384-term dot products, two different machines), and such a voxel changes
    its label only when it sits exactly on a class threshold or on a tie between two classes."""
    ann, _, got, ref = sims512
    lab_gpu = vt.assign_labels(got)
    lab_ref = osim.assign_labels([ref[k] for k in ann])
    mism = int((lab_gpu != lab_ref).sum())
    n_lsb = sum(int((got[k] != ref[k]).sum()) for k in ann)
    print(f'labels: {mism} of {lab_ref.size} voxels differ ({mism / lab_ref.size:.2e}); the 5 maps differ on {n_lsb} voxels')
    assert len(np.unique(lab_ref)) >= 3
    assert mism <= n_lsb and mism <= 2e-5 * lab_ref.size
    assert np.array_equal(lab_gpu, osim.assign_labels([got[k] for k in ann]))      # bit-exact on identical maps


def test_config4_residue_pinned_to_fp64_truth(gpu, sims512):
    """The +-1 LSB voxels (and with them the differing labels) against GROUND TRUTH: the per-class value of every voxel of
    the 64^3 grid in fp64 -- from the same fp16 features and the same fp32 query vectors -- through predict_ntf.py:65,
    71-72 (dot, >= 0.25, ** 2.5, mean over the class) and :98-99 (255 / (0.99 max), truncation, wrap-around).
      * the GPU maps and the oracle maps are EACH within one quantisation step of the truth everywhere, and both are off
        on well under 1e-4 of the voxels: neither is biased (a systematic error of the hi + lo query split, or of the
        MFMA summation order, would put the GPU off on many voxels and in one direction);
      * wherever GPU and oracle disagree (or either misses the truth), the voxel is one that fp32 arithmetic cannot decide:
        its true scaled value sits on a quantisation boundary (|scaled - nearest integer| <= 2^-20 relative), or one of the
        class's 1024 dot products sits on the 0.25 threshold of predict_ntf.py:71 (|dot - 0.25| <= 2e-6: keeping or dropping
        that one term moves the scaled value by a few hundredths, across a boundary that is not otherwise close) -- which
        side such a voxel falls on is decided by the summation order of a 384-term fp32 dot product, on any two machines;
      * the labels differ only on voxels where some class map differs."""
    ann, feat, got, ref = sims512
    vol_shape = (512, 512, 512)
    dev = gpu
    f64 = feat.to(dev).double().reshape(feat.shape[0], -1)                          # (384, 64^3), the fp16 values exactly
    coords = torch.cat([torch.as_tensor(v) for v in ann.values()])
    qf = osim.sample_features(feat.float().cpu(), osim.rel_coords(coords, vol_shape), 'bilinear')   # fp32 queries (A, F)
    n_off_gpu = n_off_ref = n_dis = n_thr = 0
    worst = 0.0
    start = 0
    signs = []
    for k, v in ann.items():
        n = torch.as_tensor(v).shape[0]
        q64 = qf[start:start + n].to(dev).double()
        start += n
        acc = torch.zeros(f64.shape[1], dtype=torch.float64, device=dev)
        thr = torch.full((f64.shape[1],), 1.0, dtype=torch.float64, device=dev)     # distance of the nearest dot from 0.25
        for a0 in range(0, n, 128):                                                 # (128, 262144) fp64 blocks
            d = q64[a0:a0 + 128] @ f64
            acc += torch.where(d >= 0.25, d, torch.zeros((), dtype=torch.float64, device=dev)).pow(2.5).sum(0)
            thr = torch.minimum(thr, (d - 0.25).abs().min(0).values)
        sim = acc / n
        scaled = 255.0 / (0.99 * sim.max()) * sim
        truth = (scaled.floor().long() % 256).reshape(64, 64, 64)
        g = got[k][::4, ::4, ::4].to(dev).long()                                    # nearest resize 64 -> 256: out i <- src i // 4
        r = ref[k][::4, ::4, ::4].to(dev).long()
        assert torch.equal(got[k].to(dev).long(), g.repeat_interleave(4, 0).repeat_interleave(4, 1).repeat_interleave(4, 2))

        def wrap(d):
            d = d.abs()
            return torch.minimum(d, 256 - d)
        dg, dr = wrap(g - truth), wrap(r - truth)
        assert int(dg.max()) <= 1 and int(dr.max()) <= 1, k
        n_off_gpu += int((dg > 0).sum()); n_off_ref += int((dr > 0).sum())
        signs.append(int(((g - truth + 128) % 256 - 128).sum()))                    # net direction of the GPU's misses
        dis = (g != r).reshape(-1)
        n_dis += int(dis.sum())
        # every voxel where the two disagree, or where either side misses the truth: undecidable in fp32 (see above)
        odd = dis | ((dg > 0) | (dr > 0)).reshape(-1)
        if bool(odd.any()):
            sc = scaled[odd]
            rel = (sc - sc.round()).abs() / sc.abs().clamp_min(1.0)
            on_threshold = thr[odd] <= 2e-6
            n_thr += int(on_threshold.sum())
            undecidable = (rel <= 2.0 ** -20) | on_threshold
            assert bool(undecidable.all()), (k, rel[~undecidable].tolist(), thr[odd][~undecidable].tolist())
            worst = max(worst, float(rel[~on_threshold].max()) if bool((~on_threshold).any()) else 0.0)
    nvox = 5 * 64 ** 3
    print(f'vs fp64 truth over {nvox} voxel-classes: GPU off by 1 LSB on {n_off_gpu}, oracle on {n_off_ref}; they disagree on '
          f'{n_dis}; net direction of the GPU misses per class {signs}; {n_thr} of these voxels have a dot product on the 0.25 '
          f'threshold, the others lie within {worst:.2e} (relative) of a quantisation boundary')
    assert n_off_gpu <= 1e-4 * nvox and n_off_ref <= 1e-4 * nvox
    assert n_off_gpu <= 4 * max(n_off_ref, 8)                   # the GPU is as right as the oracle, not merely close to it
    lab_gpu, lab_ref = vt.assign_labels(got), osim.assign_labels([ref[k] for k in ann])
    differs = np.zeros(lab_ref.shape, dtype=bool)
    for k in ann:
        differs |= (got[k] != ref[k]).numpy()
    assert not (lab_gpu != lab_ref)[~differs].any()


def test_config3_768_feature_volume_5x1024_queries(gpu, ct512):
    """The 1024-queries-per-class preset on a ViT-B/8-sized feature volume (768 x 64^3 fp16, 403 MB): the matrix-core kernel
    (two 384-feature units per query chunk) against the oracle's einsum formulation on the host."""
    vol, label = ct512
    ann = _annotations_5x1024(label)
    g = torch.Generator().manual_seed(11)
    base = torch.randn(768, 8, 8, 8, generator=g)
    feat = torch.nn.functional.interpolate(base[None], size=(64, 64, 64), mode='trilinear', align_corners=False)[0]
    feat = feat + 0.35 * torch.randn(768, 64, 64, 64, generator=g)           # smooth structure + noise: maps with a wide range
    feat = F.normalize(feat, dim=0).half()
    got = vt.compute_similarities(vol, feat, ann)
    assert vt._lib.kernel_name('similarity') == 'sim_mfma_kernel<F 768>'
    ref = osim.similarity_maps(tuple(vol.shape), feat.float(), ann)
    for k in ann:
        assert got[k].shape == (256, 256, 256) and got[k].dtype == torch.uint8
        d = (got[k].int() - ref[k].int()).abs()
        d = torch.minimum(d, 256 - d)
        frac = float((d > 0).float().mean())
        print(f'768 features, {k}: {int((d > 0).sum())} of {d.numel()} voxels differ by 1 LSB ({frac:.2e}), max |diff| {int(d.max())}, '
              f'map max {int(ref[k].max())}')
        assert int(d.max()) <= 1 and frac <= 1e-3, k


def test_config4_bilateral_solver_at_size(gpu, sims512):
    """The --bilateral-solver branch at configs[4]'s size: 512^3 volume -> 256^3 maps, two of the five classes against the
    CPU restatement of the solver (fp64 CG on both sides; the fp32 trilinear resizes may move a voxel across a uint8 /
    luma-bin edge)."""
    ann, feat, _, _ = sims512
    sub = {k: ann[k] for k in ('ntf2', 'ntf3')}
    # The grey reference for the solver is a smooth 512^3 volume with mild noise, NOT the CT-like benchmark volume: on
    # that one the reference's solver returns an all-zero map (a bilateral vertex whose voxels all have zero Sobel
    # confidence makes the initial guess b / splat(c) NaN, the NaN spreads through the CG dot products and nan_to_num
    # (bilateral_solver3d.py:245) turns the whole crop into zeros; the oracle and the GPU path both reproduce that --
    # checked in round 2 -- but a comparison of two zero maps tests nothing).
    ax = torch.linspace(-1, 1, 512)
    vol = 800 * torch.exp(-3 * (ax - 0.1) ** 2).view(-1, 1, 1) * torch.exp(-3 * (ax + 0.2) ** 2).view(1, -1, 1) * \
        torch.exp(-3 * ax ** 2).view(1, 1, -1) - 300
    vol = vol + 15 * torch.randn(vol.shape, generator=torch.Generator().manual_seed(5))
    got = vt.compute_similarities(vol, feat, sub, bilateral_solver=True)
    ref = osim.similarity_maps(tuple(vol.shape), feat.float().cpu(), sub, volume=vol)
    for k in sub:
        assert got[k].shape == (256, 256, 256) and got[k].dtype == torch.uint8
        d = (got[k].int() - ref[k].int()).abs()
        d = torch.minimum(d, 256 - d)
        frac = float((d > 0).float().mean())
        print(f'bilateral {k}: {frac:.2e} of the voxels differ, max |diff| {int(d.max())}, '
              f'{float((d > 1).float().mean()):.2e} by more than 1 step')
        assert int(ref[k].max()) >= 250 and frac <= 0.01 and float((d > 1).float().mean()) <= 1e-3, k
