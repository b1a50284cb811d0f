"""Drop-in for the reference's ``predict_ntf.py``: per-class similarity volumes + label volume, MI355X-native.

Same function and CLI surface as /root/reference/predict_ntf.py (compute_similarities :24-101, flags
:105-112, directory contract :119-172, label assignment :203-215, outputs :216, :249-250); the query
sampling, the voxel x query contraction, threshold / power / class mean, quantisation and the label
arg-max run in libvittf's HIP kernels (``--gpu`` is implied: there is no CPU path), and so does the bilateral-solver
post-process (``--bilateral-solver``, bilateral.hip).
"""
import json
import sys
import time
from argparse import ArgumentParser
from pathlib import Path
from pprint import pprint

import numpy as np
import torch
import torch.nn.functional as F

import vit_tf_amd as vt
from vit_tf_amd.samplers import sample_uniform, sample_surface, sample_both, device_labels
from infer import make_3d, make_4d, make_5d, sample_features3d, norm_minmax   # noqa: F401  (re-exported like the reference)

sampling_modes = {
    'uniform': sample_uniform,
    'surface': sample_surface,
    'both': sample_both,
    'annotated': lambda *args, **kwargs: None,
}

ct_org_names = ['liver', 'bladder', 'lung', 'kidney', 'bone']
ct_org_thresholds = [0.486, 0.264, 0.236, 0.68, 0.291]


def compute_similarities(volume, features, annotations, bilateral_solver=False, keep_on_device=False):
    """(:24-101) volume (W, H, D), features (F, W', H', D'), annotations {name: (N, 3)} ->
    {name: uint8 (W//2, H//2, D//2)} (CPU tensors like the reference's, unless keep_on_device)."""
    return vt.compute_similarities(volume, features, annotations, bilateral_solver=bilateral_solver,
                                   keep_on_device=keep_on_device)


def assign_labels(similarities):
    """(:203-215) running maximum over the class maps with the CT-ORG thresholds -> uint8 numpy volume."""
    return vt.assign_labels(similarities, ct_org_thresholds)


def _metrics(labels, pred, names):
    """(:228-246) accuracy, per-class precision / recall / F1 / IoU and the confusion matrix (counted on the GPU)."""
    acc, prec, rec, f1, iou, cm = vt.scores.scores(labels, pred)
    return {
        'mAcc': float(acc),
        'precision': dict(zip(names, prec.tolist())), 'mPrec': float(prec.mean()),
        'recall': dict(zip(names, rec.tolist())), 'mRec': float(rec.mean()),
        'f1': dict(zip(names, f1.tolist())), 'mF1': float(f1.mean()),
        'iou': dict(zip(names, iou.tolist())), 'mIoU': float(iou.mean()),
        'confusion_matrix': dict(zip(names, cm.tolist())),
    }


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('--data', type=str, help='directory holding volume, features and annotations / labels')
    parser.add_argument('--bilateral-solver', action='store_true', help='refine every class map with the 3-D bilateral solver')
    parser.add_argument('--load-sims', action='store_true', help='reuse a similarities file written by an earlier run')
    parser.add_argument('--num-samples', type=float, default=0.0, help='annotations sampled per class from the labels (0: use the annotation file)')
    parser.add_argument('--sampling-mode', type=str, choices=['uniform', 'surface', 'both'], default='both', help='where samples are drawn from')
    parser.add_argument('--gpu', action='store_true', help='Use GPU (always on in this build)')
    args = parser.parse_args(argv)
    d = Path(args.data)
    if args.num_samples == 0.0:
        args.sampling_mode = 'annotated'
    bls = 'bls' if args.bilateral_solver else ''
    tag = f'{args.num_samples}{args.sampling_mode}{bls}'
    if (d / f'ntf_pred{tag}.npy').exists():
        print(f'Already inferred NTF preds for {d} using sampling mode {args.sampling_mode} and {args.num_samples} samples')
        sys.exit(0)
    print(f'Inferring for {d} using sampling mode {args.sampling_mode} and {args.num_samples} samples')

    feat_fns = [p for p in d.iterdir() if 'features' in str(p) and 'pred' not in str(p)]
    if not feat_fns:
        raise ValueError(f'No features found in {d}')
    feat_fn = max(feat_fns, key=lambda p: p.stat().st_size)
    if len(feat_fns) > 1:
        print(f'Found multiple features in {d}. Using largest one {feat_fn.name}.')

    volume = np.flip(np.load(d / 'volume.npy', allow_pickle=True).astype(np.float32), axis=-3).copy()
    labels = None
    if (d / 'labels.npy').exists():
        labels = np.flip(np.load(d / 'labels.npy', allow_pickle=True)[()], axis=-3).copy()
    else:
        assert args.num_samples == 0.0, 'Cannot sample labels if they are not provided'
    features = np.load(feat_fn, allow_pickle=True)[()]
    features = torch.as_tensor(features['k'] if isinstance(features, dict) else features).squeeze()

    if args.num_samples == 0.0:
        annotations = np.load(d / 'annotations.npy', allow_pickle=True)[()]
    elif args.num_samples > 0.0:
        draw = sampling_modes[args.sampling_mode]
        annotations = {}
        labels_dev = device_labels(labels)                 # one upload; every class draws from it
        for i in range(1, int(labels.max()) + 1):
            total = int((labels_dev == i).sum().item())
            n = min(int(args.num_samples), total) if args.num_samples > 1.0 else int(args.num_samples * total)
            if n > 0:
                annotations[f'ntf{i}'] = draw(labels_dev, n, thin_to_reasonable=True, class_id=i)
    else:
        raise Exception(f'Invalid value for --num-samples: {args.num_samples}')

    print(f'Computing similarties for {tuple(volume.shape)} with features {tuple(features.shape)}')
    t0 = time.time()
    t1 = t0
    if args.load_sims:
        similarities = {k: torch.as_tensor(v) for k, v in np.load(d / 'similarities.npy', allow_pickle=True)[()].items()}
        t2 = t1
    else:
        t1 = time.time()
        if sum(int(torch.as_tensor(v).shape[0]) for v in annotations.values()) > 10000:     # (:185-187) one class per call
            similarities = {k: compute_similarities(volume, features, {k: v}, bilateral_solver=args.bilateral_solver,
                                                    keep_on_device=True)[k] for k, v in annotations.items()}
        else:
            similarities = compute_similarities(volume, features, annotations, bilateral_solver=args.bilateral_solver,
                                                keep_on_device=True)
        torch.cuda.synchronize()
        t2 = time.time()
    print('Similarities:', {k: v.shape for k, v in similarities.items()})
    pred = assign_labels(similarities)
    np.save(d / f'ntf_pred{tag}.npy', pred)
    if tuple(pred.shape[-3:]) != tuple(volume.shape[-3:]):
        pred = vt.scores.resize_nearest_u8(pred, tuple(volume.shape[-3:]))       # (:217-218) nearest up-sample, on the GPU
    print('Pred:', pred.shape, pred.min(), pred.max())
    print('NTF fit time:', t1 - t0)
    print('NTF predict time:', t2 - t1)
    if labels is None:
        sys.exit(0)
    ntf_metrics = _metrics(labels, pred, ['background'] + list(annotations.keys()))
    ntf_metrics['fit_time'] = t1 - t0
    ntf_metrics['predict_time'] = t2 - t1
    print('NTF Metrics:')
    pprint(ntf_metrics)
    with open(d / f'ntf_metrics{tag}.json', 'w') as f:
        json.dump(ntf_metrics, f)


if __name__ == '__main__':
    main()
