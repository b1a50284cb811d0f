"""Drop-in for the reference's ``evaluate_similarities.py`` (:37-83): scores exported ``predictions.npy``
against a label volume and writes ``metrics.json``.  The confusion matrices are counted on the GPU
(vittf_confusion_matrix); kept so the workflow around the hot path runs unchanged.  Works without icecream."""
import json
from argparse import ArgumentParser
from pathlib import Path
from pprint import pprint

import numpy as np
import torch
import torch.nn.functional as F

import vit_tf_amd as vt

label2idx = {'background': 0, 'liver': 1, 'bladder': 2, 'lung': 3, 'kidney': 4, 'bone': 5}
idx2label = ['liver', 'bladder', 'lung', 'kidney', 'bone']


def binary_scores(target, pred):
    """precision / recall / f1 / iou per class [0, 1], 2x2 confusion matrix, accuracy (:65-68); counted on the GPU."""
    acc, prec, rec, f1, iou, cm = vt.scores.scores(target, pred)
    return acc, prec.tolist(), rec.tolist(), f1.tolist(), iou.tolist(), cm.tolist()


def evaluate(data_dir, label_fn, label_names):
    data_dir, label_fn = Path(data_dir), Path(label_fn)
    assert (data_dir / 'predictions.npy').exists()
    assert label_fn.exists()
    assert (data_dir / 'metadata.json').exists()
    with (data_dir / 'metadata.json').open('r', encoding='UTF-8') as f:
        metadata = json.load(f)
    labels_orig = torch.as_tensor(np.load(label_fn, allow_pickle=True)[()])
    preds = {k: torch.as_tensor(v) for k, v in np.load(data_dir / 'predictions.npy', allow_pickle=True)[()].items()}
    results = {}
    labels_dev = vt.samplers.device_labels(labels_orig)          # one upload, every class resized from it
    for name, key in zip(label_names, sorted(preds.keys())):
        p = preds[key]
        # (:63) F.interpolate((labels == idx)[None, None], p.shape, mode='nearest'): mask + resize in one kernel
        target = vt.scores.resize_nearest_u8(labels_dev, p.shape[-3:], equals=label2idx[name], keep_on_device=True)
        acc, prec, rec, f1, iou, cm = binary_scores(target.reshape(-1), p.reshape(-1).numpy())
        results[name] = {'accuracy': acc, 'precision': prec, 'recall': rec, 'f1': f1, 'iou': iou,
                         'confusion_matrix': cm, 'annotation_time': metadata[key]['time'],
                         'num_annotations': metadata[key]['num_annotations']}
    return results


if __name__ == '__main__':
    parser = ArgumentParser()
    parser.add_argument('--data', type=Path, help='Path to features, annotations, volume etc.')
    parser.add_argument('--label', type=Path, default='userstudy/labels-10.npy', help='Path to label volume')
    parser.add_argument('--labels', type=str, nargs='+', default=['lung', 'liver', 'kidney'], help='Labels found in predictions (in order)')
    args = parser.parse_args()
    results = evaluate(args.data, args.label, args.labels)
    pprint(results)
    with open(Path(args.data) / 'metrics.json', 'w') as f:
        json.dump(results, f)
