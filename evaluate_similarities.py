"""Drop-in for the reference's ``evaluate_similarities.py`` (:37-83): scores exported ``predictions.npy``
against a label volume and writes ``metrics.json``.  Scoring only -- O(Nvox) CPU bookkeeping, no kernel
(SURVEY.md 3.3); kept so the workflow around the hot path runs unchanged.  Works without icecream."""
import json
from argparse import ArgumentParser
from pathlib import Path
from pprint import pprint

import numpy as np
import torch
import torch.nn.functional as F

label2idx = {'background': 0, 'liver': 1, 'bladder': 2, 'lung': 3, 'kidney': 4, 'bone': 5}
idx2label = ['liver', 'bladder', 'lung', 'kidney', 'bone']


def binary_scores(target, pred):
    """precision / recall / f1 / iou per class [0, 1], 2x2 confusion matrix, accuracy (:65-68)."""
    try:
        from sklearn.metrics import precision_recall_fscore_support, jaccard_score, confusion_matrix, accuracy_score
        prec, rec, f1, _ = precision_recall_fscore_support(target, pred, average=None)
        return (accuracy_score(target, pred), prec.tolist(), rec.tolist(), f1.tolist(),
                jaccard_score(target, pred, average=None).tolist(), confusion_matrix(target, pred).tolist())
    except ImportError:
        t, p = np.asarray(target).astype(np.int64), np.asarray(pred).astype(np.int64)
        k = int(max(t.max(), p.max())) + 1
        cm = np.zeros((k, k), dtype=np.int64)
        np.add.at(cm, (t, p), 1)
        tp = np.diag(cm).astype(np.float64)
        with np.errstate(divide='ignore', invalid='ignore'):
            prec = np.nan_to_num(tp / cm.sum(0)); rec = np.nan_to_num(tp / cm.sum(1))
            f1 = np.nan_to_num(2 * prec * rec / (prec + rec)); iou = np.nan_to_num(tp / (cm.sum(0) + cm.sum(1) - tp))
        return float(tp.sum() / cm.sum()), prec.tolist(), rec.tolist(), f1.tolist(), iou.tolist(), cm.tolist()


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
    for name, key in zip(label_names, sorted(preds.keys())):
        p = preds[key]
        mask = (labels_orig == label2idx[name]).to(torch.uint8)[None, None]
        target = F.interpolate(mask, p.shape[-3:], mode='nearest').reshape(-1)
        acc, prec, rec, f1, iou, cm = binary_scores(target.numpy(), p.reshape(-1).numpy())
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
