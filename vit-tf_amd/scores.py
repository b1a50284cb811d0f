"""Scores of a predicted label volume against the ground truth (predict_ntf.py:228-246, evaluate_similarities.py:63-68):
the confusion matrix is counted on the GPU (vittf_confusion_matrix), and precision / recall / F1 / IoU / accuracy follow
from it exactly as sklearn derives them (labels = the sorted values present in either volume; 0/0 -> 0)."""
import numpy as np
import torch

from . import _lib


def confusion_matrix(target, pred, classes=None):
    """int64 (classes, classes) numpy array, rows = target, columns = prediction; uint8-valued inputs of equal size
    (numpy / CPU / GPU tensors).  classes defaults to max value + 1."""
    lib = _lib.require_device()
    dev = torch.device('cuda', torch.cuda.current_device())

    def dev_u8(a):
        t = torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a)
        return t.to(dev, torch.uint8).contiguous().reshape(-1)
    t, p = dev_u8(target), dev_u8(pred)
    if t.numel() != p.numel():
        raise ValueError(f'target has {t.numel()} voxels, prediction {p.numel()}')
    if classes is None:
        classes = (int(max(t.max().item(), p.max().item())) + 1) if t.numel() else 1
    counts = torch.empty(classes * classes + 1, dtype=torch.int64, device=dev)
    _lib.check(lib.vittf_confusion_matrix(_lib.ptr(t), _lib.ptr(p), t.numel(), int(classes), _lib.ptr(counts),
                                          _lib.stream_ptr()), 'vittf_confusion_matrix')
    counts = counts.cpu().numpy()
    if counts[-1]:
        raise ValueError(f'{int(counts[-1])} voxels hold a label >= {classes}')
    return counts[:-1].reshape(classes, classes)


def scores_from_confusion(cm):
    """(accuracy, precision, recall, f1, iou, confusion matrix) with sklearn's conventions: classes absent from both
    volumes are dropped, undefined ratios are 0."""
    cm = np.asarray(cm, dtype=np.int64)
    present = (cm.sum(0) + cm.sum(1)) > 0
    cm = cm[present][:, present]
    tp = np.diag(cm).astype(np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        prec = np.nan_to_num(tp / cm.sum(0))
        rec = np.nan_to_num(tp / cm.sum(1))
        f1 = np.nan_to_num(2 * prec * rec / (prec + rec))
        iou = np.nan_to_num(tp / (cm.sum(0) + cm.sum(1) - tp))
    return float(tp.sum() / cm.sum()), prec, rec, f1, iou, cm


def scores(target, pred, classes=None):
    return scores_from_confusion(confusion_matrix(target, pred, classes))
