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


def resize_nearest_u8(vol, out_shape, equals=None, keep_on_device=False):
    """F.interpolate(make_5d(vol), out_shape, mode='nearest') of a uint8 volume on the GPU (vittf_resize_nearest_u8):
    the label up-sample of predict_ntf.py:217-218, or -- with `equals` -- the resized class mask
    F.interpolate((labels == equals).to(uint8)[None, None], out_shape, mode='nearest') of evaluate_similarities.py:63.
    vol: numpy / CPU / GPU uint8-valued (n0, n1, n2).  Returns a uint8 numpy array (or device tensor)."""
    lib = _lib.require_device()
    dev = torch.device('cuda', torch.cuda.current_device())
    t = torch.as_tensor(np.ascontiguousarray(vol) if isinstance(vol, np.ndarray) else vol).squeeze()
    if t.ndim != 3:
        raise ValueError(f'expected a 3-D volume, got {tuple(t.shape)}')
    t = t.to(dev, torch.uint8).contiguous()
    out_shape = tuple(int(x) for x in out_shape)
    out = torch.empty(out_shape, dtype=torch.uint8, device=dev)
    _lib.check(lib.vittf_resize_nearest_u8(_lib.ptr(t), t.shape[0], t.shape[1], t.shape[2], _lib.ptr(out), out_shape[0],
                                           out_shape[1], out_shape[2], -1 if equals is None else int(equals),
                                           _lib.stream_ptr()), 'vittf_resize_nearest_u8')
    return out if keep_on_device else out.cpu().numpy()
