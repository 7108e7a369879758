"""Volume metrics of the reference (utils/metrics.py:69-127) for host-side (numpy) volumes.
The device-side counterpart of ``evaluation`` is ojf_volume_evaluate (ops.volume_evaluate)."""
import numpy as np


def evaluation(est, target, mask=None):
    """mse / mad / occupancy iou / sign accuracy on volumes clipped to +-0.04 (metrics.py:111-127)."""
    eps = 1.e-10
    est = np.clip(np.nan_to_num(est.astype(np.float32)), -0.04, 0.04)
    target = np.clip(np.nan_to_num(target.astype(np.float32)), -0.04, 0.04)
    if mask is None:
        mask = np.ones(est.shape, dtype=bool)
    m = mask > 0
    n = float(np.nansum(mask))
    diff = est - target
    mse = float(np.nansum(mask * np.power(diff, 2))) / (n + eps)
    mad = float(np.nansum((mask * np.abs(diff)).astype(np.float32))) / (n + eps)
    occ_e, occ_t = est < 0, target < 0
    tp = float(np.sum(occ_e & occ_t & m))
    fp = float(np.sum(occ_e & ~occ_t & m))
    fn = float(np.sum(~occ_e & occ_t & m))
    tn = float(np.sum(~occ_e & ~occ_t & m))
    return {'mse': mse, 'mad': mad, 'iou': tp / (tp + fp + fn + eps), 'acc': (tp + tn) / (n + eps)}


def semantic_evaluation(est, target, mask, n_class):
    """Mean accuracy / mean IoU over the classes present in the scene, class 0 excluded
    (metrics.py:69-108)."""
    eps = np.finfo(np.float32).eps
    est = est.flatten() * mask.flatten()
    target = target.flatten() * mask.flatten()
    est_ids = np.bincount(np.unique(est), minlength=n_class)
    gt_ids = np.bincount(np.unique(target), minlength=n_class)
    ok = (target >= 0) & (target < n_class)
    hist = np.bincount(n_class * target[ok].astype(np.uint16) + est[ok], minlength=n_class * n_class)
    hist = hist.reshape(n_class, n_class)  # rows: target, cols: estimate
    tp = np.diag(hist)
    fp = hist.sum(axis=0) - tp
    fn = hist.sum(axis=1) - tp
    valid_ids = np.sum(gt_ids) - 1
    acc = tp / (tp + fn + eps)
    iou = tp / (tp + fn + fp + eps)
    present = np.where(est_ids | gt_ids)[0]
    metrics = {'Mean Acc': np.sum(acc[1:]) / valid_ids, 'Mean IoU': np.sum(iou[1:]) / valid_ids}
    return metrics, dict(zip(present, iou[present]))
