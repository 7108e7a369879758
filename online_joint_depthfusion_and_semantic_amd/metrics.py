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
    est = est.flatten() * mask.flatten()
    target = target.flatten() * mask.flatten()
    est_ids = np.bincount(np.unique(est), minlength=n_class)
    gt_ids = np.bincount(np.unique(target), minlength=n_class)
    ok = (target >= 0) & (target < n_class)
    hist = np.bincount(n_class * target[ok].astype(np.uint16) + est[ok], minlength=n_class * n_class)
    hist = hist.reshape(n_class, n_class)  # rows: target, cols: estimate
    return semantic_metrics_from_counts(hist, est_ids, gt_ids)


def semantic_metrics_from_counts(hist, est_ids, gt_ids):
    """The arithmetic of metrics.py:89-108 on a confusion matrix (rows = target) and the label-presence vectors;
    shared by the host path above and the device path (ops.volume_confusion)."""
    eps = np.finfo(np.float32).eps
    est_ids, gt_ids = np.asarray(est_ids).astype(np.int64), np.asarray(gt_ids).astype(np.int64)
    tp = np.diag(hist)
    fp = hist.sum(axis=0) - tp
    fn = hist.sum(axis=1) - tp
    valid_ids = np.sum(gt_ids) - 1
    acc = tp / (tp + fn + eps)
    iou = tp / (tp + fn + fp + eps)
    present = np.where(est_ids | gt_ids)[0]
    metrics = {'Mean Acc': np.sum(acc[1:]) / valid_ids, 'Mean IoU': np.sum(iou[1:]) / valid_ids}
    return metrics, dict(zip(present, iou[present]))


# ---- reconstruction F-score ------------------------------------------------------------------------
# The reference quotes F-scores (README.md:6) but contains no F-score code; this is
# the engine's own definition, applied identically to oracle volumes and HIP volumes for parity:
# surface samples = zero crossings of the TSDF along the three grid axes (linear interpolation between
# two adjacent observed voxels of opposite sign), F = harmonic mean of precision / recall at distance tau.

def surface_points(tsdf, mask, origin, resolution):
    """Zero-crossing points [M,3] (world units) of a TSDF volume; ``mask`` marks observed voxels."""
    t = np.nan_to_num(np.asarray(tsdf, dtype=np.float32))
    m = np.asarray(mask, dtype=bool)
    pts = []
    for axis in range(3):
        a = [slice(None)] * 3
        b = [slice(None)] * 3
        a[axis], b[axis] = slice(0, -1), slice(1, None)
        ta, tb = t[tuple(a)], t[tuple(b)]
        cross = m[tuple(a)] & m[tuple(b)] & ((ta < 0) != (tb < 0))
        idx = np.argwhere(cross).astype(np.float64)
        if idx.size == 0:
            continue
        va, vb = ta[cross].astype(np.float64), tb[cross].astype(np.float64)
        frac = va / (va - vb)  # position of the zero between the two voxel centres
        idx[:, axis] += frac
        pts.append((idx + 0.5) * float(resolution) + np.asarray(origin, dtype=np.float64))
    return np.concatenate(pts, axis=0) if pts else np.zeros((0, 3))


def f_score(points_est, points_gt, tau):
    """Precision, recall and F-score of two point sets at distance threshold ``tau``."""
    from scipy.spatial import cKDTree
    if len(points_est) == 0 or len(points_gt) == 0:
        return {'precision': 0.0, 'recall': 0.0, 'fscore': 0.0}
    d_e = cKDTree(points_gt).query(points_est)[0]
    d_g = cKDTree(points_est).query(points_gt)[0]
    p, r = float((d_e <= tau).mean()), float((d_g <= tau).mean())
    return {'precision': p, 'recall': r, 'fscore': 0.0 if p + r == 0 else 2 * p * r / (p + r)}


def reconstruction_f_score(est, gt, weights, origin, resolution, tau=None):
    """F-score of the fused TSDF against the ground-truth TSDF restricted to the observed region
    (weights > 0), threshold tau (default: 1.5 voxels).  Volumes that live on the device are scored there
    (mesh.reconstruction_f_score: same definition, ojf_points_within instead of a k-d tree)."""
    import torch
    if torch.is_tensor(est) and est.is_cuda:
        from . import mesh
        dev = est.device
        return mesh.reconstruction_f_score(est, torch.as_tensor(gt).to(dev), torch.as_tensor(weights).to(dev), origin,
                                           resolution, tau)
    mask = np.asarray(weights) > 0
    tau = 1.5 * float(resolution) if tau is None else tau
    return f_score(surface_points(est, mask, origin, resolution), surface_points(gt, mask, origin, resolution), tau)
