"""-m gpu: full-grid Database passes on device against their host definitions (bit-exact for u8/fp16 work)."""
import numpy as np
import pytest
import torch
from scipy.ndimage import median_filter

from online_joint_depthfusion_and_semantic_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(32, 32, 32), (17, 9, 30), (64, 40, 8), (5, 5, 5), (3, 2, 1)])
def test_median5_matches_scipy(cuda, shape):
    rng = np.random.default_rng(sum(shape))
    # blobs of labels plus salt noise: the regime filter_semantics is used in
    vol = (rng.integers(0, 40, size=[(s + 3) // 4 for s in shape]).repeat(4, 0).repeat(4, 1).repeat(4, 2))[:shape[0], :shape[1], :shape[2]]
    vol = np.where(rng.random(shape) < 0.2, rng.integers(0, 256, size=shape), vol).astype(np.uint8)
    vol = np.ascontiguousarray(vol)
    want = median_filter(vol, size=5)  # modules/database.py:116
    got = ops.volume_median5(torch.from_numpy(vol).to(cuda)).cpu().numpy()
    assert np.array_equal(got, want)


def test_fill_filter_known_answers(cuda):
    t = torch.empty((9, 7, 13), dtype=torch.float16, device=cuda)
    ops.volume_fill(t, 0.1)
    assert torch.all(t == torch.tensor(0.1, dtype=torch.float16))
    w = torch.tensor(np.linspace(0, 4, 9 * 7 * 13).reshape(9, 7, 13).astype(np.float16)).to(cuda)
    w0 = w.clone()
    t.fill_(-0.02)
    ops.volume_filter(t, w, 2.0, 0.1)
    low = w0 < 2.0
    assert torch.all(t[low] == torch.tensor(0.1, dtype=torch.float16)) and torch.all(w[low] == 0)
    assert torch.all(t[~low] == torch.tensor(-0.02, dtype=torch.float16)) and torch.equal(w[~low], w0[~low])
    u = torch.ones((5, 5, 5), dtype=torch.uint8, device=cuda)
    ops.volume_fill(u, 0)
    assert int(u.sum()) == 0


@pytest.mark.parametrize('shape,n_classes,offset', [((32, 32, 32), 30, 0), ((17, 9, 31), 40, 0), ((64, 40, 9), 30, 3),
                                                    ((5, 5, 5), 12, 1), ((33, 20, 16), 100, 0), ((16, 16, 16), 256, 0)])
def test_confusion_counts_and_semantic_metrics(cuda, shape, n_classes, offset):
    """ojf_volume_confusion == the counts inside utils/metrics.py:69-108 (np.bincount of target * C + est on masked
    volumes, np.unique presence); the metrics computed from them are then the same floats.  ``offset`` misaligns the
    device pointers (scalar path), C > 64 takes the global-atomics path."""
    from online_joint_depthfusion_and_semantic_amd import metrics
    rng = np.random.default_rng(sum(shape) + n_classes)
    n = int(np.prod(shape))
    hi = min(n_classes, 255)
    est = rng.integers(0, hi, size=n).astype(np.uint8)
    gt = np.where(rng.random(n) < 0.7, est, rng.integers(0, hi, size=n)).astype(np.uint8)
    wgt = np.where(rng.random(n) < 0.4, rng.random(n) * 5, 0).astype(np.float16)

    def dev(a):
        buf = torch.zeros(n + 8, dtype=torch.from_numpy(a).dtype, device=cuda)
        view = buf[offset:offset + n]
        view.copy_(torch.from_numpy(a))
        return view.view(shape)
    hist, e_ids, g_ids = ops.volume_confusion(dev(est), dev(gt), dev(wgt), n_classes)
    m = wgt > 0
    e, g = est * m, gt * m
    want = np.bincount(n_classes * g.astype(np.uint16).astype(np.int64) + e, minlength=n_classes * n_classes).reshape(n_classes, n_classes)
    assert np.array_equal(hist, want) and hist.sum() == n
    assert np.array_equal(np.flatnonzero(e_ids), np.unique(e)) and np.array_equal(np.flatnonzero(g_ids), np.unique(g))
    want_m, want_iou = metrics.semantic_evaluation(est.reshape(shape), gt.reshape(shape), m.reshape(shape), n_classes)
    got_m, got_iou = metrics.semantic_metrics_from_counts(hist, e_ids[:n_classes], g_ids[:n_classes])
    assert got_m == want_m and got_iou.keys() == want_iou.keys() and all(got_iou[k] == want_iou[k] for k in want_iou)


def test_database_evaluate_semantics_on_device(cuda):
    """Database.evaluate_semantics (modules/database.py:311-349) with device-resident volumes == the host path."""
    from online_joint_depthfusion_and_semantic_amd.config import default_config, database_config
    from online_joint_depthfusion_and_semantic_amd.database import Database
    from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream
    cfg = default_config(24, 32, semantics=True)
    cfg.SETTINGS.device = str(cuda)
    st = SyntheticStream(24, 32, 32, 5)
    db = Database(st, database_config(cfg))
    s = st.scene
    rng = np.random.default_rng(2)
    gt = db.ids_gt[s].volume.cpu().numpy()
    est = np.where(rng.random(gt.shape) < 0.8, gt, rng.integers(0, 30, size=gt.shape)).astype(np.uint8)
    db.ids_est[s].volume = torch.from_numpy(est).to(cuda)
    db.fusion_weights[s] = torch.from_numpy((rng.random(gt.shape) < 0.3).astype(np.float16) * 2).to(cuda)
    db.state[s] = True
    quiet = type('W', (), {'log': staticmethod(lambda *a: None)})
    on_dev, iou_dev = db.evaluate_semantics(mode='test', workspace=quiet)
    db.to_numpy()
    on_host, iou_host = db.evaluate_semantics(mode='test', workspace=quiet)
    assert on_dev == on_host and on_dev['Mean IoU'] > 0.3
    assert iou_dev[s].keys() == iou_host[s].keys() and all(iou_dev[s][k] == iou_host[s][k] for k in iou_host[s])
