"""Thin torch-tensor front-end of the libojf C ABI (device pointers + current HIP stream).

PyTorch is plumbing here: it owns the HBM allocations and the stream; all arithmetic of the hot
path happens in the HIP kernels behind include/ojf.h.  Every function requires CUDA(HIP) tensors
and raises if libojf.so or the GPU is missing.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import MODE_FAST, MODE_PARITY  # noqa: F401  (re-exported)


def camera_arrays(intrinsics, extrinsics):
    """Host-side camera preparation: (Kinv f32[9], E f32[12]) as numpy arrays.

    Kinv follows the reference exactly: ``intrinsics.float().inverse()``
    (modules/extractor.py:39,104); E is the first three rows of ``extrinsics.float()``
    (extractor.py:40,115-117; ScanNet hands over 4x4 poses, Replica 3x4)."""
    K = torch.as_tensor(intrinsics).detach().cpu().reshape(3, 3).float()
    Ki = K.inverse().float().numpy().reshape(9).copy()
    E = torch.as_tensor(extrinsics).detach().cpu().float()
    E = E.reshape(-1, 4)[:3].contiguous().numpy().reshape(12).copy()
    return Ki, E


def _origin_array(origin):
    if torch.is_tensor(origin):
        origin = origin.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(origin, dtype=np.float64).reshape(3))


def _vol16(t):
    assert t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and t.dim() == 3
    return t


def extract(depth, Ki, E, origin, resolution, tsdf, weights, n_points=9, pad_value=-0.1,
            out_values=None, out_weights=None, out_stride=None, debug=False, planes=False):
    """Gather along the depth rays.  depth: cuda f32 [h,w] (or [1,h,w]).  Returns a dict with
    fusion_values / fusion_weights [h*w, n_points] (plus indices / weights / points / pcl when
    ``debug``).  ``out_values``/``out_weights``/``out_stride`` let the caller place the result
    inside a wider row buffer; ``planes=True`` selects the coalesced sample-plane layout
    [n_points, h*w] (element (n,k) at [k*stride + n]) that Pipeline.fuse feeds to the net."""
    _lib.require_gpu()
    lib = _lib.load()
    depth = depth.reshape(depth.shape[-2], depth.shape[-1])
    assert depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous()
    h, w = depth.shape
    N = h * w
    X, Y, Z = _vol16(tsdf).shape
    assert _vol16(weights).shape == tsdf.shape
    dev = depth.device
    if out_values is None:
        out_stride = N if planes else n_points
        shape = (n_points, N) if planes else (N, n_points)
        out_values = torch.empty(shape, dtype=torch.float32, device=dev)
        out_weights = torch.empty(shape, dtype=torch.float32, device=dev)
    origin = _origin_array(origin)
    dbg = {}
    if debug:
        dbg['indices'] = torch.empty((N, n_points, 8, 3), dtype=torch.int64, device=dev)
        dbg['weights'] = torch.empty((N, n_points, 8), dtype=torch.float64, device=dev)
        dbg['points'] = torch.empty((N, n_points, 3), dtype=torch.float64, device=dev)
        dbg['pcl'] = torch.empty((N, 3), dtype=torch.float32, device=dev)
    ov = out_values if isinstance(out_values, int) else _lib.ptr(out_values)  # raw device address allowed
    ow = out_weights if isinstance(out_weights, int) else _lib.ptr(out_weights)
    rc = lib.ojf_extract(_lib.ptr(depth), _lib.ptr(Ki), _lib.ptr(E), _lib.ptr(origin),
                         float(resolution), _lib.ptr(tsdf), _lib.ptr(weights), X, Y, Z, h, w,
                         n_points, float(pad_value), ov, ow,
                         int(out_stride), 1 if planes else 0, _lib.ptr(dbg.get('indices')), _lib.ptr(dbg.get('weights')),
                         _lib.ptr(dbg.get('points')), _lib.ptr(dbg.get('pcl')), _lib.stream_ptr(dev))
    _lib.check(rc, 'ojf_extract')
    out = dict(fusion_values=out_values, fusion_weights=out_weights)
    out.update(dbg)
    return out


class IntegrateWorkspace:
    """Device scratch of ojf_integrate for one (grid, frame size, mode); reusable across scenes."""

    def __init__(self, shape, h, w, n_tail, mode, device):
        _lib.require_gpu()
        lib = _lib.load()
        X, Y, Z = shape
        self.key = (tuple(shape), h, w, n_tail, mode)
        self.bytes = int(lib.ojf_integrate_workspace_bytes(X, Y, Z, h, w, n_tail, mode))
        if self.bytes == 0:
            raise _lib.OjfError('ojf_integrate_workspace_bytes: unsupported sizes/mode: ' +
                                lib.ojf_last_error().decode())
        self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        self.stats = torch.zeros(4, dtype=torch.int32, device=device)
        rc = lib.ojf_integrate_workspace_init(_lib.ptr(self.buf), self.bytes, X, Y, Z, h, w, n_tail,
                                              mode, _lib.stream_ptr(device))
        _lib.check(rc, 'ojf_integrate_workspace_init')


def extract_to_net(depth, Ki, E, origin, resolution, tsdf, weights, engine, *, pad_value=-0.1):
    """``extract`` + ``engine.prepare_input`` in one launch (ojf_extract_to_net): the gathered values / weights and the raw
    depth land in the fusion net's input planes; nothing else is written.  Only for engines with ``fused_input``."""
    lib = _lib.load()
    assert depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous()
    h, w = depth.shape
    X, Y, Z = _vol16(tsdf).shape
    assert _vol16(weights).shape == tsdf.shape
    rc = lib.ojf_extract_to_net(_lib.ptr(depth), _lib.ptr(Ki), _lib.ptr(E), _lib.ptr(_origin_array(origin)), float(resolution),
                                _lib.ptr(tsdf), _lib.ptr(weights), X, Y, Z, h, w, int(engine.n_points), float(pad_value),
                                engine.handle, _lib.stream_ptr(depth.device))
    _lib.check(rc, 'ojf_extract_to_net')


def integrate(depth_filtered, Ki, E, origin, resolution, est, tsdf, weights, workspace,
              n_points=9, n_tail=7, trunc=0.1, est_stride=None, sem_ids=None, sem_scores=None,
              id_vol=None, score_vol=None, mode=MODE_FAST, stats=False, mask=None):
    """Scatter the clamped net output into the volumes, in place.  est: cuda f32 rows with
    ``est_stride`` floats per pixel (default: est.shape[-1]).  stats=True fills ``workspace.stats``
    ({touched voxels, scatter entries, records, 0}; costs ~14 us per frame of same-line atomics).
    mask: optional cuda bool [h, w]; then ``depth_filtered`` is the RAW frame and the kernels apply
    ``torch.where(mask == 0, 0, frame)`` (pipeline.py:196) themselves - same result, one launch less."""
    _lib.require_gpu()
    lib = _lib.load()
    depth_filtered = depth_filtered.reshape(depth_filtered.shape[-2], depth_filtered.shape[-1])
    assert depth_filtered.is_cuda and depth_filtered.dtype == torch.float32 and depth_filtered.is_contiguous()
    h, w = depth_filtered.shape
    X, Y, Z = _vol16(tsdf).shape
    assert _vol16(weights).shape == tsdf.shape
    assert est.is_cuda and est.dtype == torch.float32 and est.is_contiguous()
    if est_stride is None:
        est_stride = est.shape[-1]
    if sem_ids is not None:
        assert sem_ids.dtype == torch.uint8 and sem_ids.is_contiguous() and sem_ids.numel() == h * w
        assert sem_scores.dtype == torch.float32 and sem_scores.is_contiguous() and sem_scores.numel() == h * w
        assert id_vol.dtype == torch.uint8 and id_vol.is_contiguous() and id_vol.shape == tsdf.shape
        assert score_vol.dtype == torch.float16 and score_vol.is_contiguous() and score_vol.shape == tsdf.shape
    assert workspace.key == ((X, Y, Z), h, w, n_tail, mode), 'workspace built for another configuration'
    origin = _origin_array(origin)
    if mask is not None:
        mask = mask.reshape(h, w)
        assert mask.is_cuda and mask.dtype == torch.bool and mask.is_contiguous()
    rc = lib.ojf_integrate_masked(_lib.ptr(depth_filtered), _lib.ptr(mask), _lib.ptr(Ki), _lib.ptr(E), _lib.ptr(origin),
                           float(resolution), _lib.ptr(est), int(est_stride), n_points, n_tail,
                           float(trunc), _lib.ptr(tsdf), _lib.ptr(weights), _lib.ptr(sem_ids),
                           _lib.ptr(sem_scores), _lib.ptr(id_vol), _lib.ptr(score_vol), X, Y, Z, h, w,
                           mode, _lib.ptr(workspace.buf), workspace.bytes,
                           _lib.ptr(workspace.stats) if stats else None, _lib.stream_ptr(depth_filtered.device))
    _lib.check(rc, 'ojf_integrate')


def extract_many(jobs, n_points=9, pad_value=-0.1):
    """One frame of each of several SCENES as a single launch (ojf_extract_many).  ``jobs``: dicts with depth (cuda f32 [h, w]),
    Ki, E (camera_arrays), origin, resolution, tsdf, weights and either ``engine`` (a FusionNetEngine with ``fused_input``: the
    results land in its input planes) or ``out_values`` / ``out_weights`` sample planes [n_points, h*w].  Bit for bit what the
    separate ``extract`` / ``extract_to_net`` calls write."""
    lib = _lib.load()
    n = len(jobs)
    assert 1 <= n <= _lib.MAX_SCENES
    h, w = jobs[0]['depth'].shape
    X, Y, Z = _vol16(jobs[0]['tsdf']).shape
    arr = (_lib.ExtractJob * n)()
    keep = []
    for a, j in zip(arr, jobs):
        depth = j['depth']
        assert depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous() and tuple(depth.shape) == (h, w)
        assert _vol16(j['tsdf']).shape == (X, Y, Z) and _vol16(j['weights']).shape == (X, Y, Z)
        origin = _origin_array(j['origin'])
        keep += [origin, j['Ki'], j['E']]
        a.depth_dev, a.Kinv_host, a.E_host, a.origin_host = _lib.ptr(depth), _lib.ptr(j['Ki']), _lib.ptr(j['E']), _lib.ptr(origin)
        a.resolution = float(j['resolution'])
        a.tsdf_dev, a.weights_dev = _lib.ptr(j['tsdf']), _lib.ptr(j['weights'])
        eng = j.get('engine')
        if eng is not None:
            assert eng.fused_input and int(eng.n_points) == n_points
            a.net = eng.handle
        else:
            ov, ow = j['out_values'], j['out_weights']
            assert ov.is_cuda and ov.dtype == torch.float32 and ov.is_contiguous() and tuple(ov.shape) == (n_points, h * w) == tuple(ow.shape)
            a.out_values_dev, a.out_weights_dev, a.out_stride, a.out_layout = _lib.ptr(ov), _lib.ptr(ow), h * w, 1
    rc = lib.ojf_extract_many(n, arr, X, Y, Z, h, w, int(n_points), float(pad_value), _lib.stream_ptr(jobs[0]['depth'].device))
    _lib.check(rc, 'ojf_extract_many')


def integrate_many(jobs, n_points=9, n_tail=7, trunc=0.1):
    """One frame of each of several SCENES through ojf_integrate_masked's FAST kernels as two launches (ojf_integrate_many).
    ``jobs``: dicts with depth, mask (or None), Ki, E, origin, resolution, est (rows [h*w, stride]), tsdf, weights, workspace (an
    IntegrateWorkspace of its own per job) and optionally sem_ids, sem_scores, id_vol, score_vol (all jobs or none).  The
    volumes come out bit for bit as from the separate ``integrate`` calls."""
    lib = _lib.load()
    n = len(jobs)
    assert 1 <= n <= _lib.MAX_SCENES
    h, w = jobs[0]['depth'].shape
    X, Y, Z = _vol16(jobs[0]['tsdf']).shape
    arr = (_lib.IntegrateJob * n)()
    keep = []
    for a, j in zip(arr, jobs):
        depth, est, ws = j['depth'], j['est'], j['workspace']
        assert depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous() and tuple(depth.shape) == (h, w)
        assert est.is_cuda and est.dtype == torch.float32 and est.is_contiguous()
        assert _vol16(j['tsdf']).shape == (X, Y, Z) and _vol16(j['weights']).shape == (X, Y, Z)
        assert ws.key == ((X, Y, Z), h, w, n_tail, MODE_FAST), 'workspace built for another configuration'
        mask = j.get('mask')
        if mask is not None:
            assert mask.is_cuda and mask.dtype == torch.bool and mask.is_contiguous() and mask.numel() == h * w
        origin = _origin_array(j['origin'])
        keep += [origin, j['Ki'], j['E']]
        a.depth_dev, a.mask_dev = _lib.ptr(depth), _lib.ptr(mask)
        a.Kinv_host, a.E_host, a.origin_host, a.resolution = _lib.ptr(j['Ki']), _lib.ptr(j['E']), _lib.ptr(origin), float(j['resolution'])
        a.est_dev, a.est_stride = _lib.ptr(est), int(est.shape[-1])
        a.tsdf_dev, a.weights_dev = _lib.ptr(j['tsdf']), _lib.ptr(j['weights'])
        if j.get('sem_ids') is not None:
            ids, sc, iv, sv = j['sem_ids'], j['sem_scores'], j['id_vol'], j['score_vol']
            assert ids.dtype == torch.uint8 and ids.is_contiguous() and ids.numel() == h * w
            assert sc.dtype == torch.float32 and sc.is_contiguous() and sc.numel() == h * w
            assert iv.dtype == torch.uint8 and iv.is_contiguous() and iv.shape == j['tsdf'].shape
            assert sv.dtype == torch.float16 and sv.is_contiguous() and sv.shape == j['tsdf'].shape
            a.sem_ids_dev, a.sem_scores_dev, a.id_vol_dev, a.score_vol_dev = _lib.ptr(ids), _lib.ptr(sc), _lib.ptr(iv), _lib.ptr(sv)
        a.workspace_dev, a.workspace_bytes = _lib.ptr(ws.buf), ws.bytes
    rc = lib.ojf_integrate_many(n, arr, int(n_points), int(n_tail), float(trunc), X, Y, Z, h, w, _lib.stream_ptr(jobs[0]['depth'].device))
    _lib.check(rc, 'ojf_integrate_many')


class EntryWorkspace:
    """FAST-mode scratch for ``integrate_entries``: header + 4 B/voxel head table + one 32-B record per entry
    + touched list (layout of csrc/ojf_integrate.hip)."""

    def __init__(self, shape, max_rows, device):
        _lib.require_gpu()
        X, Y, Z = shape
        nvox = X * Y * Z
        self.shape, self.max_rows = tuple(shape), int(max_rows)
        self.bytes = 512 + nvox * 4 + self.max_rows * 8 * 32 + min(self.max_rows * 8, nvox) * 4  # (header: csrc/ojf_integrate.h kHeaderBytes)
        self.buf = torch.zeros(self.bytes, dtype=torch.uint8, device=device)  # the head table starts clean
        self.stats = torch.zeros(4, dtype=torch.int32, device=device)


def integrate_entries(values, indices, weights, tsdf, weights_volume, workspace, row_ids=None, row_scores=None,
                      id_vol=None, score_vol=None):
    """Scatter materialised updates (the reference Integrator's inputs) into the volumes, in place.
    values [R] f32, indices [R,8,3] i64, weights [R,8] f64 (cuda, contiguous)."""
    _lib.require_gpu()
    lib = _lib.load()
    R = values.numel()
    assert values.is_cuda and values.dtype == torch.float32 and values.is_contiguous()
    assert indices.dtype == torch.int64 and indices.is_contiguous() and indices.numel() == R * 24
    assert weights.dtype == torch.float64 and weights.is_contiguous() and weights.numel() == R * 8
    X, Y, Z = _vol16(tsdf).shape
    assert _vol16(weights_volume).shape == tsdf.shape and workspace.shape == (X, Y, Z) and R <= workspace.max_rows
    if row_ids is not None:
        assert row_ids.dtype == torch.uint8 and row_ids.is_contiguous() and row_ids.numel() == R
        assert row_scores.dtype == torch.float32 and row_scores.is_contiguous() and row_scores.numel() == R
    rc = lib.ojf_integrate_entries(_lib.ptr(values), _lib.ptr(indices), _lib.ptr(weights), _lib.ptr(row_ids),
                                   _lib.ptr(row_scores), R, _lib.ptr(tsdf), _lib.ptr(weights_volume), _lib.ptr(id_vol),
                                   _lib.ptr(score_vol), X, Y, Z, _lib.ptr(workspace.buf), workspace.bytes,
                                   _lib.ptr(workspace.stats), _lib.stream_ptr(values.device))
    _lib.check(rc, 'ojf_integrate_entries')


def volume_fill(vol, value):
    _lib.require_gpu()
    lib = _lib.load()
    assert vol.is_cuda and vol.is_contiguous()
    if vol.dtype == torch.float16:
        rc = lib.ojf_volume_fill_f16(_lib.ptr(vol), vol.numel(), float(value), _lib.stream_ptr(vol.device))
    elif vol.dtype == torch.uint8:
        rc = lib.ojf_volume_fill_u8(_lib.ptr(vol), vol.numel(), int(value), _lib.stream_ptr(vol.device))
    else:
        raise TypeError('volume_fill: fp16 or u8 volumes only')
    _lib.check(rc, 'ojf_volume_fill')


def volume_filter(tsdf, weights, threshold, init_value):
    _lib.require_gpu()
    lib = _lib.load()
    rc = lib.ojf_volume_filter(_lib.ptr(_vol16(tsdf)), _lib.ptr(_vol16(weights)), tsdf.numel(),
                               float(threshold), float(init_value), _lib.stream_ptr(tsdf.device))
    _lib.check(rc, 'ojf_volume_filter')


def volume_evaluate(est, gt, weights):
    """utils/metrics.py:111-127 on device; returns dict(mse, mad, iou, acc) of python floats."""
    _lib.require_gpu()
    lib = _lib.load()
    sums = torch.zeros(8, dtype=torch.float64, device=est.device)
    rc = lib.ojf_volume_evaluate(_lib.ptr(_vol16(est)), _lib.ptr(_vol16(gt)), _lib.ptr(_vol16(weights)),
                                 est.numel(), _lib.ptr(sums), _lib.stream_ptr(est.device))
    _lib.check(rc, 'ojf_volume_evaluate')
    n, sq, ab, inter, union, same = sums[:6].tolist()
    eps = 1.e-10
    return {'mse': sq / (n + eps), 'mad': ab / (n + eps), 'iou': inter / (union + eps), 'acc': same / (n + eps)}


def volume_confusion(ids_est, ids_gt, weights, n_classes):
    """Confusion counts of Database.evaluate_semantics on device: (hist int64 [C, C] rows = gt, est_present bool[256],
    gt_present bool[256]) as numpy arrays (the only D2H traffic: C*C*8 + 2 KB)."""
    _lib.require_gpu()
    lib = _lib.load()
    for t in (ids_est, ids_gt):
        assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
    assert ids_est.shape == ids_gt.shape == _vol16(weights).shape
    hist = torch.empty(n_classes * n_classes, dtype=torch.int64, device=ids_est.device)
    present = torch.empty(512, dtype=torch.int32, device=ids_est.device)
    rc = lib.ojf_volume_confusion(_lib.ptr(ids_est), _lib.ptr(ids_gt), _lib.ptr(weights), ids_est.numel(), int(n_classes),
                                  _lib.ptr(hist), _lib.ptr(present), _lib.stream_ptr(ids_est.device))
    _lib.check(rc, 'ojf_volume_confusion')
    p = present.cpu().numpy() != 0
    return hist.cpu().numpy().reshape(n_classes, n_classes), p[:256], p[256:]


def volume_median5(ids):
    """scipy.ndimage.median_filter(ids, size=5) on a cuda u8 volume (Database.filter_semantics)."""
    _lib.require_gpu()
    lib = _lib.load()
    assert ids.is_cuda and ids.dtype == torch.uint8 and ids.is_contiguous() and ids.dim() == 3
    out = torch.empty_like(ids)
    X, Y, Z = ids.shape
    rc = lib.ojf_volume_median5_u8(_lib.ptr(ids), _lib.ptr(out), X, Y, Z, _lib.stream_ptr(ids.device))
    _lib.check(rc, 'ojf_volume_median5_u8')
    return out
