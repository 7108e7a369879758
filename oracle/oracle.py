"""ctypes front-end of the CPU oracle (oracle/ojf_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from the product package.  See the header of ojf_oracle.c for how
the oracle is pinned against the reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(['make', '-C', _HERE, '-s'], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libojf_oracle.so')
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.ojf_oracle_h2f.restype = ctypes.c_float
        _LIB.ojf_oracle_h2f.argtypes = [ctypes.c_uint16]
        _LIB.ojf_oracle_f2h.restype = ctypes.c_uint16
        _LIB.ojf_oracle_f2h.argtypes = [ctypes.c_float]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def camera_arrays(intrinsics, extrinsics):
    """Host-side camera preparation shared by oracle and product: K^-1 as the reference
    computes it (modules/extractor.py:39,104: ``intrinsics.float().inverse()``) and the first
    three rows of the fp32 extrinsics (extractor.py:40,115)."""
    import torch
    K = torch.as_tensor(np.asarray(intrinsics)).reshape(3, 3).float()
    Ki = K.inverse().float().numpy().astype(np.float32).reshape(9).copy()
    E = np.asarray(extrinsics, dtype=np.float64).reshape(-1, 4)[:3]
    E = np.ascontiguousarray(E.astype(np.float32)).reshape(12)
    return Ki, E


def extract(depth, Ki, E, origin, res, tsdf, wgt, n_points=9, pad_value=-0.1, debug=False):
    """depth [h,w] f32; tsdf/wgt [X,Y,Z] float16.  Returns dict like Extractor.forward."""
    h, w = depth.shape
    X, Y, Z = tsdf.shape
    N = h * w
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    tsdf = np.ascontiguousarray(tsdf).view(np.uint16)
    wgt = np.ascontiguousarray(wgt).view(np.uint16)
    origin = np.ascontiguousarray(origin, dtype=np.float64)
    vals = np.empty((N, n_points), np.float32)
    wts = np.empty((N, n_points), np.float32)
    idx = np.empty((N, n_points, 8, 3), np.int64) if debug else None
    cw = np.empty((N, n_points, 8), np.float64) if debug else None
    pts = np.empty((N, n_points, 3), np.float64) if debug else None
    pcl = np.empty((N, 3), np.float32) if debug else None
    rc = lib().ojf_oracle_extract(
        _p(depth), _p(Ki), _p(E), _p(origin), ctypes.c_double(float(res)), _p(tsdf), _p(wgt),
        X, Y, Z, h, w, n_points, ctypes.c_float(pad_value), _p(vals), _p(wts), _p(idx), _p(cw),
        _p(pts), _p(pcl))
    assert rc == 0
    out = dict(fusion_values=vals, fusion_weights=wts)
    if debug:
        out.update(indices=idx, weights=cw, points=pts, pcl=pcl)
    return out


def integrate(depth_filtered, Ki, E, origin, res, est, tsdf, wgt, n_points=9, n_tail=7, trunc=0.1,
              sem_ids=None, sem_scores=None, id_vol=None, score_vol=None):
    """In place on tsdf/wgt (float16 [X,Y,Z]) and, if given, id_vol (u8) / score_vol (float16).
    est [h*w, >=n_tail] f32.  Returns the number of distinct voxels written."""
    h, w = depth_filtered.shape
    X, Y, Z = tsdf.shape
    assert tsdf.flags.c_contiguous and wgt.flags.c_contiguous
    assert tsdf.dtype == np.float16 and wgt.dtype == np.float16
    depth_filtered = np.ascontiguousarray(depth_filtered, dtype=np.float32)
    est = np.ascontiguousarray(est, dtype=np.float32).reshape(h * w, -1)
    origin = np.ascontiguousarray(origin, dtype=np.float64)
    if sem_ids is not None:
        sem_ids = np.ascontiguousarray(sem_ids, dtype=np.uint8)
        sem_scores = np.ascontiguousarray(sem_scores, dtype=np.float32)
        assert id_vol.dtype == np.uint8 and score_vol.dtype == np.float16
        assert id_vol.flags.c_contiguous and score_vol.flags.c_contiguous
    touched = ctypes.c_int64(0)
    rc = lib().ojf_oracle_integrate(
        _p(depth_filtered), _p(Ki), _p(E), _p(origin), ctypes.c_double(float(res)), _p(est),
        est.shape[1], n_points, n_tail, ctypes.c_float(trunc), _p(tsdf.view(np.uint16)),
        _p(wgt.view(np.uint16)), _p(sem_ids), _p(sem_scores), _p(id_vol),
        _p(None if score_vol is None else score_vol.view(np.uint16)), X, Y, Z, h, w,
        ctypes.byref(touched))
    assert rc == 0
    return touched.value


def unique_voxels(depth, depth_filtered, Ki, E, origin, res, shape, n_points=9, n_tail=7):
    h, w = depth.shape
    X, Y, Z = shape
    ug, us = ctypes.c_int64(0), ctypes.c_int64(0)
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    depth_filtered = np.ascontiguousarray(depth_filtered, dtype=np.float32)
    origin = np.ascontiguousarray(origin, dtype=np.float64)
    rc = lib().ojf_oracle_unique_voxels(
        _p(depth), _p(depth_filtered), _p(Ki), _p(E), _p(origin), ctypes.c_double(float(res)),
        X, Y, Z, h, w, n_points, n_tail, ctypes.byref(ug), ctypes.byref(us))
    assert rc == 0
    return ug.value, us.value
