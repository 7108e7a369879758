"""CPU: pins the oracle (oracle/ojf_oracle.c + the torch fp32 net) against the golden vectors that
tests/golden/make_golden.py produced by running the REFERENCE's modules (bit-exact bars)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle
from helpers import n_mismatch, golden, net_from_golden, oracle_net_est, oracle_fuse, fresh_volumes, make_stream

HERE = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _replay(h, w, grid, frames, get):
    """Replays the generator's stream through the oracle; ``get(i, key)`` returns golden inputs."""
    vols = fresh_volumes(grid, True)
    st = make_stream(h, w, grid)
    for i in range(frames):
        depth, mask = get(i, 'depth'), get(i, 'mask')
        Ki, E = oracle.camera_arrays(get(i, 'intrinsics'), get(i, 'extrinsics'))
        ex = oracle.extract(depth, Ki, E, st.origin, st.resolution, vols['tsdf'], vols['wgt'], debug=True)
        fd = np.where(mask, depth, np.float32(0)).astype(np.float32)
        oracle.integrate(fd, Ki, E, st.origin, st.resolution, get(i, 'est'), vols['tsdf'], vols['wgt'],
                         sem_ids=get(i, 'sem_ids'), sem_scores=get(i, 'sem_scores'), id_vol=vols['ids'],
                         score_vol=vols['scores'])
        yield i, ex, {k: v.copy() for k, v in vols.items()}


def test_oracle_matches_reference_arrays_tiny():
    g = golden('extract_integrate_12x16_g32.npz')
    for i, ex, vols in _replay(12, 16, 32, 4, lambda i, k: g['f%d_%s' % (i, k)]):
        for key in ('fusion_values', 'fusion_weights', 'weights', 'points', 'pcl'):
            assert n_mismatch(ex[key], g['f%d_%s' % (i, key)]) == 0, (key, i)
        assert n_mismatch(ex['indices'], g['f%d_indices' % i].astype(np.int64)) == 0, i
        for key in ('tsdf', 'wgt', 'ids', 'scores'):
            assert n_mismatch(vols[key], g['f%d_%s' % (i, key)]) == 0, (key, i)


@pytest.mark.parametrize('name,h,w,grid,frames', [('A_120x160_g64', 120, 160, 64, 3), ('B_240x320_g256', 240, 320, 256, 2)])
def test_oracle_matches_reference_digests(name, h, w, grid, frames):
    with open(os.path.join(HERE, 'golden', 'extract_integrate_digests.json')) as f:
        dig = json.load(f)[name]
    st = make_stream(h, w, grid)

    def get(i, key):  # the generator's inputs are re-derived from the shared seeded stream
        b = st.frame(i)
        depth = b['tof_depth'].copy()
        if i == 1:
            depth[::5, ::3] = 0.0
        if key == 'depth':
            return depth
        if key == 'mask':
            return (depth > 0.05) & (depth < 5.0)
        if key == 'est':
            return np.random.default_rng([7, i]).uniform(-0.15, 0.15, (h * w, 9)).astype(np.float32)
        return {'intrinsics': b['intrinsics'], 'extrinsics': b['extrinsics'], 'sem_ids': b['semantic_gt'],
                'sem_scores': b['semantic_scores']}[key]
    for i, ex, vols in _replay(h, w, grid, frames, get):
        for key in ('fusion_values', 'fusion_weights', 'indices', 'weights', 'points', 'pcl'):
            assert sha(ex[key]) == dig['f%d_%s' % (i, key)], (key, i)
        for key in ('tsdf', 'wgt', 'ids', 'scores'):
            assert sha(vols[key]) == dig['f%d_%s' % (i, key)], (key, i)


@pytest.mark.parametrize('sem', [True, False])
def test_reference_pipeline_golden(sem):
    """Reference Pipeline.fuse x3 (+ the net fixture): oracle frame step reproduces it.
    The torch net here is the package's module with the golden state_dict; conv/BN arithmetic is the
    same ATen code the reference ran, so tsdf_est and hence all four volumes are bit-identical."""
    g = golden('pipeline_v3_%s_24x32_g32.npz' % ('sem' if sem else 'nosem'))
    h, w, grid = 24, 32, 32
    net = net_from_golden(g, sem, h, w)
    st = make_stream(h, w, grid)
    vols = fresh_volumes(grid, True)
    for i in range(3):
        oracle_fuse(st, i, vols, net, True)
        for key in ('tsdf', 'wgt', 'ids', 'scores'):
            assert n_mismatch(vols[key], g['f%d_%s' % (i, key)]) == 0, (key, i)
    f = st.frame(3)
    est = oracle_net_est(net, g['net_fusion_values'], g['net_fusion_weights'], f['tof_depth'], f['semantic_gt'], 30, h, w)
    assert np.abs(est - g['net_est']).max() <= 1e-6


def test_fp16_conversions_exhaustive():
    lib = oracle.lib()
    import torch
    allh = np.arange(65536, dtype=np.uint16)
    ref = torch.from_numpy(allh.view(np.float16)).float().numpy()
    got = np.array([lib.ojf_oracle_h2f(int(x)) for x in allh[::7]], dtype=np.float32)
    assert n_mismatch(got, ref[::7]) == 0
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-70000, 70000, 4000), rng.uniform(-1e-4, 1e-4, 4000), rng.uniform(-1, 1, 4000),
                         [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, np.inf, -np.inf]]).astype(np.float32)
    want = torch.from_numpy(xs).half().numpy().view(np.uint16)
    got = np.array([lib.ojf_oracle_f2h(float(x)) for x in xs], dtype=np.uint16)
    assert (got == want).all()


def test_oracle_extract_on_ground_truth_grid_matches_reference_fuse_training_at_B():
    """tests/golden/train_v3_nosem_240x320_g256.npz holds sha256 of ``tsdf_target`` of the reference's own
    ``fuse_training`` at 320x240 -> 256^3: the ground-truth grid interpolated along the rays of frame 2 (modules/
    pipeline.py:306-312).  A grid with sign changes exposes the order of torch.sum's 8-term fp64 sums (ATen row_sum:
    interleaved partial sums p_k + p_{k+4}); the oracle follows it bit for bit, and the pre-frame state it builds with two
    integrate calls is the reference's (sha256)."""
    import hashlib
    from online_joint_depthfusion_and_semantic_amd.synthetic import gt_volumes
    from helpers import frame_inputs, fresh_volumes, make_stream
    g = golden('train_v3_nosem_240x320_g256.npz')
    h, w, grid = 240, 320, 256
    st = make_stream(h, w, grid)
    vols = fresh_volumes(grid, False)
    for i in range(2):
        fi = frame_inputs(st, i)
        oracle.integrate(fi['fd'], fi['Ki'], fi['E'], st.origin, st.resolution, fi['est'], vols['tsdf'], vols['wgt'])
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(vols['tsdf']) == str(g['pre_tsdf_sha256']) and sha(vols['wgt']) == str(g['pre_wgt_sha256'])
    gt, _ = gt_volumes(grid)
    fi = frame_inputs(st, 2)
    ex = oracle.extract(fi['depth'], fi['Ki'], fi['E'], st.origin, st.resolution, gt, vols['wgt'])
    valid = fi['fd'].reshape(-1) != 0
    assert int(valid.sum()) == int(g['n_valid'])
    assert sha(ex['fusion_values'][valid]) == str(g['tsdf_target_sha256'])
