"""-m gpu: the BASELINE configurations at their own sizes, end to end.

 * configs[1] / [2] (320x240 depth -> 256^3 grid, FusionNet_v3 without / with the semantic head):
   ``Pipeline.fuse`` against fixtures produced by the REFERENCE's own ``Pipeline.fuse`` at that size
   (tests/golden/make_golden.py --full-size): weight / id / score volumes by sha256 (PARITY mode: bit for bit),
   TSDF at the touched voxels within one fp16 ulp, the reference's volume metrics; FAST mode against the same
   fixtures with its own budget; IoU / accuracy / F-score parity between oracle volumes and HIP volumes at B.
 * FusionNet_v3 at 240x320 and 480x640 in both arithmetics against the fp32 CPU net (which
   tests/test_oracle_golden.py pins on the reference's ``net_est``): |d tsdf_est| <= 1e-5.
 * configs[4] (640x480 -> 512^3, 40 classes): one frame of extract + integrate WITH semantics against the C oracle.

Tolerances are the measured figures with a small margin; they are stated next to each assert.
"""
import copy
import functools
import hashlib

import numpy as np
import pytest
import torch

from oracle import oracle
from online_joint_depthfusion_and_semantic_amd import metrics, ops
from online_joint_depthfusion_and_semantic_amd.config import default_config, database_config
from online_joint_depthfusion_and_semantic_amd.database import Database
from online_joint_depthfusion_and_semantic_amd.engine import FusionNetEngine
from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline
from helpers import (n_mismatch, f16_ulp_distance, golden, net_from_golden, oracle_fuse, fresh_volumes, make_stream,
                     frame_inputs, to_cuda)

pytestmark = pytest.mark.gpu
F16_ULP_BAND = 6.2e-5  # one fp16 ulp at the top of the +-0.1 truncation band (ulp of [0.0625, 0.125) = 6.1e-5)


def sha(t):
    a = t.cpu().numpy() if torch.is_tensor(t) else t
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ---- FusionNet_v3 at the BASELINE frame sizes -------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def _net_case(h, w, sem):
    """Seeded net (the golden state_dict of the pipeline fixtures), seeded inputs with the statistics of real
    extractor outputs, and the fp32 CPU forward (one thread: the golden-vector rule)."""
    g = golden('pipeline_v3_%s_24x32_g32.npz' % ('sem' if sem else 'nosem'))
    net = net_from_golden(g, sem, h, w)
    gen = torch.Generator().manual_seed(h * 7 + w)
    x = dict(tsdf_values=(torch.rand(1, 9, h, w, generator=gen) - 0.5) * 0.2,
             tsdf_weights=torch.rand(1, 9, h, w, generator=gen) * 4,
             tsdf_frame=torch.rand(1, 1, h, w, generator=gen) * 4)
    ids = torch.randint(0, 30, (h, w), generator=gen, dtype=torch.uint8)
    x['semantic_frame'] = ((1 + ids.float()) / 30).view(1, 1, h, w)
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        with torch.no_grad():
            ref = net(x)[0].permute(1, 2, 0).reshape(h * w, 9).contiguous()
    finally:
        torch.set_num_threads(n)
    return net, x, ids, ref


@pytest.mark.parametrize('arith', ['f16x3', 'f32'])
@pytest.mark.parametrize('sem', [False, True])
@pytest.mark.parametrize('h,w', [(240, 320), (480, 640)])
def test_fusion_net_v3_at_baseline_frame_sizes(cuda, h, w, sem, arith):
    """modules/model.py:265-283 at the frame sizes of BASELINE configs[1]-[4]: the dilation-27 border handling, the
    strip / tile raggedness and the LDS tap table all depend on (h, w)."""
    net, x, ids, ref = _net_case(h, w, sem)
    eng = FusionNetEngine(net, h, w, cuda, arithmetic=arith)
    rows = lambda t: t[0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(cuda)
    eng.prepare_input(rows(x['tsdf_values']), rows(x['tsdf_weights']), x['tsdf_frame'].reshape(h, w).contiguous().to(cuda),
                      ids.contiguous().to(cuda) if sem else None, 30)
    est = torch.empty((h * w, 9), device=cuda)
    eng.forward(est)
    eng.check()
    err = (est.cpu() - ref).abs()
    print('FusionNet_v3 %dx%d sem=%s %s: max |d est| = %.2e, mean %.2e' % (w, h, sem, arith, float(err.max()), float(err.mean())))
    assert float(err.max()) <= 1e-5  # <= 1e-5 with fp32-class MFMA arithmetic
    eng.close()


# ---- Pipeline.fuse at configs[1] / [2] against the reference's own Pipeline.fuse ---------------------------------
def _pipeline_B(cuda, use_sem, mode):
    g_small = golden('pipeline_v3_%s_24x32_g32.npz' % ('sem' if use_sem else 'nosem'))
    state = {k[len('state_'):]: torch.from_numpy(g_small[k]) for k in g_small.files if k.startswith('state_')}
    h, w, grid = 240, 320, 256
    cfg = default_config(h, w, semantics=True, use_semantics=use_sem, integrate_mode=mode)
    cfg.SETTINGS.device = str(cuda)
    st = make_stream(h, w, grid)
    db = Database(st, database_config(cfg))
    pipe = Pipeline(cfg)
    pipe._fusion_network.load_state_dict(state)
    return st, db, pipe.to(cuda).eval()


@pytest.mark.parametrize('mode', ['parity', 'fast'])
@pytest.mark.parametrize('use_sem', [False, True])
def test_pipeline_fuse_at_B_matches_reference_golden(cuda, use_sem, mode):
    g = golden('pipeline_v3_%s_240x320_g256.npz' % ('sem' if use_sem else 'nosem'))
    st, db, pipe = _pipeline_B(cuda, use_sem, mode)
    s = st.scene
    with torch.no_grad():
        for i in range(3):
            b = {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in st.batch(i).items()}
            pipe.fuse(b, db, cuda)
            # semantic ids / scores: "last writer wins" is reproduced exactly in both modes
            assert sha(db.ids_est[s].volume) == str(g['f%d_ids_sha256' % i]), i
            assert sha(db.scores[s].volume) == str(g['f%d_scores_sha256' % i]), i
            wgt = db.fusion_weights[s]
            assert int((wgt > 0).sum()) == int(g['f%d_touched' % i]), i  # same voxel set (indices are bit-exact)
            if mode == 'parity':  # the weight path never sees the net: bit for bit, every frame
                assert sha(wgt) == str(g['f%d_wgt_sha256' % i]), i
    pipe.check()
    wgt = db.fusion_weights[s].cpu().numpy()
    tsdf = db.scenes_est[s].volume.cpu().numpy()
    touched = wgt > 0
    got, want = tsdf[touched], g['last_tsdf_touched']
    assert got.shape == want.shape
    assert (np.isnan(got) == np.isnan(want)).all()
    ad = np.nan_to_num(np.abs(got.astype(np.float32) - want.astype(np.float32)))
    frac = float((ad > 0).mean())
    print('Pipeline.fuse B sem=%s %s: max |dTSDF| %.2e, %.4f %% of %d touched voxels differ'
          % (use_sem, mode, float(ad.max()), 100 * frac, got.size))
    # the HIP net differs from the reference's CPU net by <= 1e-6 in tsdf_est (stated bar 1e-5); after three frames
    # that moves the fp16 rounding of (w_old*v_old + U)/(w_old + W) by one step of the band's LARGEST ulp at most, at
    # a fraction of a percent of the touched voxels (measured: PARITY 0.14 %, FAST 0.17 %; max 3.05e-5)
    assert ad.max() <= F16_ULP_BAND
    assert frac <= 0.004
    # the reference's metrics (utils/metrics.py:111-127) on the reference's volumes vs ours on ours
    have = db.evaluate(mode='val')
    for k in ('mse', 'mad', 'iou', 'acc'):
        want_m = float(g['metric_' + k])
        print('   %s: reference %.9f  here %.9f' % (k, want_m, have[k]))
        assert abs(have[k] - want_m) <= 2e-5 * max(1.0, abs(want_m)), (k, want_m, have[k])


def test_metric_and_fscore_parity_at_B(cuda):
    """BASELINE 'F-score/IoU parity' at configs[1]: four frames through the HIP path (FAST) and through the CPU oracle
    frame step (C extract / integrate around the fp32 CPU net); mse / mad / IoU / accuracy and the reconstruction
    F-score of the two sets of volumes agree."""
    h, w, grid, frames = 240, 320, 256, 4
    cfg = default_config(h, w, semantics=False, integrate_mode='fast')
    cfg.SETTINGS.device = str(cuda)
    st = make_stream(h, w, grid)
    db = Database(st, database_config(cfg))
    g_small = golden('pipeline_v3_nosem_24x32_g32.npz')
    pipe = Pipeline(cfg)
    pipe._fusion_network.load_state_dict({k[len('state_'):]: torch.from_numpy(g_small[k]) for k in g_small.files
                                          if k.startswith('state_')})
    pipe = pipe.to(cuda).eval()
    cpu_net = copy.deepcopy(pipe._fusion_network).cpu().eval()
    vols = fresh_volumes(grid, False)
    with torch.no_grad():
        for i in range(frames):
            pipe.fuse({k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in st.batch(i).items()}, db, cuda)
            oracle_fuse(st, i, vols, cpu_net, False)
    pipe.check()
    s = st.scene
    got_t, got_w = db.scenes_est[s].volume.cpu().numpy(), db.fusion_weights[s].cpu().numpy()
    assert ((got_w > 0) == (vols['wgt'] > 0)).all()
    wd = f16_ulp_distance(got_w, vols['wgt'])
    print('   weights: %d of %d touched voxels differ by one ulp after %d frames' % (int((wd > 0).sum()), int((got_w > 0).sum()), frames))
    assert wd.max() <= 1 and (wd > 0).sum() <= 1e-3 * (got_w > 0).sum()
    gt = db.scenes_gt[s].volume.cpu().numpy()
    have = db.evaluate(mode='val')
    want = metrics.evaluation(vols['tsdf'], gt, vols['wgt'] > 0)
    for k in want:
        print('   %s: oracle %.9f  hip %.9f' % (k, want[k], have[k]))
        assert abs(want[k] - have[k]) <= 2e-5 * max(1.0, abs(want[k])), (k, want[k], have[k])
    f_want = metrics.reconstruction_f_score(vols['tsdf'], gt, vols['wgt'], st.origin, st.resolution)
    f_have = metrics.reconstruction_f_score(db.scenes_est[s].volume, db.scenes_gt[s].volume, db.fusion_weights[s],
                                            st.origin, st.resolution)
    print('   F-score: oracle %r  hip %r' % (f_want, f_have))
    assert f_want['fscore'] > 0.01
    for k in ('precision', 'recall', 'fscore'):
        assert abs(f_want[k] - f_have[k]) <= 1e-3, (k, f_want, f_have)


# ---- configs[4]: 640x480 -> 512^3, 40 classes, against the C oracle ---------------------------------------------
def test_config_C_frame_with_semantics_against_oracle(cuda):
    h, w, grid, n_classes = 480, 640, 512, 40
    st = make_stream(h, w, grid, n_classes=n_classes)
    rng = np.random.default_rng(11)
    vols = fresh_volumes(grid, True)
    fi0, fi1 = frame_inputs(st, 2), frame_inputs(st, 3)
    assert int(fi1['sem_ids'].max()) < n_classes
    # pre-frame state: one oracle frame, so that the gather sees a non-trivial volume and old ids / scores exist
    oracle.integrate(fi0['fd'], fi0['Ki'], fi0['E'], st.origin, st.resolution, fi0['est'], vols['tsdf'], vols['wgt'],
                     sem_ids=fi0['sem_ids'], sem_scores=fi0['sem_scores'], id_vol=vols['ids'], score_vol=vols['scores'])
    pre = to_cuda(vols, cuda)
    # extract: bit-exact
    ref = oracle.extract(fi1['depth'], fi1['Ki'], fi1['E'], st.origin, st.resolution, vols['tsdf'], vols['wgt'])
    out = ops.extract(_t(fi1['depth'], cuda), fi1['Ki'], fi1['E'], st.origin, st.resolution, pre['tsdf'], pre['wgt'])
    for key in ref:
        assert n_mismatch(out[key].cpu().numpy(), ref[key]) == 0, key
    assert float(np.abs(ref['fusion_weights']).max()) > 0
    # integrate with semantics from the common pre-frame state
    want = {k: v.copy() for k, v in vols.items()}
    touched = oracle.integrate(fi1['fd'], fi1['Ki'], fi1['E'], st.origin, st.resolution, fi1['est'], want['tsdf'], want['wgt'],
                               sem_ids=fi1['sem_ids'], sem_scores=fi1['sem_scores'], id_vol=want['ids'], score_vol=want['scores'])
    assert touched > 500000
    for mode in (ops.MODE_PARITY, ops.MODE_FAST):
        g = {k: v.clone() for k, v in pre.items()}
        ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, mode, cuda)
        ops.integrate(_t(fi1['fd'], cuda), fi1['Ki'], fi1['E'], st.origin, st.resolution, _t(fi1['est'], cuda),
                      g['tsdf'], g['wgt'], ws, mode=mode, stats=True,
                      sem_ids=_t(fi1['sem_ids'].reshape(-1), cuda), sem_scores=_t(fi1['sem_scores'].reshape(-1), cuda),
                      id_vol=g['ids'], score_vol=g['scores'])
        assert int(ws.stats[0].item()) == touched
        assert n_mismatch(g['ids'].cpu().numpy(), want['ids']) == 0, mode
        assert n_mismatch(g['scores'].cpu().numpy(), want['scores']) == 0, mode
        for key in ('tsdf', 'wgt'):
            got = g[key].cpu().numpy()
            assert (np.isnan(got) == np.isnan(want[key])).all()
            ulp = np.where(np.isnan(got), 0, f16_ulp_distance(got, want[key]))
            print('config C %s %s: max %d ulp, %d of %d touched voxels differ' % (
                'parity' if mode == ops.MODE_PARITY else 'fast', key, int(ulp.max()), int((ulp > 0).sum()), touched))
            if mode == ops.MODE_PARITY:
                assert ulp.max() == 0, key
            else:  # <= 1 fp16 ulp on <= 0.05 % of the touched voxels (the FAST budget of test_extract_integrate_gpu.py)
                assert ulp.max() <= 1 and (ulp > 0).sum() <= 5e-4 * touched, (key, int((ulp > 0).sum()))
        del ws, g
