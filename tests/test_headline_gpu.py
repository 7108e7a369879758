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


# ---- configs[3]: the training frame step at 320x240 -> 256^3 against the reference's own fuse_training -----------
@pytest.mark.parametrize('engine', ['hip', 'torch'])
def test_fuse_training_at_B_matches_reference_golden(cuda, engine):
    """modules/pipeline.py:251-363 driven as train_fusion.py:166-177 does, at BASELINE configs[3]'s frame and grid size.
    Fixture = tests/golden/make_golden.py::run_training_full_size: the reference's ``Pipeline.fuse_training`` (eval()
    mode) on frame 2 of the stream, from a pre-frame state that the C oracle reproduces bit for bit (two frames with
    the seeded stand-in of helpers.frame_inputs; sha256 asserted), then ``loss.backward()`` of
    mean|fused - target| + 10 mean (fused - target)^2.  Compared: tsdf_target bit for bit, tsdf_est / tsdf_fused within
    2e-5 (every 7th row + float64 column sums over all 76 800 rows), the valid-pixel count, EVERY parameter gradient
    (per tensor <= 2e-3 of max(its own scale, 1e-3 of the largest gradient): eval()-mode fp32 nets differ among
    themselves by ~2e-4 there because |fused - target| changes sign under rounding; measured figures are printed),
    the post-frame weight volume bit for bit (PARITY integrate) and the post-frame TSDF within one fp16 ulp."""
    g = golden('train_v3_nosem_240x320_g256.npz')
    g_small = golden('pipeline_v3_nosem_24x32_g32.npz')
    h, w, grid = 240, 320, 256
    cfg = default_config(h, w, semantics=False, use_semantics=False, integrate_mode='parity')
    cfg.SETTINGS.device = str(cuda)
    cfg.FUSION_MODEL.train_engine = engine
    st = make_stream(h, w, grid)
    db = Database(st, database_config(cfg))
    pipe = Pipeline(cfg)
    pipe._fusion_network.load_state_dict({k[len('state_'):]: torch.from_numpy(g_small[k]) for k in g_small.files if k.startswith('state_')})
    pipe = pipe.to(cuda).eval()
    s = st.scene
    vols = fresh_volumes(grid, False)
    for i in range(2):
        fi = frame_inputs(st, i)
        oracle.integrate(fi['fd'], fi['Ki'], fi['E'], st.origin, st.resolution, fi['est'], vols['tsdf'], vols['wgt'])
    assert sha(vols['tsdf']) == str(g['pre_tsdf_sha256']) and sha(vols['wgt']) == str(g['pre_wgt_sha256'])
    b = {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in st.batch(2).items()}
    # The hip engine's backward pass runs backward-data and the weight gradients in split-fp16 (round 3: from the second pass
    # on, which is why the frame is stepped more than once here; since round 4 every pass, under factors derived in the pass
    # itself).  The same frame is stepped twice from the same pre-frame volumes (gradients dropped in between, so the second
    # pass also meets the executor's steady state: packed weights cached, gradient tensors in place) and EVERY pass is held
    # against the reference's golden gradients directly.
    for rep in range(2 if engine == 'hip' else 1):
        db.scenes_est[s].volume.copy_(_t(vols['tsdf'], cuda))
        db.fusion_weights[s].copy_(_t(vols['wgt'], cuda))
        for p in pipe._fusion_network.parameters():
            p.grad = None
        out = pipe.fuse_training(b, db, cuda)
        _check_training_frame_against_golden(g, pipe, db, s, out, h, w, '%s pass %d' % (engine, rep + 1))


def _check_training_frame_against_golden(g, pipe, db, s, out, h, w, engine):
    assert out['tsdf_fused'].requires_grad
    assert out['tsdf_fused'].shape == (1, int(g['n_valid']), 9) and out['tsdf_est'].shape == (1, h * w, 9)
    assert sha(out['tsdf_target'].detach()[0]) == str(g['tsdf_target_sha256'])
    for k in ('tsdf_est', 'tsdf_fused'):
        a = out[k].detach()[0].cpu().numpy()
        d_rows = float(np.abs(a[::7] - g[k + '_rows7']).max())
        d_sum = float(np.abs(a.astype(np.float64).sum(0) - g[k + '_colsum']).max())
        print('fuse_training B %s %s: max |d| on every 7th row %.2e, column sums differ by %.2e (of %.1f)'
              % (engine, k, d_rows, d_sum, float(np.abs(g[k + '_colsum']).max())))
        assert d_rows <= 2e-5, (k, d_rows)
        assert d_sum <= 2e-5 * a.shape[0] * 0.05, (k, d_sum)  # rounding differences do not add up coherently
    diff = out['tsdf_fused'] - out['tsdf_target']
    loss = diff.abs().mean() + 10 * (diff ** 2).mean()
    assert abs(float(loss) - float(g['loss'])) <= 1e-6
    loss.backward()
    grads = {n: p.grad for n, p in pipe._fusion_network.named_parameters()}
    gmax = max(float(np.abs(g['grad_' + n]).max()) for n in grads if g['grad_' + n].size)
    worst = (0.0, None)
    for n, got in grads.items():
        want = g['grad_' + n]
        assert (got is None) == (want.size == 0), n
        if got is None:
            continue
        scale = max(float(np.abs(want).max()), 1e-3 * gmax)
        e = float(np.abs(got.cpu().numpy() - want).max()) / scale
        worst = max(worst, (e, n))
        assert e <= 2e-3, (n, e, scale)
    print('fuse_training B %s: worst parameter-gradient deviation %.2e of its scale (%s), largest gradient %.3e' % (engine, worst[0], worst[1], gmax))
    wgt = db.fusion_weights[s]
    assert int((wgt > 0).sum()) == int(g['post_touched'])
    assert sha(wgt) == str(g['post_wgt_sha256'])
    got = db.scenes_est[s].volume.cpu().numpy()[wgt.cpu().numpy() > 0]
    want = g['post_tsdf_touched']
    assert (np.isnan(got) == np.isnan(want)).all()
    ad = np.nan_to_num(np.abs(got.astype(np.float32) - want.astype(np.float32)))
    print('   post-frame TSDF: max |d| %.2e, %.4f %% of %d touched voxels differ' % (float(ad.max()), 100 * float((ad > 0).mean()), got.size))
    assert ad.max() <= F16_ULP_BAND and float((ad > 0).mean()) <= 0.004


# ---- configs[3]: a multi-step training TRAJECTORY against the reference (VERDICT r3 missing #4) ----------------------
@pytest.mark.parametrize('engine', ['hip', 'torch'])
def test_training_trajectory_matches_reference_golden(cuda, engine):
    """train_fusion.py:145-189 for 2 windows x 4 frames: fuse_training -> loss.backward() -> clip_grad_norm_ per frame,
    RMSprop step / zero_grad / PolynomialLR step per window (tests/golden/make_golden.py::run_training_trajectory: the
    reference's own Pipeline and optimizer, eval() mode, lr 1e-3).  What a single-frame fixture cannot see: the executor
    re-packing its weight copies after each optimizer step, the split-fp16 backward running under factors measured on
    the PREVIOUS frame across steps, gradient accumulation over a window through the staging buffers, the volumes the
    later frames extract from.  The two steps move tsdf_est by up to ~0.3 (printed), so a stale weight copy or a lost
    frame of gradient is far outside the bars.  RMSprop's first steps are sign-like (g / sqrt(0.01 g^2)): an element
    whose gradient is ~0 may land on the other side in any fp32 implementation - final parameters are compared
    element-wise with a small allowance for such elements (the reference against itself with 1e-5 relative gradient
    noise: tsdf_est 7e-5, 2.3e-3 on the worst parameter element)."""
    from online_joint_depthfusion_and_semantic_amd.loss import PolynomialLR
    g = golden('train_trajectory_v3_nosem_48x64_g64.npz')
    g_small = golden('pipeline_v3_nosem_24x32_g32.npz')
    h, w, grid = 48, 64, 64
    frames, accum, lr = int(g['frames']), int(g['accum']), float(g['lr'])
    cfg = default_config(h, w, semantics=False, use_semantics=False, integrate_mode='parity')
    cfg.SETTINGS.device = str(cuda)
    cfg.FUSION_MODEL.train_engine = engine
    st = make_stream(h, w, grid)
    db = Database(st, database_config(cfg))
    pipe = Pipeline(cfg)
    state0 = {k[len('state_'):]: torch.from_numpy(g_small[k]) for k in g_small.files if k.startswith('state_')}
    pipe._fusion_network.load_state_dict(state0)
    pipe = pipe.to(cuda).eval()
    net = pipe._fusion_network
    opt = torch.optim.RMSprop(net.parameters(), lr=lr, momentum=0.9, weight_decay=0.01, eps=1e-9)
    sched = PolynomialLR(opt, 50000)
    s = st.scene
    est = {}
    for i in range(frames):
        b = {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in st.batch(i).items()}
        out = pipe.fuse_training(b, db, cuda)
        assert out['tsdf_fused'].shape[1] == int(g['n_valid'][i]), i
        diff = out['tsdf_fused'] - out['tsdf_target']
        loss = diff.abs().mean() + 10 * (diff ** 2).mean()
        loss.backward()
        assert abs(float(loss) - float(g['loss'][i])) <= 2e-4 * float(g['loss'][i]), (i, float(loss), float(g['loss'][i]))
        if i in (accum, frames - 1):
            est[i] = out['tsdf_est'].detach()[0].cpu().numpy()
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1., norm_type=2)
        if (i + 1) % accum == 0 or i == frames - 1:
            opt.step()
            opt.zero_grad()
            sched.step()
    moved = float(np.abs(g['f%d_tsdf_est' % (frames - 1)] - g['f%d_tsdf_est' % accum]).max())
    for i, a in est.items():
        d = float(np.abs(a - g['f%d_tsdf_est' % i]).max())
        print('training trajectory %s: frame %d tsdf_est max |d| %.2e (the steps move it by %.2e)' % (engine, i, d, moved))
        assert d <= 1e-3, (i, d)
    worst, off, total = 0.0, 0, 0
    for name, p in net.named_parameters():
        want = g['final_' + name]
        ad = np.abs(p.detach().cpu().numpy() - want)
        worst = max(worst, float(ad.max()))
        off += int((ad > 1e-3).sum())
        total += ad.size
        assert float(np.abs(want - state0[name].numpy()).max()) > 0 or want.size == 0
    print('   final parameters: max |d| %.2e, %d of %d elements beyond 1e-3 (two steps move an element by up to %.1e)'
          % (worst, off, total, 2 * 10 * lr * 1.9))
    assert off <= 2e-3 * total and worst <= 4.5e-2
    wgt = db.fusion_weights[s].cpu().numpy()
    assert np.array_equal(wgt, g['post_wgt'])  # the geometry of eight frames, bit for bit (PARITY integrate)
    got, want = db.scenes_est[s].volume.cpu().numpy(), g['post_tsdf']
    assert (np.isnan(got) == np.isnan(want)).all()
    ad = np.nan_to_num(np.abs(got.astype(np.float32) - want.astype(np.float32)))
    print('   post-trajectory TSDF: max |d| %.2e on %d voxels of %d touched' % (float(ad.max()), int((ad > 0).sum()), int((wgt > 0).sum())))
    assert ad.max() <= 1e-3


# ---- configs[2] with PREDICTED labels at its own size against the reference's Pipeline.fuse + AdapNet -------------
@pytest.mark.parametrize('engine', ['hip', 'torch'])
def test_fuse_predict_strategy_at_B_matches_reference_golden(cuda, engine):
    """modules/pipeline.py:42-60,181-185 composed with the fusion path at 320x240 -> 256^3, 30 classes, two frames
    (tests/golden/make_golden_adapnet.py::predict_pipeline_full_size: the reference's own AdapNet around the stand-in
    ResNet-50, seeded weights, dropout flags off).  Per frame: softmax scores within 1e-6, arg-max ids equal wherever
    the reference's top-1 / top-2 margin exceeds 1e-4, the weight volume bit for bit (PARITY); after the last frame id /
    score / TSDF volumes at the touched voxels (a flipped near-tie pixel may change the <= 56 entries it writes and,
    through the semantic input channel of the two-head net, TSDF values)."""
    from adapnet_golden_util import randomise_net
    g = golden('pipeline_predict_240x320_g256.npz')
    small = golden('pipeline_v3_sem_24x32_g32.npz')
    h, w, grid, n_classes = 240, 320, 256, 30
    cfg = default_config(h, w, semantics=True, use_semantics=True, n_classes=n_classes, integrate_mode='parity')
    cfg.SETTINGS.device = str(cuda)
    cfg.DATA.semantic_strategy = 'predict'
    cfg.SEMANTIC_2D_MODEL.engine = engine
    st = make_stream(h, w, grid, n_classes=n_classes)
    db = Database(st, database_config(cfg))
    pipe = Pipeline(cfg)
    pipe._fusion_network.load_state_dict({k[len('state_'):]: torch.from_numpy(small[k]) for k in small.files if k.startswith('state_')})
    randomise_net(pipe._semantic_2d_network, 31)
    pipe = pipe.to(cuda).eval()
    pipe.device = torch.device(cuda)
    s = st.scene
    flips_total = 0
    with torch.no_grad():
        for i in range(2):
            b = {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in st.batch(i).items()}
            ids, scores = pipe._frame_semantics(b)
            scores, ids = scores.reshape(h, w).cpu().numpy(), ids.reshape(h, w).cpu().numpy()
            clear = g['f%d_seg_margin' % i].astype(np.float32) > 1e-4
            flips = int((ids != g['f%d_seg_ids' % i]).sum())
            flips_total += flips
            ds = float(np.abs(scores - g['f%d_seg_scores' % i]).max())
            print('predict B %s frame %d: max |d score| %.2e, %d arg-max flips (%d pixels with margin < 1e-4)'
                  % (engine, i, ds, flips, int((~clear).sum())))
            # the HIP engine is bit-reproducible (measured 1.3e-7); the torch engine runs MIOpen, whose algorithm choice -
            # and with it the rounding - varies with what ran before in the process (seen in full-suite runs only)
            assert ds <= (1e-6 if engine == 'hip' else 2e-5)
            assert (ids[clear] == g['f%d_seg_ids' % i][clear]).all() and flips <= (~clear).sum()
            pipe.fuse(b, db, cuda)
            wgt = db.fusion_weights[s]
            assert int((wgt > 0).sum()) == int(g['f%d_touched' % i])
            assert sha(wgt) == str(g['f%d_wgt_sha256' % i]), i
    pipe.check()
    touched = db.fusion_weights[s].cpu().numpy() > 0
    got_ids = db.ids_est[s].volume.cpu().numpy()[touched]
    got_sc = db.scores[s].volume.cpu().numpy()[touched]
    got_t = db.scenes_est[s].volume.cpu().numpy()[touched]
    id_bad = int((got_ids != g['last_ids_touched']).sum())
    sc_ulp = f16_ulp_distance(got_sc, g['last_scores_touched'])
    assert (np.isnan(got_t) == np.isnan(g['last_tsdf_touched'])).all()
    td = np.nan_to_num(np.abs(got_t.astype(np.float32) - g['last_tsdf_touched'].astype(np.float32)))
    print('   volumes: %d of %d touched voxels with another id, score ulps max %d (%d voxels > 0), max |dTSDF| %.2e, %.4f %% differ'
          % (id_bad, int(touched.sum()), int(sc_ulp.max()), int((sc_ulp > 0).sum()), float(td.max()), 100 * float((td > 0).mean())))
    if engine == 'hip':
        assert id_bad <= 56 * flips_total
        assert sc_ulp.max() <= 1 or (sc_ulp > 1).sum() <= 56 * flips_total
        assert td.max() <= (F16_ULP_BAND if flips_total == 0 else 2e-3)
        assert float((td > 0).mean()) <= (0.004 if flips_total == 0 else 0.02)
    else:
        # MIOpen's convolutions are not bit-reproducible between processes: scores move by up to ~1e-6, which can turn the
        # integrator's "new score > stored score" comparisons (modules/integrator.py:113-117) at near-ties without any
        # arg-max flip in the frame - a few voxel ids and, through the semantic input channel, TSDF values follow
        assert id_bad <= 56 * flips_total + 1e-3 * touched.sum()
        assert (sc_ulp > 1).sum() <= 56 * flips_total + 1e-3 * touched.sum()
        assert td.max() <= 2e-3 and float((td > F16_ULP_BAND).mean()) <= 0.02


# ---- the BATCHED 2-D pass (look-ahead chunks, what drivers.test_fusion runs by default) against the same reference golden ----
@pytest.mark.parametrize('B', [4, 8])
def test_batched_2d_pass_at_B_matches_reference_golden(cuda, B):
    """modules/pipeline.py:42-60,181-185 at 320x240 -> 256^3, 30 classes, with the labels predicted as ONE batched AdapNet++
    pass of B frames (SegEngine.predict_many: at this size the GEMM-shaped kernel picks tile forms a single frame never uses).
    The two frames the reference's own ``Pipeline.fuse`` + AdapNet golden holds sit at DIFFERENT batch positions, the other
    positions carry frames of a second scene: per golden frame softmax scores within 1e-6, arg-max ids equal wherever the
    reference's top-1 / top-2 margin exceeds 1e-4; then ``fuse_sequence`` over the same chunk (stream order: the golden scene
    sees frame 0, then frame 1) - the golden scene's weight volume by sha256 (PARITY integrate), id / score / TSDF volumes at the
    touched voxels as in the frame-at-a-time test."""
    from adapnet_golden_util import randomise_net
    from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticDataset
    g = golden('pipeline_predict_240x320_g256.npz')
    small = golden('pipeline_v3_sem_24x32_g32.npz')
    h, w, grid, n_classes = 240, 320, 256, 30
    cfg = default_config(h, w, semantics=True, use_semantics=True, n_classes=n_classes, integrate_mode='parity')
    cfg.SETTINGS.device = str(cuda)
    cfg.DATA.semantic_strategy = 'predict'
    scenes = ('room_0', 'room_1')  # room_0 = the golden's stream (seed 1911, 20 frames), room_1 fills the other positions
    ds = SyntheticDataset(h, w, grid, 20, scenes=scenes, n_classes=n_classes)
    db = Database(ds, database_config(cfg))
    pipe = Pipeline(cfg)
    pipe._fusion_network.load_state_dict({k[len('state_'):]: torch.from_numpy(small[k]) for k in small.files if k.startswith('state_')})
    randomise_net(pipe._semantic_2d_network, 31)
    pipe = pipe.to(cuda).eval()
    pipe.device = torch.device(cuda)
    pos = {4: (1, 3), 8: (2, 6)}[B]  # batch positions of golden frames 0 and 1

    def batch(s, i):
        return {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in ds.streams[s].batch(i).items()}
    chunk, filler = [], 0
    for p in range(B):
        if p in pos:
            chunk.append(batch('room_0', pos.index(p)))
        else:
            chunk.append(batch('room_1', filler))
            filler += 1
    flips_total = 0
    with torch.no_grad():
        sems = pipe._frame_semantics_many(chunk)
        assert pipe.__dict__['_seg_graph_many']['graph'] is not None  # the batched pass, captured and replayed
        for i, p in enumerate(pos):
            ids, scores = sems[p]
            scores, ids = scores.reshape(h, w).cpu().numpy(), ids.reshape(h, w).cpu().numpy()
            clear = g['f%d_seg_margin' % i].astype(np.float32) > 1e-4
            flips = int((ids != g['f%d_seg_ids' % i]).sum())
            flips_total += flips
            ds_ = float(np.abs(scores - g['f%d_seg_scores' % i]).max())
            print('batched 2-D pass B = %d, golden frame %d at position %d: max |d score| %.2e, %d arg-max flips (%d pixels with margin < 1e-4)'
                  % (B, i, p, ds_, flips, int((~clear).sum())))
            assert ds_ <= 1e-6
            assert (ids[clear] == g['f%d_seg_ids' % i][clear]).all() and flips <= (~clear).sum()
        pipe.fuse_sequence(chunk, db, cuda)
    pipe.check()
    s = 'room_0'
    wgt = db.fusion_weights[s]
    assert int((wgt > 0).sum()) == int(g['f1_touched'])
    assert sha(wgt) == str(g['f1_wgt_sha256'])
    touched = wgt.cpu().numpy() > 0
    got_ids = db.ids_est[s].volume.cpu().numpy()[touched]
    got_sc = db.scores[s].volume.cpu().numpy()[touched]
    got_t = db.scenes_est[s].volume.cpu().numpy()[touched]
    id_bad = int((got_ids != g['last_ids_touched']).sum())
    sc_ulp = f16_ulp_distance(got_sc, g['last_scores_touched'])
    assert (np.isnan(got_t) == np.isnan(g['last_tsdf_touched'])).all()
    td = np.nan_to_num(np.abs(got_t.astype(np.float32) - g['last_tsdf_touched'].astype(np.float32)))
    print('   volumes: %d of %d touched voxels with another id, score ulps max %d (%d voxels > 0), max |dTSDF| %.2e, %.4f %% differ'
          % (id_bad, int(touched.sum()), int(sc_ulp.max()), int((sc_ulp > 0).sum()), float(td.max()), 100 * float((td > 0).mean())))
    assert id_bad <= 56 * flips_total
    assert sc_ulp.max() <= 1 or (sc_ulp > 1).sum() <= 56 * flips_total
    assert td.max() <= (F16_ULP_BAND if flips_total == 0 else 2e-3)
    assert float((td > 0).mean()) <= (0.004 if flips_total == 0 else 0.02)
    assert float((db.fusion_weights['room_1'].float() > 0).sum()) > 1000  # (the filler frames were fused too, into their own scene)
