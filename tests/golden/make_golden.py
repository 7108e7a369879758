#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Only runs in the build container (needs /root/reference, which never travels to the GPU box).
The reference has no tests or fixtures for this path, so its Python modules,
imported unmodified, are the ground truth:
    modules/extractor.py  Extractor.forward
    modules/integrator.py Integrator.forward
    modules/pipeline.py   Pipeline.fuse / fuse_training (torchvision.models stubbed: AdapNet is
                          imported by pipeline.py but never constructed for semantic_strategy 'gt')
    modules/model.py      FusionNet_v3
Rules: torch.set_num_threads(1) so that the reference's duplicate-index writes are
deterministic ("highest entry wins"), seeded inputs from the shared synthetic generator, eval mode.

    python tests/golden/make_golden.py            # writes *.npz / *.json next to this file
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# the reference imports torchvision.models.resnet50 at module scope of modules/adapnet.py
tv = types.ModuleType('torchvision')
tv.models = types.ModuleType('torchvision.models')
tv.models.resnet50 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('stub'))
sys.modules.setdefault('torchvision', tv)
sys.modules.setdefault('torchvision.models', tv.models)

torch.set_num_threads(1)

from modules.extractor import Extractor  # noqa: E402
from modules.integrator import Integrator  # noqa: E402
from modules.pipeline import Pipeline as RefPipeline  # noqa: E402
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream  # noqa: E402


class NS(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def ref_config(h, w, semantics, use_semantics):
    return NS(SETTINGS=NS(gpu=False, device='cpu', implementation='standard'),
              FUSION_MODEL=NS(name='v3', output_scale=1.0, n_points=9, n_tail_points=7, growth_factor=6,
                              use_semantics=use_semantics),
              SEMANTIC_2D_MODEL=NS(stage=2, n_classes=30),
              DATA=NS(semantics='class30' if semantics else None, semantic_strategy='gt', input='tof_depth',
                      resx=w, resy=h, init_value=0.1))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_extract_integrate(h, w, grid, frames, keep_arrays):
    """Reference Extractor + Integrator over a stream with a seeded stand-in for the net output."""
    cfg = ref_config(h, w, True, False)
    ex, ig = Extractor(cfg), Integrator(cfg)
    st = SyntheticStream(h, w, grid, 20)
    tsdf = torch.full((grid,) * 3, 0.1, dtype=torch.float16)
    wgt = torch.zeros((grid,) * 3, dtype=torch.float16)
    ids = torch.zeros((grid,) * 3, dtype=torch.uint8)
    sc = torch.zeros((grid,) * 3, dtype=torch.float16)
    origin = torch.from_numpy(st.origin)
    out = {}
    for i in range(frames):
        b = st.batch(i)
        depth = b['tof_depth']
        if i == 1:  # exercise zero-depth pixels that are NOT masked out by noise alone
            depth[0, ::5, ::3] = 0.0
            b['mask'] = (depth > 0.05) & (depth < 5.0)
        values = ex.forward(depth, b['extrinsics'], b['intrinsics'], tsdf, wgt, origin, st.resolution)
        rng = np.random.default_rng([7, i])
        est = torch.from_numpy(rng.uniform(-0.15, 0.15, (1, h * w, 9)).astype(np.float32))
        fd = torch.where(b['mask'], depth, torch.zeros_like(depth)).view(1, h * w, 1)
        valid = (fd != 0.).nonzero()[:, 1]
        rep = lambda t: t.view(1, h * w, 1).unsqueeze(-2).repeat(1, 1, 9, 1)[:, valid, :7]
        updates = dict(values=torch.clamp(est[:, valid, :7], -0.1, 0.1), indices=values['indices'][:, valid, :7],
                       weights=values['weights'][:, valid, :7], semantics=rep(b['semantic_gt']),
                       scores=rep(b['semantic_scores']))
        tsdf, wgt, ids, sc = ig.forward(updates, tsdf, wgt, sc, ids)
        pre = 'f%d_' % i
        rec = dict(depth=depth[0].numpy(), mask=b['mask'][0].numpy(), extrinsics=b['extrinsics'][0].numpy(),
                   intrinsics=b['intrinsics'][0].numpy(), est=est[0].numpy(), sem_ids=b['semantic_gt'][0].numpy(),
                   sem_scores=b['semantic_scores'][0].numpy(),
                   fusion_values=values['fusion_values'][0].numpy(), fusion_weights=values['fusion_weights'][0].numpy(),
                   indices=values['indices'][0].numpy(), weights=values['weights'][0].numpy(),
                   points=values['points'][0].numpy(), pcl=values['pcl'][0].numpy(),
                   tsdf=tsdf.numpy().copy(), wgt=wgt.numpy().copy(), ids=ids.numpy().copy(), scores=sc.numpy().copy())
        if keep_arrays:
            rec['indices'] = rec['indices'].astype(np.int16)
            for k, v in rec.items():
                out[pre + k] = v
        else:
            for k in ('fusion_values', 'fusion_weights', 'indices', 'weights', 'points', 'pcl', 'tsdf', 'wgt', 'ids', 'scores'):
                out[pre + k] = sha(rec[k])
    return out


class DuckDatabase:
    """The five dict attributes + __getitem__ the reference Pipeline touches."""

    def __init__(self, st, semantics, gt):
        g = st.grid
        s = st.scene
        vol = lambda a: types.SimpleNamespace(volume=a)
        self.origin = torch.from_numpy(st.origin)
        self.resolution = st.resolution
        self.scenes_est = {s: vol(torch.full((g,) * 3, 0.1, dtype=torch.float16))}
        self.fusion_weights = {s: torch.zeros((g,) * 3, dtype=torch.float16)}
        self.ids_est = {s: vol(torch.zeros((g,) * 3, dtype=torch.uint8))}
        self.scores = {s: vol(torch.zeros((g,) * 3, dtype=torch.float16))}
        self.gt = torch.from_numpy(gt)
        self.state = {s: False}
        self.semantics = semantics

    def __getitem__(self, s):
        d = dict(origin=self.origin, resolution=self.resolution, gt=self.gt, current=self.scenes_est[s].volume,
                 weights=self.fusion_weights[s], ids_est=None, scores=None)
        if self.semantics:
            d.update(ids_est=self.ids_est[s].volume, scores=self.scores[s].volume)
        return d


def seeded_state(net, seed):
    torch.manual_seed(seed)
    for m in net.modules():  # train_fusion.py:29-31 xavier init, then randomised BN statistics
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.xavier_normal_(m.weight)
            m.bias.data.normal_(0, 0.05)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)


def run_pipeline(h, w, grid, frames, use_semantics):
    """Reference Pipeline.fuse over a stream, then one fuse_training frame from the final state."""
    from online_joint_depthfusion_and_semantic_amd.synthetic import gt_volumes
    cfg = ref_config(h, w, True, use_semantics)
    pipe = RefPipeline(cfg)
    seeded_state(pipe._fusion_network, 11)
    pipe.eval()
    st = SyntheticStream(h, w, grid, 20)
    gt, _ = gt_volumes(grid)
    db = DuckDatabase(st, True, gt)
    out = {'state_' + k: v.numpy() for k, v in pipe._fusion_network.state_dict().items()}
    with torch.no_grad():
        for i in range(frames):
            b = st.batch(i)
            pipe.fuse(b, db, torch.device('cpu'))
            s = st.scene
            for k, v in (('tsdf', db.scenes_est[s].volume), ('wgt', db.fusion_weights[s]),
                         ('ids', db.ids_est[s].volume), ('scores', db.scores[s].volume)):
                out['f%d_%s' % (i, k)] = v.numpy().copy()
    # net-only fixture on the last frame's extracted inputs (eval mode)
    b = st.batch(frames)
    vol = db[st.scene]
    with torch.no_grad():
        vals = pipe._extractor.forward(b['tof_depth'], b['extrinsics'], b['intrinsics'], vol['current'], vol['weights'],
                                       vol['origin'], vol['resolution'])
        pipe._shape = b['image'].shape
        pipe.device = torch.device('cpu')
        tin = pipe._prepare_fusion_input(b['tof_depth'], vals, b['semantic_gt'].long() if use_semantics else None)
        est = pipe._fusion(tin, vals)
    out['net_fusion_values'] = vals['fusion_values'][0].numpy()
    out['net_fusion_weights'] = vals['fusion_weights'][0].numpy()
    out['net_est'] = est[0].numpy()
    # one training frame (eval-mode BN so that the fixture does not depend on batch statistics)
    o = pipe.fuse_training(st.batch(frames), db, torch.device('cpu'))
    for k, v in o.items():
        out['train_' + k] = v.detach()[0].numpy()
    s = st.scene
    out['train_tsdf'] = db.scenes_est[s].volume.numpy().copy()
    out['train_wgt'] = db.fusion_weights[s].numpy().copy()
    return out


def run_pipeline_full_size(h, w, grid, frames, use_semantics, small):
    """Reference Pipeline.fuse at a BASELINE size (configs[1] / [2]: 320x240 -> 256^3).  Whole volumes would be
    134 MB per frame, so the fixture keeps, per frame, sha256 of the weight / id / score volumes (net-independent:
    a HIP PARITY-mode run must reproduce them bit for bit) and, for the last frame, the fp16 TSDF at the touched
    voxels (C order of ``wgt > 0``) plus the reference's volume metrics.  The net state is the small fixture's
    (same seed, asserted)."""
    from online_joint_depthfusion_and_semantic_amd.synthetic import gt_volumes
    sys.path.insert(0, REF)
    from utils import metrics as ref_metrics
    cfg = ref_config(h, w, True, use_semantics)
    pipe = RefPipeline(cfg)
    seeded_state(pipe._fusion_network, 11)
    pipe.eval()
    for k, v in pipe._fusion_network.state_dict().items():
        assert np.array_equal(v.numpy(), small['state_' + k]), k
    st = SyntheticStream(h, w, grid, 20)
    gt, _ = gt_volumes(grid)
    db = DuckDatabase(st, True, gt)
    s = st.scene
    out = {}
    with torch.no_grad():
        for i in range(frames):
            pipe.fuse(st.batch(i), db, torch.device('cpu'))
            for k, v in (('wgt', db.fusion_weights[s]), ('ids', db.ids_est[s].volume), ('scores', db.scores[s].volume),
                         ('tsdf', db.scenes_est[s].volume)):
                out['f%d_%s_sha256' % (i, k)] = np.array(sha(v.numpy()))
            out['f%d_touched' % i] = np.array(int((db.fusion_weights[s] > 0).sum()))
    tsdf, wgt = db.scenes_est[s].volume.numpy(), db.fusion_weights[s].numpy()
    out['last_tsdf_touched'] = tsdf[wgt > 0]
    ev = ref_metrics.evaluation(tsdf, gt, wgt > 0)  # utils/metrics.py:111-127 on the reference's own volumes
    for k, v in ev.items():
        out['metric_' + k] = np.array(float(v))
    return out


def run_training_full_size(h=240, w=320, grid=256):
    """BASELINE configs[3]'s frame step at its own size: ONE ``fuse_training`` call of the reference
    (modules/pipeline.py:251-363) at 320x240 -> 256^3, followed by ``loss.backward()``.

    Pre-frame state = two frames of the reference's Extractor + Integrator fed with the seeded stand-in of
    tests/helpers.py::frame_inputs (rng [7, i], uniform +-0.15) - a state the C oracle reproduces bit for bit on the GPU
    box (sha256 kept here), so the fixture does not depend on any net.  Then frame 2 through the reference's
    ``Pipeline.fuse_training`` with the seeded state_dict of the small fixtures, eval() mode (dropout and batch statistics
    cannot be pinned), and the gradients of  mean|fused - target| + 10 mean (fused - target)^2  (utils/loss.py's
    FusionLoss raises under this torch; any differentiable scalar pins the backward pass).
    Kept: every 7th row of tsdf_est / tsdf_fused + float64 column sums of all rows, sha256 of tsdf_target, the valid
    count, ALL parameter gradients (fp32, 1.4 MB), sha256 of the post-frame weight volume and the post-frame TSDF at
    the touched voxels."""
    from online_joint_depthfusion_and_semantic_amd.synthetic import gt_volumes
    cfg = ref_config(h, w, False, False)
    ex, ig = Extractor(cfg), Integrator(cfg)
    st = SyntheticStream(h, w, grid, 20)
    gt, _ = gt_volumes(grid)
    db = DuckDatabase(st, False, gt)
    s = st.scene
    origin = torch.from_numpy(st.origin)
    tsdf, wgt = db.scenes_est[s].volume, db.fusion_weights[s]
    for i in range(2):
        b = st.batch(i)
        depth = b['tof_depth']
        values = ex.forward(depth, b['extrinsics'], b['intrinsics'], tsdf, wgt, origin, st.resolution)
        rng = np.random.default_rng([7, i])
        est = torch.from_numpy(rng.uniform(-0.15, 0.15, (1, h * w, 9)).astype(np.float32))
        fd = torch.where(b['mask'], depth, torch.zeros_like(depth)).view(1, h * w, 1)
        valid = (fd != 0.).nonzero()[:, 1]
        updates = dict(values=torch.clamp(est[:, valid, :7], -0.1, 0.1), indices=values['indices'][:, valid, :7],
                       weights=values['weights'][:, valid, :7])
        tsdf, wgt, _, _ = ig.forward(updates, tsdf, wgt, None, None)
    db.scenes_est[s].volume, db.fusion_weights[s] = tsdf, wgt
    out = {'pre_tsdf_sha256': np.array(sha(tsdf.numpy())), 'pre_wgt_sha256': np.array(sha(wgt.numpy()))}
    pipe = RefPipeline(cfg)
    seeded_state(pipe._fusion_network, 11)
    small = np.load(os.path.join(HERE, 'pipeline_v3_nosem_24x32_g32.npz'))
    for k, v in pipe._fusion_network.state_dict().items():
        assert np.array_equal(v.numpy(), small['state_' + k]), k
    pipe.eval()
    o = pipe.fuse_training(st.batch(2), db, torch.device('cpu'))
    diff = o['tsdf_fused'] - o['tsdf_target']
    loss = diff.abs().mean() + 10 * (diff ** 2).mean()
    loss.backward()
    out['loss'] = np.array(float(loss))
    out['n_valid'] = np.array(int(o['tsdf_fused'].shape[1]))
    for k in ('tsdf_est', 'tsdf_fused'):
        a = o[k].detach()[0].numpy()
        out[k + '_rows7'] = a[::7].copy()
        out[k + '_colsum'] = a.astype(np.float64).sum(0)
        out[k + '_abssum'] = np.array(np.abs(a.astype(np.float64)).sum())
    out['tsdf_target_sha256'] = np.array(sha(o['tsdf_target'].detach()[0].numpy()))
    for name, p in pipe._fusion_network.named_parameters():
        out['grad_' + name] = p.grad.numpy().copy() if p.grad is not None else np.zeros(0, np.float32)
    tsdf, wgt = db.scenes_est[s].volume.numpy(), db.fusion_weights[s].numpy()
    out['post_wgt_sha256'] = np.array(sha(wgt))
    out['post_touched'] = np.array(int((wgt > 0).sum()))
    out['post_tsdf_touched'] = tsdf[wgt > 0]
    return out


def run_training_trajectory(h=48, w=64, grid=64, frames=8, accum=4, lr=1e-3, perturb=0.0):
    """A multi-step training trajectory of the reference: ``Pipeline.fuse_training`` driven the way train_fusion.py:145-189
    drives it - per frame loss.backward() and clip_grad_norm_(max_norm=1), every ``accum`` frames optimizer.step() /
    zero_grad() / scheduler.step() - for 2 windows x 4 frames, RMSprop(momentum 0.9, weight_decay 0.01, eps 1e-9) and
    PolynomialLR(max_iter 50000) of configs/fusion/replica_accuracy.yaml:30-48 (lr raised from 1e-5 to 1e-3 so that two
    steps move the output far beyond the comparison tolerance: a stale packed weight copy cannot hide).  eval() mode
    (dropout / batch statistics cannot be pinned), loss = mean|d| + 10 mean d^2 (utils/loss.py's FusionLoss raises under this
    torch).  Kept: per-frame loss and valid count, tsdf_est of the frame after the first step and of the last frame, ALL
    final parameters, the post-trajectory volumes.  ``perturb``: relative noise on every gradient (sensitivity probe only)."""
    from online_joint_depthfusion_and_semantic_amd.synthetic import gt_volumes
    from utils.schedulers import PolynomialLR
    cfg = ref_config(h, w, False, False)
    st = SyntheticStream(h, w, grid, 20)
    gt, _ = gt_volumes(grid)
    db = DuckDatabase(st, False, gt)
    s = st.scene
    pipe = RefPipeline(cfg)
    seeded_state(pipe._fusion_network, 11)
    pipe.eval()
    net = pipe._fusion_network
    opt = torch.optim.RMSprop(net.parameters(), lr=lr, momentum=0.9, weight_decay=0.01, eps=1e-9)
    sched = PolynomialLR(opt, 50000)
    out = {'lr': np.array(lr), 'accum': np.array(accum), 'frames': np.array(frames)}
    gen = torch.Generator().manual_seed(5)
    losses, nvalid = [], []
    for i in range(frames):
        o = pipe.fuse_training(st.batch(i), db, torch.device('cpu'))
        diff = o['tsdf_fused'] - o['tsdf_target']
        loss = diff.abs().mean() + 10 * (diff ** 2).mean()
        if loss.grad_fn:
            loss.backward()
        if perturb:
            for p in net.parameters():
                if p.grad is not None:
                    p.grad.mul_(1 + perturb * torch.randn(p.grad.shape, generator=gen))
        losses.append(float(loss))
        nvalid.append(int(o['tsdf_fused'].shape[1]))
        if i in (accum, frames - 1):
            out['f%d_tsdf_est' % i] = o['tsdf_est'].detach()[0].numpy().copy()
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1., norm_type=2)
        if (i + 1) % accum == 0 or i == frames - 1:
            opt.step()
            opt.zero_grad()
            sched.step()
    out['loss'] = np.array(losses, np.float64)
    out['n_valid'] = np.array(nvalid)
    for name, p in net.named_parameters():
        out['final_' + name] = p.detach().numpy().copy()
    out['post_tsdf'] = db.scenes_est[s].volume.numpy().copy()
    out['post_wgt'] = db.fusion_weights[s].numpy().copy()
    return out


def run_training_train_mode(h=48, w=64, grid=64):
    """ONE ``fuse_training`` frame of the reference with its FusionNet_v3 (modules/model.py:219-283) in train() mode - batch
    statistics in every BatchNorm2d, running-statistics update - and only the ``Dropout2d`` modules in eval() (their random
    stream cannot be pinned), followed by ``loss.backward()`` (every other training fixture is eval mode).

    A 46-layer net with batch statistics amplifies rounding differences: torch's own fp32 run deviates from exact
    arithmetic by up to 1e-2 of a gradient's scale.  So the fixture holds the reference twice from the same pre-frame state:
    as it runs (fp32), and with the SAME module tree converted to float64 (``net.double()``; the reference's glue around
    it - pipeline.py:104-135 - then runs in float64 by promotion).  A consumer is judged against the float64 values, with
    the reference's own fp32 deviation as the yardstick.  Pre-frame state: two frames of the reference's ``Pipeline.fuse``
    in eval() mode.  Kept: tsdf_est / tsdf_fused (all rows), sha256 of tsdf_target, loss, ALL parameter gradients, ALL
    BatchNorm buffers after the step - both precisions - and the state_dict."""
    import copy
    from online_joint_depthfusion_and_semantic_amd.synthetic import gt_volumes
    cfg = ref_config(h, w, False, False)
    st = SyntheticStream(h, w, grid, 20)
    gt, _ = gt_volumes(grid)
    pipe = RefPipeline(cfg)
    seeded_state(pipe._fusion_network, 11)
    out = {'state_' + k: v.numpy().copy() for k, v in pipe._fusion_network.state_dict().items()}
    db0 = DuckDatabase(st, False, gt)
    pipe.eval()
    with torch.no_grad():
        for i in range(2):
            pipe.fuse(st.batch(i), db0, torch.device('cpu'))
    s = st.scene
    out['pre_tsdf'] = db0.scenes_est[s].volume.numpy().copy()
    out['pre_wgt'] = db0.fusion_weights[s].numpy().copy()
    state0 = copy.deepcopy(pipe._fusion_network.state_dict())

    def one(double):
        db = copy.deepcopy(db0)
        net = pipe._fusion_network
        net.float()
        net.load_state_dict(state0)
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.eval()
        net.zero_grad()
        for p_ in net.parameters():
            p_.grad = None
        if double:
            net.double()
            fwd = net.forward
            net.forward = lambda x: fwd({k: v.double() for k, v in x.items()})
        try:
            o = pipe.fuse_training(st.batch(2), db, torch.device('cpu'))
        finally:
            if double:
                del net.forward
        diff = o['tsdf_fused'] - o['tsdf_target']
        loss = diff.abs().mean() + 10 * (diff ** 2).mean()
        loss.backward()
        tag = '64_' if double else '32_'
        r = {tag + 'loss': np.array(float(loss)), tag + 'n_valid': np.array(int(o['tsdf_fused'].shape[1]))}
        for k in ('tsdf_est', 'tsdf_fused'):
            r[tag + k] = o[k].detach()[0].numpy().copy()
        r[tag + 'tsdf_target_sha256'] = np.array(sha(o['tsdf_target'].detach()[0].float().numpy()))
        for name, p_ in net.named_parameters():
            r[tag + 'grad_' + name] = p_.grad.numpy().copy() if p_.grad is not None else np.zeros(0, np.float32)
        for name, b in net.named_buffers():
            r[tag + 'buf_' + name] = b.detach().numpy().copy()
        r[tag + 'post_wgt'] = db.fusion_weights[s].numpy().copy()
        r[tag + 'post_tsdf'] = db.scenes_est[s].volume.numpy().copy()
        return r
    out.update(one(False))
    out.update(one(True))
    # keep the file small: float64 arrays as they are (few MB compressed), volumes only once
    del out['64_post_wgt'], out['64_post_tsdf']
    return out


def probe_matmul():
    """Documents the fp32 accumulation order of the reference's two torch.matmul calls here."""
    from oracle import oracle
    st = SyntheticStream(120, 160, 64, 20)
    res = {}
    for hw in ((12, 16), (120, 160), (240, 320), (480, 640)):
        s2 = SyntheticStream(hw[0], hw[1], 64, 20)
        f = s2.frame(3)
        cfg = ref_config(hw[0], hw[1], False, False)
        ex = Extractor(cfg)
        b = s2.batch(3)
        pw = ex.compute_coordinates(b['tof_depth'], b['extrinsics'].float(), b['intrinsics'].float(), None, None)
        Ki, E = oracle.camera_arrays(f['intrinsics'], f['extrinsics'])
        tiny = np.full((2, 2, 2), 0.1, np.float16)
        o = oracle.extract(f['tof_depth'], Ki, E, s2.origin, s2.resolution, tiny, tiny, debug=True)
        res['%dx%d' % hw] = int((o['pcl'].view(np.uint32) != pw[0].numpy().view(np.uint32)).sum())
    return res


def full_size():
    """BASELINE configs[1] / [2] end to end through the reference's Pipeline.fuse (about 2 minutes, one thread)."""
    for tag, use_sem in (('sem', True), ('nosem', False)):
        small = np.load(os.path.join(HERE, 'pipeline_v3_%s_24x32_g32.npz' % tag))
        np.savez_compressed(os.path.join(HERE, 'pipeline_v3_%s_240x320_g256.npz' % tag),
                            **run_pipeline_full_size(240, 320, 256, 3, use_sem, small))


def main():
    if '--probe-matmul' in sys.argv:
        print(probe_matmul())
        return
    if '--full-size' in sys.argv:  # only the 320x240 -> 256^3 pipeline fixtures
        full_size()
        return
    if '--train-trajectory' in sys.argv:  # only the multi-step training fixture (round 4)
        np.savez_compressed(os.path.join(HERE, 'train_trajectory_v3_nosem_48x64_g64.npz'), **run_training_trajectory())
        return
    if '--train-trajectory-probe' in sys.argv:  # how far does fp32-level gradient noise move the trajectory?
        a, b = run_training_trajectory(), run_training_trajectory(perturb=1e-5)
        for k in a:
            if k.startswith('f') and k.endswith('est'):
                print(k, 'max |d|', float(np.abs(a[k] - b[k]).max()), 'scale', float(np.abs(a[k]).max()))
        worst = 0.0
        for k in a:
            if k.startswith('final_'):
                worst = max(worst, float(np.abs(a[k] - b[k]).max()))
        print('final parameters: max |d|', worst, 'loss', a['loss'], b['loss'])
        return
    if '--train-mode' in sys.argv:  # only the train()-mode frame step fixture (round 5)
        np.savez_compressed(os.path.join(HERE, 'train_mode_v3_nosem_48x64_g64.npz'), **run_training_train_mode())
        return
    if '--train-full-size' in sys.argv:  # only the configs[3] frame step at 320x240 -> 256^3
        np.savez_compressed(os.path.join(HERE, 'train_v3_nosem_240x320_g256.npz'), **run_training_full_size())
        return
    tiny = run_extract_integrate(12, 16, 32, 4, keep_arrays=True)
    np.savez_compressed(os.path.join(HERE, 'extract_integrate_12x16_g32.npz'), **tiny)
    digests = {'A_120x160_g64': run_extract_integrate(120, 160, 64, 3, keep_arrays=False),
               'B_240x320_g256': run_extract_integrate(240, 320, 256, 2, keep_arrays=False)}
    with open(os.path.join(HERE, 'extract_integrate_digests.json'), 'w') as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, 'pipeline_v3_sem_24x32_g32.npz'), **run_pipeline(24, 32, 32, 3, True))
    np.savez_compressed(os.path.join(HERE, 'pipeline_v3_nosem_24x32_g32.npz'), **run_pipeline(24, 32, 32, 3, False))
    full_size()
    np.savez_compressed(os.path.join(HERE, 'train_v3_nosem_240x320_g256.npz'), **run_training_full_size())
    np.savez_compressed(os.path.join(HERE, 'train_trajectory_v3_nosem_48x64_g64.npz'), **run_training_trajectory())
    np.savez_compressed(os.path.join(HERE, 'train_mode_v3_nosem_48x64_g64.npz'), **run_training_train_mode())
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
