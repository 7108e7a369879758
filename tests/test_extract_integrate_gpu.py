"""-m gpu parity tests proper: HIP extract / integrate through the C ABI vs the CPU oracle on the
same seeded inputs.  Bars: indices, corner weights, fusion_values/weights bit-exact;
PARITY-mode volumes bit-exact; FAST-mode TSDF/weights <= 1 fp16 ulp on <= 0.05 % (256^3) / 0.1 % (coarse grids) of touched voxels
from a common pre-frame state; semantic ids / scores bit-exact in both modes."""
import numpy as np
import pytest
import torch

from oracle import oracle
from online_joint_depthfusion_and_semantic_amd import ops
from helpers import (bits, n_mismatch, f16_ulp_distance, fresh_volumes, frame_inputs, to_cuda, make_stream)

pytestmark = pytest.mark.gpu

CASES = [(12, 16, 32, 4), (120, 160, 64, 4), (240, 320, 256, 3)]
# FAST-mode budget: <= 1 fp16 ulp on a small fraction of the touched voxels per frame from a common pre-frame state.
# Measured: TSDF 0.034 % at 256^3 (53 of 155 780), 0.06 % at 64^3, <= 1 voxel at 32^3; weights 0.036 % at 64^3.  (SURVEY.md
# §8c hoped for < 0.01 %: the exact fixed-point sum rounds once where the reference rounds after each of the ~10 fp32
# adds of a voxel, so P(move) ~ n * 2^-24 / 2^-11.)  Bars = measured + margin.
FAST_MOVED_FRACTION = {32: 1e-3, 64: 1e-3, 256: 5e-4}


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize('h,w,grid,frames', CASES)
def test_extract_bit_exact(cuda, h, w, grid, frames):
    st = make_stream(h, w, grid)
    rng = np.random.default_rng(3)
    tsdf = rng.uniform(-0.1, 0.1, (grid,) * 3).astype(np.float16)
    wgt = rng.uniform(0, 6, (grid,) * 3).astype(np.float16)
    g_tsdf, g_wgt = _t(tsdf, cuda), _t(wgt, cuda)
    debug = h * w <= 19200
    for i in range(frames):
        fi = frame_inputs(st, i)
        ref = oracle.extract(fi['depth'], fi['Ki'], fi['E'], st.origin, st.resolution, tsdf, wgt, debug=debug)
        out = ops.extract(_t(fi['depth'], cuda), fi['Ki'], fi['E'], st.origin, st.resolution, g_tsdf, g_wgt, debug=debug)
        for key in ref:
            assert n_mismatch(out[key].cpu().numpy(), ref[key]) == 0, (key, i)


def test_extract_out_of_volume_and_zero_depth(cuda):
    # a grid much smaller than the room: most rays leave the volume (pad value -0.1 / weight 0),
    # zeroed pixels collapse onto the camera centre
    st = make_stream(24, 32, 16)
    st.resolution = 0.08  # 16^3 * 8 cm = 1.28 m cube around the origin corner
    rng = np.random.default_rng(5)
    tsdf = rng.uniform(-0.1, 0.1, (16,) * 3).astype(np.float16)
    wgt = rng.uniform(0, 3, (16,) * 3).astype(np.float16)
    fi = frame_inputs(st, 2)
    fi['depth'][::3, ::2] = 0.0
    ref = oracle.extract(fi['depth'], fi['Ki'], fi['E'], st.origin, st.resolution, tsdf, wgt, debug=True)
    out = ops.extract(_t(fi['depth'], cuda), fi['Ki'], fi['E'], st.origin, st.resolution, _t(tsdf, cuda), _t(wgt, cuda), debug=True)
    for key in ref:
        assert n_mismatch(out[key].cpu().numpy(), ref[key]) == 0, key
    assert (ref['fusion_weights'] == 0).any() and np.isclose(ref['fusion_values'], -0.1).any()


def _run_integrate(st, fi, vols_gpu, ws, mode, semantics, cuda):
    kw = {}
    if semantics:
        kw = dict(sem_ids=_t(fi['sem_ids'].reshape(-1), cuda), sem_scores=_t(fi['sem_scores'].reshape(-1), cuda),
                  id_vol=vols_gpu['ids'], score_vol=vols_gpu['scores'])
    ops.integrate(_t(fi['fd'], cuda), fi['Ki'], fi['E'], st.origin, st.resolution, _t(fi['est'], cuda),
                  vols_gpu['tsdf'], vols_gpu['wgt'], ws, mode=mode, stats=True, **kw)


def _oracle_integrate(st, fi, vols, semantics):
    kw = {}
    if semantics:
        kw = dict(sem_ids=fi['sem_ids'], sem_scores=fi['sem_scores'], id_vol=vols['ids'], score_vol=vols['scores'])
    return oracle.integrate(fi['fd'], fi['Ki'], fi['E'], st.origin, st.resolution, fi['est'], vols['tsdf'], vols['wgt'], **kw)


@pytest.mark.parametrize('semantics', [False, True])
@pytest.mark.parametrize('h,w,grid,frames', CASES)
def test_integrate_parity_mode_bit_exact(cuda, h, w, grid, frames, semantics):
    st = make_stream(h, w, grid)
    vols = fresh_volumes(grid, semantics)
    g = to_cuda(vols, cuda)
    ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, ops.MODE_PARITY, cuda)
    for i in range(frames):
        fi = frame_inputs(st, i)
        touched = _oracle_integrate(st, fi, vols, semantics)
        _run_integrate(st, fi, g, ws, ops.MODE_PARITY, semantics, cuda)  # state carried on both sides
        for key in vols:
            assert n_mismatch(g[key].cpu().numpy(), vols[key]) == 0, (key, i)
        assert int(ws.stats[0].item()) == touched


@pytest.mark.parametrize('semantics', [False, True])
@pytest.mark.parametrize('h,w,grid,frames', CASES)
def test_integrate_fast_mode_tolerance(cuda, h, w, grid, frames, semantics):
    st = make_stream(h, w, grid)
    vols = fresh_volumes(grid, semantics)
    ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, ops.MODE_FAST, cuda)
    for i in range(frames):
        fi = frame_inputs(st, i)
        g = to_cuda(vols, cuda)  # common pre-frame state
        touched = _oracle_integrate(st, fi, vols, semantics)
        _run_integrate(st, fi, g, ws, ops.MODE_FAST, semantics, cuda)
        assert int(ws.stats[0].item()) == touched
        for key in ('tsdf', 'wgt'):
            got = g[key].cpu().numpy()
            ulp = f16_ulp_distance(got, vols[key])
            nan_mismatch = np.isnan(got) != np.isnan(vols[key])
            assert not nan_mismatch.any(), (key, i)
            ulp = np.where(np.isnan(got), 0, ulp)
            assert ulp.max() <= 1, (key, i, int(ulp.max()))
            print('fast-mode %s %dx%d->%d^3 frame %d: %d of %d touched voxels moved by one ulp' % (key, w, h, grid, i, int((ulp > 0).sum()), touched))
            assert (ulp > 0).sum() <= max(2, FAST_MOVED_FRACTION[grid] * touched), (key, i, int((ulp > 0).sum()), touched)
        if semantics:
            assert n_mismatch(g['ids'].cpu().numpy(), vols['ids']) == 0, i
            assert n_mismatch(g['scores'].cpu().numpy(), vols['scores']) == 0, i
    # the workspace must be left clean: an all-masked frame touches nothing
    fi = frame_inputs(st, 0)
    fi['fd'][:] = 0
    g = to_cuda(vols, cuda)
    _run_integrate(st, fi, g, ws, ops.MODE_FAST, semantics, cuda)
    assert int(ws.stats[0].item()) == 0
    for key in vols:
        assert n_mismatch(g[key].cpu().numpy(), vols[key]) == 0, key


def test_integrate_fast_is_deterministic(cuda):
    h, w, grid = 120, 160, 64
    st = make_stream(h, w, grid)
    ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, ops.MODE_FAST, cuda)
    outs = []
    for rep in range(3):
        g = to_cuda(fresh_volumes(grid, True), cuda)
        for i in range(3):
            _run_integrate(st, frame_inputs(st, i), g, ws, ops.MODE_FAST, True, cuda)
        outs.append({k: v.cpu().numpy() for k, v in g.items()})
    for key in outs[0]:
        assert n_mismatch(outs[0][key], outs[1][key]) == 0 and n_mismatch(outs[0][key], outs[2][key]) == 0, key


_DIGEST_SCRIPT = r"""
import hashlib, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
from online_joint_depthfusion_and_semantic_amd import ops
from helpers import fresh_volumes, frame_inputs, to_cuda, make_stream
import numpy as np
dev = torch.device('cuda:0')
h, w, grid = 120, 160, 64
st = make_stream(h, w, grid)
ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, ops.MODE_FAST, dev)
g = to_cuda(fresh_volumes(grid, True), dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for i in range(3):
    fi = frame_inputs(st, i)
    ops.integrate(t(fi['fd']), fi['Ki'], fi['E'], st.origin, st.resolution, t(fi['est']), g['tsdf'], g['wgt'], ws, mode=ops.MODE_FAST,
                  sem_ids=t(fi['sem_ids'].reshape(-1)), sem_scores=t(fi['sem_scores'].reshape(-1)), id_vol=g['ids'], score_vol=g['scores'])
torch.cuda.synchronize()
print('DIGEST', ' '.join(hashlib.sha256(g[k].cpu().numpy().tobytes()).hexdigest() for k in sorted(g)))
"""


def test_wave_combined_accumulate_gives_the_same_bits(cuda):
    # OJF_INTEGRATE_WAVE_COMBINE=1 (colliding writes combined inside the wave before the LDS hash: measured slower, off by
    # default) is read once per process: two child processes, same frames, same volume digests
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for on in (False, True):
        env = dict(os.environ)
        env.pop('OJF_INTEGRATE_WAVE_COMBINE', None)
        if on:
            env['OJF_INTEGRATE_WAVE_COMBINE'] = '1'
        out = subprocess.run([sys.executable, '-c', _DIGEST_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith('DIGEST')]
        assert out.returncode == 0 and line, out.stderr[-2000:]
        digests.append(line[0])
    assert digests[0] == digests[1]


def test_known_answer_constant_update_on_empty_volume(cuda):
    # constant est == v on an empty volume (w_old = 0) gives TSDF = fp16(v) and
    # weight = fp16(sum of corner weights) at every touched voxel; untouched voxels keep the init value
    h, w, grid = 60, 80, 64
    st = make_stream(h, w, grid)
    fi = frame_inputs(st, 1)
    fi['est'][:] = 0.05
    for mode in (ops.MODE_FAST, ops.MODE_PARITY):
        g = to_cuda(fresh_volumes(grid, False), cuda)
        ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, mode, cuda)
        _run_integrate(st, fi, g, ws, mode, False, cuda)
        tsdf, wgt = g['tsdf'].cpu().numpy(), g['wgt'].cpu().numpy()
        touched = wgt > 0
        assert touched.sum() > 1000
        expect = np.float16(np.float32(0.05))
        near = f16_ulp_distance(tsdf[touched], np.full(touched.sum(), expect, np.float16))
        assert near.max() <= 1
        zero_w = (wgt == 0)
        ok = (tsdf[zero_w] == np.float16(0.1)) | np.isnan(tsdf[zero_w])
        assert ok.all()


def test_error_paths(cuda):
    st = make_stream(12, 16, 32)
    fi = frame_inputs(st, 0)
    vols = to_cuda(fresh_volumes(32, False), cuda)
    with pytest.raises(Exception):
        ops.extract(_t(fi['depth'], cuda), fi['Ki'], fi['E'], st.origin, st.resolution, vols['tsdf'], vols['wgt'], n_points=8)
    with pytest.raises(Exception):
        ops.extract(_t(fi['depth'], cuda), fi['Ki'], fi['E'], st.origin, -1.0, vols['tsdf'], vols['wgt'])
    ws = ops.IntegrateWorkspace((32,) * 3, 12, 16, 7, ops.MODE_FAST, cuda)
    with pytest.raises(AssertionError):  # workspace built for another frame size
        ops.integrate(_t(fi['fd'][:6], cuda), fi['Ki'], fi['E'], st.origin, st.resolution, _t(fi['est'][:96], cuda),
                      vols['tsdf'], vols['wgt'], ws)


@pytest.mark.parametrize('semantics', [False, True])
def test_reference_style_extractor_integrator_modules(cuda, semantics):
    """The drop-in Extractor / Integrator modules with the reference's own call signatures
    (extractor.py:24, integrator.py:15): Extractor.forward's full dict feeds the reference-style
    _prepare_volume_update slicing (pipeline.py:137-171) and Integrator.forward(updates, ...)."""
    from online_joint_depthfusion_and_semantic_amd.config import default_config
    from online_joint_depthfusion_and_semantic_amd.extractor import Extractor
    from online_joint_depthfusion_and_semantic_amd.integrator import Integrator
    h, w, grid = 60, 80, 64
    cfg = default_config(h, w, semantics=semantics)
    cfg.SETTINGS.device = str(cuda)
    ex, ig = Extractor(cfg), Integrator(cfg)
    st = make_stream(h, w, grid)
    vols = fresh_volumes(grid, semantics)
    g = to_cuda(vols, cuda)
    for i in range(3):
        b = st.batch(i)
        fi = frame_inputs(st, i)
        out = ex.forward(b['tof_depth'].to(cuda), b['extrinsics'], b['intrinsics'], g['tsdf'], g['wgt'],
                         torch.from_numpy(st.origin), st.resolution)
        ref = oracle.extract(fi['depth'], fi['Ki'], fi['E'], st.origin, st.resolution, vols['tsdf'], vols['wgt'], debug=True)
        assert n_mismatch(out['fusion_values'][0].cpu().numpy(), ref['fusion_values']) == 0
        assert n_mismatch(out['indices'][0].cpu().numpy(), ref['indices']) == 0
        assert n_mismatch(out['weights'][0].cpu().numpy(), ref['weights']) == 0
        est = _t(fi['est'], cuda).view(1, h * w, 9)
        valid = (_t(fi['fd'], cuda).view(1, h * w, 1) != 0).nonzero()[:, 1]
        updates = dict(values=torch.clamp(est[:, valid, :7], -0.1, 0.1), indices=out['indices'][:, valid, :7],
                       weights=out['weights'][:, valid, :7])
        if semantics:
            rep = lambda t: t.view(1, h * w, 1).unsqueeze(-2).repeat(1, 1, 9, 1)[:, valid, :7]
            updates['semantics'] = rep(_t(fi['sem_ids'], cuda))
            updates['scores'] = rep(_t(fi['sem_scores'], cuda))
        pre = {k: v.copy() for k, v in vols.items()}
        touched = _oracle_integrate(st, fi, vols, semantics)
        gg = to_cuda(pre, cuda)  # common pre-frame state
        r = ig.forward(updates, gg['tsdf'], gg['wgt'], gg.get('scores'), gg.get('ids'))
        assert r[0] is gg['tsdf'] and r[1] is gg['wgt']
        assert int(ig._entry_ws.stats[0].item()) == touched
        for key in ('tsdf', 'wgt'):
            got = gg[key].cpu().numpy()
            ulp = np.where(np.isnan(got), 0, f16_ulp_distance(got, vols[key]))
            assert ulp.max() <= 1 and (np.isnan(got) == np.isnan(vols[key])).all(), (key, i)
        if semantics:
            assert n_mismatch(gg['ids'].cpu().numpy(), vols['ids']) == 0
            assert n_mismatch(gg['scores'].cpu().numpy(), vols['scores']) == 0
        g = to_cuda(vols, cuda)


def test_extract_sample_plane_layout(cuda):
    """out_layout 1 (sample planes, the layout Pipeline.fuse uses) holds the same bits as the row layout."""
    st = make_stream(24, 32, 32)
    rng = np.random.default_rng(9)
    tsdf = _t(rng.uniform(-0.1, 0.1, (32,) * 3).astype(np.float16), cuda)
    wgt = _t(rng.uniform(0, 4, (32,) * 3).astype(np.float16), cuda)
    fi = frame_inputs(st, 1)
    d = _t(fi['depth'], cuda)
    rows = ops.extract(d, fi['Ki'], fi['E'], st.origin, st.resolution, tsdf, wgt)
    pl = ops.extract(d, fi['Ki'], fi['E'], st.origin, st.resolution, tsdf, wgt, planes=True)
    assert pl['fusion_values'].shape == (9, 24 * 32)
    assert torch.equal(pl['fusion_values'].t().contiguous(), rows['fusion_values'])
    assert torch.equal(pl['fusion_weights'].t().contiguous(), rows['fusion_weights'])


@pytest.mark.parametrize('h,w,n_points', [(13, 15, 9), (13, 15, 1), (13, 15, 17), (7, 9, 5)])
def test_extract_ragged_sizes_and_sample_counts(cuda, h, w, n_points):
    """Frames whose pixel count is not a multiple of the 64-pixel block, and ray sample counts on both sides of
    the tile kernel's limit (16; above it the one-lane-per-item kernel runs): bit-exact like the rest."""
    st = make_stream(h, w, 32)
    rng = np.random.default_rng(9)
    tsdf = rng.uniform(-0.1, 0.1, (32,) * 3).astype(np.float16)
    wgt = rng.uniform(0, 6, (32,) * 3).astype(np.float16)
    fi = frame_inputs(st, 1)
    fi['depth'][2, 3] = 0.0
    ref = oracle.extract(fi['depth'], fi['Ki'], fi['E'], st.origin, st.resolution, tsdf, wgt, n_points=n_points, debug=True)
    out = ops.extract(_t(fi['depth'], cuda), fi['Ki'], fi['E'], st.origin, st.resolution, _t(tsdf, cuda), _t(wgt, cuda),
                      n_points=n_points, debug=True)
    for key in ref:
        assert n_mismatch(out[key].cpu().numpy(), ref[key]) == 0, key
    assert out['fusion_values'].shape[-1] == n_points


def test_integrate_ragged_frame(cuda):
    """A frame that ends inside a 64-pixel block and inside 8x8 tiles on both axes (13x15)."""
    h, w, grid = 13, 15, 32
    st = make_stream(h, w, grid)
    vols = fresh_volumes(grid, False)
    ws = {m: ops.IntegrateWorkspace((grid,) * 3, h, w, 7, m, cuda) for m in (ops.MODE_FAST, ops.MODE_PARITY)}
    for mode in (ops.MODE_PARITY, ops.MODE_FAST):
        ref = {k: v.copy() for k, v in vols.items()}
        g = to_cuda(vols, cuda)
        for i in range(3):
            fi = frame_inputs(st, i)
            touched = _oracle_integrate(st, fi, ref, False)
            _run_integrate(st, fi, g, ws[mode], mode, False, cuda)
            assert int(ws[mode].stats[0].item()) == touched
        for key in ref:
            ulp = f16_ulp_distance(g[key].cpu().numpy(), ref[key])
            assert ulp.max() <= (0 if mode == ops.MODE_PARITY else 1), (mode, key)


def test_full_size_properties_config_C(cuda):
    """BASELINE configs[4] size (640x480 depth into a 512^3 grid), where the scalar oracle would take minutes: the
    size-independent properties instead.  (1) extract of an empty volume returns exactly (init value, 0) for
    every in-volume sample; (2) a constant update on the empty volume gives TSDF = fp16(v) at every touched voxel
    and leaves the rest alone; (3) FAST and PARITY agree to one fp16 ulp; (4) the call is deterministic;
    (5) extracting the fused volume back along the same rays returns v (within the interpolation of touched and
    untouched corners: never outside [min(v, init), max(v, init)])."""
    h, w, grid = 480, 640, 512
    st = make_stream(h, w, grid)
    fi = frame_inputs(st, 2)
    v = np.float32(-0.03)
    est = np.full((h * w, 9), v, np.float32)
    depth, fd = _t(fi['depth'], cuda), _t(fi['fd'], cuda)
    outs = {}
    for mode in (ops.MODE_FAST, ops.MODE_PARITY, ops.MODE_FAST):
        tsdf = torch.full((grid,) * 3, 0.1, dtype=torch.float16, device=cuda)
        wgt = torch.zeros((grid,) * 3, dtype=torch.float16, device=cuda)
        if not outs:
            ex = ops.extract(depth, fi['Ki'], fi['E'], st.origin, st.resolution, tsdf, wgt)
            fv, fw = ex['fusion_values'].cpu().numpy(), ex['fusion_weights'].cpu().numpy()
            assert np.all(fw == 0) and np.all((fv == np.float32(np.float16(0.1))) | (fv < np.float32(0.1)))
        ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, mode, cuda)
        ops.integrate(fd, fi['Ki'], fi['E'], st.origin, st.resolution, _t(est, cuda), tsdf, wgt, ws, mode=mode, stats=True)
        touched = wgt > 0
        n_touched = int(touched.sum())
        assert n_touched == int(ws.stats[0].item()) and n_touched > 200000
        t = tsdf[touched].float().cpu().numpy()
        assert f16_ulp_distance(t.astype(np.float16), np.full(t.shape, np.float16(v), np.float16)).max() <= 1
        rest = tsdf[~touched]
        assert bool(((rest == 0.1) | torch.isnan(rest)).all())
        outs.setdefault(mode, []).append((tsdf.cpu().numpy(), wgt.cpu().numpy()))
        del ws
    (t0, w0), (t1, w1) = outs[ops.MODE_FAST]
    assert n_mismatch(t0, t1) == 0 and n_mismatch(w0, w1) == 0  # deterministic
    tp, wp = outs[ops.MODE_PARITY][0]
    assert f16_ulp_distance(t0, tp).max() <= 1 and f16_ulp_distance(w0, wp).max() <= 1
    ex = ops.extract(depth, fi['Ki'], fi['E'], st.origin, st.resolution, _t(tp, cuda), _t(wp, cuda))
    fv = ex['fusion_values'].cpu().numpy()
    valid = fi['fd'].reshape(-1) != 0
    mid = fv[valid][:, 1:8]  # the seven integrated samples of valid rays
    assert mid.min() >= float(np.float16(v)) - 1e-3 and mid.max() <= 0.1 + 1e-3
    assert np.median(mid) < 0.0  # most of what the rays see is the fused band


@pytest.mark.parametrize('semantics', [False, True])
def test_integrate_hash_overflow_path(cuda, semantics):
    """Coarse frame into a fine grid with 15 integrated samples per ray (48x64 rays into 256^3: ~4000 distinct voxels
    per 8x8 tile against 2048 hash slots): the entries that find their tile's hash full take the counter-allocated
    record path behind the tile slices.  Same bars as everywhere else."""
    h, w, grid, P, T = 48, 64, 256, 17, 15
    st = make_stream(h, w, grid)
    vols = fresh_volumes(grid, semantics)
    tiles = (h // 8) * (w // 8)
    for mode in (ops.MODE_PARITY, ops.MODE_FAST):
        ref = {k: v.copy() for k, v in vols.items()}
        g = to_cuda(vols, cuda)
        ws = ops.IntegrateWorkspace((grid,) * 3, h, w, T, mode, cuda)
        for i in range(2):
            fi = frame_inputs(st, i, n_points=P)
            if mode == ops.MODE_FAST:
                g = to_cuda(ref, cuda)  # FAST is specified per frame from a common pre-frame state
            kw_o, kw_g = {}, {}
            if semantics:
                kw_o = dict(sem_ids=fi['sem_ids'], sem_scores=fi['sem_scores'], id_vol=ref['ids'], score_vol=ref['scores'])
                kw_g = dict(sem_ids=_t(fi['sem_ids'].reshape(-1), cuda), sem_scores=_t(fi['sem_scores'].reshape(-1), cuda),
                            id_vol=g['ids'], score_vol=g['scores'])
            touched = oracle.integrate(fi['fd'], fi['Ki'], fi['E'], st.origin, st.resolution, fi['est'], ref['tsdf'], ref['wgt'],
                                       n_points=P, n_tail=T, **kw_o)
            ops.integrate(_t(fi['fd'], cuda), fi['Ki'], fi['E'], st.origin, st.resolution, _t(fi['est'], cuda), g['tsdf'], g['wgt'],
                          ws, n_points=P, n_tail=T, mode=mode, stats=True, **kw_g)
            assert int(ws.stats[0].item()) == touched
            if mode == ops.MODE_FAST:
                assert int(ws.stats[2].item()) > tiles * 2048  # more records than the tile slices alone can hold
            for key in ('tsdf', 'wgt'):
                ulp = f16_ulp_distance(g[key].cpu().numpy(), ref[key])
                assert ulp.max() <= (0 if mode == ops.MODE_PARITY else 1), (mode, key, i)
            if semantics:
                assert n_mismatch(g['ids'].cpu().numpy(), ref['ids']) == 0
                assert n_mismatch(g['scores'].cpu().numpy(), ref['scores']) == 0


@pytest.mark.parametrize('mode', [ops.MODE_FAST, ops.MODE_PARITY])
@pytest.mark.parametrize('semantics', [False, True])
def test_masked_integrate_equals_filtered_frame(cuda, mode, semantics):
    """ojf_integrate_masked (raw frame + validity mask, pipeline.py:196 formed inside the kernels) is bit-identical to
    ojf_integrate on torch.where(mask, frame, 0), in both modes, over several frames."""
    h, w, grid = 29, 37, 32
    st = make_stream(h, w, grid)
    vols = fresh_volumes(grid, semantics)
    a, b = to_cuda(vols, cuda), to_cuda(vols, cuda)
    ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, mode, cuda)
    rng = np.random.default_rng(4)
    for i in range(3):
        fi = frame_inputs(st, i)
        raw = fi['depth'].astype(np.float32).copy()
        mask = rng.random((h, w)) > 0.3
        raw[~mask & (rng.random((h, w)) > 0.5)] = np.nan  # what the mask hides may be anything
        filt = np.where(mask, raw, np.float32(0)).astype(np.float32)
        kw_a, kw_b = {}, {}
        if semantics:
            ids, sc = _t(fi['sem_ids'].reshape(-1), cuda), _t(fi['sem_scores'].reshape(-1), cuda)
            kw_a = dict(sem_ids=ids, sem_scores=sc, id_vol=a['ids'], score_vol=a['scores'])
            kw_b = dict(sem_ids=ids, sem_scores=sc, id_vol=b['ids'], score_vol=b['scores'])
        est = _t(fi['est'], cuda)
        ops.integrate(_t(filt, cuda), fi['Ki'], fi['E'], st.origin, st.resolution, est, a['tsdf'], a['wgt'], ws, mode=mode, **kw_a)
        ops.integrate(_t(raw, cuda), fi['Ki'], fi['E'], st.origin, st.resolution, est, b['tsdf'], b['wgt'], ws, mode=mode,
                      mask=torch.from_numpy(mask).to(cuda), **kw_b)
    assert int((a['wgt'].float() > 0).sum()) > 200
    for key in a:
        x, y = a[key], b[key]
        assert torch.equal(x.view(torch.uint8) if x.dtype != torch.uint8 else x, y.view(torch.uint8) if y.dtype != torch.uint8 else y), key


def test_fast_mode_weight_overflow_saturates_like_the_reference(cuda):
    """a degenerate frame (voxels far larger than the scene: every ray lands in the same few voxels) pushes a
    voxel's per-frame weight beyond the 19 integer bits of the 2^-44 fixed point.  The 64-bit sums wrap; finalize
    detects it and redoes them in fp64, so the weight saturates to fp16 infinity exactly where the reference's does
    and the TSDF stays a weighted mean."""
    h, w, grid, P, T = 480, 640, 4, 17, 15
    st = make_stream(h, w, grid)
    st.resolution = 40.0                      # 4^3 voxels of 40 m ...
    st.origin = np.array([-60.0, -60.0, -60.0])  # ... with the whole room at the centre of voxel (1, 1, 1): corner weights ~1
    vols = fresh_volumes(grid, False)
    fi = frame_inputs(st, 1, n_points=P)
    ref = {k: v.copy() for k, v in vols.items()}
    oracle.integrate(fi['fd'], fi['Ki'], fi['E'], st.origin, st.resolution, fi['est'], ref['tsdf'], ref['wgt'], n_points=P, n_tail=T)
    assert np.isinf(ref['wgt'].astype(np.float32)).any()  # 4.6e6 entries of weight ~0.9 on one voxel: beyond 2^19 and beyond fp16
    g = to_cuda(vols, cuda)
    ws = ops.IntegrateWorkspace((grid,) * 3, h, w, T, ops.MODE_FAST, cuda)
    ops.integrate(_t(fi['fd'], cuda), fi['Ki'], fi['E'], st.origin, st.resolution, _t(fi['est'], cuda), g['tsdf'], g['wgt'], ws,
                  n_points=P, n_tail=T, mode=ops.MODE_FAST)
    got_w, got_t = g['wgt'].cpu().numpy(), g['tsdf'].cpu().numpy()
    assert (np.isinf(got_w.astype(np.float32)) == np.isinf(ref['wgt'].astype(np.float32))).all()
    fin = np.isfinite(ref['wgt'].astype(np.float32))
    assert f16_ulp_distance(got_w[fin], ref['wgt'][fin]).max() <= 1
    touched = ref['wgt'] > 0
    # the reference's sequential fp32 sum of 5e5 terms carries ~1e-3 relative error of its own; ours is the exact sum
    assert np.abs(got_t[touched].astype(np.float32) - ref['tsdf'][touched].astype(np.float32)).max() <= 2e-3


def test_fast_integrate_inside_a_graph(cuda):
    """include/ojf.h: OJF_MODE_FAST alternates two counter sets per call; the phase lives in the workspace header (device
    state, round 4), so a captured call replays correctly - here with hash-full frames (a tiny grid: every tile overflows
    its LDS hash into the counter-allocated lists, the part the alternation exists for) against direct calls."""
    h, w, grid = 24, 32, 16
    st = make_stream(h, w, grid)
    fi = frame_inputs(st, 0)
    fd, est = _t(fi['fd'], cuda), _t(fi['est'], cuda)
    vols = [to_cuda(fresh_volumes(grid, False), cuda) for _ in range(2)]
    wss = [ops.IntegrateWorkspace((grid,) * 3, h, w, 7, ops.MODE_FAST, cuda) for _ in range(2)]
    run = lambda k: ops.integrate(fd, fi['Ki'], fi['E'], st.origin, st.resolution, est, vols[k]['tsdf'], vols[k]['wgt'], wss[k], mode=ops.MODE_FAST)
    for k in range(2):
        run(k)  # (warm-up outside the capture; both sides have integrated the frame once)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run(0)
    for _ in range(3):  # an odd number of replays: both counter phases are exercised
        graph.replay()
        run(1)
    torch.cuda.synchronize()
    # (the capture itself does not execute: side 0 ran 1 + 3 times, side 1 ran 1 + 3 times)
    for key in ('tsdf', 'wgt'):
        assert n_mismatch(vols[0][key].cpu().numpy(), vols[1][key].cpu().numpy()) == 0, key


# ---- round 6: the frames of several scenes as single launches (ojf_extract_many / ojf_integrate_many) -----------------------
@pytest.mark.parametrize('semantics', [False, True])
@pytest.mark.parametrize('h,w,grid,S', [(60, 80, 64, 3), (13, 15, 32, 2), (240, 320, 128, 4), (48, 64, 64, 8)])
def test_many_scene_launches_equal_the_separate_calls(cuda, h, w, grid, S, semantics):
    """modules/extractor.py:24-79 / modules/integrator.py:15-124 for S scenes per launch (blockIdx.y = scene): sample planes,
    TSDF / weight / id / score volumes bit for bit those of S separate ojf_extract / ojf_integrate_masked calls, over three
    frames per scene from different pre-frame states (scene s starts s frames into its stream), with masks."""
    n_tail = 7
    streams = [make_stream(h, w, grid, scene='room_%d' % s, seed=1911 + 17 * s) for s in range(S)]
    one = [to_cuda(fresh_volumes(grid, semantics), cuda) for _ in range(S)]
    many = [to_cuda(fresh_volumes(grid, semantics), cuda) for _ in range(S)]
    ws_one = ops.IntegrateWorkspace((grid,) * 3, h, w, n_tail, ops.MODE_FAST, cuda)
    ws_many = [ops.IntegrateWorkspace((grid,) * 3, h, w, n_tail, ops.MODE_FAST, cuda) for _ in range(S)]
    for step in range(3):
        fis = [frame_inputs(st, step + s) for s, st in enumerate(streams)]
        depth = [_t(fi['depth'], cuda) for fi in fis]
        mask = [_t(fi['fd'] != 0, cuda) for fi in fis]
        est = [_t(fi['est'], cuda) for fi in fis]
        ids = [_t(fi['sem_ids'].reshape(-1), cuda) for fi in fis]
        sc = [_t(fi['sem_scores'].reshape(-1), cuda) for fi in fis]
        # gather
        ref = [ops.extract(depth[s], fis[s]['Ki'], fis[s]['E'], streams[s].origin, streams[s].resolution, one[s]['tsdf'], one[s]['wgt'], planes=True)
               for s in range(S)]
        outs = [(torch.empty((9, h * w), device=cuda), torch.empty((9, h * w), device=cuda)) for _ in range(S)]
        ops.extract_many([dict(depth=depth[s], Ki=fis[s]['Ki'], E=fis[s]['E'], origin=streams[s].origin, resolution=streams[s].resolution,
                               tsdf=many[s]['tsdf'], weights=many[s]['wgt'], out_values=outs[s][0], out_weights=outs[s][1]) for s in range(S)])
        for s in range(S):
            assert torch.equal(ref[s]['fusion_values'].view(torch.int32), outs[s][0].view(torch.int32)), (step, s)
            assert torch.equal(ref[s]['fusion_weights'].view(torch.int32), outs[s][1].view(torch.int32)), (step, s)
        # scatter
        for s in range(S):
            kw = dict(sem_ids=ids[s], sem_scores=sc[s], id_vol=one[s]['ids'], score_vol=one[s]['scores']) if semantics else {}
            ops.integrate(depth[s], fis[s]['Ki'], fis[s]['E'], streams[s].origin, streams[s].resolution, est[s], one[s]['tsdf'], one[s]['wgt'],
                          ws_one, n_tail=n_tail, mask=mask[s], **kw)
        ops.integrate_many([dict(depth=depth[s], mask=mask[s], Ki=fis[s]['Ki'], E=fis[s]['E'], origin=streams[s].origin,
                                 resolution=streams[s].resolution, est=est[s], tsdf=many[s]['tsdf'], weights=many[s]['wgt'], workspace=ws_many[s],
                                 **(dict(sem_ids=ids[s], sem_scores=sc[s], id_vol=many[s]['ids'], score_vol=many[s]['scores']) if semantics else {}))
                            for s in range(S)], n_tail=n_tail)
        for s in range(S):
            for k in one[s]:
                a, b = one[s][k], many[s][k]
                assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), (step, s, k)
    assert float((many[0]['wgt'].float() > 0).sum()) > 100
    assert not torch.equal(many[0]['wgt'], many[1]['wgt'])  # (the scenes differ)


def test_many_scene_launches_refuse_shared_state(cuda):
    from online_joint_depthfusion_and_semantic_amd import _lib
    h, w, grid = 24, 32, 32
    st = make_stream(h, w, grid)
    fi = frame_inputs(st, 0)
    v = to_cuda(fresh_volumes(grid, False), cuda)
    ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, ops.MODE_FAST, cuda)
    job = dict(depth=_t(fi['depth'], cuda), mask=None, Ki=fi['Ki'], E=fi['E'], origin=st.origin, resolution=st.resolution,
               est=_t(fi['est'], cuda), tsdf=v['tsdf'], weights=v['wgt'], workspace=ws)
    with pytest.raises(_lib.OjfError, match='share'):
        ops.integrate_many([job, dict(job)])
    out = torch.empty((9, h * w), device=cuda)
    ej = dict(depth=job['depth'], Ki=fi['Ki'], E=fi['E'], origin=st.origin, resolution=st.resolution, tsdf=v['tsdf'], weights=v['wgt'],
              out_values=out, out_weights=torch.empty_like(out))
    with pytest.raises(_lib.OjfError, match='same output'):
        ops.extract_many([ej, dict(ej)])


@pytest.mark.parametrize('h,w,grid', [(120, 160, 64), (13, 15, 32), (48, 64, 64)])
def test_extract_bit_exact_for_a_rolled_camera(cuda, h, w, grid):
    """extract_tile_kernel comes in two tile orientations (lanes down the image columns / along the rows) and the host picks per frame
    the image axis that runs along the volume's contiguous z axis (csrc/ojf_extract.hip extract_columns).  An upright camera takes the
    column tiles (every other extract test); the same stream filmed with the camera ROLLED by 90 degrees takes the row tiles: all
    outputs - indices, corner weights, points, sample rows, sample planes, and the net-input form through Pipeline - bit for bit the oracle's."""
    st = make_stream(h, w, grid)
    rng = np.random.default_rng(5)
    tsdf = rng.uniform(-0.1, 0.1, (grid,) * 3).astype(np.float16)
    wgt = rng.uniform(0, 6, (grid,) * 3).astype(np.float16)
    g_tsdf, g_wgt = _t(tsdf, cuda), _t(wgt, cuda)
    roll = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], dtype=np.float32)
    for i in range(2):
        fi = frame_inputs(st, i)
        E = fi['E'].reshape(3, 4).copy()
        assert abs(E[2, 1]) > abs(E[2, 0])  # upright: the camera's y axis carries the volume's z
        E[:, :3] = E[:, :3] @ roll
        assert abs(E[2, 0]) > abs(E[2, 1])  # rolled: now the x axis does
        E = np.ascontiguousarray(E.reshape(12))
        ref = oracle.extract(fi['depth'], fi['Ki'], E, st.origin, st.resolution, tsdf, wgt, debug=True)
        out = ops.extract(_t(fi['depth'], cuda), fi['Ki'], E, st.origin, st.resolution, g_tsdf, g_wgt, debug=True)
        for key in ref:
            assert n_mismatch(out[key].cpu().numpy(), ref[key]) == 0, (key, i)
        planes = ops.extract(_t(fi['depth'], cuda), fi['Ki'], E, st.origin, st.resolution, g_tsdf, g_wgt, planes=True)
        assert n_mismatch(planes['fusion_values'].t().contiguous().cpu().numpy(), ref['fusion_values']) == 0
        assert n_mismatch(planes['fusion_weights'].t().contiguous().cpu().numpy(), ref['fusion_weights']) == 0
