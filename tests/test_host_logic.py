"""CPU: host-side logic of the package (no kernels): config, synthetic streams, the state_dict
schema / BN folding of the fusion nets, metrics, Database bookkeeping on host tensors."""
import os

import numpy as np
import pytest
import torch

from online_joint_depthfusion_and_semantic_amd import model, metrics
from online_joint_depthfusion_and_semantic_amd.config import AttrDict, default_config, database_config
from online_joint_depthfusion_and_semantic_amd.database import Database
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream, scene_sdf
from online_joint_depthfusion_and_semantic_amd import ops
from helpers import golden, net_from_golden


def test_config_attrdict():
    c = default_config(24, 32, semantics=True)
    assert c.FUSION_MODEL.n_points == 9 and c.DATA.resx == 32 and c['DATA']['semantics'] == 'class30'
    c.FUSION_MODEL.resx = 5
    assert c.FUSION_MODEL['resx'] == 5
    d = database_config(c)
    assert d.n_classes == 30 and d.init_value == 0.1 and d.semantic_grid is True
    with pytest.raises(AttributeError):
        _ = AttrDict(a=1).b


def test_synthetic_stream_is_seeded_and_sane():
    a, b = SyntheticStream(24, 32, 32, 20), SyntheticStream(24, 32, 32, 20)
    fa, fb = a.frame(3), b.frame(3)
    for k in ('tof_depth', 'mask', 'extrinsics', 'semantic_gt'):
        assert np.array_equal(fa[k], fb[k])
    assert not np.array_equal(a.frame(4)['tof_depth'], fa['tof_depth'])
    assert fa['tof_depth'].dtype == np.float32 and fa['extrinsics'].shape == (3, 4)
    R = fa['extrinsics'][:, :3]
    assert np.allclose(R.T @ R, np.eye(3), atol=1e-12)
    d = fa['depth_gt']
    assert (d > 0.3).all() and (d < 7).all()
    # surface points re-projected with the analytic depth lie on the zero level set of the GT SDF
    K, E = fa['intrinsics'], fa['extrinsics']
    v, u = np.mgrid[0:24, 0:32]
    pc = np.stack([(u - K[0, 2]) / K[0, 0] * d, (v - K[1, 2]) / K[1, 1] * d, d], -1)
    pw = pc @ E[:, :3].T + E[:, 3]
    assert np.abs(scene_sdf(pw)).max() < 1e-6  # depth_gt is stored as float32
    bt = a.batch(0)
    assert bt['tof_depth'].shape == (1, 24, 32) and bt['frame_id'][0].startswith('room_0/')


def test_state_dict_schema_and_param_counts():
    for sem, n_params, n_keys in ((False, 360591, 389), (True, 571833, 585)):
        cfg = AttrDict(n_points=9, growth_factor=6, use_semantics=sem, output_scale=1.0, resx=32, resy=24)
        net = model.FusionNet_v3(cfg)
        sd = net.state_dict()
        assert sum(p.numel() for p in net.parameters()) == n_params and len(sd) == n_keys
        for k in ('block0.4.block.4.weight', 'vortex0.gave_pool.1.weight', 'vortex3.branches.3.9.bias',
                  'vortex3.final.1.running_var', 'pred.4.pred.6.weight'):
            assert k in sd, k
        assert ('block2.0.block.0.weight' in sd) == sem
        assert sd['vortex3.final.0.weight'].shape == (114, 570, 1, 1)
        assert sd['vortex3.branches.0.0.weight'].shape[1] == (228 if sem else 114)
        assert len(model.fold_layers(net)) == (85 if sem else 57)


def test_bn_folding_matches_eval_forward():
    g = golden('pipeline_v3_nosem_24x32_g32.npz')
    net = net_from_golden(g, False, 24, 32)
    blk = net.block0[0].block
    x = torch.randn(1, 19, 24, 32)
    with torch.no_grad():
        want = blk[2](blk[1](blk[0](x)))
        w, b, k, d = model.fold_layers(net)[0]
        got = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), padding=1), 0.01)
    assert k == 3 and d == 1 and float((want - got).abs().max()) < 1e-5
    dil = [l[3] for l in model.fold_layers(net)[10:28]]
    assert dil == [1, 1, 1, 1, 1, 1, 3, 3, 1, 1, 9, 9, 1, 1, 27, 27, 1, 1]


def test_metrics_known_answers():
    est = np.array([-0.05, -0.01, 0.02, 0.03, 0.01], np.float16)
    gt = np.array([-0.02, 0.01, 0.02, -0.03, 0.5], np.float16)
    mask = np.array([1, 1, 1, 1, 0], bool)
    r = metrics.evaluation(est, gt, mask)
    e, t = np.clip(est.astype(np.float32), -0.04, 0.04), np.clip(gt.astype(np.float32), -0.04, 0.04)
    assert abs(r['mse'] - np.mean((e[:4] - t[:4]) ** 2)) < 1e-9
    assert abs(r['iou'] - 1 / 3) < 1e-6 and abs(r['acc'] - 0.5) < 1e-6
    ids_e = np.array([1, 1, 2, 0, 2], np.uint8)
    ids_t = np.array([1, 2, 2, 0, 1], np.uint8)
    m, per = metrics.semantic_evaluation(ids_e, ids_t, np.ones(5, bool), 4)
    assert abs(m['Mean IoU'] - (1 / 3 + 1 / 3) / 2) < 1e-5 and set(per) == {0, 1, 2}


def test_database_on_host_tensors():
    cfg = default_config(24, 32, semantics=True)
    cfg.SETTINGS.device = 'cpu'
    st = SyntheticStream(24, 32, 16, 5)
    db = Database(st, database_config(cfg))
    s = st.scene
    v = db[s]
    assert v['current'].dtype == torch.float16 and v['ids_est'].dtype == torch.uint8 and tuple(v['gt'].shape) == (16,) * 3
    assert v['origin'].dtype == torch.float64 and len(db) == 1 and db.state[s] is False
    db.fusion_weights[s][:2] = 3.0
    db.scenes_est[s].volume[:4] = -0.02
    db.state[s] = True
    db.filter(value=2.0)
    assert float(db.scenes_est[s].volume[1, 0, 0]) == pytest.approx(-0.02, abs=1e-3)
    assert float(db.scenes_est[s].volume[3, 0, 0]) == pytest.approx(0.1, abs=1e-3)
    db.to_numpy()
    r = db.evaluate(mode='val')
    assert set(r) == {'mse', 'mad', 'iou', 'acc'}
    db.to_torch()
    db.reset(s)
    assert db.state[s] is False and float(db.fusion_weights[s].abs().sum()) == 0


def test_database_save_to_workspace_file_layout(tmp_path):
    """modules/database.py:141-177 + utils/setup.py:224-274: file names and dataset keys of the workspace export
    ('tsdf' mode needs no device); only scenes with integrated frames are written; the default save mode of ``save`` /
    ``save_to_workspace`` is 'ply' like the reference's (:141, :180)."""
    import inspect
    from online_joint_depthfusion_and_semantic_amd.drivers import Workspace
    cfg = default_config(24, 32, semantics=True)
    cfg.SETTINGS.device = 'cpu'

    class Two(SyntheticStream):
        pass
    st = Two(24, 32, 16, 5)
    st.scenes = ['room_0', 'room_1/extra']
    db = Database(st, database_config(cfg))
    db.state['room_1/extra'] = True
    db.fusion_weights['room_1/extra'][:3] = 2.0
    ws = Workspace(str(tmp_path / 'exp'))
    db.save_to_workspace(ws, mode='latest_val', save_mode='tsdf')
    names = sorted(os.listdir(ws.output_path))
    stems = sorted({n.rsplit('.', 1)[0] for n in names})
    assert stems == ['room_1.extra.semantic_latest_val', 'room_1.extra.tsdf_latest_val', 'room_1.extra.weights_latest_val']
    wname = [n for n in names if '.weights_' in n][0]
    if wname.endswith('.npz'):  # no h5py in this image: same key, npz container
        got = np.load(os.path.join(ws.output_path, wname))['weights']
    else:
        import h5py
        got = np.array(h5py.File(os.path.join(ws.output_path, wname), 'r')['weights'])
    assert got.dtype == np.float16 and got[:3].min() == 2.0 and got[3:].max() == 0
    assert inspect.signature(Database.save).parameters['save_mode'].default == 'ply'
    assert inspect.signature(Database.save_to_workspace).parameters['save_mode'].default == 'ply'
    ws.save_model_state({'epoch': 1}, is_best=True, name='best.pth.tar')
    ws.save_model_state({'epoch': 2})
    assert sorted(os.listdir(ws.model_path)) == ['best.pth.tar', 'last.pth.tar']
    ws.log('hello', mode='val')
    ws.writer.add_scalar('Val/iou', 0.5, global_step=3)
    assert open(os.path.join(ws.log_path, 'validation.logs')).read() == 'hello\n'
    assert open(os.path.join(ws.log_path, 'scalars.csv')).read() == 'Val/iou,3,0.5\n'


def test_camera_arrays_follow_reference_host_math():
    st = SyntheticStream(24, 32, 16, 5)
    f = st.frame(2)
    Ki, E = ops.camera_arrays(torch.from_numpy(f['intrinsics']), torch.from_numpy(f['extrinsics']))
    want = torch.from_numpy(f['intrinsics']).float().inverse().numpy().reshape(9)
    assert np.array_equal(Ki, want) and Ki.dtype == np.float32
    E44 = np.vstack([f['extrinsics'], [0, 0, 0, 1]])
    _, E2 = ops.camera_arrays(f['intrinsics'], E44)  # ScanNet-style 4x4 poses
    assert np.array_equal(E, E2) and np.array_equal(E, f['extrinsics'].astype(np.float32).reshape(12))


def test_fusion_loss_and_schedule():
    from online_joint_depthfusion_and_semantic_amd.loss import FusionLoss, PolynomialLR
    torch.manual_seed(0)
    e, t = torch.randn(1, 57, 9) * 0.1, torch.randn(1, 57, 9) * 0.1
    x1 = torch.sign(e).reshape(1, 9, 57)[0].T
    x2 = torch.sign(t).reshape(1, 9, 57)[0].T
    l3 = torch.nn.CosineEmbeddingLoss(margin=0.0, reduction='mean')(x1, x2, torch.ones(57))
    want = (e - t).abs().mean() + 10 * ((e - t) ** 2).mean() + 0.1 * l3  # utils/loss.py:65-103
    assert abs(float(FusionLoss()(e, t)) - float(want)) < 1e-7
    empty = FusionLoss()(e[:, :0], t[:, :0])
    assert float(empty) == 1.0 and empty.grad_fn is None
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = PolynomialLR(opt, max_iter=100)
    for _ in range(10):
        opt.step()
        sch.step()
    assert abs(opt.param_groups[0]['lr'] - (1 - 10 / 100) ** 0.9) < 1e-12  # utils/schedulers.py:19-21


def test_synthetic_dataset_and_checkpoint_keys():
    from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticDataset
    from online_joint_depthfusion_and_semantic_amd.drivers import remove_parent
    ds = SyntheticDataset(12, 16, 16, 3, scenes=['a', 'b'])
    assert len(ds) == 6 and ds.scenes == ['a', 'b']
    s = ds[4]
    assert s['frame_id'].startswith('b/') and s['tof_depth'].shape == (12, 16) and s['extrinsics'].shape == (3, 4)
    b = next(iter(torch.utils.data.DataLoader(ds, batch_size=1)))
    assert b['tof_depth'].shape == (1, 12, 16) and b['frame_id'][0].startswith('a/') and b['mask'].dtype == torch.bool
    g = ds.get_grid('b', 0.1, True)
    assert g[0].volume.shape == (16, 16, 16) and g[1].volume.dtype == np.uint8
    st = remove_parent({'_fusion_network.pred.0.w': 1, 'x': 2}, '_fusion_network')
    assert st == {'pred.0.w': 1, 'x': 2}


def test_f_score_known_answers():
    # a plane x = 1.03 sampled on an 8 cm grid: the zero crossings of its exact SDF recover the plane
    g, res, origin = 24, 0.08, np.array([0.0, 0.0, 0.0])
    xs = (np.arange(g) + 0.5) * res
    sdf = np.broadcast_to((1.03 - xs)[:, None, None], (g, g, g)).astype(np.float32)
    pts = metrics.surface_points(sdf, np.ones((g, g, g), bool), origin, res)
    assert len(pts) == g * g and np.abs(pts[:, 0] - 1.03).max() < 1e-6
    same = metrics.reconstruction_f_score(sdf, sdf, np.ones_like(sdf), origin, res)
    assert same == {'precision': 1.0, 'recall': 1.0, 'fscore': 1.0}
    shifted = np.broadcast_to((1.03 + 0.5 - xs)[:, None, None], (g, g, g)).astype(np.float32)
    far = metrics.reconstruction_f_score(shifted, sdf, np.ones_like(sdf), origin, res)
    assert far['fscore'] == 0.0
    near = metrics.reconstruction_f_score(sdf + 0.05, sdf, np.ones_like(sdf), origin, res)  # 5 cm < 1.5 voxels
    assert near['fscore'] == 1.0
    assert metrics.f_score(np.zeros((0, 3)), pts, 0.1)['fscore'] == 0.0


def test_bench_gpus_n_without_launcher_does_not_die_in_argument_handling():
    """``python bench.py --gpus 8`` (no torch.distributed.run around it) used to exit with
    "launch with torch.distributed.run"; it now becomes the launcher itself.  Without a HIP device the only acceptable
    failure is the loud "needs an MI355X" one."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    if torch.cuda.is_available():
        pytest.skip('argument handling without a device is what this test covers')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '20', '--warmup', '5'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert 'needs an MI355X' in r.stderr and 'torch.distributed.run' not in r.stderr
