"""-m gpu: the build-owned train / test drivers (counterparts of train_fusion.py / test_fusion.py)."""
import os

import numpy as np
import pytest
import torch

from online_joint_depthfusion_and_semantic_amd.config import default_config
from online_joint_depthfusion_and_semantic_amd.drivers import train_fusion, test_fusion as run_test_fusion, _training_defaults
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticDataset

pytestmark = pytest.mark.gpu


def test_train_then_test_round_trip(cuda, tmp_path):
    h, w, grid = 48, 64, 32
    cfg = _training_defaults(default_config(h, w))
    cfg.SETTINGS.device = str(cuda)
    cfg.TRAINING.optimization.accumulation_steps = 2
    cfg.TRAINING.optimizer.lr = 1e-3
    ds = SyntheticDataset(h, w, grid, 8, scenes=['room_0', 'room_1'])
    pipe, db, losses = train_fusion(cfg, ds, cuda, max_steps=12, checkpoint_dir=str(tmp_path), log=lambda *a: None)
    assert len(losses) == 12 and np.all(np.isfinite(losses))
    assert np.mean(losses[-4:]) < np.mean(losses[:4])  # the fusion net learns on the synthetic stream
    ck = torch.load(os.path.join(str(tmp_path), 'last.pth.tar'), map_location='cpu')
    assert set(ck) == {'epoch', 'model_state', 'optimizer_state', 'scheduler_state'}  # train_fusion.py:245-250
    assert any(db.state.values())
    results, per_scene, db2 = run_test_fusion(cfg, ds, cuda, state_dict=ck['model_state'], log=lambda *a: None)
    assert set(results) == {'mse', 'mad', 'iou', 'acc'} and all(np.isfinite(v) for v in results.values())
    assert set(per_scene) == {'room_0', 'room_1'}


def test_train_with_validation_best_last_resume(cuda, tmp_path):
    """train_fusion.py:110-122 (resume), :154-157 (hybrid trajectory reset), :191-251 (validation pass every eval_freq
    steps, best / last checkpoints, validation volumes saved to the workspace)."""
    h, w, grid = 48, 64, 32
    cfg = _training_defaults(default_config(h, w))
    cfg.SETTINGS.device = str(cuda)
    cfg.SETTINGS.eval_freq = 4
    cfg.SETTINGS.log_freq = 2
    cfg.SETTINGS.save_mode = 'tsdf'
    cfg.DATA.data_load_strategy = 'hybrid'
    cfg.TRAINING.optimization.accumulation_steps = 2
    cfg.TRAINING.optimization.reset_strategy = False
    cfg.TRAINING.optimizer.lr = 1e-3
    class PlainFrameNumbers(SyntheticDataset):  # the reference's loaders name frames 'scene/trajectory/<int>'
        def __getitem__(self, item):
            d = super().__getitem__(item)
            parts = d['frame_id'].split('/')
            d['frame_id'] = '/'.join(parts[:-1] + [str(int(parts[-1]))])
            return d
    train = PlainFrameNumbers(h, w, grid, 8, scenes=['room_0'])
    val = SyntheticDataset(h, w, grid, 4, scenes=['room_9'], seed=7)
    resets = []
    from online_joint_depthfusion_and_semantic_amd.database import Database
    orig_reset = Database.reset

    def spy(self, scene_id=None):
        resets.append(scene_id)
        return orig_reset(self, scene_id)
    Database.reset = spy
    try:
        out = str(tmp_path / 'run')
        pipe, db, losses = train_fusion(cfg, train, cuda, checkpoint_dir=out, val_dataset=val, log=lambda *a: None)
    finally:
        Database.reset = orig_reset
    assert len(losses) == 8
    assert 'room_0' in resets  # frame '.../000000' of a hybrid stream resets its scene's grid
    files = sorted(os.listdir(out))
    assert 'best.pth.tar' in files and 'last.pth.tar' in files
    best = torch.load(os.path.join(out, 'best.pth.tar'), map_location='cpu')
    last = torch.load(os.path.join(out, 'last.pth.tar'), map_location='cpu')
    assert set(best) == {'epoch', 'model_state', 'best_iou'} and 0 <= best['best_iou'] <= 1  # train_fusion.py:234-238
    assert set(last) == {'epoch', 'model_state', 'optimizer_state', 'scheduler_state'}
    exported = os.listdir(os.path.join(out, 'output'))
    assert any(n.startswith('room_9.tsdf_latest_val') for n in exported) and any(n.startswith('room_9.tsdf_best_val') for n in exported)
    scal = open(os.path.join(out, 'logs', 'scalars.csv')).read()
    assert 'Val/iou,4,' in scal and 'Val/iou,8,' in scal and 'Train/loss,2,' in scal and 'Train/iou,8,' in scal
    # resume: epoch counter, optimizer and scheduler state come back; one more epoch runs
    cfg.TRAINING.resume = os.path.join(out, 'last.pth.tar')
    cfg.TRAINING.n_epochs = 2
    pipe2, _, losses2 = train_fusion(cfg, train, cuda, checkpoint_dir=out, val_dataset=val, log=lambda *a: None)
    assert len(losses2) == 8  # epochs [1, 2) only
    again = torch.load(os.path.join(out, 'last.pth.tar'), map_location='cpu')
    assert again['epoch'] == 2 and again['scheduler_state']['last_epoch'] == 2 * last['scheduler_state']['last_epoch']


def test_test_fusion_reports_semantics(cuda, tmp_path):
    """test_fusion.py:88-118 with DATA.semantics: median-filtered labels, evaluate + evaluate_semantics, the text log."""
    h, w, grid = 48, 64, 32
    cfg = _training_defaults(default_config(h, w, semantics=True))
    cfg.SETTINGS.device = str(cuda)
    cfg.SETTINGS.save_mode = 'tsdf'
    ds = SyntheticDataset(h, w, grid, 6, scenes=['room_0'])
    results, per_scene, db = run_test_fusion(cfg, ds, cuda, log=lambda *a: None, test_dir=str(tmp_path))
    assert {'mse', 'mad', 'iou', 'acc', 'Mean Acc', 'Mean IoU'} <= set(results)
    text = open(os.path.join(str(tmp_path), 'test.logs')).read()
    assert 'Average semantic results over test scenes' in text and 'Scene: room_0' in text and 'Mean IoU' in text


def test_test_fusion_on_a_replica_layout(cuda, tmp_path):
    """The Replica adapter (datasets.py) feeding the test driver end to end: frames written to disk in the
    reference's layout (16-bit millimetre depth, camera-matrix files, npz GT grid) are fused and evaluated; the
    result must equal fusing the same decoded samples through ``Pipeline.fuse`` directly."""
    PIL = pytest.importorskip('PIL.Image')
    from online_joint_depthfusion_and_semantic_amd import datasets, drivers
    from online_joint_depthfusion_and_semantic_amd.config import database_config
    from online_joint_depthfusion_and_semantic_amd.database import Database
    from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline
    from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream, gt_volumes, grid_spec
    h = w = 64
    grid, n = 64, 6
    root = str(tmp_path)
    st = SyntheticStream(h, w, grid, 12)
    # the list format always carries five columns: [0] GT depth, [1] ToF depth, [-3] colour, [-2] cameras, [-1] semantics
    mods = ('left_depth_gt', 'left_depth_noise_5.0', 'left_rgb', 'left_camera_matrix', 'left_class30')
    for m in mods:
        os.makedirs(os.path.join(root, st.scene, '1', m))
    for i in range(n):
        f = st.frame(i)
        PIL.fromarray(np.zeros((h, w, 3), np.uint8)).save(os.path.join(root, st.scene, '1', 'left_rgb', '%d.png' % i))
        for m, key in (('left_depth_gt', 'depth_gt'), ('left_depth_noise_5.0', 'tof_depth')):
            PIL.fromarray(np.round(f[key] * 1000.0).astype(np.uint16)).save(os.path.join(root, st.scene, '1', m, '%d.png' % i))
        np.savetxt(os.path.join(root, st.scene, '1', 'left_camera_matrix', '%d.txt' % i),
                   datasets.Replica.file_from_pose(f['extrinsics']))
    with open(os.path.join(root, 'list.txt'), 'w') as fp:
        fp.write(' '.join('{}/1/{}'.format(st.scene, m) for m in mods) + '\n')
    tsdf, _ = gt_volumes(grid, 0.1)
    _, res, bbox = grid_spec(grid)
    datasets.export_grid_npz(os.path.join(root, st.scene, 'gt_semantic_sdf', 'sdf.hdf'), tsdf.astype(np.float32), bbox, res)

    config = _training_defaults(default_config(h, w))
    config.SETTINGS.device = str(cuda)
    lst = os.path.join(root, 'list.txt')
    config.DATA.update(dataset='Replica', root_dir=root, train_scene_list=lst, val_scene_list=lst, test_scene_list=lst,
                       normalize=False, truncation_strategy='standard')
    config.TESTING.test_ratio = 1
    ds = drivers.get_data('Replica', drivers.get_data_config(config, 'test'))
    torch.manual_seed(5)
    state = Pipeline(config)._fusion_network.state_dict()
    out_dir = os.path.join(root, 'test_out')
    results, per_scene, db = run_test_fusion(config, ds, cuda, state_dict=state, log=lambda *a: None, test_dir=out_dir)
    assert set(results) >= {'iou', 'acc', 'mse', 'mad'} and all(np.isfinite(v) for v in results.values())
    # test_fusion.py:120-122: every scene exported with SETTINGS.save_mode (default 'test': volumes + ply)
    from online_joint_depthfusion_and_semantic_amd import mesh
    exported = mesh.load_ply(os.path.join(out_dir, st.scene.replace('/', '.') + '.ply'))
    assert exported['vertices'].shape[0] > 100 and exported['faces'].max() < exported['vertices'].shape[0]
    assert np.isfinite(exported['vertices']).all() and exported['normals'].shape == exported['vertices'].shape
    assert any(n.startswith(st.scene.replace('/', '.') + '.tsdf.') for n in os.listdir(out_dir))

    pipe = Pipeline(config)
    pipe._fusion_network.load_state_dict(state)
    pipe = pipe.to(cuda).eval()
    db2 = Database(ds, database_config(config))
    with torch.no_grad():
        for i in range(len(ds)):
            s = ds[i]
            batch = {k: (v.unsqueeze(0) if torch.is_tensor(v) else [v]) for k, v in s.items()}
            batch = {k: (v.to(cuda) if torch.is_tensor(v) and k not in ('extrinsics', 'intrinsics') else v) for k, v in batch.items()}
            pipe.fuse(batch, db2, cuda)
    pipe.check()
    db2.filter(value=config.TESTING.outlier_filter_val)
    a, b = db.scenes_est[st.scene].volume, db2.scenes_est[st.scene].volume
    assert torch.equal(torch.as_tensor(a).cpu().view(torch.int16), torch.as_tensor(b).cpu().view(torch.int16))
    assert int((torch.as_tensor(db.fusion_weights[st.scene]).float() > 0).sum()) > 1000


def test_test_fusion_lookahead_chunks(cuda):
    """test_fusion with ``semantic_strategy: predict``: the driver hands Pipeline.fuse_sequence chunks of TESTING.lookahead
    frames and announces the next chunk (its 2-D pass on the side stream).  Ten frames in chunks of 3 (3 + 3 + 3 + 1) and of
    8 (8 + 2) against frame at a time: the geometry does not depend on the labels here (use_semantics False) - bit for bit, i.e.
    every frame was fused exactly once and in order.  (The labels themselves are not compared: the units' always-on dropout is
    keyed by the pass counter, which counts chunks here and frames there; tests/test_pipeline_gpu.py compares them with it off.)"""
    h, w, grid, n_classes = 64, 96, 32, 12

    def run(lookahead):
        cfg = _training_defaults(default_config(h, w, semantics=True, use_semantics=False, n_classes=n_classes))
        cfg.SETTINGS.device = str(cuda)
        cfg.DATA.semantic_strategy = 'predict'
        cfg.TESTING.lookahead = lookahead
        ds = SyntheticDataset(h, w, grid, 10, scenes=['room_0'], n_classes=n_classes)
        torch.manual_seed(5)  # the same randomly initialised networks in every run
        return run_test_fusion(cfg, ds, cuda, log=lambda *a: None)
    r1, _, db1 = run(1)
    for lookahead in (3, 8):
        r, _, db = run(lookahead)
        assert torch.equal(db.scenes_est['room_0'].volume.view(torch.int16), db1.scenes_est['room_0'].volume.view(torch.int16))
        assert torch.equal(db.fusion_weights['room_0'].view(torch.int16), db1.fusion_weights['room_0'].view(torch.int16))
        assert r['mse'] == pytest.approx(r1['mse'], rel=1e-12) and r['iou'] == pytest.approx(r1['iou'], rel=1e-12)
        assert int(db.ids_est['room_0'].volume.max()) < n_classes


def test_train_fusion_with_the_backward_pass_beside_the_next_frame_is_the_serial_loop(cuda):
    """drivers.train_fusion runs with FUSION_MODEL.train_overlap by default (a frame's backward pass on the gradient stream beside the next
    frame's forward stage; clip / all-reduce / optimizer step inside ``pipeline.gradients()``, a join in front of every evaluation):
    losses, parameters, BatchNorm buffers and the scene volume of eleven frames (accumulation over 3, an evaluation every 5) are bit for
    bit those of ``train_overlap: False``."""
    h, w, grid = 48, 64, 32

    def run(overlap):
        cfg = _training_defaults(default_config(h, w))
        cfg.SETTINGS.device = str(cuda)
        cfg.SETTINGS.seed = 5
        cfg.SETTINGS.eval_freq = 5
        if overlap is not None:
            cfg.FUSION_MODEL.train_overlap = overlap
        cfg.TRAINING.optimization.accumulation_steps = 3
        cfg.TRAINING.optimization.reset_strategy = False
        cfg.TRAINING.optimizer.lr = 1e-3
        ds = SyntheticDataset(h, w, grid, 12, scenes=['room_0'])
        pipe, db, losses = train_fusion(cfg, ds, cuda, max_steps=11, log=lambda *a: None)
        torch.cuda.synchronize()
        tn = pipe.__dict__['_hip_train']
        assert tn.overlap == (overlap is not False)
        net = pipe._fusion_network
        scene = next(iter(db.scenes_est))
        return (losses, [v.detach().clone() for v in net.state_dict().values()],
                torch.as_tensor(db.scenes_est[scene].volume).clone(), torch.as_tensor(db.fusion_weights[scene]).clone())
    a, b = run(None), run(False)
    assert len(a[0]) == 11 and a[0] == b[0]
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)
    assert torch.equal(a[2].view(torch.int16), b[2].view(torch.int16)) and torch.equal(a[3].view(torch.int16), b[3].view(torch.int16))
