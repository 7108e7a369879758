"""Build-owned counterparts of the reference's drivers (test_fusion.py:24-122, train_fusion.py:35-255)
around the drop-in ``Pipeline`` / ``Database``: same loop structure, checkpoint layout
({epoch, model_state, optimizer_state, scheduler_state}, train_fusion.py:234-251), post-processing
(filter -> evaluate, test_fusion.py:82-94) - plus what the reference lacks: one process per GPU with
scenes sharded across ranks and ONE flat RCCL all-reduce of the fusion-net gradients per
accumulation boundary.

    python -m online_joint_depthfusion_and_semantic_amd.drivers test  [--config cfg.yaml]
    python -m torch.distributed.run --nproc-per-node N -m online_joint_depthfusion_and_semantic_amd.drivers train
"""
import argparse
import contextlib
import os

import numpy as np
import torch

from .config import default_config, database_config, load_config_from_yaml
from .database import Database
from .distributed import FlatGradientAllReduce, ShardedScenes, init_from_env
from .loss import FusionLoss, PolynomialLR
from .pipeline import Pipeline
from .synthetic import SyntheticDataset


def remove_parent(state, parent):
    """utils/loading.py:197-200: strip an optional '<parent>.' prefix from checkpoint keys."""
    pre = parent + '.'
    return {(k[len(pre):] if k.startswith(pre) else k): v for k, v in state.items()}


def weights_init(m):
    if isinstance(m, torch.nn.Conv2d):  # train_fusion.py:27-31
        torch.nn.init.xavier_normal_(m.weight)


def _host_pose_batch(batch, device):
    """transform.to_device (utils/transform.py:33-37) for everything but the poses, which are kernel
    arguments and stay on the host."""
    if batch.get('_ojf_on_device') == str(device):  # (idempotent: the training loop prepares a batch once and announces THAT object)
        return batch
    out = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) and k not in ('extrinsics', 'intrinsics') else v)
           for k, v in batch.items()}
    out['_ojf_on_device'] = str(device)
    return out


def _loader(dataset, scenes, shuffle=False):
    if hasattr(dataset, 'scene_of_item'):  # Replica / ScanNet adapters
        idx = [i for i in range(len(dataset)) if dataset.scene_of_item(i) in scenes]
    else:
        idx = [i for i in range(len(dataset)) if dataset.scenes[i // dataset.frames_per_scene] in scenes]
    return torch.utils.data.DataLoader(torch.utils.data.Subset(dataset, idx), batch_size=1, shuffle=shuffle)


class Workspace:
    """utils/setup.py:224-274: experiment directory with model / logs / output sub-directories, the two text loggers
    and the data writers ``Database.save_to_workspace`` calls.  Scalars go to ``logs/scalars.csv`` (tag, step, value)
    instead of a tensorboard event file (tensorboard glue is out of scope); ``writer.add_scalar`` keeps
    the SummaryWriter call shape so the loop reads like train_fusion.py:178-224."""

    class _Scalars:
        def __init__(self, path):
            self.path = path

        def add_scalar(self, tag, value, global_step=None):
            with open(self.path, 'a') as f:
                f.write('{},{},{}\n'.format(tag, global_step, float(value)))

    def __init__(self, path):
        self.workspace_path = path
        self.model_path = os.path.join(path, 'model')
        self.log_path = os.path.join(path, 'logs')
        self.output_path = os.path.join(path, 'output')
        for d in (self.workspace_path, self.model_path, self.log_path, self.output_path):
            os.makedirs(d, exist_ok=True)
        self.writer = Workspace._Scalars(os.path.join(self.log_path, 'scalars.csv'))

    def log(self, message, mode='train'):
        name = {'train': 'training', 'val': 'validation'}.get(mode, mode)
        with open(os.path.join(self.log_path, name + '.logs'), 'a') as f:
            f.write(str(message) + '\n')

    def save_config(self, config):
        import json
        with open(os.path.join(self.workspace_path, 'config.json'), 'w') as f:
            json.dump(config, f, default=str)

    def save_model_state(self, state, is_best=False, name=None):
        """utils/saving.py:67-91: 'last.pth.tar', or ``name`` / 'best.pth.tar' when ``is_best``."""
        torch.save(state, os.path.join(self.model_path, (name or 'best.pth.tar') if is_best else 'last.pth.tar'))

    def _save_volume(self, file, key, data):
        from .datasets import save_volume_hdf
        save_volume_hdf(os.path.join(self.output_path, file), key, data)

    def save_tsdf_data(self, file, data):
        self._save_volume(file, 'TSDF', data)

    def save_weights_data(self, file, data):
        self._save_volume(file, 'weights', data)

    def save_semantic_data(self, file, data):
        self._save_volume(file, 'semantics', data)

    def save_ply_data(self, file, data, resolution=0.01):
        """utils/saving.py:42-48 (hard-coded 1 cm spacing there); the mesh comes from the HIP kernel (mesh.py)."""
        from . import mesh
        vol = data if torch.is_tensor(data) else torch.from_numpy(np.ascontiguousarray(data))
        m = mesh.extract_mesh(vol.to(device='cuda', dtype=torch.float16).contiguous(), resolution=float(resolution))
        mesh.save_ply(os.path.join(self.output_path, file), m['vertices'], m['faces'], m['normals'])


def test_fusion(config, dataset, device, rank=0, world=1, state_dict=None, log=print, test_dir=None):
    """Fuses this rank's scenes frame by frame, then filter -> filter_semantics -> evaluate -> evaluate_semantics
    (test_fusion.py:73-118, written to ``test_dir``/test.logs like the reference's logger) and, when ``test_dir`` is
    given, exports every scene with ``SETTINGS.save_mode`` ('test' | 'ply' | 'tsdf', test_fusion.py:120-122)."""
    shard = ShardedScenes(dataset, rank, world)
    database = Database(shard, database_config(config))
    pipeline = Pipeline(config)
    if state_dict is not None:
        pipeline._fusion_network.load_state_dict(remove_parent(state_dict, '_fusion_network'))
    pipeline = pipeline.to(device).eval()
    # semantic_strategy 'predict': the 2-D network does not depend on the volumes, so the labels of TESTING.lookahead (default 8: 1075 frames/s against 940 at 4 and 511 frame at a time)
    # consecutive frames are predicted as one batched pass (Pipeline.fuse_sequence); the frame steps themselves stay in order
    lookahead = int(config.TESTING.get('lookahead', 8)) if (config.DATA.semantics and config.DATA.semantic_strategy == 'predict') else 1
    with torch.no_grad():
        chunk, ahead = [], None  # `ahead`: the chunk in front of `chunk`, fused once `chunk` is complete (its 2-D pass then runs beside ahead's frame steps)
        for batch in _loader(dataset, shard.scenes):
            if not torch.all(torch.isfinite(batch['extrinsics'])):
                continue
            if lookahead <= 1:
                pipeline.fuse(_host_pose_batch(batch, device), database, device)
                continue
            chunk.append(_host_pose_batch(batch, device))
            if len(chunk) == lookahead:
                if ahead is not None:
                    pipeline.fuse_sequence(ahead, database, device, prefetch=chunk)
                ahead, chunk = chunk, []
        if ahead is not None:
            pipeline.fuse_sequence(ahead, database, device, prefetch=chunk or None)
        if chunk:
            pipeline.fuse_sequence(chunk, database, device)
    pipeline.check()  # loud if the split-fp16 range guard fired
    database.filter(value=config.TESTING.outlier_filter_val)  # on device; to_numpy() only for export
    semantics = bool(config.DATA.semantics)
    if semantics:
        database.filter_semantics(value=5)  # test_fusion.py:88-89

    lines = []

    def emit(msg):
        lines.append(msg)
        log('rank {} {}'.format(rank, msg))
    results, per_scene = database.evaluate(mode='test')
    emit('Average test results over test scenes:')
    for k, v in results.items():
        emit('{}: {}'.format(k, v))
    emit('Per scene results')
    for scene, r in per_scene.items():
        emit('Scene: ' + scene)
        for k, v in r.items():
            emit('{}: {}'.format(k, v))
    if semantics and config.DATA.get('semantic_grid', False):  # test_fusion.py:108-118
        sem_results, class_iou = database.evaluate_semantics(mode='test')
        emit('Average semantic results over test scenes')
        for k, v in sem_results.items():
            emit('{:12}:\t{}'.format(k, v))
        emit('Per scene semantic results:')
        for scene, r in class_iou.items():
            emit('Scene: ' + scene)
            for k, v in r.items():
                emit('{}: {}'.format(k, v))
        results = dict(results, **sem_results)
    if test_dir is not None:
        os.makedirs(test_dir, exist_ok=True)
        with open(os.path.join(test_dir, 'test.logs' if world == 1 else 'test.rank%d.logs' % rank), 'a') as f:
            f.write('\n'.join(lines) + '\n')
        for scene_id in database.scenes_est.keys():
            database.save(path=test_dir, save_mode=config.SETTINGS.get('save_mode', 'test'), scene_id=scene_id)
    return results, per_scene, database


def _dist_on():
    d = torch.distributed
    return d.is_available() and d.is_initialized() and d.get_world_size() > 1


def _agree(values, op='max'):
    """A number (or a fixed-length list of numbers) agreed on by all ranks in ONE all-reduce (gloo: CPU tensor,
    RCCL: device tensor)."""
    scalar = not isinstance(values, (list, tuple))
    vals = [float(values)] if scalar else [float(v) for v in values]
    if _dist_on():
        d = torch.distributed
        dev = torch.device('cuda', torch.cuda.current_device()) if d.get_backend() == 'nccl' else torch.device('cpu')
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        d.all_reduce(t, op={'max': d.ReduceOp.MAX, 'sum': d.ReduceOp.SUM, 'min': d.ReduceOp.MIN}[op])
        vals = t.tolist()
    return vals[0] if scalar else vals


def sync_buffers(net):
    """BatchNorm running statistics drift apart across ranks (batch 1, per-rank frames): average the float buffers
    (one flat all-reduce) and take the max of the step counters before anything is checkpointed or validated, so
    that every rank holds - and rank 0 saves - the same network."""
    if not _dist_on():
        return
    d = torch.distributed
    fl = [b for b in net.buffers() if b.is_floating_point()]
    if fl:
        flat = torch.cat([b.detach().reshape(-1).float() for b in fl])
        d.all_reduce(flat, op=d.ReduceOp.SUM)
        flat /= d.get_world_size()
        off = 0
        for b in fl:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    for b in net.buffers():
        if not b.is_floating_point():
            d.all_reduce(b, op=d.ReduceOp.MAX)


def _default_frame_step(pipeline, criterion, batch, database, device):
    out = pipeline.fuse_training(_host_pose_batch(batch, device), database, device)
    return criterion(out['tsdf_fused'], out['tsdf_target'])


VAL_KEYS = ('mse', 'mad', 'iou', 'acc')


def validate(config, pipeline, dataset, database, scenes, device, workspace=None):
    """train_fusion.py:201-214: fuse the validation frames in eval mode, filter(0.5), evaluate.  Returns the metric
    SUMS over this rank's scenes (always the four VAL_KEYS, so that the cross-rank reduction has one shape) and the
    scene count (the caller averages over all ranks' scenes, like database.py:304 does over one process's)."""
    database.reset()
    pipeline.eval()
    with torch.no_grad():
        for batch in _loader(dataset, scenes):
            if not torch.all(torch.isfinite(batch['extrinsics'])):
                continue
            pipeline.fuse(_host_pose_batch(batch, device), database, device)
    pipeline.check()
    database.filter(value=0.5)
    ev = database.evaluate(mode='val', workspace=workspace)
    n = len(database.scenes_est.keys())
    return [ev.get(k, 0.) * n for k in VAL_KEYS], n


def train_fusion(config, dataset, device, rank=0, world=1, max_steps=None, log=print, checkpoint_dir=None,
                 val_dataset=None, workspace=None, frame_step=None):
    """Online learning loop (train_fusion.py:35-255) on this rank's scenes.

    Collective schedule: the reference is single-process; here every rank walks ITS OWN frame stream, so shards
    of different length, ranks without scenes and frames skipped for a non-finite pose must not change how often
    ``grads.reduce()`` is called.  Every epoch the ranks agree on ``n_steps = max(len(local loader))``; step ``i``
    runs on every rank (a rank that has no frame i, or skips it, contributes its zero / partial gradient), and the
    accumulation boundary ``(i + 1) % accumulation_steps == 0 or i == n_steps - 1`` (train_fusion.py:186) as well as
    the evaluation boundary (:191) are functions of the common ``i`` alone.

    ``val_dataset``: validation frames (fused every ``SETTINGS.eval_freq`` steps and at the end of the epoch, best /
    last checkpoints, train_fusion.py:191-251); ``workspace``: a ``Workspace`` (default: ``checkpoint_dir`` when
    given); ``frame_step(pipeline, criterion, batch, database, device) -> loss`` replaces the default
    ``fuse_training`` + criterion (tests, custom losses)."""
    if config.SETTINGS.seed:
        np.random.seed(config.SETTINGS.seed + rank)
        torch.manual_seed(config.SETTINGS.seed)  # identical replicas on every rank
    frame_step = frame_step or _default_frame_step
    shard = ShardedScenes(dataset, rank, world)
    database = Database(shard, database_config(config))
    val_shard = val_database = None
    if val_dataset is not None:
        val_shard = ShardedScenes(val_dataset, rank, world)
        val_database = Database(val_shard, database_config(config))
    if workspace is None and checkpoint_dir:  # every rank exports its own validation scenes into the shared directory
        workspace = Workspace(checkpoint_dir)
        workspace.model_path = checkpoint_dir  # checkpoints directly under checkpoint_dir
    chief = workspace is not None and rank == 0  # rank 0 alone writes logs, scalars and model checkpoints
    # this function's loop is written for it (gradient work inside ``with pipeline.gradients():``, joins before evaluation and
    # checkpoints): a frame's backward pass runs beside the next frame's forward stage - same bits, +14 % frames/s.
    # (Set here, not in _training_defaults: a loop that clips / steps outside that context must keep the serial contract.)
    config.FUSION_MODEL.setdefault('train_overlap', True)
    pipeline = Pipeline(config)
    pipeline.apply(weights_init)
    net = pipeline._fusion_network
    if config.FUSION_MODEL.get('pretrained'):  # train_fusion.py:84-86
        net.load_state_dict(torch.load(config.FUSION_MODEL.pretrained, map_location='cpu')['model_state'])
    pipeline = pipeline.to(device)
    opt_cfg = config.TRAINING.optimizer
    optimizer = torch.optim.RMSprop(net.parameters(), lr=opt_cfg.lr, momentum=opt_cfg.momentum,
                                    weight_decay=opt_cfg.weight_decay, eps=opt_cfg.eps)
    scheduler = PolynomialLR(optimizer, config.TRAINING.scheduler.max_iter)
    criterion = FusionLoss(w_l1=config.TRAINING.loss.w_l1, w_l2=config.TRAINING.loss.w_l2, w_cos=config.TRAINING.loss.w_cos)

    start_epoch = 0
    resume = config.TRAINING.get('resume')
    if resume:  # train_fusion.py:110-122
        if os.path.isfile(resume):
            log('Loading model and optimizer from checkpoint {}'.format(resume))
            ck = torch.load(resume, map_location=device)
            net.load_state_dict(remove_parent(ck['model_state'], '_fusion_network'))
            optimizer.load_state_dict(ck['optimizer_state'])
            scheduler.load_state_dict(ck['scheduler_state'])
            start_epoch = ck['epoch']
        else:
            log('No checkpoint found at {}'.format(resume))

    grads = FlatGradientAllReduce(net)  # p.grad are views into one flat buffer
    opt = config.TRAINING.optimization
    accum = opt.accumulation_steps
    eval_freq = int(config.SETTINGS.get('eval_freq', 0) or 0)
    log_freq = int(config.SETTINGS.get('log_freq', 0) or 0)
    hybrid = config.DATA.get('data_load_strategy', None) == 'hybrid'
    save_mode = config.SETTINGS.get('save_mode', 'tsdf')
    best_score = 0.
    losses, step, window = [], 0, 0.
    done = False
    for epoch in range(start_epoch, config.TRAINING.n_epochs):
        if chief:
            workspace.log('Training epoch {}/{}'.format(epoch, config.TRAINING.n_epochs), mode='train')
        database.reset()
        net.train()
        loader = _loader(dataset, shard.scenes)
        n_steps = int(_agree(len(loader), 'max'))  # the common step count of this epoch
        frames = iter(loader)
        ahead = next(frames, None)
        for i in range(n_steps):
            batch, ahead = ahead, next(frames, None)
            if batch is not None:
                batch = _host_pose_batch(batch, device)
            if (ahead is not None and config.SETTINGS.get('announce_frames', False) and torch.all(torch.isfinite(ahead['extrinsics']))
                    and hasattr(pipeline, 'announce_training_frame')):
                # one frame of look-ahead for the ONE host read of the frame step (Pipeline.announce_training_frame)
                ahead = _host_pose_batch(ahead, device)
                pipeline.announce_training_frame(ahead, device)
            boundary = (i + 1) % accum == 0 or i == n_steps - 1  # decided on the common counter, before any skip
            evaluate_now = (eval_freq > 0 and (i + 1) % eval_freq == 0) or i == n_steps - 1
            if batch is not None and torch.all(torch.isfinite(batch['extrinsics'])):
                scene = batch['frame_id'][0].split('/')[0]
                if hybrid and batch['frame_id'][0].split('/')[-1] == '0':  # new trajectory (train_fusion.py:154-157)
                    if chief:
                        workspace.log('Resetting grid for scene {} at step {}'.format(scene, i), mode='train')
                    database.reset(scene)
                if opt.reset_strategy and np.random.random_sample() <= opt.reset_prob:
                    database.reset(scene)
                loss = frame_step(pipeline, criterion, batch, database, device)
                if loss.grad_fn is not None:
                    loss.backward()
                    # train_fusion.py:172 reads loss.item() on every frame; the values are kept on the device and read at the
                    # logging / evaluation boundaries instead (a host read per frame drains the queue: the frame step runs
                    # asynchronously otherwise, pipeline.py::fuse_training)
                    losses.append(loss.detach())
                    window = window + losses[-1]
            if log_freq and (i + 1) % log_freq == 0:
                if chief:
                    workspace.writer.add_scalar('Train/loss', float(window) / log_freq, global_step=i + 1 + epoch * n_steps)
                window = 0.
            # (FUSION_MODEL.train_overlap: this frame's backward pass runs on the pipeline's gradient stream beside the next frame's
            # forward stage; everything that touches the gradients or steps the weights goes behind it on that stream)
            # and, with FUSION_MODEL.train_overlap_thread, is enqueued by the pipeline's gradient thread: this thread waits for it only
            # at the boundaries, where the weights change)
            def gradient_step(boundary=boundary):
                if opt.clipping:
                    grads.clip_(1.0)  # clip_grad_norm_(net.parameters(), 1., 2) on the flat buffer the gradients live in
                if boundary:
                    grads.reduce()  # the single exchange step of the whole training path: every rank, every boundary
                    optimizer.step()
                    grads.zero()    # keeps p.grad aliased to the flat buffer (no set_to_none)
                    scheduler.step()
            if hasattr(pipeline, 'gradient_work'):
                pipeline.gradient_work(gradient_step, join=boundary)
            else:
                gradient_step()
            step += 1
            done = max_steps is not None and step >= max_steps
            if evaluate_now or done:
                if hasattr(pipeline, 'join_gradients'):
                    pipeline.join_gradients()
                grads.zero()
                sync_buffers(net)
                gstep = i + 1 + epoch * n_steps
                train_eval = database.evaluate(mode='train', workspace=workspace if chief else _Quiet)  # on device
                if chief:
                    for k in ('mse', 'acc', 'iou', 'mad'):
                        if k in train_eval:
                            workspace.writer.add_scalar('Train/' + k, train_eval[k], global_step=gstep)
                score = None
                if val_dataset is not None:
                    sums, n = validate(config, pipeline, val_dataset, val_database, val_shard.scenes, device,
                                       workspace if chief else _Quiet)
                    tot = _agree(sums + [n], 'sum')
                    val_eval = {k: tot[j] / max(tot[-1], 1) for j, k in enumerate(VAL_KEYS)}
                    if chief:
                        for k, v in val_eval.items():
                            workspace.writer.add_scalar('Val/' + k, v, global_step=gstep)
                    score = (val_eval.get('iou', 0.) + val_eval.get('acc', 0.)) / 2
                    if score >= best_score:  # train_fusion.py:226-240
                        best_score = score
                        if workspace is not None:
                            val_database.save_to_workspace(workspace, mode='best_val', save_mode=save_mode)
                        if chief:
                            workspace.log('Found new best model with score {} at epoch {}'.format(best_score, epoch), mode='val')
                            workspace.save_model_state({'epoch': epoch + 1, 'model_state': net.state_dict(),
                                                        'best_iou': best_score}, is_best=True, name='best.pth.tar')
                    if workspace is not None:
                        val_database.save_to_workspace(workspace, mode='latest_val', save_mode=save_mode)
                if chief:  # train_fusion.py:245-251
                    workspace.save_model_state({'epoch': epoch + 1, 'model_state': net.state_dict(),
                                                'optimizer_state': optimizer.state_dict(),
                                                'scheduler_state': scheduler.state_dict()}, is_best=False)
                net.train()
            if done:
                break
        if done:
            break
    if hasattr(pipeline, 'join_gradients'):
        pipeline.join_gradients()
    losses = [float(v) for v in torch.stack(losses).cpu()] if losses else []  # one transfer for the whole run
    log('rank {} mean loss {:.6f} over {} frames'.format(rank, float(np.mean(losses)) if losses else float('nan'), len(losses)))
    return pipeline, database, losses


class _Quiet:
    """Workspace stand-in for the ranks that do not log."""

    @staticmethod
    def log(message, mode='train'):
        pass


def _training_defaults(config):
    t = config.TRAINING
    t.setdefault('n_epochs', 1)
    t.setdefault('optimizer', {'name': 'rmsprop', 'lr': 1.e-05, 'momentum': 0.9, 'weight_decay': 0.01, 'eps': 1.e-09})
    t.setdefault('scheduler', {'name': 'poly_lr', 'max_iter': 50000})
    t.setdefault('loss', {'name': 'fusion', 'w_l1': 1., 'w_l2': 10, 'w_cos': 0.1})
    t.optimization.setdefault('reset_strategy', True)
    t.optimization.setdefault('reset_prob', 0.01)
    from .config import AttrDict
    config.TRAINING = AttrDict(dict(t))
    return config


def get_data_config(config, mode):
    """utils/setup.py:28-70: the flat per-split config the dataset classes read."""
    import copy
    from .config import AttrDict
    from .datasets import ToTensor
    d = AttrDict(copy.deepcopy(dict(config.DATA)))
    d.device = config.SETTINGS.device
    d.implementation = config.SETTINGS.get('implementation', None)
    d.mode = mode
    if mode == 'train':
        d.scene_list, d.frame_ratio = d.train_scene_list, config.TRAINING.train_ratio
    elif mode == 'val':
        d.scene_list, d.frame_ratio = d.val_scene_list, config.TRAINING.val_ratio
    else:
        d.scene_list, d.frame_ratio = d.test_scene_list, config.TESTING.test_ratio
    for key, default in (('fusion_strategy', None), ('data_load_strategy', None), ('intensity_grad', False)):
        d.setdefault(key, default)
    d.n_classes = config.get('SEMANTIC_2D_MODEL', {}).get('n_classes', None)
    d.augmentations = None  # photometric / geometric augmentations of the 2-D segmentation training are out of scope
    d.transform = ToTensor()
    return d


def get_data(name, data_config):
    """utils/setup.py:73-77: ``DATA.dataset`` -> dataset object (``Replica`` | ``ScanNet``)."""
    from . import datasets
    if name not in ('Replica', 'ScanNet'):
        raise ValueError('unknown dataset {!r} (Replica | ScanNet)'.format(name))
    return getattr(datasets, name)(data_config)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('mode', choices=['train', 'test'])
    ap.add_argument('--config')
    ap.add_argument('--height', type=int, default=120)
    ap.add_argument('--width', type=int, default=160)
    ap.add_argument('--grid', type=int, default=64)
    ap.add_argument('--frames', type=int, default=20)
    ap.add_argument('--scenes', type=int, default=None, help='number of synthetic scenes (default: one per rank)')
    ap.add_argument('--checkpoint')
    args = ap.parse_args()
    rank, world, local = init_from_env()
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    config = load_config_from_yaml(args.config) if args.config else default_config(args.height, args.width)
    config.SETTINGS.device = str(device)
    config = _training_defaults(config)
    n_scenes = args.scenes or world
    if config.DATA.get('dataset', 'synthetic') in ('Replica', 'ScanNet'):  # real data in the reference's layout
        dataset = get_data(config.DATA.dataset, get_data_config(config, args.mode))
    else:
        dataset = SyntheticDataset(config.DATA.resy, config.DATA.resx, args.grid, args.frames,
                                   scenes=['room_%d' % i for i in range(n_scenes)])
    if args.mode == 'test':
        state = torch.load(args.checkpoint, map_location='cpu')['model_state'] if args.checkpoint else None
        test_fusion(config, dataset, device, rank, world, state)
    else:
        train_fusion(config, dataset, device, rank, world, checkpoint_dir=os.path.dirname(args.checkpoint) if args.checkpoint else None)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
