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
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) and k not in ('extrinsics', 'intrinsics') else v)
            for k, v in batch.items()}


def _loader(dataset, scenes, shuffle=False):
    if hasattr(dataset, 'scene_of_item'):  # Replica / ScanNet adapters
        idx = [i for i in range(len(dataset)) if dataset.scene_of_item(i) in scenes]
    else:
        idx = [i for i in range(len(dataset)) if dataset.scenes[i // dataset.frames_per_scene] in scenes]
    return torch.utils.data.DataLoader(torch.utils.data.Subset(dataset, idx), batch_size=1, shuffle=shuffle)


def test_fusion(config, dataset, device, rank=0, world=1, state_dict=None, log=print, test_dir=None):
    """Fuses this rank's scenes frame by frame, then filter -> evaluate (test_fusion.py:73-118) and, when ``test_dir``
    is given, exports every scene with ``SETTINGS.save_mode`` ('test' | 'ply' | 'tsdf', test_fusion.py:120-122)."""
    shard = ShardedScenes(dataset, rank, world)
    database = Database(shard, database_config(config))
    pipeline = Pipeline(config)
    if state_dict is not None:
        pipeline._fusion_network.load_state_dict(remove_parent(state_dict, '_fusion_network'))
    pipeline = pipeline.to(device).eval()
    with torch.no_grad():
        for batch in _loader(dataset, shard.scenes):
            if not torch.all(torch.isfinite(batch['extrinsics'])):
                continue
            pipeline.fuse(_host_pose_batch(batch, device), database, device)
    pipeline.check()  # loud if the split-fp16 range guard fired
    database.filter(value=config.TESTING.outlier_filter_val)  # on device; to_numpy() only for export
    results, per_scene = database.evaluate(mode='test')
    for k, v in results.items():
        log('rank {} {}: {}'.format(rank, k, v))
    if test_dir is not None:
        os.makedirs(test_dir, exist_ok=True)
        for scene_id in database.scenes_est.keys():
            database.save(path=test_dir, save_mode=config.SETTINGS.get('save_mode', 'test'), scene_id=scene_id)
    return results, per_scene, database


def train_fusion(config, dataset, device, rank=0, world=1, max_steps=None, log=print, checkpoint_dir=None):
    """Online learning loop (train_fusion.py:133-189) on this rank's scenes."""
    if config.SETTINGS.seed:
        np.random.seed(config.SETTINGS.seed + rank)
        torch.manual_seed(config.SETTINGS.seed)  # identical replicas on every rank
    shard = ShardedScenes(dataset, rank, world)
    database = Database(shard, database_config(config))
    pipeline = Pipeline(config)
    pipeline.apply(weights_init)
    pipeline = pipeline.to(device)
    net = pipeline._fusion_network
    opt_cfg = config.TRAINING.optimizer
    optimizer = torch.optim.RMSprop(net.parameters(), lr=opt_cfg.lr, momentum=opt_cfg.momentum,
                                    weight_decay=opt_cfg.weight_decay, eps=opt_cfg.eps)
    scheduler = PolynomialLR(optimizer, config.TRAINING.scheduler.max_iter)
    criterion = FusionLoss(w_l1=config.TRAINING.loss.w_l1, w_l2=config.TRAINING.loss.w_l2, w_cos=config.TRAINING.loss.w_cos)
    grads = FlatGradientAllReduce(net)  # p.grad are views into one flat buffer
    accum = config.TRAINING.optimization.accumulation_steps
    losses, step = [], 0
    for epoch in range(config.TRAINING.n_epochs):
        database.reset()
        net.train()
        loader = _loader(dataset, shard.scenes)
        n_batches = len(loader)
        for i, batch in enumerate(loader):
            if not torch.all(torch.isfinite(batch['extrinsics'])):
                continue
            opt = config.TRAINING.optimization
            if opt.reset_strategy and np.random.random_sample() <= opt.reset_prob:
                database.reset(batch['frame_id'][0].split('/')[0])
            out = pipeline.fuse_training(_host_pose_batch(batch, device), database, device)
            loss = criterion(out['tsdf_fused'], out['tsdf_target'])
            if loss.grad_fn is not None:
                loss.backward()
                losses.append(float(loss.item()))
            if opt.clipping:
                torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1., norm_type=2)
            if (i + 1) % accum == 0 or i == n_batches - 1:
                grads.reduce()  # the single exchange step of the whole training path
                optimizer.step()
                grads.zero()    # keeps p.grad aliased to the flat buffer (no set_to_none)
                scheduler.step()
            step += 1
            if max_steps is not None and step >= max_steps:
                break
        if checkpoint_dir and rank == 0:
            os.makedirs(checkpoint_dir, exist_ok=True)
            torch.save({'epoch': epoch + 1, 'model_state': net.state_dict(), 'optimizer_state': optimizer.state_dict(),
                        'scheduler_state': scheduler.state_dict()}, os.path.join(checkpoint_dir, 'last.pth.tar'))
        if max_steps is not None and step >= max_steps:
            break
    log('rank {} mean loss {:.6f} over {} frames'.format(rank, float(np.mean(losses)) if losses else float('nan'), len(losses)))
    return pipeline, database, losses


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
