"""Multi-GPU layer: one process per GPU, scenes sharded across ranks, RCCL only for the
training-time gradient exchange.

* Inference needs NO collective: every rank holds a replica of the net weights and the volumes of
  its own scenes (``shard_scenes``); a scene is never split across GPUs (its rays hit the same voxels).
* Training (train_fusion.py:145-189 accumulates 8 frames, clips, steps): the only exchange step is
  ONE all-reduce(sum) over a single flat fp32 buffer holding all gradients of the fusion net
  (360 591 / 571 833 elements = 1.44 / 2.29 MB) at each accumulation boundary
  (``FlatGradientAllReduce``), followed by the identical optimizer step on every rank.  On the xGMI
  mesh this message is latency-bound (~26 us of wire time for a ring), so one fused buffer - not
  per-parameter buckets - is the right shape.
Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force_group=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).
    Returns (rank, world_size, local_rank).  World size 1 needs no process group; ``force_group`` (or
    ``OJF_DIST_FORCE_GROUP=1``) creates one anyway - the same RCCL load / communicator set-up / all-reduce launch as on the
    8-GPU node, on a one-GPU box (bench.py --train --force-group, tests/test_bench_gpu.py)."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if force_group is None:
        force_group = os.environ.get('OJF_DIST_FORCE_GROUP', '0') not in ('', '0')
    if (world > 1 or force_group) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if world == 1:  # nobody launched us: a rendezvous with ourselves
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if 'MASTER_PORT' not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def pin_rank_to_cores(rank, world):
    """One disjoint slice of the process's allowed cores per rank (cores / ranks, contiguous): the training step is host-paced
    (292 launches per frame), and N unpinned Python processes migrating over one socket's cores show up as a scaling loss that
    is not the GPUs'.  Returns the slice as a string for the bench line ('' where the platform has no affinity call)."""
    if world <= 1 or not hasattr(os, 'sched_setaffinity'):
        return ''
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // world
    if per < 1:
        return ''
    mine = cores[rank * per:(rank + 1) * per]
    os.sched_setaffinity(0, mine)
    try:
        torch.set_num_threads(max(1, min(torch.get_num_threads(), per)))
    except RuntimeError:
        pass
    return '%d-%d' % (mine[0], mine[-1]) if mine == list(range(mine[0], mine[-1] + 1)) else ','.join(map(str, mine))


def shard_scenes(scenes, rank, world):
    """Scene s (in list order) lives on rank ``index mod world``."""
    return [s for i, s in enumerate(scenes) if i % world == rank]


class ShardedScenes:
    """Dataset facade exposing only this rank's scenes to ``Database`` (which reads
    ``dataset.scenes`` and ``dataset.get_grid``, modules/database.py:48-53)."""

    def __init__(self, dataset, rank, world):
        self._dataset = dataset
        self.scenes = shard_scenes(list(dataset.scenes), rank, world)

    def get_grid(self, *a, **k):
        return self._dataset.get_grid(*a, **k)

    def create_grid(self, *a, **k):
        return self._dataset.create_grid(*a, **k)


class FlatGradientAllReduce:
    """Gradients of ``module`` live as views into one contiguous fp32 buffer; ``reduce()`` issues a
    single all-reduce over it (averaging by default so that the step matches a world-size-times
    larger accumulation window)."""

    def __init__(self, module, average=True, group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.average = average
        self.group = group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:  # p.grad becomes a view: autograd accumulates straight into the buffer
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def zero(self):
        self.flat.zero_()

    def clip_(self, max_norm=1.0, eps=1e-6):
        """``torch.nn.utils.clip_grad_norm_(module.parameters(), max_norm, 2)`` (train_fusion.py:182-183) on the flat
        buffer: the 2-norm of all gradients is the 2-norm of the buffer, so the clip is three small launches and no
        Python walk over 230 tensors (the per-frame clip cost the training step ~0.3 ms of host time).  Same formula
        (coef = max_norm / (norm + 1e-6), clamped to 1); the norm is summed in one pass instead of per tensor, i.e. it
        differs from torch's in the last bits only.  Returns the norm (a 0-d tensor: no host read)."""
        norm = torch.linalg.vector_norm(self.flat, 2)
        self.flat.mul_(torch.clamp(max_norm / (norm + eps), max=1.0))
        return norm

    def reduce(self):
        # (a group of ONE rank - init_from_env(force_group=True) - still goes through the collective: the identity, and the
        # first RCCL launch of this code path on a one-GPU box)
        if dist.is_available() and dist.is_initialized():
            if self.flat.is_cuda and dist.get_backend(self.group) == 'gloo':
                # dry runs of the schedule on one device (bench.py --dist-backend gloo, tests): gloo reduces host tensors
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average and dist.get_world_size(self.group) > 1:
                self.flat.div_(dist.get_world_size(self.group))
        return self.flat
