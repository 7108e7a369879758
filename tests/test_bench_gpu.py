"""-m gpu: bench.py's launch paths.  ``python bench.py --gpus N`` must start its own N ranks when no
launcher did (the driver's SCALE command is the plain ``python bench.py --gpus N ...``), print exactly one JSON line
from rank 0, and ``--train`` must time BASELINE configs[3]'s frame step including the gradient all-reduce.  Two ranks on
RCCL need two devices: those cases skip on a one-GPU box, where the same code path is exercised over gloo (both ranks
share device 0 - what is validated is launch / rendezvous / barrier / max-over-ranks, not the wire)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--height', '120', '--width', '160', '--grid', '64', '--cpu-frames', '0', '--secondary', '0']


def _bench(*flags, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags), capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_single_gpu_line_has_contract_fields_and_repeats(cuda):
    out = _bench('--steps', '6', '--warmup', '3', '--repeats', '3', *SMALL)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'roofline_net', 'roofline_hbm', 'kernels'):
        assert k in out, k
    assert out['n_gpus'] == 1 and out['steps'] == 6 and out['warmup'] == 3 and out['repeats'] == 3
    assert out['value_min'] <= out['value'] <= out['value_max']
    assert abs(out['value'] * out['ms_per_step'] / 1e3 - 1.0) < 1e-6


def test_gpus_2_launches_its_own_ranks_over_gloo_on_one_device(cuda):
    out = _bench('--gpus', '2', '--dist-backend', 'gloo', '--steps', '6', '--warmup', '3', '--repeats', '2', *SMALL)
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak'
    assert out['config']['parallelism'] == 'scene-sharded x2'
    assert abs(out['value'] * out['ms_per_step'] / 1e3 - 2.0) < 1e-6  # whole-job frames/s = 2 ranks x steps / max time


def test_gpus_more_than_devices_fails_loudly_on_rccl(cuda):
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and 'HIP device(s) visible' in r.stderr


def test_train_flag_times_the_configs3_frame_step(cuda):
    out = _bench('--train', '--steps', '8', '--warmup', '8', '--repeats', '2', '--height', '48', '--width', '64', '--grid', '64')
    assert out['n_gpus'] == 1 and 'configs[3]' in out['config']['workload']
    assert out['allreduce_calls_in_timed_region'] == 2 and out['gradients_finite'] is True
    assert out['value'] > 0 and out['gradient_bytes'] > 1e6


def test_train_gpus_2_over_gloo_dry_runs_the_training_schedule_on_one_device(cuda):
    """the TRAINING schedule (self-launch, barrier, accumulation boundary on the common counter, the
    all-reduce call, max over ranks) as a two-rank gloo dry run on one device; the gradient buffer is staged through the host
    for gloo and the JSON line says so."""
    out = _bench('--train', '--gpus', '2', '--dist-backend', 'gloo', '--steps', '8', '--warmup', '8', '--repeats', '2',
                 '--height', '48', '--width', '64', '--grid', '64')
    assert out['n_gpus'] == 2 and 'configs[3]' in out['config']['workload']
    assert out['allreduce_calls_in_timed_region'] == 2 and out['gradients_finite'] is True
    assert out['allreduce_backend'].startswith('gloo') and out['allreduce_us'] > 0
    assert abs(out['value'] * out['ms_per_step'] / 1e3 - 2.0) < 1e-6


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='two ranks on RCCL need two HIP devices')
@pytest.mark.parametrize('train', [False, True])
def test_gpus_2_over_rccl(cuda, train):
    flags = ['--gpus', '2', '--steps', '8', '--warmup', '8', '--repeats', '2'] + SMALL + (['--train'] if train else [])
    out = _bench(*flags)
    assert out['n_gpus'] == 2
    if train:
        assert out['allreduce_backend'] == 'rccl' and out['allreduce_us'] > 0 and out['gradients_finite'] is True


def test_train_force_group_runs_the_gradient_all_reduce_through_rccl_on_one_rank(cuda):
    """the first RCCL load + communicator + all-reduce launch of this code on hardware - a process group
    of ONE rank (nccl backend) around the training frame step; the flat 1.44-MB gradient buffer goes through
    ``dist.all_reduce`` at every accumulation boundary (FlatGradientAllReduce.reduce; insertion point in the reference:
    train_fusion.py:186-189).  The line says which backend ran and how long the collective took."""
    out = _bench('--train', '--force-group', '--steps', '16', '--warmup', '8', '--repeats', '2')
    assert out['n_gpus'] == 1 and out['allreduce_backend'].startswith('rccl') and 'one rank' in out['allreduce_backend']
    assert out['allreduce_calls_in_timed_region'] == 4 and out['gradients_finite'] is True
    assert out['allreduce_us'] > 0 and 1.4e6 < out['gradient_bytes'] < 1.5e6
    assert len(out['per_rank']['ms_per_step']) == 1 and out['per_rank']['ms_per_step_min'] > 0
    print('RCCL on one rank: all-reduce of %d bytes %.1f us, training step %.1f frames/s' % (out['gradient_bytes'], out['allreduce_us'], out['value']))


def test_gpus_2_pins_the_ranks_to_disjoint_cores(cuda):
    """N > 1 ranks pin themselves to cores / ranks disjoint slices (the training leg is host-paced)."""
    if len(os.sched_getaffinity(0)) < 2:
        pytest.skip('one core')
    out = _bench('--train', '--gpus', '2', '--dist-backend', 'gloo', '--steps', '8', '--warmup', '8', '--repeats', '2',
                 '--height', '48', '--width', '64', '--grid', '64')
    assert out['config']['cpu_affinity'] not in ('', 'not pinned')
    assert len(out['per_rank']['ms_per_step']) == 2 and out['per_rank']['ms_per_step_max'] >= out['per_rank']['ms_per_step_min'] > 0
