"""Profiling helper (not a test): the HOST's share of the training frame step with an EMPTY device queue - every frame is
synchronised before it starts, so a launch never waits for queue space and the count read never waits for earlier frames:
what is left is Python + ctypes + the runtime's launch path.  Split by phase of the step (train_fusion.py:166-189)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd import _lib
from online_joint_depthfusion_and_semantic_amd.config import default_config, database_config
from online_joint_depthfusion_and_semantic_amd.database import Database
from online_joint_depthfusion_and_semantic_amd.distributed import FlatGradientAllReduce
from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline
from online_joint_depthfusion_and_semantic_amd.loss import FusionLoss
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream
from online_joint_depthfusion_and_semantic_amd.drivers import _training_defaults
dev = torch.device('cuda:0')
h, w, grid, n = 240, 320, 256, 40
cfg = _training_defaults(default_config(h, w)); cfg.SETTINGS.device = str(dev)
st = SyntheticStream(h, w, grid, n)
from bench import BenchStream  # (constant GT grid: no minute of host time for the analytic scene)
st = BenchStream(h, w, grid, n)
db = Database(st, database_config(cfg))
pipe = Pipeline(cfg).to(dev).train()
crit = FusionLoss(w_l1=cfg.TRAINING.loss.w_l1, w_l2=cfg.TRAINING.loss.w_l2, w_cos=cfg.TRAINING.loss.w_cos)
grads = FlatGradientAllReduce(pipe._fusion_network)
opt = torch.optim.RMSprop(pipe._fusion_network.parameters(), lr=1e-4)
bs = []
for i in range(n):
    f = st.frame(i)
    bs.append({'image': torch.zeros((1, 3, h, w), device=dev), 'frame_id': [f['frame_id']],
               'tof_depth': torch.from_numpy(f['tof_depth'])[None].to(dev), 'mask': torch.from_numpy(f['mask'])[None].to(dev),
               'extrinsics': torch.from_numpy(f['extrinsics'])[None], 'intrinsics': torch.from_numpy(f['intrinsics'])[None]})
T = {'fuse_training (count, 2 x extract, net forward, fuse output, integrate)': 0.0, 'loss forward': 0.0, 'loss.backward (net backward)': 0.0,
     'clip': 0.0, 'all-reduce + optimizer step + zero (per 8 frames, amortised)': 0.0, 'device drain after the last enqueue': 0.0}
keys = list(T)
def step(i, timed):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    out = pipe.fuse_training(bs[i], db, dev); t.append(time.perf_counter())
    loss = crit.forward(out['tsdf_fused'], out['tsdf_target']); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    grads.clip_(1.0); t.append(time.perf_counter())
    if (i + 1) % 8 == 0:
        grads.reduce(); opt.step(); grads.zero()
    t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    if timed:
        for k, a, b in zip(keys, t[:-1], t[1:]): T[k] += b - a
for i in range(8): step(i, False)
for i in range(8, n): step(i, True)
m = n - 8
host = sum(T[k] for k in keys[:-1]) / m * 1e3
print('training frame step, one frame at a time on an empty queue (%d frames):' % m)
for k in keys: print('  %-78s %.3f ms/frame' % (k, T[k] / m * 1e3))
print('  host enqueue total %.3f ms/frame; + drain %.3f = %.3f ms/frame serial' % (host, T[keys[-1]] / m * 1e3, host + T[keys[-1]] / m * 1e3))
tn = pipe.__dict__.get('_hip_train')
if tn is not None and tn._trainers:
    tr = next(iter(tn._trainers.values()))
    print('  executor launches: forward %d, backward %d' % (tr.fwd_launches, tr.bwd_launches))
