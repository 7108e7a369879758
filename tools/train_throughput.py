"""Profiling helper (not a test): frames/s of the training frame step (fuse_training + FusionLoss + backward +
RMSprop step every 8 frames, train_fusion.py:166-189) at 320x240 -> 256^3 on one GPU."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.config import default_config, database_config
from online_joint_depthfusion_and_semantic_amd.database import Database
from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline
from online_joint_depthfusion_and_semantic_amd.loss import FusionLoss
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream
from online_joint_depthfusion_and_semantic_amd.drivers import _training_defaults
dev = torch.device('cuda:0')
h, w, grid, n = 240, 320, 256, 48
cfg = _training_defaults(default_config(h, w)); cfg.SETTINGS.device = str(dev)
cfg.FUSION_MODEL.train_overlap = os.environ.get('OJF_TRAIN_OVERLAP', '1') != '0'  # like drivers.train_fusion (0: the serial loop)
st = SyntheticStream(h, w, grid, n)
db = Database(st, database_config(cfg))
pipe = Pipeline(cfg).to(dev).train()
crit = FusionLoss(w_l1=cfg.TRAINING.loss.w_l1, w_l2=cfg.TRAINING.loss.w_l2, w_cos=cfg.TRAINING.loss.w_cos)
opt = torch.optim.RMSprop(pipe._fusion_network.parameters(), lr=1e-4)
bs = []
for i in range(n):
    f = st.frame(i)
    bs.append({'image': torch.zeros((1, 3, h, w), device=dev), 'frame_id': [f['frame_id']],
               'tof_depth': torch.from_numpy(f['tof_depth'])[None].to(dev), 'mask': torch.from_numpy(f['mask'])[None].to(dev),
               'extrinsics': torch.from_numpy(f['extrinsics'])[None], 'intrinsics': torch.from_numpy(f['intrinsics'])[None]})
def step(i):
    out = pipe.fuse_training(bs[i], db, dev)
    loss = crit.forward(out['tsdf_fused'], out['tsdf_target']) if out['tsdf_fused'].shape[1] else None
    if loss is not None: loss.backward()
    if (i + 1) % 8 == 0:
        with pipe.gradients():
            opt.step(); opt.zero_grad(set_to_none=False)
for i in range(8): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(8, n): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('training frame step: %.1f frames/s (%.2f ms/frame; host enqueue alone %.2f ms/frame)' % ((n - 8) / dt, dt / (n - 8) * 1e3, (t1 - t0) / (n - 8) * 1e3))
tn = pipe.__dict__.get('_hip_train')
if tn is not None and tn._trainers:
    from online_joint_depthfusion_and_semantic_amd import _lib
    print('executor launches in the last backward pass: %d' % _lib.load().ojf_trainer_launch_count(next(iter(tn._trainers.values())).handle))
