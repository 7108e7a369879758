"""Profiling helper (not a test): host-side enqueue time of Pipeline.fuse vs GPU time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.config import default_config, database_config
from online_joint_depthfusion_and_semantic_amd.database import Database
from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream
dev = torch.device('cuda:0')
h, w, grid = 240, 320, 256
cfg = default_config(h, w); cfg.SETTINGS.device = str(dev)
st = SyntheticStream(h, w, grid, 140)
db = Database(st, database_config(cfg))
pipe = Pipeline(cfg).to(dev).eval()
bs = []
img = torch.zeros((1, 3, h, w), device=dev)
for i in range(120):
    f = st.frame(i)
    bs.append({'image': img, 'frame_id': [f['frame_id']], 'tof_depth': torch.from_numpy(f['tof_depth'])[None].to(dev),
               'mask': torch.from_numpy(f['mask'])[None].to(dev), 'extrinsics': torch.from_numpy(f['extrinsics'])[None],
               'intrinsics': torch.from_numpy(f['intrinsics'])[None]})
with torch.no_grad():
    for i in range(20): pipe.fuse(bs[i], db, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20, 120): pipe.fuse(bs[i], db, dev)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host enqueue %.3f ms/frame, total %.3f ms/frame' % ((t1 - t0) * 10, (t2 - t0) * 10))
