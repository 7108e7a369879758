"""Profiling helper (not a test): where the host time of Pipeline.fuse goes (cProfile, GPU never the bottleneck here)."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.config import default_config, database_config
from online_joint_depthfusion_and_semantic_amd.database import Database
from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream
dev = torch.device('cuda:0')
h, w, grid = 120, 160, 64
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
    torch.cuda.synchronize()
    eng = pipe._engine
    t0 = time.perf_counter()
    for i in range(200): eng.forward(pipe._est)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('net forward only: host %.1f us/call, total %.1f us/call' % ((t1 - t0) * 5e3, (t2 - t0) * 5e3))
    pr = cProfile.Profile(); pr.enable()
    for i in range(20, 120): pipe.fuse(bs[i], db, dev)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
