"""Profiling helper (not a test): times ojf_integrate's accumulate stage under OJF_ABLATE / OJF_INTEGRATE_DIRECT."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from online_joint_depthfusion_and_semantic_amd import ops
from helpers import frame_inputs, make_stream, fresh_volumes, to_cuda

dev = torch.device('cuda:0')
h, w, grid = 240, 320, 256
st = make_stream(h, w, grid, 40)
g = to_cuda(fresh_volumes(grid, False), dev)
ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, ops.MODE_FAST, dev)
fis = [frame_inputs(st, i) for i in range(8)]
dv = [(torch.from_numpy(f['fd']).to(dev), torch.from_numpy(f['est']).to(dev)) for f in fis]
def run(i):
    f = fis[i % 8]
    ops.integrate(dv[i % 8][0], f['Ki'], f['E'], st.origin, st.resolution, dv[i % 8][1], g['tsdf'], g['wgt'], ws)
for i in range(8): run(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(80): run(i)
torch.cuda.synchronize()
print('ABLATE=%s DIRECT=%s: %.1f us per integrate call' % (os.environ.get('OJF_ABLATE'), os.environ.get('OJF_INTEGRATE_DIRECT'), (time.perf_counter() - t0) / 80 * 1e6))
