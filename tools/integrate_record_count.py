"""Profiling helper (not a test): records per touched voxel of the FAST integrate on the bench stream (stats counters)."""
import sys, torch, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import frame_inputs, fresh_volumes, to_cuda
from online_joint_depthfusion_and_semantic_amd import ops
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream
dev = torch.device('cuda:0')
h, w, grid = 240, 320, 256
st = SyntheticStream(h, w, grid, 40, scene='room_0', seed=1911)
vols = to_cuda(fresh_volumes(grid, False), dev)
ws = ops.IntegrateWorkspace((grid,) * 3, h, w, 7, ops.MODE_FAST, dev)
for i in (0, 3, 10, 20, 30):
    fi = frame_inputs(st, i)
    ops.integrate(torch.from_numpy(fi['fd']).to(dev), fi['Ki'], fi['E'], st.origin, st.resolution, torch.from_numpy(fi['est']).to(dev), vols['tsdf'], vols['wgt'], ws, stats=True)
    s = ws.stats.cpu().numpy()
    print('frame %d: touched %d entries %d records %d -> %.2f records per touched voxel' % (i, s[0], s[1], s[2], s[2] / max(s[0], 1)))
