#!/usr/bin/env python
"""Build-container-only calibration of the cpu_baseline port against the imported reference:
same outputs, comparable wall time.  Needs /root/reference.

    python oracle/calibrate_port.py [h w grid frames]
"""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as mg  # noqa: E402  (stubs torchvision, imports the reference Pipeline)
from oracle import torch_port  # noqa: E402
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream, gt_volumes  # noqa: E402


def main():
    h, w, grid, frames = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (240, 320, 256, 4)))
    torch.set_num_threads(int(os.environ.get('OJF_THREADS', torch.get_num_threads())))
    cfg = mg.ref_config(h, w, False, False)
    pipe = mg.RefPipeline(cfg)
    mg.seeded_state(pipe._fusion_network, 11)
    pipe.eval()
    st = SyntheticStream(h, w, grid, 20)
    db = mg.DuckDatabase(st, False, np.zeros((2, 2, 2), np.float16))
    vols = dict(tsdf=torch.full((grid,) * 3, 0.1, dtype=torch.float16), wgt=torch.zeros((grid,) * 3, dtype=torch.float16))
    t_ref, t_port = [], []
    with torch.no_grad():
        for i in range(frames):
            b = st.batch(i)
            t0 = time.perf_counter()
            pipe.fuse(b, db, torch.device('cpu'))
            t1 = time.perf_counter()
            b = st.batch(i)
            t2 = time.perf_counter()
            torch_port.fuse(b, vols, pipe._fusion_network, torch.from_numpy(st.origin), st.resolution)
            t3 = time.perf_counter()
            t_ref.append(t1 - t0)
            t_port.append(t3 - t2)
            same = bool((db.scenes_est[st.scene].volume.view(torch.int16) == vols['tsdf'].view(torch.int16)).all())
            print('frame %d: reference %.3f s, port %.3f s, identical TSDF volume: %s' % (i, t1 - t0, t3 - t2, same))
    r, p = np.mean(t_ref[1:]), np.mean(t_port[1:])
    print('threads=%d  reference %.3f s/frame (%.3f fps)  port %.3f s/frame (%.3f fps)  ratio port/reference = %.3f'
          % (torch.get_num_threads(), r, 1 / r, p, 1 / p, p / r))


if __name__ == '__main__':
    main()
