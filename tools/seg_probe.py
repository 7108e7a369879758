"""Profiling helper (not a test): the SEGCONV engine alone at 320x240 (or H W) - N eager forwards or N graph replays,
for rocprofv3 --kernel-trace / --pmc passes (python tools/seg_probe.py [eager|graph] [N] [H W])."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd import _lib
from online_joint_depthfusion_and_semantic_amd.adapnet import AdapNet
from online_joint_depthfusion_and_semantic_amd.adapnet_engine import SegEngine
from online_joint_depthfusion_and_semantic_amd.config import default_config
mode = sys.argv[1] if len(sys.argv) > 1 else 'graph'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (240, 320)
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device('cuda:0')
cfg = default_config(H, W, semantics=True)
torch.manual_seed(0)
net = AdapNet(cfg.SEMANTIC_2D_MODEL).to(dev).eval()
net.no_resn50_dropout()
img = torch.rand(1, 3, H, W, device=dev) * 255; dep = torch.rand(1, H, W, device=dev) * 3
with torch.no_grad():
    eng = SegEngine(net)
    fn = (lambda: eng.predict(img, dep)) if B == 1 else (lambda: eng.predict_many([img] * B, [dep] * B))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    if mode == 'graph':
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            res = fn()
        run = g.replay
    else:
        run = fn
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): run()
    torch.cuda.synchronize()
    print('seg engine %s, %d frame(s) per pass: %.3f ms per pass (%d), guard rc %d' % (mode, B, (time.perf_counter() - t0) / N * 1e3, N,
          _lib.load().ojf_net_check(_lib.stream_ptr(dev))))
