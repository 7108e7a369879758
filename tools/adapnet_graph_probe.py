"""Profiling helper (not a test): AdapNet++ forward at 320x240 - eager vs hipGraph replay vs channels_last / fp16."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.adapnet import AdapNet
from online_joint_depthfusion_and_semantic_amd.config import default_config
dev = torch.device('cuda:0')
cfg = default_config(240, 320, semantics=True)
net = AdapNet(cfg.SEMANTIC_2D_MODEL).to(dev).eval()
img = torch.randn(1, 3, 240, 320, device=dev); dep = torch.rand(1, 3, 240, 320, device=dev) * 3

def run(n, x, y):
    with torch.no_grad():
        return torch.softmax(n(x, y)[0], dim=1).max(dim=1)

def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

print('eager fp32          %.2f ms' % timeit(lambda: run(net, img, dep)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): run(net, img, dep)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run(net, img, dep)
print('graph replay fp32   %.2f ms' % timeit(lambda: g.replay()))
net_cl = net.to(memory_format=torch.channels_last)
x_cl, y_cl = img.contiguous(memory_format=torch.channels_last), dep.contiguous(memory_format=torch.channels_last)
print('eager channels_last %.2f ms' % timeit(lambda: run(net_cl, x_cl, y_cl)))
with torch.autocast('cuda', dtype=torch.float16):
    print('eager fp16 autocast %.2f ms' % timeit(lambda: run(net_cl, x_cl, y_cl)))

def capture(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out

def half_run():
    with torch.autocast('cuda', dtype=torch.float16):
        return run(net_cl, x_cl, y_cl)
g2, out2 = capture(lambda: run(net_cl, x_cl, y_cl))
print('graph channels_last fp32 %.2f ms' % timeit(lambda: g2.replay()))
g3, out3 = capture(half_run)
print('graph channels_last fp16 %.2f ms' % timeit(lambda: g3.replay()))
ref = run(net_cl, x_cl, y_cl)
g3.replay(); torch.cuda.synchronize()
print('fp16 vs fp32: max score diff %.2e, id agreement %.4f' % (float((out3[0].float() - ref[0]).abs().max()), float((out3[1] == ref[1]).float().mean())))

import copy
try:
    from torch.fx.experimental.optimization import fuse
    fused = fuse(copy.deepcopy(net).eval())
    ref = run(net, img, dep)
    outf = run(fused, img, dep)
    print('fx conv-bn fused: max score diff %.2e' % float((outf[0] - ref[0]).abs().max()))
    g4, out4 = capture(lambda: run(fused, img, dep))
    print('graph fused fp32    %.2f ms' % timeit(lambda: g4.replay()))
except Exception as e:
    print('fx fuse failed:', repr(e)[:300])
