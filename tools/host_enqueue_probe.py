"""Profiling helper (not a test): host enqueue time per frame of the two-head (semantics) workload in a fresh process and
after another engine lived in the same process (bench.py's secondary legs run that way)."""
import os, sys, time, gc, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
base = dict(h=240, w=320, grid=256, semantics=True, strategy='gt', seg_engine='hip', mode='fast', arith='f16x3', n_classes=30)
def probe(tag, c):
    case = bench.Case(c, dev, 0, 80)
    with torch.no_grad():
        for i in range(20): case.fuse(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(200): case.fuse(20 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print('%-28s host enqueue %.3f ms/frame, with drain %.3f ms/frame' % (tag, (t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
    del case
    gc.collect()
    torch.cuda.empty_cache()
probe('fresh process', base)
probe('second engine', base)
probe('after an f32 engine', dict(base, arith='f32'))
probe('third engine', base)
