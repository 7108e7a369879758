"""Profiling helper (not a test): where the host threads of the overlapped training loop spend a frame (bench.TrainCase's step with timers):
the caller's thread per stage, and the gradient thread's jobs."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from online_joint_depthfusion_and_semantic_amd import train as T

def main():
    dev = torch.device('cuda:0')
    case = bench.TrainCase(240, 320, 256, dev, 0, 48)
    acc = {}
    def add(k, dt):
        acc[k] = acc.get(k, 0.0) + dt
    orig_submit = T.HipTrainNet._submit
    def timed_submit(self, fn):
        def wrapped():
            t0 = time.perf_counter()
            try:
                return fn()
            finally:
                add('gradient thread: job run time', time.perf_counter() - t0)
        return orig_submit(self, wrapped)
    T.HipTrainNet._submit = timed_submit
    stamps = []
    orig_bwd = T._NetFn.backward
    def bwd(ctx, dest):
        t0 = time.perf_counter()
        try:
            return orig_bwd(ctx, dest)
        finally:
            stamps.append(('netfn.backward', t0, time.perf_counter(), threading.current_thread().name))
    T._NetFn.backward = staticmethod(bwd)
    orig_submit2 = T.HipTrainNet._submit
    def stamped_submit(self, fn):
        def wrapped():
            t0 = time.perf_counter()
            try:
                return fn()
            finally:
                stamps.append(('job', t0, time.perf_counter(), threading.current_thread().name))
        return orig_submit2(self, wrapped)
    T.HipTrainNet._submit = stamped_submit
    def step(i):
        t0 = time.perf_counter()
        out = case.pipe.fuse_training(case.batches[i % len(case.batches)], case.db, case.dev)
        t1 = time.perf_counter()
        loss = case.crit.forward(out['tsdf_fused'], out['tsdf_target'])
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        stamps.append(('loss.backward', t2, t3, 'main'))
        boundary = (i + 1) % case.accum == 0
        def gs():
            if case.cfg.TRAINING.optimization.clipping:
                case.grads.clip_(1.0)
            if boundary:
                case.grads.reduce(); case.opt.step(); case.grads.zero()
        case.pipe.gradient_work(gs, join=boundary)
        t4 = time.perf_counter()
        add('fuse_training', t1 - t0); add('loss forward', t2 - t1); add('loss.backward', t3 - t2); add('gradient_work (submit / join at boundaries)', t4 - t3)
    for i in range(16):
        step(i)
    case.pipe.join_gradients(); torch.cuda.synchronize()
    acc.clear()
    n = 48
    t0 = time.perf_counter()
    for i in range(16, 16 + n):
        step(i)
    case.pipe.join_gradients()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    tn = case.pipe.__dict__['_hip_train']
    print('overlap %s thread %s: %.3f ms per frame (host loop %.3f)' % (tn.overlap, tn.overlap_thread, 1e3 * t / n, 1e3 * th / n))
    for k, v in acc.items():
        print('  %-50s %.3f ms/frame' % (k, 1e3 * v / n))
    base = [x for x in stamps if x[0] == 'loss.backward'][-6][1]
    for name, a, b, th in sorted(stamps, key=lambda x: x[1]):
        if a >= base:
            print('    %-16s %-22s start %8.3f ms  end %8.3f ms  (%.3f)' % (name, th, 1e3 * (a - base), 1e3 * (b - base), 1e3 * (b - a)))

main()
