"""Profiling helper (not a test): two independent training frame steps (two scenes, two pipelines, two streams) enqueued alternately by one
host thread - how much of a training step's device time is dependent-launch latency that a second chain can fill."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    dev = torch.device('cuda:0')
    h, w, grid = 240, 320, 256
    steps, warm = 48, 16
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cases = [bench.TrainCase(h, w, grid, dev, r, steps) for r in range(n)]
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    def loop(a, b):
        for i in range(a, b):
            for c, s in zip(cases, streams):
                with torch.cuda.stream(s):
                    c.step(i)
    loop(0, warm)
    torch.cuda.synchronize()
    res = []
    at = warm
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(at, at + steps)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        res.append((n * steps / t, 1e3 * t / steps, 1e3 * th / steps))
        at += steps
    res.sort()
    print('chains %d: aggregate %.1f frames/s, %.3f ms per round of %d frames, host loop %.3f ms per round' % ((n,) + res[1][:2] + (n, res[1][2])))

main()
