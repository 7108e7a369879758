"""Profiling helper (not a test): cProfile of the host side of the training frame step (which Python lines the enqueue
time goes to)."""
import os, sys, runpy, cProfile, pstats, torch
sys.argv = ['train_throughput.py']
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_throughput.py'))
step = ns['step']
pr = cProfile.Profile()
pr.enable()
for i in range(8, 40):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
