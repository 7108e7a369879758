"""Profiling helper (not a test): which torch operators (not libojf launches) the training frame step still runs, by
device time, with the Python line that issued them (torch.profiler)."""
import os, sys, runpy, torch
sys.argv = ['train_throughput.py']
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_throughput.py'))
step = ns['step']
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(8, 16):
        step(i)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by='cuda_time_total', row_limit=14, max_src_column_width=110))
