"""Profiling helper (not a test): known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this
box (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").  Run under
rocprofv3 --pmc FETCH_SIZE (or WRITE_SIZE); tools/pmc_calibrate_summary.py reads the result.
 * copy:   y.copy_(x), 256 MB read + 256 MB written per launch, 16 B per lane, coalesced
 * fill:   y.fill_(1), 256 MB written
 * rec32:  32-byte records written by single lanes at shuffled record indices (the access pattern of the integrate
           kernel's record stores): 64 MB of records per launch"""
import torch
dev = torch.device('cuda:0')
n = 64 * 1024 * 1024
x = torch.ones(n, device=dev)
y = torch.empty(n, device=dev)
for _ in range(10):
    y.copy_(x)
for _ in range(10):
    y.fill_(1.0)
m = 2 * 1024 * 1024                       # records of 8 floats
perm = torch.randperm(m, device=dev)
src = torch.ones(m, 8, device=dev)
dst = torch.empty(m, 8, device=dev)
for _ in range(10):
    dst.index_copy_(0, perm, src)         # row r of src -> row perm[r] of dst: scattered 32-byte writes
torch.cuda.synchronize()
