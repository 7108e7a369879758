"""Profiling helper (not a test): what a plain streaming kernel achieves on this device - torch copies / reads of buffers that fit the
256-MB Infinity Cache (what the frame's activation tensors do) and of buffers that do not.  The yardstick for the 'bytes at the
fabric / achievable rate' column of DESIGN.md 6.2."""
import time, torch
dev = torch.device('cuda:0')
def rate(fn, nbytes, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e12
for mb in (24, 64, 128, 1024, 4096):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device=dev).normal_(); b = torch.empty_like(a)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): b.copy_(a)
    cp = rate(g.replay, 2 * 4 * n * 10, 20)
    g2 = torch.cuda.CUDAGraph()
    s = torch.zeros((), device=dev)
    with torch.cuda.graph(g2):
        for _ in range(10): s = a.sum()
    rd = rate(g2.replay, 4 * n * 10, 20)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        for _ in range(10): b.fill_(1.0)
    wr = rate(g3.replay, 4 * n * 10, 20)
    print('%5d MB buffer: copy (read + write) %.2f TB/s, read-only (sum) %.2f TB/s, write-only (fill) %.2f TB/s' % (mb, cp, rd, wr))
