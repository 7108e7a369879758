"""Profiling helper (not a test): fwd+bwd time of every convolution shape of FusionNet_v3 in torch (MIOpen) at 320x240,
to find the ones MIOpen serves with naive fall-back kernels (training path)."""
import time, torch, torch.nn.functional as F
dev = torch.device('cuda:0')
def bench(fn, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
shapes = [(19 * k, 19, 3, 1) for k in range(1, 6)] + [(19, 19, 3, d) for d in (3, 9, 27)] + \
         [(114, 19, 1, 1), (19, 114, 1, 1), (570, 114, 1, 1), (114, 114, 1, 1), (114, 95, 1, 1), (95, 76, 1, 1), (38, 19, 1, 1), (19, 9, 1, 1)]
for cin, cout, k, d in shapes:
    x = torch.randn(1, cin, 240, 320, device=dev, requires_grad=True)
    w = torch.randn(cout, cin, k, k, device=dev, requires_grad=True); b = torch.randn(cout, device=dev, requires_grad=True)
    def f():
        y = F.conv2d(x, w, b, padding=d * (k // 2), dilation=d); y.sum().backward()
    def g():
        with torch.no_grad(): F.conv2d(x, w, b, padding=d * (k // 2), dilation=d)
    print('%4d -> %3d k%d d%-2d  fwd %.2f ms  fwd+bwd %.2f ms' % (cin, cout, k, d, bench(g), bench(f)))
