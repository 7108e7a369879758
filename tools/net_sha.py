"""Profiling helper (not a test): sha256 of the HIP fusion net's output at a frame size + its time per forward pass and the per-kernel
profile (python tools/net_sha.py [h w [sem]]): A/B of switches that must not change a bit (OJF_CONV_ROW_PERM, OJF_NO_XCD_BAND, ...)."""
import hashlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.config import default_config
from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline
dev = torch.device('cuda:0')
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (240, 320)
sem = len(sys.argv) > 3 and sys.argv[3] == 'sem'
cfg = default_config(h, w, semantics=sem, use_semantics=sem)
torch.manual_seed(0)
pipe = Pipeline(cfg)
for m in pipe._fusion_network.modules():
    if isinstance(m, torch.nn.Conv2d): torch.nn.init.xavier_normal_(m.weight)
pipe = pipe.to(dev).eval()
eng = pipe._get_engine(h, w, dev)
g = torch.Generator(device=dev); g.manual_seed(1)
fv = (torch.rand(9, h * w, device=dev, generator=g) * 0.2 - 0.1).contiguous()
fw = torch.rand(9, h * w, device=dev, generator=g).contiguous()
d = (torch.rand(h, w, device=dev, generator=g) * 3).contiguous()
ids = (torch.rand(h * w, device=dev, generator=g) * 30).to(torch.uint8)
est = torch.empty(h * w, 9, device=dev)
for _ in range(5):
    eng.prepare_input(fv, fw, d, ids if sem else None, 30 if sem else 0, planes=True)
    eng.forward(est)
torch.cuda.synchronize()
sha = hashlib.sha256(est.cpu().numpy().tobytes()).hexdigest()[:16]
t0 = time.perf_counter()
for _ in range(200): eng.forward(est)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 200 * 1e6
prof = {}
for _ in range(5):
    for name, t in eng.profile(est):
        prof.setdefault(name, []).append(t)
print('%dx%d%s: est sha %s  %.1f us per forward  %s' % (h, w, ' sem' if sem else '', sha, us,
      ' '.join('%s=%s' % (a, b) for a, b in sorted(os.environ.items()) if a.startswith('OJF_'))))
n = 5
print('   ' + ' | '.join('%s %.1f' % (k[:34], sum(v) / n) for k, v in sorted(prof.items(), key=lambda kv: -sum(kv[1]))))
