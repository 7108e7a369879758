"""Profiling helper (not a test): a few forwards of the HIP fusion net at config B for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.config import default_config
from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline
dev = torch.device('cuda:0')
h, w = 240, 320
cfg = default_config(h, w)
pipe = Pipeline(cfg)
torch.manual_seed(0)
for m in pipe._fusion_network.modules():
    if isinstance(m, torch.nn.Conv2d): torch.nn.init.xavier_normal_(m.weight)
pipe = pipe.to(dev).eval()
eng = pipe._get_engine(h, w, dev)
fv = torch.rand(h * w, 9, device=dev) * 0.1
fw = torch.rand(h * w, 9, device=dev)
d = torch.rand(h, w, device=dev) * 3
est = torch.empty(h * w, 9, device=dev)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    eng.prepare_input(fv, fw, d)
    eng.forward(est)
torch.cuda.synchronize()
print('ok', float(est.abs().mean()))
