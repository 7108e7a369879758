"""Profiling helper (not a test): SegEngine forward x12 for rocprofv3 --kernel-trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.adapnet import AdapNet
from online_joint_depthfusion_and_semantic_amd.adapnet_engine import SegEngine
from online_joint_depthfusion_and_semantic_amd.config import default_config
dev = torch.device('cuda:0')
cfg = default_config(240, 320, semantics=True)
torch.manual_seed(0)
net = AdapNet(cfg.SEMANTIC_2D_MODEL).to(dev).eval()
net.no_resn50_dropout()
img = torch.randn(1, 3, 240, 320, device=dev); dep = torch.rand(1, 3, 240, 320, device=dev) * 3
with torch.no_grad():
    eng = SegEngine(net)
    for _ in range(12):
        torch.softmax(eng(img, dep), 1).max(1)
torch.cuda.synchronize()
