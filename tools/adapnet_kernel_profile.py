"""Profiling helper (not a test): AdapNet++ eager forward at 320x240 for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.adapnet import AdapNet
from online_joint_depthfusion_and_semantic_amd.config import default_config
dev = torch.device('cuda:0')
cfg = default_config(240, 320, semantics=True)
net = AdapNet(cfg.SEMANTIC_2D_MODEL).to(dev).eval()
img = torch.randn(1, 3, 240, 320, device=dev); dep = torch.rand(1, 3, 240, 320, device=dev) * 3
with torch.no_grad():
    for _ in range(20):
        torch.softmax(net(img, dep)[0], dim=1).max(dim=1)
torch.cuda.synchronize()
