"""Profiling helper (not a test): SegEngine forward x12 for rocprofv3 --kernel-trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.adapnet import AdapNet
from online_joint_depthfusion_and_semantic_amd.adapnet_engine import SegEngine
from online_joint_depthfusion_and_semantic_amd.config import default_config
dev = torch.device('cuda:0')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (240, 320)
cfg = default_config(H, W, semantics=True)
torch.manual_seed(0)
net = AdapNet(cfg.SEMANTIC_2D_MODEL).to(dev).eval()
net.no_resn50_dropout()
img = torch.randn(1, 3, H, W, device=dev); dep = torch.rand(1, 3, H, W, device=dev) * 3
with torch.no_grad():
    eng = SegEngine(net)
    for _ in range(12):
        torch.softmax(eng(img, dep), 1).max(1)
torch.cuda.synchronize()
