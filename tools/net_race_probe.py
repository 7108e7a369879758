"""Stress helper (not a test): the generic (non-chain) two-head topology of tests/test_net_gpu.py::test_fusion_net_other_topologies,
many forwards of one engine and many engines in one process; any forward whose output differs from the first is reported."""
import os, sys, torch
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd import model
from online_joint_depthfusion_and_semantic_amd.engine import FusionNetEngine
cuda = torch.device('cuda:0')
h, w = 40, 56
bad = 0
noise_stream = torch.cuda.Stream()
A = torch.randn(2048, 2048, device=cuda)
NOISE = len(sys.argv) > 3 and sys.argv[3] == 'noise'
for version, sem, n_points, growth in (('v3', True, 3, 3), ('v3', False, 5, 4), ('v2', True, 7, 5)):
    for arith in ('f16x3', 'f32'):
        cfg = NS(n_points=n_points, growth_factor=growth, use_semantics=sem, output_scale=1.0, resx=w, resy=h)
        torch.manual_seed(11)
        net = getattr(model, 'FusionNet_' + version)(cfg)
        for m in net.modules():
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.xavier_normal_(m.weight); m.bias.data.normal_(0, 0.05)
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
        net = net.eval()
        g = torch.Generator().manual_seed(2)
        fv = ((torch.rand(h * w, n_points, generator=g) - 0.5) * 0.2).to(cuda)
        fw = (torch.rand(h * w, n_points, generator=g) * 4).to(cuda)
        fr = (torch.rand(h, w, generator=g) * 4).to(cuda)
        sem_ids = torch.randint(0, 30, (h, w), generator=g, dtype=torch.uint8).to(cuda)
        first = None
        for e in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
            junk = torch.full((1 << 22,), float('nan'), device=cuda)  # poison what the allocator hands out next
            del junk
            eng = FusionNetEngine(net, h, w, cuda, arithmetic=arith)
            for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
                if NOISE and it % 3 == 0:
                    with torch.cuda.stream(noise_stream):
                        B = A @ A
                eng.prepare_input(fv, fw, fr, sem_ids if sem else None, 30)
                est = torch.full((h * w, n_points), float('nan'), device=cuda)
                eng.forward(est)
                try:
                    eng.check()
                except Exception as ex:
                    print('CHECK RAISED', version, sem, arith, e, it, ex); bad += 1
                if first is None:
                    first = est.clone()
                elif not torch.equal(first, est):
                    d = (first - est).abs()
                    print('DIFFERS', version, sem, arith, 'engine', e, 'iter', it, 'max', float(d.nan_to_num(1e9).max()), 'n', int((d != 0).sum()), 'nan', int(torch.isnan(est).sum()))
                    bad += 1
            eng.close()
        print('done', version, sem, arith, 'bad so far', bad, flush=True)
print('TOTAL BAD', bad)
