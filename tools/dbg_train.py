"""Profiling / debugging helper (not a test): HIP training path and torch's own fp32 autograd, both against the float64 module."""
import sys, copy, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_train_gpu import _net
from online_joint_depthfusion_and_semantic_amd.train import HipTrainNet
cuda = torch.device('cuda:0')
h, w = 40, 56
def loss(e, t): return (e - t).abs().mean() + 10 * ((e - t) ** 2).mean()
for training in (False, True):
    net = _net('v3', False, h, w)
    ref = copy.deepcopy(net).double(); t32 = copy.deepcopy(net).to(cuda)
    net = net.to(cuda)
    for m in (net, ref, t32): m.train(training)
    g = torch.Generator().manual_seed(11)
    x = dict(tsdf_values=(torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2, tsdf_weights=torch.rand(1, 9, h, w, generator=g) * 4,
             tsdf_frame=torch.rand(1, 1, h, w, generator=g) * 4)
    target = (torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2
    est_ref = ref({k: v.double() for k, v in x.items()}); loss(est_ref, target.double()).backward()
    xc = {k: v.to(cuda) for k, v in x.items()}
    est = HipTrainNet(net)(xc); loss(est, target.to(cuda)).backward()
    with torch.backends.cudnn.flags(enabled=False):
        est32 = t32(xc); loss(est32, target.to(cuda)).backward()
    print('training', training, 'est err hip %.3e torch32 %.3e' % (float((est.detach().cpu().double() - est_ref).abs().max()), float((est32.detach().cpu().double() - est_ref).abs().max())))
    worst = [0, 0]; per = {}
    for (name, p), (_, q), (_, r) in zip(net.named_parameters(), ref.named_parameters(), t32.named_parameters()):
        if q.grad is None: continue
        s = float(q.grad.abs().max())
        e = float((p.grad.cpu().double() - q.grad).abs().max()) / max(s, 1e-12); e32 = float((r.grad.cpu().double() - q.grad).abs().max()) / max(s, 1e-12)
        worst = [max(worst[0], e if s > 1e-6 else 0), max(worst[1], e32 if s > 1e-6 else 0)]
        k = '.'.join(name.split('.')[:3]) if name.startswith('vortex') else name.split('.')[0]
        if s > 1e-6: per[k] = [max(per.get(k, [0, 0])[0], e), max(per.get(k, [0, 0])[1], e32)]
    for k, v in per.items(): print('  %-24s hip %.3e torch32 %.3e' % (k, v[0], v[1]))
