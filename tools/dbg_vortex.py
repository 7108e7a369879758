"""Debug helper: one VortexPooling through HipTrainNet._vortex vs the module in float64, including the input gradient."""
import sys, copy, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_train_gpu import _net, slotted
from online_joint_depthfusion_and_semantic_amd.train import HipTrainNet, to_c4, from_c4
cuda = torch.device('cuda:0')
h, w = 40, 56
for training in (False, True):
    net = _net('v3', False, h, w)
    ref = copy.deepcopy(net).double()
    net = net.to(cuda); net.train(training); ref.train(training)
    tn = HipTrainNet(net)
    g = torch.Generator().manual_seed(5)
    for name, group, slot, C in (('vortex3', 114, 116, 114), ('vortex0', 19, 20, 114)):
        x = torch.randn(1, C, h, w, generator=g, dtype=torch.float64)
        dout = torch.randn(1, 114, h, w, generator=g, dtype=torch.float64) * 1e-4
        xr = x.clone().requires_grad_(True)
        yr = getattr(ref, name)(xr); yr.backward(dout)
        xs = to_c4(slotted(x, group, slot).float().to(cuda)).requires_grad_(True)
        y = tn._vortex(xs, getattr(net, name), group, slot)
        y.backward(to_c4(slotted(dout, 114, 116).float().to(cuda)))
        dx = from_c4(xs.grad, xs.shape[0] * 4).cpu().double()
        want = slotted(xr.grad, group, slot)
        e = (dx - want).abs()
        print(training, name, 'out err %.2e' % float((from_c4(y, 114).detach().cpu().double() - yr).abs().max()),
              'dx err %.3e scale %.3e' % (float(e.max()), float(want.abs().max())), 'worst at', [int(i) for i in torch.nonzero(e == e.max())[0]])
        # which part: border rows/cols?
        interior = e[:, :, 3:-3, 3:-3].max(); print('   interior max err %.3e' % float(interior))

import torch.nn.functional as F
print('---- pieces (eval mode) ----')
net = _net('v3', False, h, w); ref = copy.deepcopy(net).double(); net = net.to(cuda).eval(); ref.eval(); tn = HipTrainNet(net)
g = torch.Generator().manual_seed(7)
x = torch.randn(1, 114, h, w, generator=g, dtype=torch.float64)
def check(tag, f_hip, f_ref, cout):
    dout = torch.randn(1, cout, h, w, generator=g, dtype=torch.float64) * 1e-4
    xr = x.clone().requires_grad_(True); yr = f_ref(xr); yr.backward(dout)
    xs = to_c4(slotted(x, 114, 116).float().to(cuda)).requires_grad_(True)
    y = f_hip(xs); y.backward(to_c4(slotted(dout, cout, (cout + 3) // 4 * 4).float().to(cuda)))
    dx = from_c4(xs.grad, 116).cpu().double(); want = slotted(xr.grad, 114, 116)
    print(tag, 'out err %.2e' % float((from_c4(y, cout).detach().cpu().double() - yr).abs().max()), 'dx err %.3e scale %.3e' % (float((dx - want).abs().max()), float(want.abs().max())))
v, vr = net.vortex3, ref.vortex3
check('branch0 first unit only', lambda t: tn._sequential(t, torch.nn.Sequential(*list(v.branches[0])[:3]), 114, 116), lambda t: torch.nn.Sequential(*list(vr.branches[0])[:3])(t), 19)
check('branch0 two units', lambda t: tn._sequential(t, torch.nn.Sequential(*list(v.branches[0])[:6]), 114, 116), lambda t: torch.nn.Sequential(*list(vr.branches[0])[:6])(t), 19)
check('branch0 all', lambda t: tn._sequential(t, v.branches[0], 114, 116), lambda t: vr.branches[0](t), 114)
check('pool + branch1', lambda t: tn._sequential(F.avg_pool2d(t.permute(0, 3, 1, 2), 3, stride=1, padding=1).permute(0, 2, 3, 1).contiguous(), v.branches[1], 114, 116), lambda t: vr.branches[1](vr.pool1(t)), 114)
