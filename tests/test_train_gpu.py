"""-m gpu: the HIP training path of the fusion net (train.py / include/ojf.h ojf_train_*) against torch autograd.

 * ``LayerUnit`` (conv -> BatchNorm2d -> activation -> Dropout2d scale): output, input gradient, weight / bias / gamma /
   beta gradients and the running-statistics update vs the same unit written with torch ops in float64, for every
   layer geometry of the net (1x1, dilated 3x3, slotted concatenations, wide outputs).
 * whole nets (v3 with / without the semantic head, v2): est and ALL parameter gradients vs ``loss.backward()`` through
   the module's own forward (float64 on the CPU) in train() mode (batch statistics; dropout probability 0, its random
   stream cannot be shared) and in eval() mode; running statistics after the step.
Stated tolerance: 1e-4 of the largest magnitude of each tensor, measured ~1e-6."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from online_joint_depthfusion_and_semantic_amd import model
from online_joint_depthfusion_and_semantic_amd.train import HipTrainNet, LayerUnit, to_c4, from_c4

pytestmark = pytest.mark.gpu
REL = 1e-4


def close(got, want, what):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err <= REL * max(scale, 1e-30) + 1e-12, (what, err, scale)
    return err / max(scale, 1e-30)


def slotted(x, group, slot):
    """logical [1, C, H, W] -> physical channels: `group`-wide tensors in `slot`-wide slots."""
    _, C, H, W = x.shape
    n = (C + group - 1) // group
    out = x.new_zeros(1, n * slot, H, W)
    for s in range(n):
        k = min(group, C - s * group)
        out[:, s * slot:s * slot + k] = x[:, s * group:s * group + k]
    return out


@pytest.mark.parametrize('IC,OC,k,dil,group,slot,act,bn,training', [
    (19, 19, 3, 1, 19, 20, 'leaky', True, True), (57, 19, 3, 1, 19, 20, 'leaky', True, True), (95, 19, 3, 1, 19, 20, 'leaky', True, False),
    (19, 19, 3, 27, 19, 20, 'relu', True, True), (19, 19, 3, 9, 19, 20, 'relu', True, True), (114, 19, 1, 1, 19, 20, 'relu', True, True),
    (114, 19, 1, 1, 114, 116, 'relu', True, True), (19, 114, 1, 1, 19, 20, 'relu', True, True), (570, 114, 1, 1, 114, 116, None, True, True),
    (114, 95, 1, 1, 114, 116, 'leaky', True, True), (19, 19, 1, 1, 19, 20, 'leaky', False, True), (19, 9, 1, 1, 19, 20, 'tanh', False, True),
    (228, 19, 1, 1, 114, 116, 'relu', True, False)])
def test_layer_unit_against_torch(cuda, IC, OC, k, dil, group, slot, act, bn, training):
    H, W = 37, 45
    g = torch.Generator().manual_seed(IC * 7 + OC + k + dil)
    x = torch.randn(1, IC, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(OC, IC, k, k, generator=g, dtype=torch.float64) / np.sqrt(IC * k * k)
    b = torch.randn(OC, generator=g, dtype=torch.float64) * 0.1
    gamma = torch.rand(OC, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(OC, generator=g, dtype=torch.float64) * 0.1
    rm, rv = torch.randn(OC, generator=g, dtype=torch.float64) * 0.1, torch.rand(OC, generator=g, dtype=torch.float64) + 0.5
    drop = ((torch.rand(OC, generator=g) < 0.8).double() / 0.8) if act != 'tanh' else None
    dout = torch.randn(1, OC, H, W, generator=g, dtype=torch.float64) * 1e-5  # loss gradients are tiny
    scale = 0.7 if act == 'tanh' else 1.0

    # torch reference in float64
    xr, wr, br, gr, ber = (t.clone().requires_grad_(True) for t in (x, w, b, gamma, beta))
    rmr, rvr = rm.clone(), rv.clone()
    y = F.conv2d(xr, wr, br, padding=dil * (k // 2), dilation=dil)
    if bn:
        y = F.batch_norm(y, rmr, rvr, gr, ber, training, 0.1, 1e-5)
    y = {'relu': F.relu, 'leaky': lambda t: F.leaky_relu(t, 0.01), 'tanh': torch.tanh, None: lambda t: t}[act](y) * scale
    if drop is not None:
        y = y * drop.view(1, -1, 1, 1)
    y.backward(dout)

    f = lambda t: t.float().to(cuda)
    xs = to_c4(f(slotted(x, group, slot))).requires_grad_(True)
    wp, bp = f(w).requires_grad_(True), f(b).requires_grad_(True)
    gp, bep = (f(gamma).requires_grad_(True), f(beta).requires_grad_(True)) if bn else (None, None)
    bnm = None
    if bn:
        bnm = torch.nn.BatchNorm2d(OC).to(cuda)
        bnm.running_mean.copy_(f(rm)); bnm.running_var.copy_(f(rv))
    meta = dict(group=group, slot=slot, dil=dil, act=act, scale=scale, bn=bnm, drop=f(drop) if drop is not None else None, training=training)
    out = LayerUnit.apply(xs, wp, bp, gp, bep, meta)
    out.backward(to_c4(f(dout)))
    close(from_c4(out, OC), y, 'out')
    if out.shape[0] * 4 > OC:  # padding channels stay exactly zero
        assert float(out.detach().permute(0, 3, 1, 2).reshape(-1, H, W)[OC:].abs().max()) == 0
    dx_log = from_c4(xs.grad, xs.shape[0] * 4)
    close(dx_log, slotted(xr.grad, group, slot), 'dx')
    close(wp.grad, wr.grad, 'dW')
    if bn and training:  # sum of dy vanishes under batch statistics: compare on the scale of the unnormalised sum
        assert float(bp.grad.abs().max()) <= 1e-4 * float(dout.abs().sum() / OC) + 1e-12
    else:
        close(bp.grad, br.grad, 'db')
    if bn:
        close(gp.grad, gr.grad, 'dgamma')
        close(bep.grad, ber.grad, 'dbeta')
        close(bnm.running_mean, rmr, 'running_mean')
        close(bnm.running_var, rvr, 'running_var')
        assert int(bnm.num_batches_tracked) == (1 if training else 0)


def _net(version, sem, h, w, seed=3, n_points=9):
    cfg = type('C', (), dict(n_points=n_points, growth_factor=6, use_semantics=sem, output_scale=0.9, resx=w, resy=h))()
    torch.manual_seed(seed)
    net = getattr(model, 'FusionNet_' + version)(cfg)
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.xavier_normal_(m.weight)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0  # the random stream of torch's dropout cannot be shared: parity runs without it
    return net


def _whole_net_gradient_case(cuda, version, sem, training, path, h, w, noise_factor=2.0):
    graph = path == 'graph'
    net = _net(version, sem, h, w)
    ref, ref32 = copy.deepcopy(net).double(), copy.deepcopy(net)
    net = net.to(cuda)
    for m in (net, ref, ref32):
        m.train(training)
    g = torch.Generator().manual_seed(11)
    x = dict(tsdf_values=(torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2, tsdf_weights=torch.rand(1, 9, h, w, generator=g) * 4,
             tsdf_frame=torch.rand(1, 1, h, w, generator=g) * 4, semantic_frame=torch.randint(1, 31, (1, 1, h, w), generator=g).float() / 30)
    target = (torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2

    def loss(e, t):
        return (e - t).abs().mean() + 10 * ((e - t) ** 2).mean()
    est_ref = ref({k: v.double() for k, v in x.items()})
    loss(est_ref, target.double()).backward()
    est32 = ref32(x)
    loss(est32, target).backward()
    tn = HipTrainNet(net, graph=graph, inplace_grads=not graph, executor=path.startswith('executor'), arithmetic='f32' if path == 'executor_f32' else 'f16x3')
    est = tn({k: v.to(cuda) for k, v in x.items()})
    if graph:
        sig = next(iter(tn._graphs.values()))
        assert sig.ok, getattr(sig, 'error', None)  # the device-graph path really ran
    loss(est, target.to(cuda)).backward()

    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    # torch fp32's own worst relative deviation over all gradients: the rounding noise level of this net
    noise = max(float((q32.grad.double() - q.grad).abs().max()) / max(float(q.grad.abs().max()), 1e-3 * gmax)
                for q32, q in zip(ref32.parameters(), ref.parameters()) if q.grad is not None)

    def bar(got, fp32, truth, floor, what):
        got, fp32, truth = got.detach().cpu().double(), fp32.detach().double(), truth.detach()
        e, e32 = float((got - truth).abs().max()), float((fp32 - truth).abs().max())
        assert e <= max(REL * floor, 2.5 * e32, noise_factor * noise * floor), (what, e, e32, floor, noise)
        return e / max(e32, REL * floor)
    worst = bar(est, est32, est_ref, float(est_ref.abs().max()), 'est')
    for (name, p), (_, q32), (_, q) in zip(net.named_parameters(), ref32.named_parameters(), ref.named_parameters()):
        assert (p.grad is None) == (q.grad is None), name
        if q.grad is None:
            continue
        # gradients that vanish analytically (conv biases in front of a batch-statistics BatchNorm, the global-average
        # branch's conv) are judged on the scale of the largest gradient of the net
        worst = max(worst, bar(p.grad, q32.grad, q.grad, max(float(q.grad.abs().max()), 1e-3 * gmax), name))
    for (name, b), (_, c32), (_, c) in zip(net.named_buffers(), ref32.named_buffers(), ref.named_buffers()):
        if b.dtype.is_floating_point:
            bar(b, c32, c, float(c.abs().max()), name)
        else:
            assert int(b) == int(c), name
    print('whole net %s sem=%s training=%s path=%s %dx%d: worst deviation = %.2f x torch fp32\'s own' % (version, sem, training, path, h, w, worst))


@pytest.mark.parametrize('path', ['executor', 'executor_f32', 'units', 'graph'])
@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('version,sem', [('v3', False), ('v3', True), ('v2', True), ('v2', False)])
def test_whole_net_gradients_against_torch_autograd(cuda, version, sem, training, path):
    """est, every parameter gradient and every BatchNorm buffer after one loss.backward() through the whole net.  Truth =
    the module's own forward in float64.  A 46-layer net with batch statistics amplifies rounding: torch's OWN fp32
    autograd (the module on the CPU) deviates from float64 by 1e-2 of a gradient's scale in train() mode and 2e-4 in
    eval() mode (the L1 term's sign flips), so the bar per tensor is: within 1e-4 of its scale, or no further from the
    float64 truth than 2.5x torch's fp32 deviation on that tensor, or than twice torch fp32's worst relative deviation over
    all gradients (the rounding-noise level of the net; GPU torch fp32 sits at 1x - 1.5x of it, tools/dbg_train.py).
    Paths: 'executor' = the whole pass as two libojf calls (ojf_trainer_*, the default), 'units' = one autograd node per
    layer unit, 'graph' = the unit path captured into device graphs (HipTrainNet(graph=True))."""
    _whole_net_gradient_case(cuda, version, sem, training, path, 40, 56)


@pytest.mark.parametrize('h,w', [(13, 15), (23, 37)])
def test_whole_net_gradients_on_ragged_frames(cuda, h, w):
    """Frames whose pixel count is no multiple of the 16-pixel MFMA tiles, the 64-pixel weight-gradient chunks, the 32 x 8
    pooling tiles or the 64 reduction slabs (and, at 13 x 15, smaller than the dilation-27 reach): the executor's padding
    lanes and partial slabs against float64 torch.  eval() mode only: with batch statistics over 195 / 851 pixels the
    comparison is dominated by how rounding differences are amplified (2-3 noise levels, varying with the host that
    computes the fp32 reference), while eval() sees the same lanes and slabs and sits at 0.01-0.02 of torch fp32's own
    deviation."""
    _whole_net_gradient_case(cuda, 'v3', True, False, 'executor', h, w)


@pytest.mark.parametrize('version,sem,training', [('v3', False, True), ('v3', True, False)])
def test_whole_net_gradients_at_baseline_frame_size(cuda, version, sem, training):
    """The same comparison at BASELINE configs[3]'s frame size, 240x320 = 76 800 pixels: the weight-gradient kernel's
    pixel-slab split, the fp64 slab reductions of the BatchNorm statistics and the dilation-27 borders all depend on
    h * w.  Truth = float64 torch on the host cores (about 40 s per case)."""
    _whole_net_gradient_case(cuda, version, sem, training, 'executor', 240, 320)


def test_graph_warm_up_leaves_parameter_gradients_alone(cuda):
    """building the device graphs runs two warm-up passes; they must not leave anything in ``p.grad`` - not
    when it is None, not when it is a pre-existing buffer (train_fusion points every p.grad into the flat all-reduce
    buffer) - and replayed passes accumulate exactly like eager ones, over two frames."""
    h, w = 24, 32
    net = _net('v3', False, h, w).to(cuda).eval()
    g = torch.Generator().manual_seed(5)
    frames = [dict(tsdf_values=((torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2).to(cuda), tsdf_weights=(torch.rand(1, 9, h, w, generator=g) * 4).to(cuda),
                   tsdf_frame=(torch.rand(1, 1, h, w, generator=g) * 4).to(cuda)) for _ in range(2)]
    eager = HipTrainNet(net, executor=False)  # the same unit kernels, launched one by one
    for x in frames:
        eager(x).pow(2).mean().backward()
    want = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    # (a) from p.grad = None
    net.zero_grad(set_to_none=True)
    tn = HipTrainNet(net, graph=True)
    for x in frames:
        tn(x).pow(2).mean().backward()
    assert next(iter(tn._graphs.values())).ok
    for n, p in net.named_parameters():
        if n in want:
            assert torch.allclose(p.grad, want[n], rtol=1e-5, atol=1e-7 * float(want[n].abs().max())), n
    # (b) into a pre-existing buffer holding a known value, graphs built while it is in place
    net.zero_grad(set_to_none=True)
    holders = {}
    for n, p in net.named_parameters():
        p.grad = torch.full_like(p, 0.25)
        holders[n] = p.grad
    tn = HipTrainNet(net, graph=True)
    for x in frames:
        tn(x).pow(2).mean().backward()
    for n, p in net.named_parameters():
        assert p.grad is holders[n], n
        if n in want:
            assert torch.allclose(p.grad - 0.25, want[n], rtol=1e-4, atol=1e-6 * max(float(want[n].abs().max()), 1.0)), n


def test_gradients_accumulate_in_place_like_autograd(cuda):
    """The kernels write parameter gradients into ``p.grad`` themselves (first backward: plain store, later ones: add;
    train.py::_grad_target).  Two backward passes of the same frame in eval() mode (no state changes between them) must
    leave exactly twice the gradient of one pass, in the SAME tensors (optimizers and the flat all-reduce buffer hold on
    to them), and ``zero_grad(set_to_none=True)`` must start over."""
    h, w = 24, 32
    net = _net('v3', True, h, w).to(cuda).eval()
    g = torch.Generator().manual_seed(5)
    x = dict(tsdf_values=((torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2).to(cuda), tsdf_weights=(torch.rand(1, 9, h, w, generator=g) * 4).to(cuda),
             tsdf_frame=(torch.rand(1, 1, h, w, generator=g) * 4).to(cuda),
             semantic_frame=(torch.randint(1, 31, (1, 1, h, w), generator=g).float() / 30).to(cuda))
    eng = HipTrainNet(net, inplace_grads=True, arithmetic='f32')  # (bit-equality of pass 1 and 2: one arithmetic for both)
    eng(x).pow(2).mean().backward()
    once = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    holders = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert len(once) > 100 and all(float(v.abs().max()) >= 0 for v in once.values())
    eng(x).pow(2).mean().backward()
    for n, p in net.named_parameters():
        if n in once:
            assert p.grad is holders[n], n                      # accumulated in place
            assert torch.equal(p.grad, once[n] + once[n]), n    # g + g, bit for bit
    net.zero_grad(set_to_none=True)
    eng(x).pow(2).mean().backward()
    for n, p in net.named_parameters():
        if n in once:
            assert torch.equal(p.grad, once[n]), n


def test_dropout_channels_in_train_mode(cuda):
    """Dropout2d semantics of a unit in train() mode: whole channels are zeroed with probability p, survivors scaled by
    1 / (1 - p); eval() mode is deterministic."""
    h, w = 24, 32
    net = _net('v3', False, h, w).to(cuda)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.2
    x = dict(tsdf_values=torch.rand(1, 9, h, w, device=cuda) * 0.1, tsdf_weights=torch.rand(1, 9, h, w, device=cuda), tsdf_frame=torch.rand(1, 1, h, w, device=cuda))
    torch.manual_seed(1)
    for executor in (True, False):
        tn = HipTrainNet(net.train(), executor=executor)
        a = tn(x)
        b = tn(x)
        assert not torch.equal(a, b)  # fresh masks per call
    blk = net.block0[0].block
    zeros = 0
    for _ in range(50):
        y = tn._sequential(to_c4(torch.cat([x['tsdf_values'], x['tsdf_weights'], x['tsdf_frame']], 1)), blk, 19, 20)
        per_channel = from_c4(y, 19).abs().amax(dim=(0, 2, 3))
        zeros += int((per_channel == 0).sum())
    assert 0.12 <= zeros / (50 * 19) <= 0.28  # p = 0.2 on the second stage's 19 channels
    tn2 = HipTrainNet(net.eval())
    assert torch.equal(tn2(x), tn2(x))



def test_autograd_grad_returns_parameter_gradients_by_default(cuda):
    """without ``inplace_grads`` the units behave like any autograd.Function - ``torch.autograd.grad`` returns
    every parameter's gradient and no ``.grad`` is touched."""
    h, w = 24, 32
    net = _net('v3', False, h, w).to(cuda).eval()
    x = dict(tsdf_values=torch.rand(1, 9, h, w, device=cuda) * 0.1, tsdf_weights=torch.rand(1, 9, h, w, device=cuda), tsdf_frame=torch.rand(1, 1, h, w, device=cuda))
    params = [p for p in net.parameters() if p.requires_grad]
    grads = torch.autograd.grad(HipTrainNet(net)(x).pow(2).mean(), params, allow_unused=True)
    assert all(p.grad is None for p in params)
    assert sum(g is not None and float(g.abs().sum()) > 0 for g in grads) > 100
    HipTrainNet(net, inplace_grads=True)(x).pow(2).mean().backward()
    for p, gr in zip(params, grads):
        if gr is not None:
            assert torch.allclose(p.grad, gr, rtol=1e-5, atol=1e-9), 'in-place and returned gradients differ'


def test_frozen_batchnorm_keeps_running_statistics(cuda):
    """batch vs running statistics follow each BatchNorm2d's own ``training`` flag."""
    h, w = 24, 32
    net = _net('v3', False, h, w).to(cuda).train()
    frozen = net.block0[0].block[1]
    assert isinstance(frozen, torch.nn.BatchNorm2d)
    frozen.eval()
    before = (frozen.running_mean.clone(), frozen.running_var.clone(), int(frozen.num_batches_tracked))
    other = net.block0[1].block[1]
    other_before = other.running_mean.clone()
    x = dict(tsdf_values=torch.rand(1, 9, h, w, device=cuda) * 0.1, tsdf_weights=torch.rand(1, 9, h, w, device=cuda), tsdf_frame=torch.rand(1, 1, h, w, device=cuda))
    got = HipTrainNet(net)(x)
    assert torch.equal(frozen.running_mean, before[0]) and torch.equal(frozen.running_var, before[1])
    assert int(frozen.num_batches_tracked) == before[2]
    assert not torch.equal(other.running_mean, other_before)
    import copy as _copy
    ref = _copy.deepcopy(net)
    # same flags, module forward on the device (buffers were already updated once by the HIP pass: compare outputs only)
    ref.load_state_dict(net.state_dict())
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d) and m.training:
            m.momentum = 0.0  # keep the running statistics of the comparison net out of the picture
    # (a frozen layer on batch statistics would move est by ~0.1; fp32 train-mode nets differ among themselves by ~1e-4)
    assert torch.allclose(got, ref(x), atol=2e-3)


def test_fuse_output_and_fusion_loss_kernels_match_the_tensor_formulas(cuda):
    """ojf_train_fuse_output / ojf_train_fusion_loss (one launch each way) against the tensor-op statements of
    modules/pipeline.py:104-127 and utils/loss.py:65-103 (the package's FusionLoss on CPU tensors runs those), values
    and gradients: fused rows bit for bit (the same three roundings), loss within 1e-6 relative (fp64 partial sums vs
    torch's fp32 reductions), gradients within 1e-6 of their scale."""
    from online_joint_depthfusion_and_semantic_amd.loss import FusionLoss
    from online_joint_depthfusion_and_semantic_amd.train import FuseOutput
    g = torch.Generator().manual_seed(3)
    P, n, init = 9, 56 * 40, 0.1
    est = ((torch.rand(1, P, n, generator=g) - 0.5) * 0.4).requires_grad_(True)   # beyond +-init on both sides
    fv = (torch.rand(1, P, n, generator=g) - 0.5) * 0.2
    fw = torch.rand(1, P, n, generator=g) * 3 - 0.3                                # some negative weights
    valid = torch.nonzero(torch.rand(n, generator=g) > 0.2)[:, 0]
    target = (torch.rand(1, valid.numel(), P, generator=g) - 0.5) * 0.2
    target[0, :5] = 0.0  # sign(0) rows
    crit = FusionLoss(w_l1=1.0, w_l2=10.0, w_cos=0.1)
    # tensor-op reference on the CPU
    fwc = torch.clamp_min(fw, 0)
    fused_ref = ((fwc * fv + torch.clamp(est, -init, init)) / (fwc + 1)).transpose(1, 2)[:, valid, :]
    loss_ref = crit(fused_ref, target)
    loss_ref.backward()
    grad_ref = est.grad.clone()
    # kernels
    est_d = est.detach().to(cuda).requires_grad_(True)
    fused = FuseOutput.apply(est_d, fv.to(cuda), fw.to(cuda), valid.to(cuda), init)
    assert torch.equal(fused.cpu(), fused_ref.detach())
    loss = crit(fused, target.to(cuda))
    assert loss.shape == () and abs(float(loss) - float(loss_ref)) <= 1e-6 * abs(float(loss_ref))
    loss.backward()
    scale = float(grad_ref.abs().max())
    assert float((est_d.grad.cpu() - grad_ref).abs().max()) <= 1e-6 * scale
    assert float(est_d.grad[:, :, ~torch.isin(torch.arange(n), valid).to(cuda)].abs().max()) == 0.0  # masked pixels receive nothing
    # an upstream factor reaches the gradient (loss * 3)
    est_d.grad = None
    fused = FuseOutput.apply(est_d, fv.to(cuda), fw.to(cuda), valid.to(cuda), init)
    (crit(fused, target.to(cuda)) * 3.0).backward()
    assert float((est_d.grad.cpu() - 3.0 * grad_ref).abs().max()) <= 3e-6 * scale


def _second_pass_case(cuda, version, sem, training, h, w):
    net = _net(version, sem, h, w).to(cuda).train(training)
    g = torch.Generator().manual_seed(17)
    x = dict(tsdf_values=((torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2).to(cuda), tsdf_weights=(torch.rand(1, 9, h, w, generator=g) * 4).to(cuda),
             tsdf_frame=(torch.rand(1, 1, h, w, generator=g) * 4).to(cuda), semantic_frame=(torch.randint(1, 31, (1, 1, h, w), generator=g).float() / 30).to(cuda))
    target = ((torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2).to(cuda)
    params = [p for p in net.parameters()]
    tn = HipTrainNet(net)
    passes = []
    for _ in range(3):
        e = tn(x)
        loss = (e - target).abs().mean() + 10 * ((e - target) ** 2).mean()
        passes.append(torch.autograd.grad(loss, params, allow_unused=True))
    gmax = max(float(a.abs().max()) for a in passes[0] if a is not None)
    worst = 0.0
    for k in (1, 2):
        for p0, pk in zip(passes[0], passes[k]):
            if p0 is None:
                continue
            scale = max(float(p0.abs().max()), 1e-3 * gmax)
            worst = max(worst, float((p0 - pk).abs().max()) / scale)
    print('split-fp16 backward %s sem=%s training=%s %dx%d: worst deviation from the fp32 pass %.2e of a tensor\'s scale' % (version, sem, training, h, w, worst))
    return worst


@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('version,sem', [('v3', True), ('v2', False)])
def test_executor_backward_data_in_split_fp16_from_the_second_pass(cuda, version, sem, training):
    """The executor's first backward pass runs backward-data and the weight gradients on fp32-input MFMAs and measures the
    magnitude of every dy tensor; from the second pass on dy is stored with a power-of-two factor and both run in the
    split-fp16 arithmetic (ojf_trainer_backward: conv_f16x3_kernel on the transposed weights, train_wgrad_mfma_kernel<true>).
    Same frame, same weights (eval() mode, or train() mode without dropout: batch statistics do not depend on the running
    buffers): passes 2 and 3 must reproduce pass 1's gradients to fp32-class accuracy - per tensor within 2e-5 of max(its
    scale, 1e-3 of the largest gradient) in eval() mode; 2e-4 in train() mode, where the batch-statistics chain amplifies any
    rounding difference (torch's own fp32 runs differ by 1e-2 there)."""
    worst = _second_pass_case(cuda, version, sem, training, 40, 56)
    assert worst <= (2e-4 if training else 2e-5)


def test_split_fp16_backward_at_baseline_frame_size(cuda):
    """The same comparison at BASELINE configs[3]'s frame size: 76 800 pixels is where the weight gradient's pixel slabs, its
    64-pixel chunks (K of the split-fp16 MFMAs) and the fp64 slab reductions take the shape they have in a training run."""
    worst = _second_pass_case(cuda, 'v3', False, False, 240, 320)
    assert worst <= 2e-5


# ---- the split-fp16 backward pass has no range to outgrow ------------------------
def _guard_inputs(cuda, h, w, sem):
    g = torch.Generator().manual_seed(23)
    x = dict(tsdf_values=((torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2).to(cuda), tsdf_weights=(torch.rand(1, 9, h, w, generator=g) * 4).to(cuda),
             tsdf_frame=(torch.rand(1, 1, h, w, generator=g) * 4).to(cuda), semantic_frame=(torch.randint(1, 31, (1, 1, h, w), generator=g).float() / 30).to(cuda))
    target = ((torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2).to(cuda)
    return x, target


@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('inplace', [True, False])
def test_gradient_jumps_of_any_size_stay_in_range(cuda, inplace, training):
    """Round 3 stored dy under the PREVIOUS pass's power-of-two factor with 12 binades of headroom: a loss 2^14 times larger
    than the frame before (first frame of a new scene, a handful of valid pixels, a loss-scale change) turned dy into +-inf
    halves and NaN sums, which the range guard could not even see (this test, written for the guard, showed it).  The factor
    now comes from a bound computed in the SAME pass, so there is nothing to outgrow: the loss is multiplied by 2^14, then
    by 2^-40, then by 2^30 between frames, and every pass must give exactly the scaled gradients of the first one - a power
    of two commutes with every rounding of the pass (the factor moves with it), so the comparison is bit for bit."""
    h, w = 40, 56
    net = _net('v3', False, h, w).to(cuda).train(training)
    x, target = _guard_inputs(cuda, h, w, False)
    params = [p for p in net.parameters()]
    bufs = [b.clone() for b in net.buffers()]
    tn = HipTrainNet(net, inplace_grads=inplace)

    def one(factor):
        for b, saved in zip(net.buffers(), bufs):  # (train mode: the same running statistics in front of every pass)
            b.copy_(saved)
        e = tn(x)
        loss = ((e - target).abs().mean() + 10 * ((e - target) ** 2).mean()) * factor
        if inplace:
            for p in params:
                p.grad = None
            loss.backward()
            return [None if p.grad is None else p.grad.clone() for p in params]
        return torch.autograd.grad(loss, params, allow_unused=True)

    first = one(1.0)
    for f in (2.0 ** 14, 2.0 ** -40, 2.0 ** 30, 1.0):
        got = one(f)
        for a, b in zip(first, got):
            if a is None:
                continue
            assert torch.isfinite(b).all(), f
            assert torch.equal(a * f, b), (f, float((a * f - b).abs().max()), float(a.abs().max()))
    tn(x)  # and the next forward pass is not refused (round 3: a set flag surfaced there, for the rest of the process)


def test_backward_arithmetic_f32_keeps_the_backward_convolutions_on_fp32_mfma(cuda):
    """FUSION_MODEL.train_arithmetic_bwd: 'f32' keeps every backward pass on the fp32-input MFMA path (the choice is
    exposed); its gradients agree with the split-fp16 backward to the fp32 bar."""
    h, w = 24, 40
    net = _net('v3', False, h, w).to(cuda).eval()
    x, target = _guard_inputs(cuda, h, w, False)
    params = [p for p in net.parameters()]
    tn = HipTrainNet(net, backward_arithmetic='f32')
    grads = []
    for f in (1.0, 1.0, 2.0 ** 14):
        e = tn(x)
        grads.append(torch.autograd.grad(((e - target) ** 2).mean() * f, params, allow_unused=True))
    tn16 = HipTrainNet(net)
    e = tn16(x)
    g16 = torch.autograd.grad(((e - target) ** 2).mean(), params, allow_unused=True)
    gmax = max(float(a.abs().max()) for a in grads[0] if a is not None)
    for a, b, c, d in zip(*grads, g16):
        if a is None:
            continue
        assert torch.equal(a, b)                      # the same fp32 arithmetic twice: the same bits
        assert torch.equal(a * 2.0 ** 14, c)          # a power of two commutes with every rounding
        assert float((a - d).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-3 * gmax)


def test_two_frame_sizes_interleaved_keep_their_own_tables(cuda):
    """forward(A), forward(B), backward(A) used to read B's dropout factors / BN flags from a shared layer table."""
    net = _net('v3', False, 24, 40).to(cuda).eval()
    xa, ta = _guard_inputs(cuda, 24, 40, False)
    xb, tb = _guard_inputs(cuda, 32, 48, False)
    params = [p for p in net.parameters()]
    tn = HipTrainNet(net)
    ea = tn(xa)
    want = torch.autograd.grad(((ea - ta) ** 2).mean(), params, allow_unused=True)
    ea = tn(xa)
    eb = tn(xb)
    got = torch.autograd.grad(((ea - ta) ** 2).mean(), params, allow_unused=True)
    torch.autograd.grad(((eb - tb) ** 2).mean(), params, allow_unused=True)
    gmax = max(float(a.abs().max()) for a in want if a is not None)
    for a, b in zip(want, got):
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-3 * gmax)


def test_weights_edited_through_data_need_invalidate(cuda):
    """``p.data.mul_()`` bumps no version counter; ``HipTrainNet.invalidate()`` is the documented hook, and an
    ordinary in-place update (optimizer step) is seen without it."""
    h, w = 24, 40
    net = _net('v3', False, h, w).to(cuda).eval()
    x, _ = _guard_inputs(cuda, h, w, False)
    tn = HipTrainNet(net)
    with torch.no_grad():
        e0 = tn(x).clone()
        net.pred[-1].pred[-2].weight.mul_(0.5)      # in place through the parameter: version counter moves
        e1 = tn(x).clone()
        assert not torch.equal(e0, e1)
        net.pred[-1].pred[-2].weight.data.mul_(2.0)  # through .data: invisible ...
        tn.invalidate()                              # ... without this
        e2 = tn(x).clone()
    assert float((e2 - e0).abs().max()) <= 1e-6


def test_wide_nets_up_to_the_executor_limit(cuda):
    """the BatchNorm-backward apply kernel scans the bound words of its scale group with one 256-thread
    block; a unit with more than 256 channels would have left channels out of the dy factor.  The scan is strided now - and
    such a unit cannot be built: ojf_trainer_create rejects units of more than 128 output channels (and dense 3x3 convolutions
    beyond the 128-superstep tap table), so every accepted net stays far below 256 bound words per scale group.  Checked here:
    the rejection is loud, and the widest accepted net (n_points 10: 21-channel slots, 126-channel units, four stacked branch
    units of 24 physical channels) reproduces its fp32 backward pass in the split-fp16 passes with the largest gradients
    sitting in the LAST channels of the widest units."""
    from online_joint_depthfusion_and_semantic_amd import _lib
    h, w = 24, 40
    with pytest.raises(_lib.OjfError, match='unsupported layer shape|more than 128 output channels'):
        HipTrainNet(_net('v3', False, h, w, n_points=23).to(cuda).train(True))(
            dict(tsdf_values=torch.zeros(1, 23, h, w, device=cuda), tsdf_weights=torch.zeros(1, 23, h, w, device=cuda),
                 tsdf_frame=torch.zeros(1, 1, h, w, device=cuda)))
    P = 10
    net = _net('v3', False, h, w, n_points=P).to(cuda).train(True)
    with torch.no_grad():
        for vp in (net.vortex0, net.vortex3):
            vp.final[1].weight[110:] *= 1e4
    g = torch.Generator().manual_seed(29)
    x = dict(tsdf_values=((torch.rand(1, P, h, w, generator=g) - 0.5) * 0.2).to(cuda), tsdf_weights=(torch.rand(1, P, h, w, generator=g) * 4).to(cuda),
             tsdf_frame=(torch.rand(1, 1, h, w, generator=g) * 4).to(cuda))
    target = ((torch.rand(1, P, h, w, generator=g) - 0.5) * 0.2).to(cuda)
    params = [p for p in net.parameters()]
    bufs = [b.clone() for b in net.buffers()]
    tn = HipTrainNet(net)
    passes = []
    for _ in range(3):
        for b, saved in zip(net.buffers(), bufs):
            b.copy_(saved)
        e = tn(x)
        loss = (e - target).abs().mean() + 10 * ((e - target) ** 2).mean()
        passes.append(torch.autograd.grad(loss, params, allow_unused=True))
    gmax = max(float(a.abs().max()) for a in passes[0] if a is not None)
    for k in (1, 2):
        for (name, _), p0, pk in zip(net.named_parameters(), passes[0], passes[k]):
            if p0 is None:
                continue
            assert torch.isfinite(pk).all(), (k, name)
            assert float((p0 - pk).abs().max()) <= 2e-3 * max(float(p0.abs().max()), 1e-3 * gmax), (k, name)


def _replay_run(cuda, replay, version, sem, h, w, frames=10):
    net = _net(version, sem, h, w, seed=11).to(cuda).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.2
    g = torch.Generator().manual_seed(7)
    xs = []
    for _ in range(3):  # three frames whose tensors stay where they are (a loader's ring of device buffers)
        x = dict(tsdf_values=((torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2).to(cuda), tsdf_weights=(torch.rand(1, 9, h, w, generator=g) * 4).to(cuda),
                 tsdf_frame=(torch.rand(1, 1, h, w, generator=g) * 4).to(cuda))
        if sem:
            x['semantic_frame'] = (torch.randint(1, 31, (1, 1, h, w), generator=g).float() / 30).to(cuda)
        xs.append(x)
    eng = HipTrainNet(net, inplace_grads=True, replay=replay)
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-4)
    torch.manual_seed(123)  # the dropout draws
    outs = []
    for i in range(frames):
        est = eng(xs[i % 3])
        outs.append(est.detach().clone())
        est.pow(2).mean().backward()
        if i % 4 == 3:
            opt.step()
            opt.zero_grad(set_to_none=False)
    torch.cuda.synchronize()
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    return outs, state, grads, eng.replays


@pytest.mark.gpu
@pytest.mark.parametrize('version,sem,h,w', [('v3', False, 24, 32), ('v3', True, 23, 37), ('v2', True, 24, 32)])
def test_executor_passes_replayed_as_device_graphs_change_no_bit(cuda, version, sem, h, w):
    """ojf_trainer_set_graph: forward / backward passes served by hipGraphLaunch run the same kernels with the same arguments -
    ten train()-mode frames with Dropout2d, in-place gradient accumulation and an RMSprop step every fourth frame (weights repacked in
    front of the next replay, `accumulate` toggling between two captured backward passes) leave the same outputs, parameters, BatchNorm
    statistics and gradients as plain launches, bit for bit; and the replays really ran."""
    plain = _replay_run(cuda, False, version, sem, h, w)
    graph = _replay_run(cuda, True, version, sem, h, w)
    assert plain[3] == 0 and graph[3] >= 8, (plain[3], graph[3])
    for a, b in zip(plain[0], graph[0]):
        assert torch.equal(a, b)
    for k in plain[1]:
        assert torch.equal(plain[1][k], graph[1][k]), k
    assert plain[2].keys() == graph[2].keys() and len(plain[2]) > 100
    for k in plain[2]:
        assert torch.equal(plain[2][k], graph[2][k]), k
