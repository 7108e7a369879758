"""-m gpu: the HIP fusion net against the fp32 torch-CPU reference of the same layers, in both arithmetic
modes of the MFMA kernels (include/ojf.h: split-fp16 'f16x3' = default, fp32-input 'f32').
Stated tolerance: |tsdf_est difference| <= 1e-5 absolute, for both."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from online_joint_depthfusion_and_semantic_amd import _lib, model
from online_joint_depthfusion_and_semantic_amd.engine import FusionNetEngine, conv2d_rows

pytestmark = pytest.mark.gpu
TOL = 1e-5


class NS:
    def __init__(self, **k):
        self.__dict__.update(k)


def seeded_net(version, sem, h, w, seed=0):
    cfg = NS(n_points=9, growth_factor=6, use_semantics=sem, output_scale=1.0, resx=w, resy=h)
    torch.manual_seed(seed)
    net = getattr(model, 'FusionNet_' + version)(cfg)
    for m in net.modules():  # train_fusion.py:29-31 xavier init; randomised BN statistics
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.xavier_normal_(m.weight)
            m.bias.data.normal_(0, 0.05)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    return net.eval()


@pytest.mark.parametrize('cin,cout,k,dil,act', [
    (19, 19, 3, 1, _lib.ACT_LEAKY), (38, 19, 3, 1, _lib.ACT_LEAKY), (19, 19, 3, 3, _lib.ACT_RELU),
    (19, 19, 3, 9, _lib.ACT_RELU), (19, 19, 3, 27, _lib.ACT_RELU), (114, 95, 1, 1, _lib.ACT_LEAKY),
    (19, 114, 1, 1, _lib.ACT_RELU), (228, 19, 1, 1, _lib.ACT_NONE), (19, 9, 1, 1, _lib.ACT_TANH), (64, 48, 3, 2, _lib.ACT_NONE),
    (57, 19, 3, 1, _lib.ACT_LEAKY), (95, 19, 3, 1, _lib.ACT_LEAKY), (32, 32, 3, 1, _lib.ACT_RELU)])  # LDS-tiled 3x3 (wide inputs)
@pytest.mark.parametrize('h,w', [(24, 32), (37, 53)])
@pytest.mark.parametrize('arith', ['f16x3', 'f32'])
def test_conv2d_layer(cuda, arith, cin, cout, k, dil, act, h, w):
    _lib.check(_lib.load().ojf_net_set_arithmetic(_lib.ARITHMETIC[arith]), 'ojf_net_set_arithmetic')
    g = torch.Generator().manual_seed(cin * 1000 + cout + k + dil)
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, wt, b, padding=dil * (k // 2), dilation=dil)
    ref = {_lib.ACT_NONE: lambda t: t, _lib.ACT_RELU: F.relu, _lib.ACT_LEAKY: lambda t: F.leaky_relu(t, 0.01),
           _lib.ACT_TANH: torch.tanh}[act](ref)
    cin_p, cout_p = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    in_off, out_off = 4, 8  # embedded in wider rows, like the dense-growth buffers
    xr = torch.full((h * w, cin + 9), 7.0)  # junk around the window must not leak in
    xr[:, in_off:in_off + cin] = x[0].permute(1, 2, 0).reshape(h * w, cin)
    xr = xr.to(cuda)
    out = torch.full((h * w, cout_p + 12), -3.0, device=cuda)
    conv2d_rows(xr, in_off, cin, wt.numpy(), b.numpy(), out, out_off, h, w, dilation=dil, act=act)
    got = out.cpu()
    assert torch.all(got[:, :out_off] == -3.0) and torch.all(got[:, out_off + cout_p:] == -3.0)
    assert torch.all(got[:, out_off + cout:out_off + cout_p] == 0)  # pad channels carry zeros
    y = got[:, out_off:out_off + cout].reshape(h, w, cout).permute(2, 0, 1)
    err = float((y - ref[0]).abs().max())
    print('conv2d', arith, cin, cout, k, dil, 'max err %.2e' % err)
    assert err <= TOL


def _inputs(h, w, seed=1):
    g = torch.Generator().manual_seed(seed)
    return dict(tsdf_values=(torch.rand(1, 9, h, w, generator=g) - 0.5) * 0.2,
                tsdf_weights=torch.rand(1, 9, h, w, generator=g) * 4,
                tsdf_frame=torch.rand(1, 1, h, w, generator=g) * 4,
                sem_ids=torch.randint(0, 30, (h, w), generator=g, dtype=torch.uint8))


@pytest.mark.parametrize('version,sem', [('v3', False), ('v3', True), ('v2', False), ('v2', True)])
@pytest.mark.parametrize('h,w', [(24, 32), (60, 80), (120, 160), (45, 77), (5, 7)])  # ragged strips / tiles; a frame smaller than a tile
@pytest.mark.parametrize('arith', ['f16x3', 'f32'])
def test_fusion_net_forward(cuda, arith, version, sem, h, w):
    net = seeded_net(version, sem, h, w)
    x = _inputs(h, w)
    x['semantic_frame'] = ((1 + x['sem_ids'].float()) / 30).view(1, 1, h, w)
    with torch.no_grad():
        ref = net(x)[0].permute(1, 2, 0).reshape(h * w, 9)
    eng = FusionNetEngine(net, h, w, cuda, arithmetic=arith)
    assert eng.lib.ojf_net_get_arithmetic(eng.handle) == _lib.ARITHMETIC[arith]
    fv = x['tsdf_values'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(cuda)
    fw = x['tsdf_weights'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(cuda)
    eng.prepare_input(fv, fw, x['tsdf_frame'].reshape(h, w).contiguous().to(cuda),
                      x['sem_ids'].contiguous().to(cuda) if sem else None, 30)
    for stride in (9, 12):
        est = torch.full((h * w, stride), 5.0, device=cuda)
        eng.forward(est)
        got = est.cpu()
        err = float((got[:, :9] - ref).abs().max())
        print('net', arith, version, sem, h, w, 'max err %.2e' % err)
        assert err <= TOL, (version, sem, h, w, err)
        if stride > 9:
            assert torch.all(got[:, 9:] == 5.0)
    assert eng.macs_per_pixel == (326876 if not sem else (508820 if version == 'v3' else None)) or version == 'v2'
    eng.close()


def _large_weight_inputs(h, w):
    x = _inputs(h, w, seed=5)
    g = torch.Generator().manual_seed(9)
    x['tsdf_weights'] = torch.rand(1, 9, h, w, generator=g) * 60000.0
    x['tsdf_weights'][0, :, :4] = 65504.0  # fp16 max: what a saturated fp16 weight volume hands over
    return x


def _run(eng, x, h, w, cuda):
    fv = x['tsdf_values'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(cuda)
    fw = x['tsdf_weights'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(cuda)
    eng.prepare_input(fv, fw, x['tsdf_frame'].reshape(h, w).contiguous().to(cuda), None, 0)
    est = torch.empty((h * w, 9), device=cuda)
    eng.forward(est)
    return est


@pytest.mark.parametrize('arith', ['f16x3', 'f32'])
def test_fusion_net_large_volume_weights(cuda, arith):
    """The weight channels come from fp16 volumes and reach tens of thousands after long streams.  With weights
    that normalise them (what BN statistics of a trained net do: here the columns of the weight channels are
    scaled by 1e-4) both arithmetics agree with the fp32 reference to the stated tolerance."""
    h, w = 48, 64
    net = seeded_net('v3', False, h, w, seed=3)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.Conv2d) and m.in_channels % 19 == 0:
                m.weight[:, 9:18] *= 1e-4
    x = _large_weight_inputs(h, w)
    with torch.no_grad():
        ref = net(x)[0].permute(1, 2, 0).reshape(h * w, 9)
    eng = FusionNetEngine(net, h, w, cuda, arithmetic=arith)
    got = _run(eng, x, h, w, cuda).cpu()
    eng.check()
    err = float((got - ref).abs().max())
    print('net large weights', arith, 'max err %.2e' % err)
    assert err <= TOL, err
    eng.close()


def test_f16x3_range_guard(cuda):
    """Un-normalised weights drive activations beyond the fp16 range: the split-fp16 path must say so (never return
    silently wrong numbers), the fp32-input path must still be right."""
    h, w = 48, 64
    net = seeded_net('v3', False, h, w, seed=3)
    x = _large_weight_inputs(h, w)
    with torch.no_grad():
        ref = net(x)[0].permute(1, 2, 0).reshape(h * w, 9)
    eng = FusionNetEngine(net, h, w, cuda, arithmetic='f16x3')
    _run(eng, x, h, w, cuda)
    with pytest.raises(_lib.OjfError, match='fp16 range'):
        eng.check()
    eng.check()  # the check cleared the flag
    eng.close()
    eng = FusionNetEngine(net, h, w, cuda, arithmetic='f32')
    got = _run(eng, x, h, w, cuda).cpu()
    eng.check()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2e-3  # pre-activations ~1e4 here: 1e-7 relative
    eng.close()


@pytest.mark.parametrize('arith', ['f16x3', 'f32'])
@pytest.mark.parametrize('version,sem,n_points,growth', [('v3', False, 5, 4), ('v3', True, 3, 3), ('v2', True, 7, 5)])
def test_fusion_net_other_topologies(cuda, arith, version, sem, n_points, growth):
    """Channel counts other than 19/20 x 6 take the generic paths (stand-alone closing 1x1 convolutions and final
    conv instead of the fused tail, layer-by-layer prediction head): same tolerance."""
    h, w = 40, 56
    cfg = NS(n_points=n_points, growth_factor=growth, use_semantics=sem, output_scale=1.0, resx=w, resy=h)
    torch.manual_seed(11)
    net = getattr(model, 'FusionNet_' + version)(cfg)
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.xavier_normal_(m.weight)
            m.bias.data.normal_(0, 0.05)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    net = net.eval()
    g = torch.Generator().manual_seed(2)
    x = dict(tsdf_values=(torch.rand(1, n_points, h, w, generator=g) - 0.5) * 0.2,
             tsdf_weights=torch.rand(1, n_points, h, w, generator=g) * 4,
             tsdf_frame=torch.rand(1, 1, h, w, generator=g) * 4)
    sem_ids = torch.randint(0, 30, (h, w), generator=g, dtype=torch.uint8)
    x['semantic_frame'] = ((1 + sem_ids.float()) / 30).view(1, 1, h, w)
    with torch.no_grad():
        ref = net(x)[0].permute(1, 2, 0).reshape(h * w, n_points)
    eng = FusionNetEngine(net, h, w, cuda, arithmetic=arith)
    fv = x['tsdf_values'][0].permute(1, 2, 0).reshape(h * w, n_points).contiguous().to(cuda)
    fw = x['tsdf_weights'][0].permute(1, 2, 0).reshape(h * w, n_points).contiguous().to(cuda)
    eng.prepare_input(fv, fw, x['tsdf_frame'].reshape(h, w).contiguous().to(cuda), sem_ids.to(cuda) if sem else None, 30)
    est = torch.empty((h * w, n_points), device=cuda)
    eng.forward(est)
    eng.check()
    err = float((est.cpu() - ref).abs().max())
    print('net topology', arith, version, sem, n_points, growth, 'max err %.2e' % err)
    assert err <= TOL, err
    eng.close()


def test_fusion_net_graph_replay(cuda, monkeypatch):
    """OJF_NET_GRAPH=1 (opt-in): the launch sequence of a forward is captured once per output buffer and replayed;
    results must be identical to the plain launches, also after the output buffer changes."""
    h, w = 60, 80
    net = seeded_net('v3', False, h, w)
    x = _inputs(h, w)

    def run(env):
        if env:
            monkeypatch.setenv('OJF_NET_GRAPH', '1')
        else:
            monkeypatch.delenv('OJF_NET_GRAPH', raising=False)
        eng = FusionNetEngine(net, h, w, cuda)
        outs = []
        est_a, est_b = torch.empty((h * w, 9), device=cuda), torch.empty((h * w, 12), device=cuda)
        for est in (est_a, est_a, est_b, est_a):
            fv = x['tsdf_values'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(cuda)
            fw = x['tsdf_weights'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(cuda)
            eng.prepare_input(fv, fw, x['tsdf_frame'].reshape(h, w).contiguous().to(cuda), None, 0)
            eng.forward(est)
            outs.append(est[:, :9].clone().cpu())
        eng.check()
        eng.close()
        return outs

    plain, graph = run(False), run(True)
    for a, b in zip(plain, graph):
        assert torch.equal(a, b)


def test_fusion_net_forward_640x480(cuda):
    """BASELINE configs[4] frame size through the whole net (default arithmetic), against the fp32 CPU net."""
    h, w = 480, 640
    net = seeded_net('v3', False, h, w)
    x = _inputs(h, w)
    with torch.no_grad():
        ref = net(x)[0].permute(1, 2, 0).reshape(h * w, 9)
    eng = FusionNetEngine(net, h, w, cuda)
    got = _run(eng, x, h, w, cuda).cpu()
    eng.check()
    err = float((got - ref).abs().max())
    print('net 640x480 max err %.2e' % err)
    assert err <= TOL, err
    eng.close()


@pytest.mark.parametrize('arith', ['f16x3', 'f32'])
def test_nan_inputs_propagate_like_the_reference(cuda, arith):
    """The reference stores NaN TSDF in voxels it touches with zero total weight; a later frame gathers them and
    feeds NaN to the net, whose global-average branch spreads it to every output.  Same here, in both arithmetics -
    and it is not a range-guard violation."""
    h, w = 24, 32
    net = seeded_net('v3', False, h, w)
    x = _inputs(h, w)
    x['tsdf_values'][0, 4, 10, 11] = float('nan')
    with torch.no_grad():
        ref = net(x)[0]
    assert torch.isnan(ref).all()
    eng = FusionNetEngine(net, h, w, cuda, arithmetic=arith)
    got = _run(eng, x, h, w, cuda).cpu()
    eng.check()  # NaN in -> NaN out is not an error
    assert torch.isnan(got).all()
    eng.close()


@pytest.mark.parametrize('rolled', [False, True])
@pytest.mark.parametrize('version', ['v3', 'v2'])
@pytest.mark.parametrize('h,w,grid', [(60, 80, 64), (37, 53, 32), (240, 320, 128)])
def test_extract_to_net_equals_extract_plus_prepare(cuda, version, h, w, grid, rolled):
    """ojf_extract_to_net (the extractor writes the net's input planes itself) against ojf_extract + ojf_net_prepare_input on
    a partly filled volume: the net output must be identical bit for bit (ragged frame sizes: the last block of 64
    pixels is partial; rays leaving the volume: pad values)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import make_stream, frame_inputs
    from online_joint_depthfusion_and_semantic_amd import ops
    st = make_stream(h, w, grid, 4)
    fi = frame_inputs(st, 1)
    g = torch.Generator().manual_seed(grid)
    tsdf = ((torch.rand(grid, grid, grid, generator=g) - 0.5) * 0.2).half().to(cuda)
    wgt = (torch.rand(grid, grid, grid, generator=g) * 3).half().to(cuda)
    wgt[:, :, ::3] = 0
    depth = torch.from_numpy(fi['depth']).to(cuda).contiguous()
    net = seeded_net(version, False, h, w)
    eng = FusionNetEngine(net, h, w, cuda)
    assert eng.fused_input
    Ki, E = fi['Ki'], fi['E']
    if rolled:  # camera rolled by 90 degrees: the extractor takes its row tiles instead of the column tiles (csrc/ojf_extract.hip)
        import numpy as np
        E = E.reshape(3, 4).copy()
        E[:, :3] = E[:, :3] @ np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], dtype=np.float32)
        E = np.ascontiguousarray(E.reshape(12))
    origin, res = st.origin, st.resolution
    fv = torch.empty((9, h * w), device=cuda)
    fw = torch.empty((9, h * w), device=cuda)
    ops.extract(depth, Ki, E, origin, res, tsdf, wgt, n_points=9, out_values=fv, out_weights=fw, out_stride=h * w, planes=True)
    eng.prepare_input(fv, fw, depth, planes=True)
    a = eng.forward(torch.zeros((h * w, 9), device=cuda)).clone()
    eng.prepare_input(torch.zeros_like(fv), torch.zeros_like(fw), torch.zeros_like(depth), planes=True)  # wipe the slot
    ops.extract_to_net(depth, Ki, E, origin, res, tsdf, wgt, eng)
    b = eng.forward(torch.zeros((h * w, 9), device=cuda))
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0
    assert torch.equal(a, b)
    sem_eng = FusionNetEngine(seeded_net(version, True, h, w), h, w, cuda)
    assert not sem_eng.fused_input
    with pytest.raises(_lib.OjfError):
        ops.extract_to_net(depth, Ki, E, origin, res, tsdf, wgt, sem_eng)


def test_dense_chain_kernel_under_concurrent_chains(cuda):
    """dense_chain_kernel (csrc/ojf_net_chain.h) hands its (Block, tile) items out by ticket so that it makes progress with
    any number of resident blocks: three nets of the headline frame size on three streams, launched back to back without
    synchronising, leave every CU contested by three persistent kernels that wait for their neighbours' tiles.  Every
    forward pass must return the bits a lone forward pass returns (the kernel's arithmetic is order-free), and the range /
    stuck flag must stay clear."""
    h, w = 240, 320
    x = _inputs(h, w)
    engines, streams, lone = [], [], []
    for k in range(3):
        eng = FusionNetEngine(seeded_net('v3', False, h, w, seed=k), h, w, cuda)
        engines.append(eng)
        streams.append(torch.cuda.Stream(device=cuda))
        lone.append(_run(eng, x, h, w, cuda).clone())
        eng.check()
    torch.cuda.synchronize()
    outs = [[torch.empty((h * w, 9), device=cuda) for _ in range(6)] for _ in engines]
    fv = x['tsdf_values'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(cuda)
    fw = x['tsdf_weights'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(cuda)
    fr = x['tsdf_frame'].reshape(h, w).contiguous().to(cuda)
    torch.cuda.synchronize()
    for rep in range(6):
        for k, eng in enumerate(engines):
            with torch.cuda.stream(streams[k]):
                eng.prepare_input(fv, fw, fr, None, 0)
                eng.forward(outs[k][rep])
    torch.cuda.synchronize()
    for k, eng in enumerate(engines):
        eng.check()
        for rep in range(6):
            assert torch.equal(outs[k][rep], lone[k]), (k, rep, float((outs[k][rep] - lone[k]).abs().max()))
        eng.close()


_CHAIN_AB_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
import test_net_gpu as t
dev = torch.device('cuda:0')
h, w = int(sys.argv[2]), int(sys.argv[3])
sem = sys.argv[4] == '1'
net = t.seeded_net('v3', sem, h, w)
x = t._inputs(h, w)
eng = t.FusionNetEngine(net, h, w, dev)
fv = x['tsdf_values'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(dev)
fw = x['tsdf_weights'][0].permute(1, 2, 0).reshape(h * w, 9).contiguous().to(dev)
eng.prepare_input(fv, fw, x['tsdf_frame'].reshape(h, w).contiguous().to(dev), x['sem_ids'].contiguous().to(dev) if sem else None, 30)
est = torch.empty((h * w, 9), device=dev)
eng.forward(est); eng.check()
torch.save(est.cpu(), sys.argv[5])
"""


@pytest.mark.parametrize('h,w,sem', [(240, 320, False), (60, 80, True), (45, 77, False)])
def test_dense_chain_kernel_against_the_pair_kernels(cuda, tmp_path, h, w, sem):
    """One dense_chain_kernel launch per head (default) against one dense_pair_kernel launch per Block
    (OJF_NO_DENSE_CHAIN=1, read once per process: two child processes): same split-fp16 products, another summation
    order - the outputs agree far inside the 1e-5 bar both hold against the fp32 CPU net."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for off in (False, True):
        env = dict(os.environ)
        env.pop('OJF_NO_DENSE_CHAIN', None)
        if off:
            env['OJF_NO_DENSE_CHAIN'] = '1'
        path = str(tmp_path / ('est%d.pt' % off))
        out = subprocess.run([sys.executable, '-c', _CHAIN_AB_SCRIPT, root, str(h), str(w), '1' if sem else '0', path], env=env,
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        res.append(torch.load(path))
    err = float((res[0] - res[1]).abs().max())
    print('dense chain vs pair kernels %dx%d sem=%s: max difference %.2e' % (w, h, sem, err))
    assert err <= 2e-6, err


@pytest.mark.parametrize('switch', ['OJF_BRANCH_KERNEL', 'OJF_SUBCONV'])
@pytest.mark.parametrize('h,w,sem', [(240, 320, False), (60, 80, True), (45, 77, False), (480, 640, False)])
def test_branch_kernel_against_the_grouped_launches(cuda, tmp_path, h, w, sem, switch):
    """vortex_branch_kernel (opt-in, OJF_BRANCH_KERNEL=1, read once per process: two child processes) - both dilated 3x3 of the
    four VortexPooling branches as one LDS-resident launch over the r x r sub-images of every dilation r - and subconv_kernel
    (opt-in, OJF_SUBCONV=1: the same decomposition as small LDS-staged blocks, one launch per 3x3) against the two
    grouped conv_f16x3_kernel launches of the default path: the same split-fp16 products in another summation order (and a
    w_lo x_lo term for channels 16..19), so the net's outputs agree far inside the 1e-5 bar both hold against the fp32 CPU net.
    The frame sizes cover every item kind (20x16 / 20x14 tiles, stacked whole sub-images) and ragged sub-image borders."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for on in (False, True):
        env = dict(os.environ)
        env.pop('OJF_BRANCH_KERNEL', None)
        env.pop('OJF_SUBCONV', None)
        if on:
            env[switch] = '1'
        path = str(tmp_path / ('est%d.pt' % on))
        out = subprocess.run([sys.executable, '-c', _CHAIN_AB_SCRIPT, root, str(h), str(w), '1' if sem else '0', path], env=env,
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        res.append(torch.load(path))
    err = float((res[0] - res[1]).abs().max())
    print('%s=1 vs grouped launches %dx%d sem=%s: max difference %.2e' % (switch, w, h, sem, err))
    assert err > 0.0, switch + '=1 did not change the path'
    assert err <= 2e-6, err
