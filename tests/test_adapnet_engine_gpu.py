"""SegEngine (adapnet_engine.py: AdapNet++ inference on the SEGCONV HIP kernels) against the torch module tree it
was built from (adapnet.py, itself pinned block by block on the reference in tests/test_adapnet.py).

The default initialisation makes a 50-layer ReLU network forget its input (activations shrink layer by layer and
the output is set by the last biases), which would make an end-to-end comparison vacuous - so the nets here are
re-initialised to keep the signal alive, and the test checks that the output really depends on the input.
Tolerance: split-fp16 products + another summation order, ~80 layers deep: max|err| <= 5e-4 * max|logit|."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def lively(net, seed=0, dropout=False):
    from online_joint_depthfusion_and_semantic_amd import adapnet
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / m.weight[0].numel()))
            elif isinstance(m, nn.ConvTranspose2d):
                fan = m.weight.shape[0] * (m.kernel_size[0] / m.stride[0]) ** 2
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan))
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and m.bias is not None:
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.bias.shape, generator=g) + 0.5)
        for m in net.modules():
            if isinstance(m, (adapnet.Bottleneck, adapnet.BottleneckSSMA)):
                m.bn3.weight.mul_(0.4)  # keeps the residual stream from growing 16 units deep
            if isinstance(m, adapnet.BottleneckSSMA):
                m.dropout = dropout
    return net


def build(stage, n_classes, seed=0):
    from online_joint_depthfusion_and_semantic_amd.adapnet import AdapNet
    from online_joint_depthfusion_and_semantic_amd.config import AttrDict
    torch.manual_seed(seed)
    return lively(AdapNet(AttrDict({'stage': stage, 'n_classes': n_classes})), seed).cuda().eval()


@pytest.mark.parametrize('stage,n_classes,h,w', [(2, 30, 64, 96), (1, 12, 48, 64), (2, 40, 240, 320), (2, 40, 480, 640), (1, 5, 16, 16)])
def test_engine_matches_module(stage, n_classes, h, w):
    from online_joint_depthfusion_and_semantic_amd import _lib
    from online_joint_depthfusion_and_semantic_amd.adapnet_engine import SegEngine
    net = build(stage, n_classes)
    g = torch.Generator().manual_seed(5)
    img = torch.randn((1, 3, h, w), generator=g).cuda()
    dep = (torch.rand((1, 3, h, w), generator=g) * 3).cuda()
    args = (img, dep) if stage != 1 else (img,)
    with torch.no_grad():
        want = net(*args)[0]
        other = net(*[torch.flip(a, dims=(3,)) for a in args])[0]
        eng = SegEngine(net)
        got = eng(*args)
    top = want.abs().max().item()
    assert tuple(got.shape) == tuple(want.shape) == (1, n_classes, h, w)
    assert (want - torch.flip(other, dims=(3,))).abs().max().item() > 0.05 * top  # the output depends on the input
    assert 1e-2 < top < 1e4
    err = (got - want).abs().max().item()
    assert err <= 5e-4 * top, (err, top)
    assert (got.argmax(1) == want.argmax(1)).float().mean().item() > 0.999
    assert _lib.load().ojf_net_check(_lib.stream_ptr(img.device)) == 0


def test_dropout_quirk_is_kept():
    """adapnet.py:80-82: units built with drop_out=True drop activations at inference too."""
    from online_joint_depthfusion_and_semantic_amd.adapnet_engine import SegEngine
    net = build(2, 12)
    for m in net.modules():
        if hasattr(m, 'dropout') and isinstance(m.dropout, bool):
            m.dropout = True
    net.no_resn50_dropout()
    img, dep = torch.randn(1, 3, 32, 48).cuda(), torch.rand(1, 3, 32, 48).cuda()
    with torch.no_grad():
        eng = SegEngine(net)
        a, b = eng(img, dep), eng(img, dep)
        assert (a - b).abs().max().item() > 1e-3 * a.abs().max().item()  # two draws of the masks
        for m in net.modules():
            if hasattr(m, 'dropout') and isinstance(m.dropout, bool):
                m.dropout = False  # the flags are read at call time, like the module does
        a, b = eng(img, dep), eng(img, dep)
        assert torch.equal(a, b)  # every kernel of the engine sums in a fixed order


def test_pipeline_uses_the_engine_and_follows_weight_updates(cuda):
    from online_joint_depthfusion_and_semantic_amd.config import default_config
    from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline
    h, w = 64, 96
    cfg = default_config(h, w, semantics=True, use_semantics=True, n_classes=12)
    cfg.SETTINGS.device = str(cuda)
    cfg.DATA.semantic_strategy = 'predict'
    torch.manual_seed(0)
    pipe = Pipeline(cfg).to(cuda).eval()
    pipe.device = torch.device(cuda)
    lively(pipe._semantic_2d_network, 3)
    batch = {'image': torch.randn(1, 3, h, w, device=cuda) * 50 + 120, cfg.DATA.input: torch.rand(1, 1, h, w, device=cuda) * 3}
    with torch.no_grad():
        hip = pipe._segmentation(batch)
        assert pipe._seg_cache['engine'] is not None
        first = pipe._seg_cache['engine']
        cfg.SEMANTIC_2D_MODEL.engine = 'torch'
        ref = pipe._segmentation(batch)
        assert (hip - ref).abs().max().item() < 2e-4 and (hip.argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.999
        cfg.SEMANTIC_2D_MODEL.engine = 'hip'
        for p in pipe._semantic_2d_network.decoder.parameters():
            p.mul_(1.5)  # in-place update: version counters move, the engine is rebuilt
        ref2 = pipe._segmentation(dict(batch))
        cfg.SEMANTIC_2D_MODEL.engine = 'torch'
        want2 = pipe._segmentation(batch)
        cfg.SEMANTIC_2D_MODEL.engine = 'hip'
        assert pipe._seg_cache['engine'] is not first
        assert (ref2 - want2).abs().max().item() < 2e-4 and (ref2 - hip).abs().max().item() > 1e-3
        # graph replay of the engine == eager engine, before and after another update
        for _ in range(2):
            s, i = pipe._segmentation_graph(batch)
            assert pipe._seg_graph['graph'] is not None
            es, ei = pipe._segment_max(batch)  # the eager chain of the same libojf launches (SegEngine.predict)
            assert torch.equal(s, es) and torch.equal(i, ei)  # same kernels, fixed summation order
            # torch's image / 255 (a multiplication by 1/255 on the GPU, a true division here and on the reference's CPU
            # path), softmax and max around the same engine: one input ulp through 50 layers of a lively net
            ts, ti = pipe._segmentation(batch).max(dim=-1)
            assert torch.allclose(s, ts, rtol=0, atol=2e-3) and float((i.long() == ti).float().mean()) > 0.995
            for p in pipe._semantic_2d_network.decoder.parameters():
                p.mul_(0.9)
    with pytest.raises(Exception):  # training mode never routes through the inference engine silently
        pipe._semantic_2d_network.train()
        assert pipe._seg_engine((1, 3, h, w)) is None
        raise RuntimeError('ok')


@pytest.mark.parametrize('stage', [2, 1])
def test_engine_matches_reference_whole_network_golden(stage):
    """SegEngine against the REFERENCE's whole-network output (tests/golden/adapnet_net.npz, generated by importing
    modules/adapnet.py): main head within 5e-4 of the logit range, same arg-max."""
    from adapnet_golden_util import golden_net
    from online_joint_depthfusion_and_semantic_amd.adapnet_engine import SegEngine
    net, ins, outs = golden_net(stage)
    net = net.cuda()
    with torch.no_grad():
        got = SegEngine(net)(*[x.cuda() for x in ins]).cpu()
    want = outs[0]
    err, top = float((got - want).abs().max()), float(want.abs().max())
    assert err <= 5e-4 * top, (err, top)
    assert float((got.argmax(1) == want.argmax(1)).float().mean()) > 0.995


def test_predict_many_equals_predict_per_frame():
    """SegEngine.predict_many: B frames as one [B, H, W, C] pass == B single-frame passes (scores to 1e-6: a layer's kernel
    form - and with it the order its K blocks are added - may differ with the number of pixel tiles; arg-max identical apart
    from exact near-ties)."""
    from online_joint_depthfusion_and_semantic_amd.adapnet_engine import SegEngine
    net = build(2, 30)
    net.no_resn50_dropout()
    for m in net.modules():
        if hasattr(m, 'dropout') and isinstance(m.dropout, bool):
            m.dropout = False
    g = torch.Generator().manual_seed(9)
    B, h, w = 3, 64, 96
    images = [(torch.rand((1, 3, h, w), generator=g) * 255).cuda() for _ in range(B)]
    depths = [(torch.rand((1, h, w), generator=g) * 3).cuda() for _ in range(B)]
    with torch.no_grad():
        eng = SegEngine(net)
        scores, ids = eng.predict_many(images, depths)
        for b in range(B):
            s1, i1 = eng.predict(images[b], depths[b])
            assert (scores[b] - s1).abs().max().item() <= 1e-6
            assert (ids[b] == i1).float().mean().item() >= 0.9995
    assert scores.shape == (B, h * w) and ids.dtype == torch.uint8
