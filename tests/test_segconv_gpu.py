"""SEGCONV (ojf_segconv_*): every convolution shape of AdapNet++ at 320x240 against torch's fp32 conv2d (+ eval
BatchNorm, residual, activation) on the same device.  Tolerance: the split-fp16 MFMA products carry ~2^-22 relative
error per term, fp32 accumulation order differs from MIOpen's: |err| <= 2e-5 * (sum_k |w||x| scale) - checked as
max|err| <= 3e-5 * max|ref| + 1e-6 per layer."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def to_nhwc(x, pad_to=8):
    from online_joint_depthfusion_and_semantic_amd.segconv import nhwc
    c = x.shape[1]
    buf = nhwc((c + pad_to - 1) // pad_to * pad_to, x.shape[2], x.shape[3], x.device)
    buf[:, :c] = x
    return buf[:, :c]


# (c_in, c_out, k, stride, dilation, padding, h, w)  - the layer shapes of adapnet.py at a 240x320 frame
SHAPES = [
    (3, 64, 7, 2, 1, 3, 240, 320),      # stem
    (64, 64, 1, 1, 1, 0, 60, 80), (64, 64, 3, 1, 1, 1, 60, 80), (64, 256, 1, 1, 1, 0, 60, 80), (256, 64, 1, 1, 1, 0, 60, 80),
    (256, 24, 1, 1, 1, 0, 60, 80),      # skip2
    (256, 128, 1, 1, 1, 0, 60, 80), (128, 128, 3, 2, 1, 1, 60, 80), (256, 512, 1, 2, 1, 0, 60, 80), (128, 512, 1, 1, 1, 0, 30, 40),
    (128, 32, 3, 1, 2, 2, 30, 40),      # multi-scale halves of layer2
    (64, 512, 1, 1, 1, 0, 30, 40), (512, 24, 1, 1, 1, 0, 30, 40),
    (512, 256, 1, 1, 1, 0, 30, 40), (256, 256, 3, 2, 1, 1, 30, 40), (512, 1024, 1, 2, 1, 0, 30, 40), (256, 1024, 1, 1, 1, 0, 15, 20),
    (256, 128, 3, 1, 16, 16, 15, 20), (1024, 256, 1, 1, 1, 0, 15, 20),
    (1024, 512, 1, 1, 1, 0, 15, 20), (512, 256, 3, 1, 8, 8, 15, 20), (512, 2048, 1, 1, 1, 0, 15, 20), (1024, 2048, 1, 1, 1, 0, 15, 20),
    (2048, 64, 1, 1, 1, 0, 15, 20), (64, 64, 3, 1, 12, 12, 15, 20), (64, 256, 1, 1, 1, 0, 15, 20), (1280, 256, 1, 1, 1, 0, 15, 20),
    (2048, 256, 1, 1, 1, 0, 1, 1),      # eASPP image-pooling branch
    (512, 16, 3, 1, 1, 1, 15, 20), (16, 512, 3, 1, 1, 1, 15, 20), (512, 256, 3, 1, 1, 1, 15, 20),  # SSMA res
    (48, 4, 3, 1, 1, 1, 30, 40), (4, 48, 3, 1, 1, 1, 30, 40), (48, 24, 3, 1, 1, 1, 60, 80),         # SSMA skips
    (280, 256, 3, 1, 1, 1, 30, 40), (256, 256, 3, 1, 1, 1, 60, 80), (256, 30, 1, 1, 1, 0, 60, 80), (256, 24, 1, 1, 1, 0, 1, 1),
    (24, 40, 3, 1, 1, 1, 7, 5), (8, 8, 5, 3, 2, 4, 33, 17),  # odd geometry: ragged tiles, 5x5, stride 3
    # >= 256 blocks of 64 channels x 128 pixels: the LDS-shared-weights kernel (640x480 frames)
    (256, 256, 3, 1, 1, 1, 120, 160), (64, 64, 3, 1, 1, 1, 240, 320), (64, 128, 3, 2, 1, 1, 240, 320), (40, 72, 3, 1, 2, 2, 131, 157),
    (256, 30, 1, 1, 1, 0, 240, 320),
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_conv_bn_act_matches_torch(shape):
    from online_joint_depthfusion_and_semantic_amd.segconv import SegConv
    cin, cout, k, s, d, p, h, w = shape
    g = torch.Generator().manual_seed(cin * 131 + cout)
    conv = nn.Conv2d(cin, cout, k, stride=s, dilation=d, padding=p, bias=(cout % 3 == 0)).cuda()
    bn = nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / np.sqrt(cin * k * k))
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(cout, generator=g) * 2 + 0.1)
    x = (torch.randn((1, cin, h, w), generator=g) * 2).cuda()
    with torch.no_grad():
        lin = bn(conv(x))
        res = torch.randn(lin.shape, generator=g).cuda()
        gate = torch.rand(lin.shape, generator=g).cuda()
        cases = {'plain': (SegConv(conv), conv(x), {}),
                 'bn_relu': (SegConv(conv, bn), F.relu(lin), {'act': 'relu'}),
                 'bn_res_relu': (SegConv(conv, bn), F.relu(lin + res), {'act': 'relu', 'residual': to_nhwc(res)}),
                 'sigmoid_mul': (SegConv(conv, bn), torch.sigmoid(lin) * gate, {'act': 'sigmoid', 'mul': to_nhwc(gate)})}
    xin = to_nhwc(x)
    for name, (op, want, kw) in cases.items():
        got = op(xin, **kw)
        assert got.shape == want.shape, name
        err = (got - want).abs().max().item()
        assert err <= 3e-5 * want.abs().max().item() + 1e-6, (name, err, want.abs().max().item())
    from online_joint_depthfusion_and_semantic_amd import _lib
    assert _lib.load().ojf_net_check(_lib.stream_ptr(x.device)) == 0  # range guard silent


@pytest.mark.parametrize('cin,cout,stride,h,w', [(256, 256, 2, 15, 20), (256, 256, 2, 30, 40), (30, 30, 4, 60, 80), (12, 12, 4, 16, 24),
                                                 (40, 40, 4, 9, 7), (8, 6, 2, 5, 3), (16, 5, 8, 4, 6)])
def test_transposed_conv_matches_torch(cin, cout, stride, h, w):
    """ojf_segdeconv_create: ConvTranspose2d(kernel 2s, stride s, padding s/2) (+ eval BN, ReLU) - the decoder's
    upsampling layers (adapnet.py:226,236,247) and odd sizes."""
    from online_joint_depthfusion_and_semantic_amd.segconv import SegDeconv, nhwc
    g = torch.Generator().manual_seed(cin + 7 * cout + stride)
    dc = nn.ConvTranspose2d(cin, cout, 2 * stride, stride=stride, padding=stride // 2).cuda()
    bn = nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        dc.weight.copy_(torch.randn(dc.weight.shape, generator=g) / np.sqrt(cin * 4))
        dc.bias.copy_(torch.randn(cout, generator=g) * 0.2)
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(cout, generator=g) * 2 + 0.1)
        x = (torch.randn((1, cin, h, w), generator=g) * 2).cuda()
        xin = to_nhwc(x)
        for op, want, act in ((SegDeconv(dc), dc(x), None), (SegDeconv(dc, bn), F.relu(bn(dc(x))), 'relu')):
            got = op(xin, act=act)
            assert got.shape == want.shape
            assert (got - want).abs().max().item() <= 3e-5 * want.abs().max().item() + 1e-6
            assert torch.equal(got, op(xin, act=act))  # fixed summation order
        # into a channel slice of a wider buffer (the decoder's concatenations)
        cat = nhwc(cout + 24, h * stride, w * stride, x.device)
        SegDeconv(dc, bn)(xin, out=cat[:, :cout], act='relu')
        assert (cat[:, :cout] - want).abs().max().item() <= 3e-5 * want.abs().max().item() + 1e-6 and float(cat[:, cout:].abs().max()) == 0


def test_channel_slices_and_guard():
    """Concatenation by pointer + stride: two convs write halves of one buffer, a third reads a slice of it."""
    from online_joint_depthfusion_and_semantic_amd import _lib
    from online_joint_depthfusion_and_semantic_amd.segconv import SegConv, nhwc
    torch.manual_seed(0)
    a, b, c = nn.Conv2d(16, 24, 3, padding=1).cuda(), nn.Conv2d(16, 40, 1).cuda(), nn.Conv2d(40, 8, 3, padding=2, dilation=2).cuda()
    x = torch.randn(1, 16, 21, 13).cuda()
    xin = to_nhwc(x)
    cat = nhwc(64, 21, 13, x.device)
    SegConv(a)(xin, out=cat[:, :24], act='relu')
    SegConv(b)(xin, out=cat[:, 24:], act=None)
    with torch.no_grad():
        want = torch.cat((F.relu(a(x)), b(x)), 1)
        assert (cat - want).abs().max().item() <= 3e-5 * want.abs().max().item()
        y = SegConv(c)(cat[:, 24:])
        assert (y - c(want[:, 24:])).abs().max().item() <= 1e-4
    lib = _lib.load()
    # values beyond the fp16 range raise the guard instead of returning Inf silently
    big = to_nhwc(torch.full((1, 16, 21, 13), 3.0e4).cuda())
    with torch.no_grad():
        a.weight.fill_(1.0)
    SegConv(a)(big)
    assert lib.ojf_net_check(_lib.stream_ptr(x.device)) != 0 and b'fp16 range' in lib.ojf_last_error()
    assert lib.ojf_net_check(_lib.stream_ptr(x.device)) == 0  # cleared
    # argument validation
    with pytest.raises(AssertionError):
        SegConv(a)(x)  # NCHW-contiguous input is refused
    with pytest.raises(RuntimeError):
        SegConv(a)(nhwc(12, 5, 5, x.device)[:, :16] if False else nhwc(20, 5, 5, x.device)[:, 1:17])  # misaligned rows


def test_front_end_operators_match_torch(cuda):
    """csrc/ojf_seg_ops.hip against the torch operators they replace in the AdapNet++ front end: input packing
    (image / 255, depth x 3: pipeline.py:44,50), MaxPool2d(3, 2, 1), global average + broadcast / gate, softmax + max."""
    from online_joint_depthfusion_and_semantic_amd import segconv
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    img = (torch.randn(1, 3, 37, 53, generator=g) * 60 + 100).to(cuda)
    packed = segconv.pack_input(img, 255.0)  # a true division like the reference's CPU path (torch's GPU kernel multiplies by 1/255)
    assert packed.shape == (1, 8, 37, 53) and torch.equal(packed[:, :3].cpu(), img.cpu() / 255.0) and float(packed[:, 3:].abs().max()) == 0
    depth = (torch.rand(1, 37, 53, generator=g) * 4).to(cuda)
    pd = segconv.pack_input(depth, 1.0)
    assert torch.equal(pd[:, :3], depth.view(1, 1, 37, 53).repeat(1, 3, 1, 1))
    for C, H, W in ((64, 37, 53), (5, 8, 8), (24, 1, 7)):
        x = segconv.nhwc((C + 7) // 8 * 8, H, W, cuda)[:, :C]
        x.copy_(torch.randn(1, C, H, W, generator=g))
        assert torch.equal(segconv.maxpool(x), F.max_pool2d(x, 3, stride=2, padding=1))
        m = segconv.mean(x)
        want = x.mean(dim=(2, 3), keepdim=True)
        assert m.shape == want.shape and torch.allclose(m, want, rtol=0, atol=2e-6)
        out = segconv.nhwc(C + 8, H, W, cuda)
        segconv.broadcast(m, out[:, 8:])
        assert torch.equal(out[:, 8:], m.expand(1, C, H, W)) and float(out[:, :8].abs().max()) == 0
        segconv.broadcast(m, out[:, 8:], mul=x)
        assert torch.equal(out[:, 8:], m * x)
    for C in (12, 30, 40):
        logits = segconv.nhwc((C + 7) // 8 * 8, 19, 23, cuda)[:, :C]
        logits.copy_(torch.randn(1, C, 19, 23, generator=g) * 3)
        logits[0, 1, 0, 0] = logits[0, 4, 0, 0] = 9.0  # a tie: the first maximum wins (torch.max on the CPU)
        s, i = segconv.softmax_max(logits)
        ws, wi = torch.softmax(logits.cpu(), dim=1).max(dim=1)
        assert i.dtype == torch.uint8 and torch.equal(i.cpu().long().view(19, 23), wi[0])
        assert torch.allclose(s.cpu().view(19, 23), ws[0], rtol=0, atol=1e-6)


@pytest.mark.parametrize('cin,cout,k,stride,dils,h,w,same_form', [
    (256, 64, 1, 1, (1, 1), 60, 80, False),   # two encoders, one layer: split-K alone, the 64 x 64 GEMM-shaped form as a pair
    (256, 256, 3, 1, (1, 2), 60, 80, True),   # the GEMM-shaped form alone and as a pair
    (128, 32, 3, 1, (1, 2, 1, 2), 30, 40, True),    # both dilations of a multi-scale unit of both encoders
    (64, 64, 3, 1, (3, 6, 12, 3, 6, 12), 15, 20, True),  # the three cascades of two eASPPs (split-K kernel)
    (1024, 256, 1, 1, (1, 1), 15, 20, True), (64, 64, 3, 1, (1, 1), 240, 320, True)])  # ... and the LDS-shared-weights kernel
def test_grouped_launch_gives_the_bits_of_single_launches(cin, cout, k, stride, dils, h, w, same_form):
    """ojf_segconv_forward_group (blockIdx.z = member) against one ojf_segconv_forward per member: same kernels, same
    arithmetic, same order - identical bits, with residuals and ReLU, on members of different dilation.  Where the block
    count of the pair moves the launch to another kernel form (same_form False: the K sum is then split differently),
    the bar is the fp32 rounding of that sum instead."""
    from online_joint_depthfusion_and_semantic_amd import segconv
    dev = torch.device('cuda:0')
    torch.manual_seed(cin + cout + len(dils))
    convs, xs, ress = [], [], []
    for d in dils:
        conv = nn.Conv2d(cin, cout, k, stride=stride, dilation=d, padding=d * (k // 2)).to(dev)
        bn = nn.BatchNorm2d(cout).to(dev).eval()
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
        convs.append(segconv.SegConv(conv, bn))
        xs.append(to_nhwc(torch.randn(1, cin, h, w, device=dev)))
        ho, wo = convs[-1].out_size(h, w)
        ress.append(to_nhwc(torch.randn(1, cout, ho, wo, device=dev)))
    single = [c(x, act='relu', residual=r).clone() for c, x, r in zip(convs, xs, ress)]
    grouped = segconv.group(convs, xs, act='relu', residuals=ress)
    torch.cuda.synchronize()
    for a, b in zip(single, grouped):
        if same_form:
            assert torch.equal(a, b)
        else:
            assert torch.allclose(a, b, rtol=0, atol=4e-6 * float(a.abs().max()))


# ---- batches: [B, H, W, C] tensors, the frames of several scenes in one pass (round 5) ------------------------------------
def to_nhwc_batch(x, pad_to=8):
    from online_joint_depthfusion_and_semantic_amd.segconv import nhwc
    b, c = x.shape[:2]
    buf = nhwc((c + pad_to - 1) // pad_to * pad_to, x.shape[2], x.shape[3], x.device, batch=b)
    buf[:, :c] = x
    return buf[:, :c]


@pytest.mark.parametrize('shape', [(64, 64, 3, 1, 1, 1, 60, 80), (256, 512, 1, 2, 1, 0, 60, 80), (512, 256, 3, 1, 8, 8, 15, 20), (1024, 256, 1, 1, 1, 0, 15, 20),
                                   (48, 4, 3, 1, 1, 1, 30, 40), (24, 40, 3, 1, 1, 1, 7, 5), (8, 8, 5, 3, 2, 4, 33, 17), (3, 64, 7, 2, 1, 3, 48, 64)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('B', [2, 3])
def test_batched_convolution_equals_per_image_calls(shape, B):
    """A [B, H, W, C] tensor through one launch == the B images through B launches: a pixel's sum never crosses an image (taps
    outside ITS image read zeros, 15x20 = 18.75 pixel tiles: tiles straddle images), residual / gate rows and the pad
    channels follow the batch.  Equal up to the K-block order of a different kernel form (3e-6 of the layer's scale)."""
    from online_joint_depthfusion_and_semantic_amd.segconv import SegConv, group
    cin, cout, k, s, d, p, h, w = shape
    g = torch.Generator().manual_seed(cin * 7 + cout + B)
    conv = nn.Conv2d(cin, cout, k, stride=s, dilation=d, padding=p, bias=True).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / np.sqrt(cin * k * k))
    op = SegConv(conv)
    x = (torch.randn((B, cin, h, w), generator=g) * 2).cuda()
    Ho, Wo = op.out_size(h, w)
    res = torch.randn((B, cout, Ho, Wo), generator=g).cuda()
    xb, rb = to_nhwc_batch(x), to_nhwc_batch(res)
    got = op(xb, act='relu', residual=rb)
    assert got.shape == (B, cout, Ho, Wo)
    with torch.no_grad():
        ref = F.relu(conv(x) + res)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 3e-5 * scale + 1e-6
    for b in range(B):
        one = op(to_nhwc(x[b:b + 1]), act='relu', residual=to_nhwc(res[b:b + 1]))
        assert (got[b:b + 1] - one).abs().max().item() <= 3e-6 * scale
    if cout % 8:  # pad channels of own rows are zeroed for every image
        wide = got.as_strided((B, (cout + 7) // 8 * 8, Ho, Wo), got.stride())
        assert float(wide[:, cout:].abs().max()) == 0.0
    # grouped launch on batches
    outs = group([op, op], [xb, xb], act='relu', residuals=[rb, rb])  # (two members may take another kernel form than one)
    assert torch.equal(outs[0], outs[1]) and (outs[0] - got).abs().max().item() <= 3e-6 * scale


def test_batched_front_end_operators():
    """max-pool, transposed convolution (pixel-shuffle store per image), squeeze chain (a mean per image) and softmax + max on
    [B, H, W, C] tensors against the per-image calls."""
    from online_joint_depthfusion_and_semantic_amd import segconv
    from online_joint_depthfusion_and_semantic_amd.segconv import SegDeconv, PoolFC
    g = torch.Generator().manual_seed(77)
    B, h, w = 3, 15, 20
    x = torch.randn((B, 64, 2 * h, 2 * w), generator=g).cuda()
    xb = to_nhwc_batch(x)
    mp = segconv.maxpool(xb)
    assert torch.equal(mp, F.max_pool2d(x, 3, 2, 1))
    de = nn.ConvTranspose2d(64, 24, 4, stride=2, padding=1).cuda()
    with torch.no_grad():
        de.weight.copy_(torch.randn(de.weight.shape, generator=g) / 16)
    dop = SegDeconv(de)
    y = torch.randn((B, 64, h, w), generator=g).cuda()
    got = dop(to_nhwc_batch(y), act='relu')
    with torch.no_grad():
        ref = F.relu(de(y))
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 3e-5 * ref.abs().max().item() + 1e-6
    for b in range(B):
        assert (dop(to_nhwc(y[b:b + 1]), act='relu') - got[b:b + 1]).abs().max().item() <= 3e-6 * ref.abs().max().item()
    fc = nn.Conv2d(64, 24, 1).cuda()
    pf = PoolFC(fc)
    skip = torch.rand((B, 24, 2 * h, 2 * w), generator=g).cuda()
    out = segconv.nhwc(24, 2 * h, 2 * w, x.device, zero=False, batch=B)
    segconv.pool_fc([pf], [to_nhwc_batch(y)], [out], act='relu', muls=[to_nhwc_batch(skip)])
    with torch.no_grad():
        want = F.relu(fc(y.mean(dim=(2, 3), keepdim=True))) * skip
    assert (out - want).abs().max().item() <= 2e-6 * want.abs().max().item() + 1e-7
    logits = torch.randn((B, 30, 2 * h, 2 * w), generator=g).cuda()
    sc, ids = segconv.softmax_max(to_nhwc_batch(logits))
    ws, wi = torch.softmax(logits, 1).max(1)
    assert (sc.view(B, -1) - ws.reshape(B, -1)).abs().max().item() <= 1e-6 and torch.equal(ids.view(B, -1).long(), wi.reshape(B, -1))


def test_heterogeneous_multi_launch_equals_separate_calls():
    """segconv.multi: independent convolutions of DIFFERENT shapes (a unit's shortcut next to its first 1x1 and the encoder's
    skip projection; the three SSMA blocks on three map sizes) in one launch where their kernel forms allow it - every member
    equals its own single call up to the K-block order of another form (3e-6 of the layer's scale)."""
    from online_joint_depthfusion_and_semantic_amd.segconv import SegConv, multi
    g = torch.Generator().manual_seed(123)

    def conv(cin, cout, k, s=1, d=1):
        m = nn.Conv2d(cin, cout, k, stride=s, dilation=d, padding=d * (k // 2), bias=True).cuda()
        with torch.no_grad():
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) / np.sqrt(cin * k * k))
        return m
    x15 = to_nhwc((torch.randn((1, 1024, 15, 20), generator=g)).cuda())
    x30 = to_nhwc((torch.randn((1, 512, 30, 40), generator=g)).cuda())
    x60 = to_nhwc((torch.randn((1, 48, 60, 80), generator=g)).cuda())
    x30b = to_nhwc((torch.randn((1, 48, 30, 40), generator=g)).cuda())
    x15b = to_nhwc((torch.randn((1, 512, 15, 20), generator=g)).cuda())
    gate = to_nhwc(torch.rand((1, 512, 15, 20), generator=g).cuda())
    cases = [
        # layer4 unit 0: shortcut 1024 -> 2048, first 1x1 1024 -> 512 (both in-block split-K), two encoders
        [(conv(1024, 2048, 1), x15, {}), (conv(1024, 2048, 1), x15, {}), (conv(1024, 512, 1), x15, {'act': 'relu'}), (conv(1024, 512, 1), x15, {'act': 'relu'})],
        # layer3 unit 0 with the skip projection: stride-2 shortcut, 1x1, 512 -> 24
        [(conv(512, 1024, 1, s=2), x30, {}), (conv(512, 256, 1), x30, {'act': 'relu'}), (conv(512, 24, 1), x30, {})],
        # SSMA squeeze on three map sizes, then a plain-form trio (tiny K) with a gate
        [(conv(48, 4, 3), x60, {'act': 'relu'}), (conv(48, 4, 3), x30b, {'act': 'relu'}), (conv(512, 16, 3), x15b, {'act': 'relu'})],
        [(conv(16, 512, 3), to_nhwc(torch.randn((1, 16, 15, 20), generator=g).cuda()), {'act': 'sigmoid', 'mul': gate}),
         (conv(4, 48, 3), to_nhwc(torch.randn((1, 4, 30, 40), generator=g).cuda()), {'act': 'sigmoid'})],
        # a mix of forms (split-K + plain): falls back to separate launches, same results
        [(conv(512, 256, 3), x15b, {'act': 'relu'}), (conv(4, 48, 3), to_nhwc(torch.randn((1, 4, 60, 80), generator=g).cuda()), {})],
    ]
    for calls in cases:
        ops = [(SegConv(m), x, kw) for m, x, kw in calls]
        got = multi(ops)
        for (op, x, kw), y in zip(ops, got):
            one = op(x, **kw)
            scale = one.abs().max().item()
            assert y.shape == one.shape and (y - one).abs().max().item() <= 3e-6 * scale + 1e-7, (op.c_in, op.c_out)
            if op.c_out % 8:
                wide = y.as_strided((1, (op.c_out + 7) // 8 * 8, y.shape[2], y.shape[3]), y.stride())
                assert float(wide[:, op.c_out:].abs().max()) == 0.0
    from online_joint_depthfusion_and_semantic_amd import _lib
    assert _lib.load().ojf_net_check(_lib.stream_ptr(x15.device)) == 0
