"""Inference engine of the AdapNet++ front-end on the SEGCONV kernels (csrc/ojf_seg.hip).

``SegEngine(net)`` walks an ``adapnet.AdapNet`` module tree (the reference's ``modules/adapnet.py`` tree, same
state_dict), prepacks every ``Conv2d`` together with the eval-mode ``BatchNorm2d`` that follows it, and runs the
forward pass of ``AdapNet.forward`` (adapnet.py:390-415) for inference: activations stay NHWC (torch channels_last)
fp32, every convolution + BN + residual + ReLU / sigmoid(+gate) is ONE HIP launch, concatenations are written in
place through channel slices, the three transposed convolutions of the decoder run on the same kernel (phase
expansion + pixel-shuffle store: deterministic, unlike MIOpen's atomics).  Input packing (image / 255, depth x 3),
the stem's max-pool, the squeeze chains (global average -> 1x1 convolution -> broadcast / gate: two launches) and the
final softmax + max are libojf launches too (csrc/ojf_seg_ops.hip); the reference's always-on dropout quirk of the
multi-scale units (adapnet.py:80-82) rides in the epilogue of each unit's last convolution (round 5: masks from
Philox-4x32-10 keyed by (torch.initial_seed() at construction, frame, unit) - the same distribution, not torch's
stream, which the lock-step order of the two encoders had left behind anyway; ten dropout / generator-bookkeeping
launches per frame gone), off per unit under ``no_resn50_dropout``.  The auxiliary
heads (adapnet.py:299-305) do not feed the result and are skipped.  ~772 launches of the torch forward become ~165.

The engine snapshots the weights: build it after ``load_state_dict`` (``Pipeline`` rebuilds it when the
parameters change).  No fallback path: it needs libojf and a GPU.
"""
import os

import torch

from . import segconv
from .segconv import SegConv, SegDeconv, nhwc


class _Unit:
    """Bottleneck / BottleneckSSMA of the encoder."""

    def __init__(self, m):
        self.multi = hasattr(m, 'conv2a')  # BottleneckSSMA (adapnet.py:12-84) vs torchvision-style Bottleneck, by shape not class
        self.c1 = SegConv(m.conv1, m.bn1)
        if self.multi:
            self.c2a, self.c2b = SegConv(m.conv2a, m.bn2a), SegConv(m.conv2b, m.bn2b)
            self.module = m  # the dropout flag is read at call time (AdapNet.no_resn50_dropout)
        else:
            self.c2 = SegConv(m.conv2, m.bn2)
        self.c3 = SegConv(m.conv3, m.bn3)
        self.down = SegConv(m.downsample[0], m.downsample[1]) if m.downsample is not None else None

    def __call__(self, x):
        idn = x if self.down is None else self.down(x)
        y = self.c1(x, act='relu')
        if self.multi:
            half = self.c2a.c_out
            cat = nhwc(2 * half, y.shape[2], y.shape[3], x.device, zero=False, batch=y.shape[0])
            self.c2a(y, out=cat[:, :half], act='relu')
            self.c2b(y, out=cat[:, half:], act='relu')
            self.c3.set_dropout(self.rng if self.module.dropout else None, self.drop_id)  # (read at call time, like the module)
            return self.c3(cat, residual=idn, act='relu')
        y = self.c2(y, act='relu')
        return self.c3(y, residual=idn, act='relu')


def _units(units, xs, extra=()):
    """The same unit of several encoders in lock-step (``[u(x) for u, x in zip(units, xs)]``): every layer is ONE grouped
    launch over the encoders, the two dilations of a multi-scale unit ride in the same launch.  The unit's shortcut
    convolution and its first 1x1 read the same input and do not depend on each other: one heterogeneous launch
    (``segconv.multi``), together with ``extra`` - independent (conv, x, kwargs) calls on the same input, the encoder's skip
    projection - when given."""
    u0 = units[0]
    n = len(units)
    if u0.down is not None or extra:
        calls = ([(u.down, x, {}) for u, x in zip(units, xs)] if u0.down is not None else []) + \
                [(u.c1, x, {'act': 'relu'}) for u, x in zip(units, xs)] + list(extra)
        res = segconv.multi(calls)
        idns = res[:n] if u0.down is not None else xs
        ys = res[n:2 * n] if u0.down is not None else res[:n]
    else:
        idns = xs
        ys = segconv.group([u.c1 for u in units], xs, act='relu')
    if u0.multi:
        half = u0.c2a.c_out
        cats = [nhwc(2 * half, y.shape[2], y.shape[3], y.device, zero=False, batch=y.shape[0]) for y in ys]
        segconv.group([u.c2a for u in units] + [u.c2b for u in units], ys + ys,
                      outs=[c[:, :half] for c in cats] + [c[:, half:] for c in cats], act='relu')
        for u in units:
            u.c3.set_dropout(u.rng if u.module.dropout else None, u.drop_id)  # (read at call time, like the module)
        return segconv.group([u.c3 for u in units], cats, residuals=idns, act='relu')
    ys = segconv.group([u.c2 for u in units], ys, act='relu')
    return segconv.group([u.c3 for u in units], ys, residuals=idns, act='relu')


def _encoders(encs, images, skip2_outs, skip1_outs):
    """``[e(image, s2, s1) for ...]`` for encoders of one architecture (the two modalities of the fusion net), layer by
    layer in grouped launches: half the graph nodes, no side stream.  The skip projections (adapnet.py:142-147) ride in the
    first launch of the stage that follows them."""
    xs = [im if im.shape[1] == 8 else segconv.pack_input(im.contiguous()) for im in images]
    xs = segconv.group([e.stem for e in encs], [x[:, :3] for x in xs], act='relu')
    xs = [segconv.maxpool(x) for x in xs]
    for units in zip(*[e.layers[0] for e in encs]):
        xs = _units(units, xs)
    extra = [(e.skip2, x, {'out': o}) for e, x, o in zip(encs, xs, skip2_outs)]
    for units in zip(*[e.layers[1] for e in encs]):
        xs = _units(units, xs, extra)
        extra = ()
    extra = [(e.skip1, x, {'out': o}) for e, x, o in zip(encs, xs, skip1_outs)]
    for li in (2, 3):
        for units in zip(*[e.layers[li] for e in encs]):
            xs = _units(units, xs, extra)
            extra = ()
    return xs


def _easpps(aspps, xs, outs):
    """``[a(x, out) for ...]`` in grouped launches: branch 1 together with the first step of the three cascades (all read the
    eASPP input: one heterogeneous launch), step i of the cascades and the closing convolution of every eASPP one launch each."""
    a0 = aspps[0]
    h, w = xs[0].shape[2:]
    n = a0.b1.c_out
    cats = [nhwc(5 * n, h, w, x.device, zero=False, batch=x.shape[0]) for x in xs]
    depth = len(a0.cascades[0])
    n_c = len(a0.cascades)
    first = [(a.b1, x, {'out': c[:, :n], 'act': 'relu'}) for a, x, c in zip(aspps, xs, cats)] + \
            [(casc[0], x, {'act': 'relu'}) for a, x in zip(aspps, xs) for casc in a.cascades]  # (eASPP, cascade) pairs, eASPP-major
    if len(first) <= 8 and depth > 1:
        ys = segconv.multi(first)[len(aspps):]
        start = 1
    else:
        segconv.group([a.b1 for a in aspps], xs, outs=[c[:, :n] for c in cats], act='relu')
        ys = [x for x in xs for _ in a0.cascades]
        start = 0
    for i in range(start, depth):
        convs = [casc[i] for a in aspps for casc in a.cascades]
        last = i == depth - 1
        outs_i = [c[:, (k + 1) * n:(k + 2) * n] for c in cats for k in range(n_c)] if last else None
        ys = segconv.group(convs, ys, outs=outs_i, act='relu')
    # branch 5: pool -> 1x1 conv -> ReLU -> bilinear upsampling of a 1x1 map (= broadcast), both eASPPs in two launches
    segconv.pool_fc([a.b5 for a in aspps], xs, [c[:, 4 * n:] for c in cats], act='relu')
    return segconv.group([a.fin for a in aspps], cats, outs=outs, act='relu')


def _ssmas(ssmas, cats):
    """``[s(cat) for s, cat in zip(ssmas, cats)]`` - the three SSMA blocks of the decoder (skip2 at 1/4, skip1 at 1/8, the
    eASPP outputs at 1/16 resolution: adapnet.py:404-408) do not depend on each other: squeeze, excite (sigmoid x input) and
    the closing convolution of all three as one heterogeneous launch each, three launches instead of nine."""
    sq = segconv.multi([(s.squeeze, c, {'act': 'relu'}) for s, c in zip(ssmas, cats)])
    gates = segconv.multi([(s.excite, q, {'act': 'sigmoid', 'mul': c}) for s, q, c in zip(ssmas, sq, cats)])
    return segconv.multi([(s.final, g, {}) for s, g in zip(ssmas, gates)])


class _Encoder:
    def __init__(self, m):
        r = m.res_n50_enc
        self.stem = SegConv(r.conv1, r.bn1)
        self.layers = [[_Unit(u) for u in layer] for layer in (r.layer1, r.layer2, r.layer3, r.layer4)]
        self.skip2 = SegConv(m.enc_skip2_conv, m.enc_skip2_conv_bn)
        self.skip1 = SegConv(m.enc_skip1_conv, m.enc_skip1_conv_bn)

    def __call__(self, image, skip2_out, skip1_out):
        """image: the stem's packed rows (segconv.pack_input: [1,8,H,W] NHWC, 3 channels + 5 zeros) or a plain
        [1,3,H,W] tensor.  The two skip tensors are written into the given NHWC slices."""
        x = image if image.shape[1] == 8 else segconv.pack_input(image.contiguous())
        x = self.stem(x[:, :3], act='relu')
        x = segconv.maxpool(x)
        for u in self.layers[0]:
            x = u(x)
        self.skip2(x, out=skip2_out)
        for u in self.layers[1]:
            x = u(x)
        self.skip1(x, out=skip1_out)
        for layer in self.layers[2:]:
            for u in layer:
                x = u(x)
        return x


class _EASPP:
    def __init__(self, m):
        self.b1 = SegConv(m.branch1_conv, m.branch1_bn)
        self.cascades = [[SegConv(seq[i], seq[i + 1]) for i in (0, 3, 6, 9)] for seq in m.branch234]
        self.b5 = segconv.PoolFC(m.branch5_conv)  # its BatchNorm is unused by the reference (adapnet.py:209-210)
        self.fin = SegConv(m.eASPP_fin_conv, m.eASPP_fin_bn)

    def __call__(self, x, out):
        h, w = x.shape[2:]
        n = self.b1.c_out
        cat = nhwc(5 * n, h, w, x.device, zero=False, batch=x.shape[0])
        self.b1(x, out=cat[:, :n], act='relu')
        for i, convs in enumerate(self.cascades):
            y = x
            for c in convs[:-1]:
                y = c(y, act='relu')
            convs[-1](y, out=cat[:, (i + 1) * n:(i + 2) * n], act='relu')
        segconv.pool_fc([self.b5], [x], [cat[:, 4 * n:]], act='relu')  # bilinear upsampling of a 1x1 map = broadcast
        return self.fin(cat, out=out, act='relu')


class _SSMA:
    def __init__(self, m):
        self.squeeze, self.excite = SegConv(m.link[0]), SegConv(m.link[2])
        self.final = SegConv(m.final_conv[0], m.final_conv[1])

    def __call__(self, cat, out=None):
        """cat: NHWC tensor already holding (x1, x2) along the channels."""
        gate = self.excite(self.squeeze(cat, act='relu'), act='sigmoid', mul=cat)  # x * link(x)
        return self.final(gate, out=out)


class SegEngine:
    def __init__(self, net):
        # duck-typed on the attribute names of modules/adapnet.py:356-415, so the reference's own AdapNet instance
        # (torchvision backbone included) can be handed over as well as adapnet.AdapNet
        assert hasattr(net, 'encoder_mod1') and hasattr(net, 'decoder') and not net.training, 'SegEngine: an AdapNet in eval() mode'
        self.fusion, self.n_classes = net.fusion, net.n_classes
        self.enc1 = _Encoder(net.encoder_mod1)
        if self.fusion:
            self.enc2 = _Encoder(net.encoder_mod2)
            self.aspp1, self.aspp2 = _EASPP(net.eASPP_mod1), _EASPP(net.eASPP_mod2)
            self.ssma_res, self.ssma_s1, self.ssma_s2 = _SSMA(net.ssma_res), _SSMA(net.ssma_s1), _SSMA(net.ssma_s2)
        else:
            self.aspp1 = _EASPP(net.eASPP)
        d = net.decoder
        self.deconv1 = SegDeconv(d.deconv1, d.deconv1_bn)
        self.stage2 = [SegConv(d.stage2[0], d.stage2[1], pad_in=288), SegConv(d.stage2[3], d.stage2[4])]  # (280 -> 288 input channels: groups of 32)
        self.deconv2 = SegDeconv(d.stage2[6], d.stage2[7])
        self.stage3 = [SegConv(d.stage3[0], d.stage3[1], pad_in=288), SegConv(d.stage3[3], d.stage3[4]), SegConv(d.stage3[6], d.stage3[7])]
        self.deconv3 = SegDeconv(d.stage3[8], d.stage3[9])
        self.fuse1, self.fuse2 = segconv.PoolFC(d.fuse_conv1), segconv.PoolFC(d.fuse_conv2)
        # the always-on dropout of the multi-scale units (adapnet.py:80-82) rides in their last convolution's epilogue: masks
        # from a counter-based generator keyed by (seed, frame, unit) - seeded from torch's seed at construction, one stream
        # id per unit, the frame counter advanced by the last launch of a forward pass (deconv3)
        self.rng = torch.tensor([torch.initial_seed() & 0x7fffffffffffffff, 0], dtype=torch.int64, device=next(net.parameters()).device)
        encs = [self.enc1] + ([self.enc2] if self.fusion else [])
        for e, enc in enumerate(encs):
            for li, layer in enumerate(enc.layers):
                for ui, u in enumerate(layer):
                    u.rng, u.drop_id = self.rng, 1 + ui + 16 * (li + 4 * e)
        self.deconv3.set_dropout(self.rng, advance=True)

    def _skip(self, x, skip, conv, out):
        """Decoder._skip (adapnet.py:292-296): with two modalities the skip is gated by the pooled decoder state."""
        if not self.fusion:
            out.copy_(skip)
            return
        segconv.pool_fc([conv], [x], [out], act='relu', muls=[skip])

    def forward(self, mod1, mod2=None):
        """Logits [B, n_classes, H, W] (channels_last memory) = AdapNet.forward(...)[0]; B images per call run as ONE
        pass (every tensor [B, H, W, C], every layer one launch: the weights of a layer are fetched once for all images)."""
        dev = mod1.device
        B, _, H, W = mod1.shape
        assert H % 16 == 0 and W % 16 == 0 and (mod2 is None or mod2.shape[0] == B), 'SegEngine: frame sides multiples of 16'
        h4, w4, h8, w8, h16, w16 = H // 4, W // 4, H // 8, W // 8, H // 16, W // 16
        k = 2 if self.fusion else 1
        s2 = nhwc(24 * k, h4, w4, dev, zero=False, batch=B)   # skip2 of both modalities side by side = SSMA's concatenation
        s1 = nhwc(24 * k, h8, w8, dev, zero=False, batch=B)
        top = nhwc(256 * k, h16, w16, dev, zero=False, batch=B)
        # decoder stage inputs: (deconv output 256, gated skip 24, 8 ZERO channels - rows of 288 = 9 x 32 channels let the 3x3
        # convolutions on them take the scalar tap walk).  Kept per (batch, frame shape) for the life of the engine: the pad
        # is zeroed once, not per frame, and a device graph captured around a forward pass (Pipeline keeps one per batch
        # size: fuse() and fuse_many / fuse_sequence) addresses ITS pair by raw pointer - so a pair is never evicted while
        # the engine lives (a new key does not free the others: their graphs would replay into freed memory).
        key = (B, H, W, str(dev))
        bufs = self.__dict__.setdefault('_cats', {})
        if key not in bufs:
            bufs[key] = (nhwc(288, h8, w8, dev, zero=True, batch=B), nhwc(288, h4, w4, dev, zero=True, batch=B))
        cat2, cat3 = bufs[key]
        grouped = self.fusion and not os.environ.get('OJF_SEG_TWO_STREAMS')  # (A/B switch: the round-3 flow on two streams)
        if grouped:
            # The two modality encoders have one architecture: they run in lock-step, every layer ONE grouped launch
            # (blockIdx.z = modality; the dilations of a multi-scale unit and the cascades of the eASPPs ride along): 208 ->
            # ~110 graph nodes per frame and no cross-stream fork / join, on maps (15x20, 30x40) where one encoder alone
            # leaves most CUs idle.
            x, x2 = _encoders([self.enc1, self.enc2], [mod1, mod2], [s2[:, :24], s2[:, 24:]], [s1[:, :24], s1[:, 24:]])
            _easpps([self.aspp1, self.aspp2], [x, x2], [top[:, :256], top[:, 256:]])
            del x2
        elif self.fusion:
            # the two modality encoders are independent and their 15x20 / 30x40 layers leave most CUs idle: the second
            # one runs on a side stream (also inside a graph capture, where the fork / join become graph edges): 2.06 ->
            # 1.60 ms.  Finer forks (the halves of a multi-scale unit, shortcuts, eASPP branches) were measured too: every
            # fork / join edge costs more than the ~8 us launch it hides (2.06 -> 2.56 ms without the encoder fork), and
            # forks nested inside the side stream crash hipStreamEndCapture of this ROCm build.
            main = torch.cuda.current_stream(dev)
            side = self.__dict__.get('_side')
            if side is None or side.device != dev:
                side = self.__dict__['_side'] = torch.cuda.Stream(device=dev)
            side.wait_stream(main)
            if not torch.cuda.is_current_stream_capturing():  # (inside a capture the fork / join are graph edges)
                for t in (mod2, s2, s1, top):  # allocated on the main stream, used on the side stream
                    t.record_stream(side)
            with torch.cuda.stream(side):
                x2 = self.enc2(mod2, s2[:, 24:], s1[:, 24:])
                self.aspp2(x2, top[:, 256:])
                del x2
        if not grouped:
            x = self.enc1(mod1, s2[:, :24], s1[:, :24])
            self.aspp1(x, top[:, :256])
        if self.fusion:
            if not grouped:
                main.wait_stream(side)
            skip2, skip1, x = _ssmas([self.ssma_s2, self.ssma_s1, self.ssma_res], [s2, s1, top])
        else:
            skip2, skip1, x = s2, s1, top
        self.deconv1(x, out=cat2[:, :256], act='relu')
        self._skip(cat2[:, :256], skip1, self.fuse1, cat2[:, 256:280])
        y = self.stage2[1](self.stage2[0](cat2, act='relu'), act='relu')
        self.deconv2(y, out=cat3[:, :256])
        self._skip(cat3[:, :256], skip2, self.fuse2, cat3[:, 256:280])
        y = self.stage3[1](self.stage3[0](cat3, act='relu'), act='relu')
        y = self.stage3[2](y)
        return self.deconv3(y, zero_pad=False)  # (the classes' pad channels are read by nobody)

    __call__ = forward

    def predict_many(self, images, depths=None):
        """``[predict(i, d) for i, d in zip(images, depths)]`` as ONE batched pass: -> (scores f32 [B, H*W], ids u8 [B, H*W]).
        The frames of several scenes (Pipeline.fuse_many); per-pixel results equal the single-frame pass up to the rounding
        of a different K-block order where the batch changes a layer's kernel form (include/ojf.h)."""
        B = len(images)
        H, W = images[0].shape[-2:]
        dev = images[0].device
        first = nhwc(8, H, W, dev, zero=False, batch=B)
        for b, im in enumerate(images):
            segconv.pack_input(im.contiguous(), 255.0, out=first[b:b + 1])
        second = None
        if depths is not None and depths[0] is not None:
            second = nhwc(8, H, W, dev, zero=False, batch=B)
            for b, d in enumerate(depths):
                segconv.pack_input(d.contiguous(), 1.0, out=second[b:b + 1])
        if self.fusion:
            logits = self.forward(first, second if second is not None else first)
        else:
            logits = self.forward(second if second is not None else first)
        scores, ids = segconv.softmax_max(logits)
        return scores.view(B, H * W), ids.view(B, H * W)

    def predict(self, image, depth=None):
        """Pipeline._segmentation(...).max(-1) (modules/pipeline.py:42-60,183): raw batch tensors in - ``image`` [1,3,H,W]
        as the dataset hands it over (divided by 255 here, pipeline.py:44), ``depth`` [1,H,W] / [1,1,H,W] replicated to three
        channels (:50) - per-pixel (scores f32 [H*W], ids u8 [H*W]) out.  Every step is a libojf launch."""
        if self.fusion:
            first = segconv.pack_input(image.contiguous(), 255.0)
            # DATA.input == 'image': the reference feeds the image to both encoders (modules/pipeline.py:52-55)
            second = segconv.pack_input(depth.contiguous(), 1.0) if depth is not None else first
            logits = self.forward(first, second)
        elif depth is not None:  # stage 1 nets see the DATA.input modality only (pipeline.py:52-53)
            logits = self.forward(segconv.pack_input(depth.contiguous(), 1.0))
        else:
            logits = self.forward(segconv.pack_input(image.contiguous(), 255.0))
        return segconv.softmax_max(logits)
