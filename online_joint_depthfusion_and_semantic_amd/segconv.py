"""Host side of the SEGCONV kernels (csrc/ojf_seg.hip, include/ojf.h ``ojf_segconv_*``): one prepacked
``nn.Conv2d`` (+ eval-mode ``BatchNorm2d`` + residual + activation) of the AdapNet++ front-end
(modules/adapnet.py) per object, run on NHWC (torch ``channels_last``) fp32 tensors of batch 1.

No fallback: without libojf / a GPU the calls raise."""
import ctypes

import torch

from . import _lib

ACT = {None: 0, 'none': 0, 'relu': 1, 'sigmoid': 2}
ZERO_PAD = 0x100  # include/ojf.h OJF_SEG_ACT_ZERO_PAD: the launch zeroes the pad channels of rows this module allocated


def nhwc(channels, h, w, device, zero=True, batch=1):
    """[batch, channels, h, w] fp32 tensor in channels_last memory (channels padded by the caller where a consumer
    reads groups of 8)."""
    make = torch.zeros if zero else torch.empty
    return make((batch, h, w, channels), dtype=torch.float32, device=device).permute(0, 3, 1, 2)


def _rows(t):
    """(pointer to channel 0 of pixel 0, floats per pixel row) of an NHWC view [B, C, H, W] (channel slices allowed; the
    images of a batch follow each other: batch stride = H * W rows)."""
    assert t.dim() == 4 and t.dtype == torch.float32 and t.is_cuda
    B, C, H, W = t.shape
    row = t.stride(3) if W > 1 else (t.stride(2) if H > 1 else max(C, 1))
    assert (C == 1 or t.stride(1) == 1) and (H == 1 or W == 1 or t.stride(2) == W * row), 'segconv wants NHWC (channels_last) tensors'
    assert B == 1 or t.stride(0) == H * W * row, 'segconv wants the images of a batch back to back'
    return t.data_ptr(), row


def _folded(bias, bn, n):
    """(scale or None, bias) of eval-mode ``bn`` applied after a layer with ``bias``: y = (x + b - mean) * g / sqrt(var + eps) + beta."""
    bias = bias.detach().to('cpu', torch.float32) if bias is not None else torch.zeros(n)
    if bn is None:
        return None, bias.contiguous()
    inv = (bn.running_var.detach().cpu().float() + bn.eps).rsqrt()
    gamma = bn.weight.detach().cpu().float() if bn.weight is not None else torch.ones_like(inv)
    beta = bn.bias.detach().cpu().float() if bn.bias is not None else torch.zeros_like(inv)
    scale = (gamma * inv).contiguous()
    return scale, ((bias - bn.running_mean.detach().cpu().float()) * scale + beta).contiguous()


class SegConv:
    """``conv`` [+ ``bn`` in eval mode] prepacked for the device; call with NHWC tensors."""

    def __init__(self, conv, bn=None, pad_in=0):
        """pad_in > c_in: the layer is packed for rows of ``pad_in`` input channels (zero weights for the extra ones, which
        must hold finite values): a multiple of 32 lets the launch take the scalar tap walk of the GEMM-shaped kernel."""
        _lib.require_gpu()
        lib = _lib.load()
        assert isinstance(conv, torch.nn.Conv2d) and conv.groups == 1 and conv.padding_mode == 'zeros'
        k, s, d, p = conv.kernel_size, conv.stride, conv.dilation, conv.padding
        assert k[0] == k[1] and s[0] == s[1] and d[0] == d[1] and p[0] == p[1], 'square geometry only'
        w = conv.weight.detach().to('cpu', torch.float32).contiguous()
        if pad_in > w.shape[1]:
            w = torch.cat([w, torch.zeros(w.shape[0], pad_in - w.shape[1], *w.shape[2:])], dim=1).contiguous()
        scale, bias = _folded(conv.bias, bn, w.shape[0])
        self.c_out, self.c_in = int(w.shape[0]), int(w.shape[1])
        self.k, self.stride, self.dil, self.pad = int(k[0]), int(s[0]), int(d[0]), int(p[0])
        handle = ctypes.c_void_p()
        rc = lib.ojf_segconv_create(ctypes.byref(handle), w.data_ptr(), None if scale is None else scale.data_ptr(),
                                    bias.data_ptr(), self.c_in, self.c_out, self.k, self.stride, self.dil, self.pad)
        _lib.check(rc, 'ojf_segconv_create')
        self._h, self._lib = handle, lib

    def __del__(self):
        if getattr(self, '_h', None):
            self._lib.ojf_segconv_destroy(self._h)
            self._h = None

    def set_dropout(self, state, stream_id=0, advance=False):
        """The always-on ``nn.Dropout(0.5)`` of a multi-scale unit (adapnet.py:80-82) in this layer's epilogue
        (``ojf_segconv_set_dropout``): ``state`` = int64 device tensor {seed, frame} or None (off); ``advance``: this layer's
        launch increments ``frame`` instead."""
        key = (None if state is None else state.data_ptr(), int(stream_id), bool(advance))
        if getattr(self, '_drop_key', (None, 0, False)) == key:
            return
        _lib.check(self._lib.ojf_segconv_set_dropout(self._h, key[0], key[1], int(key[2])), 'ojf_segconv_set_dropout')
        self._drop_key = key

    def out_size(self, h, w):
        span = self.dil * (self.k - 1) + 1
        return (h + 2 * self.pad - span) // self.stride + 1, (w + 2 * self.pad - span) // self.stride + 1

    def __call__(self, x, out=None, act=None, residual=None, mul=None):
        """x: NHWC view [1, >=c_in, H, W] whose rows hold round_up(c_in, 8) finite channels.  out: NHWC view to
        write (default: a fresh tensor with c_out channels, padded to a multiple of 8 with zeros)."""
        H, W = x.shape[2:]
        B = x.shape[0]
        Ho, Wo = self.out_size(H, W)
        flags = 0
        if out is None:  # own rows, padded to a multiple of 8 channels: the launch itself zeroes the pad (no fill kernel)
            out = nhwc((self.c_out + 7) // 8 * 8, Ho, Wo, x.device, zero=False, batch=B)[:, :self.c_out]
            flags = ZERO_PAD if self.c_out % 8 else 0
        assert out.shape[1] == self.c_out and tuple(out.shape[2:]) == (Ho, Wo) and out.shape[0] == B
        xp, xs = _rows(x)
        op, os_ = _rows(out)
        rp, rs = _rows(residual) if residual is not None else (None, 0)
        mp, ms = _rows(mul) if mul is not None else (None, 0)
        rc = self._lib.ojf_segconv_forward_batch(self._h, B, xp, xs, op, os_, rp, rs, mp, ms, ACT[act] | flags, H, W, _lib.stream_ptr(x.device))
        _lib.check(rc, 'ojf_segconv_forward')
        return out


def group(convs, xs, outs=None, act=None, residuals=None, muls=None):
    """``[c(x, out=o, act=act, residual=r, mul=m) for ...]`` as ONE launch (``ojf_segconv_forward_group``): up to 8
    ``SegConv`` of one shape (channels, kernel size, stride; dilation / padding may differ) on same-sized inputs - the two
    modality encoders in lock-step, the dilations of a multi-scale unit, the cascades of an eASPP.  Same bits as the
    single calls.  Returns the list of outputs."""
    n = len(convs)
    assert 1 <= n <= 8 and len(xs) == n
    c0 = convs[0]
    H, W = xs[0].shape[2:]
    B = xs[0].shape[0]
    Ho, Wo = c0.out_size(H, W)
    flags = 0
    if outs is None:
        outs = [nhwc((c0.c_out + 7) // 8 * 8, Ho, Wo, xs[0].device, zero=False, batch=B)[:, :c0.c_out] for _ in range(n)]
        flags = ZERO_PAD if c0.c_out % 8 else 0
    assert len(outs) == n and all(o.shape[1] == c0.c_out and tuple(o.shape[2:]) == (Ho, Wo) and o.shape[0] == B for o in outs)
    assert all(tuple(x.shape[2:]) == (H, W) and x.shape[0] == B for x in xs)

    def rows(ts):
        if ts is None:
            return None, 0
        pr = [_rows(t) for t in ts]
        assert len(pr) == n and all(r[1] == pr[0][1] for r in pr), 'segconv.group: the members share their row strides'
        return (ctypes.c_void_p * n)(*[r[0] for r in pr]), pr[0][1]

    xp, xs_ = rows(xs)
    op, os_ = rows(outs)
    rp, rs = rows(residuals)
    mp, ms = rows(muls)
    handles = (ctypes.c_void_p * n)(*[c._h.value for c in convs])
    rc = c0._lib.ojf_segconv_forward_group_batch(n, B, handles, xp, xs_, op, os_, rp, rs, mp, ms, ACT[act] | flags, H, W, _lib.stream_ptr(xs[0].device))
    _lib.check(rc, 'ojf_segconv_forward_group')
    return list(outs)


def multi(calls):
    """``[conv(x, **kw) for conv, x, kw in calls]`` (kw: out, act, residual, mul) for up to 8 INDEPENDENT convolutions of
    any shapes as one launch where their kernel forms allow it (``ojf_segconv_forward_multi``).  Returns the outputs."""
    n = len(calls)
    assert 1 <= n <= 8
    c0 = calls[0][0]
    B = calls[0][1].shape[0]
    ptr = lambda vals: (ctypes.c_void_p * n)(*vals)
    ints = lambda vals: (ctypes.c_int * n)(*vals)
    handles, ins, in_s, outs, out_s, ress, res_s, muls, mul_s, acts, hs, ws, results = ([] for _ in range(13))
    for conv, x, kw in calls:
        assert isinstance(conv, SegConv) and x.shape[0] == B
        H, W = x.shape[2:]
        Ho, Wo = conv.out_size(H, W)
        out, flags = kw.get('out'), 0
        if out is None:
            out = nhwc((conv.c_out + 7) // 8 * 8, Ho, Wo, x.device, zero=False, batch=B)[:, :conv.c_out]
            flags = ZERO_PAD if conv.c_out % 8 else 0
        assert out.shape[1] == conv.c_out and tuple(out.shape[2:]) == (Ho, Wo) and out.shape[0] == B
        xp, xs = _rows(x)
        op, os_ = _rows(out)
        rp, rs = _rows(kw['residual']) if kw.get('residual') is not None else (None, 0)
        mp, ms = _rows(kw['mul']) if kw.get('mul') is not None else (None, 0)
        handles.append(conv._h.value); ins.append(xp); in_s.append(xs); outs.append(op); out_s.append(os_)
        ress.append(rp); res_s.append(rs); muls.append(mp); mul_s.append(ms); acts.append(ACT[kw.get('act')] | flags); hs.append(H); ws.append(W)
        results.append(out)
    rc = c0._lib.ojf_segconv_forward_multi(n, B, ptr(handles), ptr(ins), ints(in_s), ptr(outs), ints(out_s), ptr(ress), ints(res_s), ptr(muls), ints(mul_s),
                                           ints(acts), ints(hs), ints(ws), _lib.stream_ptr(calls[0][1].device))
    _lib.check(rc, 'ojf_segconv_forward_multi')
    return results


class SegDeconv:
    """``nn.ConvTranspose2d(c_in, c_out, 2*s, stride=s, padding=s//2)`` [+ eval ``bn``] on the SEGCONV kernel
    (``ojf_segdeconv_create``: 3x3 convolution to s*s phase copies + pixel-shuffle store).  Deterministic."""

    def __init__(self, deconv, bn=None):
        _lib.require_gpu()
        lib = _lib.load()
        assert isinstance(deconv, torch.nn.ConvTranspose2d) and deconv.groups == 1
        s = deconv.stride[0]
        assert deconv.stride == (s, s) and deconv.kernel_size == (2 * s, 2 * s) and deconv.padding == (s // 2, s // 2) \
            and deconv.output_padding == (0, 0) and deconv.dilation == (1, 1), 'kernel 2s, stride s, padding s/2 only'
        w = deconv.weight.detach().to('cpu', torch.float32).contiguous()  # [c_in, c_out, k, k]
        self.c_in, self.c_out, self.up = int(w.shape[0]), int(w.shape[1]), int(s)
        scale, bias = _folded(deconv.bias, bn, self.c_out)
        handle = ctypes.c_void_p()
        rc = lib.ojf_segdeconv_create(ctypes.byref(handle), w.data_ptr(), None if scale is None else scale.data_ptr(),
                                      bias.data_ptr(), self.c_in, self.c_out, self.up)
        _lib.check(rc, 'ojf_segdeconv_create')
        self._h, self._lib = handle, lib

    def __del__(self):
        if getattr(self, '_h', None):
            self._lib.ojf_segconv_destroy(self._h)
            self._h = None

    set_dropout = SegConv.set_dropout

    def __call__(self, x, out=None, act=None, zero_pad=True):
        """zero_pad=False: the pad channels of a fresh output (c_out not a multiple of 8) stay uninitialised - for outputs no
        convolution reads (the logits)."""
        H, W = x.shape[2:]
        B = x.shape[0]
        if out is None:
            out = nhwc((self.c_out + 7) // 8 * 8, H * self.up, W * self.up, x.device, zero=zero_pad and self.c_out % 8 != 0, batch=B)[:, :self.c_out]
        assert out.shape[1] == self.c_out and tuple(out.shape[2:]) == (H * self.up, W * self.up) and out.shape[0] == B
        xp, xs = _rows(x)
        op, os_ = _rows(out)
        rc = self._lib.ojf_segconv_forward_batch(self._h, B, xp, xs, op, os_, None, 0, None, 0, ACT[act], H, W, _lib.stream_ptr(x.device))
        _lib.check(rc, 'ojf_segconv_forward')
        return out


# ---- the operators around the convolutions (csrc/ojf_seg_ops.hip) -------------------------------------------------
def pack_input(src, divisor=1.0, out=None):
    """[1, 3, H, W] (contiguous NCHW) or a single [H, W] plane replicated three times (the depth modality,
    pipeline.py:50) divided by ``divisor`` -> the stem's NHWC rows: an [1, 8, H, W] channels_last tensor, channels 3..7 zero
    (``out``: one image of a batch tensor to write instead)."""
    _lib.require_gpu()
    lib = _lib.load()
    src = src.float()
    H, W = src.shape[-2:]
    planes = src.reshape(-1, H, W)
    assert planes.is_cuda and planes.is_contiguous() and planes.shape[0] in (1, 3)
    if out is None:
        out = nhwc(8, H, W, src.device, zero=False)
    op, os_ = _rows(out)
    rc = lib.ojf_seg_pack_input(planes.data_ptr(), H * W if planes.shape[0] == 3 else 0, float(divisor), H, W, op, os_,
                                _lib.stream_ptr(src.device))
    _lib.check(rc, 'ojf_seg_pack_input')
    return out


def maxpool(x):
    """nn.MaxPool2d(3, stride 2, padding 1) on an NHWC view."""
    lib = _lib.load()
    B, C, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = nhwc((C + 7) // 8 * 8, Ho, Wo, x.device, zero=C % 8 != 0, batch=B)[:, :C]
    xp, xs = _rows(x)
    op, os_ = _rows(out)
    _lib.check(lib.ojf_seg_maxpool_batch(B, xp, xs, C, H, W, op, os_, _lib.stream_ptr(x.device)), 'ojf_seg_maxpool')
    return out


def mean(x):
    """Global average over the pixels: NHWC view [1, C, H, W] -> [1, C, 1, 1] (rows padded to a multiple of 8 channels)."""
    lib = _lib.load()
    C, H, W = x.shape[1:]
    out = nhwc((C + 7) // 8 * 8, 1, 1, x.device, zero=C % 8 != 0)[:, :C]
    xp, xs = _rows(x)
    partial = torch.empty(32 * C, dtype=torch.float32, device=x.device)  # two fixed-order stages (include/ojf.h)
    _lib.check(lib.ojf_seg_mean(xp, xs, C, H * W, partial.data_ptr(), out.data_ptr(), _lib.stream_ptr(x.device)), 'ojf_seg_mean')
    return out


def broadcast(vec, out, mul=None):
    """out[:, c, y, x] = vec[0, c, 0, 0] (* mul[:, c, y, x]): bilinear upsampling of a 1x1 map / Decoder._skip's gate."""
    lib = _lib.load()
    C, H, W = out.shape[1:]
    assert vec.shape[1] == C and vec.shape[2] == vec.shape[3] == 1
    op, os_ = _rows(out)
    mp, ms = _rows(mul) if mul is not None else (None, 0)
    _lib.check(lib.ojf_seg_broadcast(vec.data_ptr(), mp, ms, C, H * W, op, os_, _lib.stream_ptr(out.device)), 'ojf_seg_broadcast')
    return out


class PoolFC:
    """Global average -> 1x1 ``conv`` (+ bias, no BatchNorm) on the 1x1 map -> ReLU -> broadcast to an output map (x gate): the
    squeeze chains of eASPP branch 5 (adapnet.py:204-210) and ``Decoder._skip`` (:292-296), two launches for up to 8
    members (``ojf_seg_pool_fc``; the 1x1 convolution runs in fp32)."""

    def __init__(self, conv):
        assert isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.groups == 1
        dev = conv.weight.device
        self.c_out, self.c_in = int(conv.weight.shape[0]), int(conv.weight.shape[1])
        self.w = conv.weight.detach().float().reshape(self.c_out, self.c_in).contiguous().clone()
        self.b = conv.bias.detach().float().contiguous().clone() if conv.bias is not None else None
        assert self.w.is_cuda, dev


def pool_fc(fcs, xs, outs, act='relu', muls=None):
    """``outs[i][:, c, y, x] = act(fc_i(mean(xs[i])))[c] (* muls[i][:, c, y, x])`` for n <= 8 members of one shape."""
    lib = _lib.load()
    B = xs[0].shape[0]
    if B > 1:  # every image of a batch has its own mean: (member, image) pairs, at most 8 per launch
        pairs = [(f, x[b:b + 1], o[b:b + 1], None if muls is None else m[b:b + 1])
                 for f, x, o, m in zip(fcs, xs, outs, muls if muls is not None else [None] * len(fcs)) for b in range(B)]
        for i in range(0, len(pairs), 8):
            part = pairs[i:i + 8]
            pool_fc([p[0] for p in part], [p[1] for p in part], [p[2] for p in part], act, None if muls is None else [p[3] for p in part])
        return outs
    n = len(fcs)
    f0 = fcs[0]
    C, H, W = xs[0].shape[1:]
    Co, Ho, Wo = outs[0].shape[1:]
    assert C == f0.c_in and Co == f0.c_out and all(f.c_in == C and f.c_out == Co for f in fcs)
    xr = [_rows(x) for x in xs]
    orr = [_rows(o) for o in outs]
    mr = [_rows(m) for m in muls] if muls is not None else None
    assert all(r[1] == xr[0][1] for r in xr) and all(r[1] == orr[0][1] for r in orr) and (mr is None or all(r[1] == mr[0][1] for r in mr))
    arr = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
    partial = torch.empty(n * 128 * C, dtype=torch.float32, device=xs[0].device)
    biases = arr([f.b.data_ptr() if f.b is not None else None for f in fcs]) if any(f.b is not None for f in fcs) else None
    rc = lib.ojf_seg_pool_fc(n, arr([r[0] for r in xr]), xr[0][1], C, H * W, arr([f.w.data_ptr() for f in fcs]), biases, Co, ACT[act],
                             arr([r[0] for r in mr]) if mr is not None else None, mr[0][1] if mr is not None else 0,
                             arr([r[0] for r in orr]), orr[0][1], Ho * Wo, partial.data_ptr(), _lib.stream_ptr(xs[0].device))
    _lib.check(rc, 'ojf_seg_pool_fc')
    return outs


def softmax_max(logits):
    """torch.softmax(logits, 1).max(1) of an NHWC view [1, C, H, W] -> (scores f32 [H*W], ids u8 [H*W])."""
    lib = _lib.load()
    B, C, H, W = logits.shape
    scores = torch.empty(B * H * W, dtype=torch.float32, device=logits.device)
    ids = torch.empty(B * H * W, dtype=torch.uint8, device=logits.device)
    lp, ls = _rows(logits)
    _lib.check(lib.ojf_seg_softmax_max(lp, ls, C, B * H * W, scores.data_ptr(), ids.data_ptr(), _lib.stream_ptr(logits.device)),
               'ojf_seg_softmax_max')
    return scores, ids
