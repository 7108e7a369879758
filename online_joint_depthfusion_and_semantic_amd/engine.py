"""HIP executor of the fusion net (ojf_net_* in include/ojf.h) fed from a torch parameter container.

``FusionNetEngine`` folds the BatchNorms of a ``FusionNet_v2``/``FusionNet_v3`` module (running
statistics, eval semantics) and hands the folded layers to libojf.  Per frame, ``prepare_input``
packs the extractor's row-major outputs + depth (+ semantic frame) into the library's private
activation layout (one small kernel instead of the reference's view/permute/contiguous chain,
modules/pipeline.py:74-102) and ``forward`` runs the convolution chain.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .model import FusionNet_v2, FusionNet_v3, fold_layers


def _layer_array(layers):
    arr = (_lib.ConvLayer * len(layers))()
    keep = []
    for i, (w, b, k, d) in enumerate(layers):
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        keep += [w, b]
        arr[i].c_in, arr[i].c_out, arr[i].ksize, arr[i].dilation = w.shape[1], w.shape[0], k, d
        arr[i].weight_host = w.ctypes.data
        arr[i].bias_host = b.ctypes.data
    return arr, keep


class FusionNetEngine:
    def __init__(self, net, h, w, device, arithmetic='f16x3'):
        """arithmetic: 'f16x3' (split-fp16 MFMA, default) | 'f32' (fp32-input MFMA); include/ojf.h OJF_ARITH_*."""
        _lib.require_gpu()
        self.lib = _lib.load()
        if arithmetic not in _lib.ARITHMETIC:
            raise ValueError('arithmetic must be one of {}'.format(sorted(_lib.ARITHMETIC)))
        self.arithmetic = arithmetic
        if isinstance(net, FusionNet_v3):
            version = 3
        elif isinstance(net, FusionNet_v2):
            version = 2
        else:
            raise TypeError('FusionNetEngine needs a FusionNet_v2 or FusionNet_v3 module')
        self.device = torch.device(device)
        self.h, self.w = h, w
        self.n_points = net.n_points
        self.use_semantics = bool(net.config.use_semantics)
        # nets without a semantic channel have one head: ops.extract_to_net can fill their input planes directly
        self.fused_input = not self.use_semantics
        layers = fold_layers(net)
        arr, keep = _layer_array(layers)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ojf_net_set_arithmetic(_lib.ARITHMETIC[arithmetic]), 'ojf_net_set_arithmetic')
            rc = self.lib.ojf_net_create(ctypes.byref(handle), version, net.n_points, net.gf,
                                         int(self.use_semantics), float(net.scale), arr, len(layers), h, w)
        _lib.check(rc, 'ojf_net_create')
        self.handle = handle
        self.macs_per_pixel = int(self.lib.ojf_net_macs_per_pixel(self.handle))

    def prepare_input(self, values, weights, depth, sem_ids=None, n_classes=0, planes=False):
        """values / weights: cuda f32 from the extractor, rows [h*w, stride] or (planes=True) sample planes
        [n_points, h*w]; depth: cuda f32 [h,w]."""
        assert values.is_cuda and values.dtype == torch.float32 and values.is_contiguous()
        assert weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous()
        assert values.shape == weights.shape and values.shape[1 if planes else 0] == self.h * self.w
        assert depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous()
        if sem_ids is not None:
            assert sem_ids.is_cuda and sem_ids.dtype == torch.uint8 and sem_ids.is_contiguous()
        rc = self.lib.ojf_net_prepare_input(self.handle, _lib.ptr(values), _lib.ptr(weights), values.shape[-1],
                                            1 if planes else 0, _lib.ptr(depth), _lib.ptr(sem_ids), int(n_classes),
                                            _lib.stream_ptr(self.device))
        _lib.check(rc, 'ojf_net_prepare_input')

    def forward(self, est):
        """est: cuda f32 [h*w, stride>=n_points] receiving output_scale * tanh(.)"""
        assert est.is_cuda and est.dtype == torch.float32 and est.is_contiguous()
        rc = self.lib.ojf_net_forward(self.handle, _lib.ptr(est), est.shape[-1], _lib.stream_ptr(self.device))
        _lib.check(rc, 'ojf_net_forward')
        return est

    @property
    def launches(self):
        """Kernel launches of the most recent forward (counted by the library at its launch sites)."""
        return int(self.lib.ojf_net_launch_count(self.handle))

    def side_streams(self):
        """Raw handles of the streams the net launches on beside the current one (ojf_net_side_streams), paired as the first
        forward pass would pair them."""
        out = (ctypes.c_void_p * 3)()
        n = self.lib.ojf_net_side_streams(self.handle, _lib.stream_ptr(self.device), out)
        if n < 0:
            _lib.check(n, 'ojf_net_side_streams')
        return [int(p) for p in out if p]

    def profile(self, est):
        """One profiled forward: [(kernel name, microseconds)] in launch order (ojf_net_profile)."""
        names = ctypes.create_string_buffer(8192)
        us = (ctypes.c_float * 256)()
        n = self.lib.ojf_net_profile(self.handle, _lib.ptr(est), est.shape[-1], _lib.stream_ptr(self.device), names, 8192, us, 256)
        if n < 0:
            _lib.check(n, 'ojf_net_profile')
        return list(zip(names.value.decode().split('\n')[:n], [float(us[i]) for i in range(n)]))

    def check(self):
        """Synchronise the stream and raise OjfError if the split-fp16 range guard fired (include/ojf.h)."""
        _lib.check(self.lib.ojf_net_check(_lib.stream_ptr(self.device)), 'ojf_net_check')

    def close(self):
        if getattr(self, 'handle', None):
            torch.cuda.synchronize(self.device)
            self.lib.ojf_net_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def conv2d_rows(x_rows, in_off, c_in, weight, bias, out_rows, out_off, h, w, dilation=1, act=_lib.ACT_NONE):
    """Stand-alone HIP convolution on NHWC rows (ojf_conv2d); used by the layer-level parity tests."""
    _lib.require_gpu()
    lib = _lib.load()
    wgt = np.ascontiguousarray(weight, dtype=np.float32)
    b = np.ascontiguousarray(bias, dtype=np.float32)
    layer = _lib.ConvLayer(c_in, wgt.shape[0], wgt.shape[2], dilation, wgt.ctypes.data, b.ctypes.data)
    rc = lib.ojf_conv2d(_lib.ptr(x_rows), x_rows.shape[-1], in_off, _lib.ptr(out_rows), out_rows.shape[-1], out_off,
                        ctypes.byref(layer), act, h, w, _lib.stream_ptr(x_rows.device))
    _lib.check(rc, 'ojf_conv2d')
