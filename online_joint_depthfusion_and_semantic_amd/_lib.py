"""ctypes binding of libojf.so (C ABI in include/ojf.h).

There is NO fallback: if the shared library is missing, or a call is made without a GPU,
this module raises.  The product path never routes through the CPU oracle.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libojf.so')
if os.environ.get('OJF_LIB_PATH'):  # A/B runs of two builds on one box (tools/): never set in production
    LIB_PATH = os.environ['OJF_LIB_PATH']

MODE_FAST = 0
MODE_PARITY = 1
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_TANH = 0, 1, 2, 3
ARITH_F32, ARITH_F16X3 = 0, 1
ARITHMETIC = {'f32': ARITH_F32, 'f16x3': ARITH_F16X3}


class OjfError(RuntimeError):
    pass


class ConvLayer(ctypes.Structure):
    _fields_ = [('c_in', ctypes.c_int), ('c_out', ctypes.c_int), ('ksize', ctypes.c_int),
                ('dilation', ctypes.c_int), ('weight_host', ctypes.c_void_p),
                ('bias_host', ctypes.c_void_p)]


class TrainLayer(ctypes.Structure):
    """include/ojf.h ojf_train_layer: one conv (+ BatchNorm) unit of the whole-net training executor."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ('weight', 'bias', 'gamma', 'beta', 'running_mean', 'running_var', 'grad_weight',
                                                 'grad_bias', 'grad_gamma', 'grad_beta', 'drop_scale')]
                + [(n, ctypes.c_int) for n in ('out_channels', 'in_channels', 'ksize', 'dilation', 'bn_training', 'accumulate')]
                + [('momentum', ctypes.c_float), ('eps', ctypes.c_float)])


_c = ctypes
_vp, _i, _f, _d, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_double, _c.c_size_t

MAX_SCENES = 8  # include/ojf.h OJF_MAX_SCENES


class ExtractJob(ctypes.Structure):
    """include/ojf.h ojf_extract_job: one scene's frame of ojf_extract_many."""
    _fields_ = [('depth_dev', _vp), ('Kinv_host', _vp), ('E_host', _vp), ('origin_host', _vp), ('resolution', _d),
                ('tsdf_dev', _vp), ('weights_dev', _vp), ('net', _vp), ('out_values_dev', _vp), ('out_weights_dev', _vp),
                ('out_stride', _i), ('out_layout', _i)]


class IntegrateJob(ctypes.Structure):
    """include/ojf.h ojf_integrate_job: one scene's frame of ojf_integrate_many."""
    _fields_ = [('depth_dev', _vp), ('mask_dev', _vp), ('Kinv_host', _vp), ('E_host', _vp), ('origin_host', _vp), ('resolution', _d),
                ('est_dev', _vp), ('est_stride', _i), ('tsdf_dev', _vp), ('weights_dev', _vp), ('sem_ids_dev', _vp),
                ('sem_scores_dev', _vp), ('id_vol_dev', _vp), ('score_vol_dev', _vp), ('workspace_dev', _vp), ('workspace_bytes', _sz)]


# name -> (restype, argtypes); must list every symbol include/ojf.h declares
SIGNATURES = {
    'ojf_version': (_c.c_char_p, []),
    'ojf_last_error': (_c.c_char_p, []),
    'ojf_device_count': (_i, []),
    'ojf_extract': (_i, [_vp, _vp, _vp, _vp, _d, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _i,
                         _vp, _vp, _vp, _vp, _vp]),
    'ojf_extract_to_net': (_i, [_vp, _vp, _vp, _vp, _d, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    'ojf_extract_many': (_i, [_i, _c.POINTER(ExtractJob), _i, _i, _i, _i, _i, _i, _f, _vp]),
    'ojf_integrate_many': (_i, [_i, _c.POINTER(IntegrateJob), _i, _i, _f, _i, _i, _i, _i, _i, _vp]),
    'ojf_integrate_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    'ojf_integrate_workspace_init': (_i, [_vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ojf_integrate': (_i, [_vp, _vp, _vp, _vp, _d, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp,
                           _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    'ojf_integrate_masked': (_i, [_vp, _vp, _vp, _vp, _vp, _d, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    'ojf_integrate_entries': (_i, [_vp, _vp, _vp, _vp, _vp, _c.c_int64, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp, _vp]),
    'ojf_net_create': (_i, [_c.POINTER(_vp), _i, _i, _i, _i, _f, _c.POINTER(ConvLayer), _i, _i, _i]),
    'ojf_net_destroy': (None, [_vp]),
    'ojf_net_layer_count': (_i, [_i, _i, _i, _i]),
    'ojf_net_prepare_input': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp]),
    'ojf_net_forward': (_i, [_vp, _vp, _i, _vp]),
    'ojf_net_macs_per_pixel': (_c.c_int64, [_vp]),
    'ojf_net_launch_count': (_i, [_vp]),
    'ojf_net_side_streams': (_i, [_vp, _vp, _vp]),
    'ojf_net_profile': (_i, [_vp, _vp, _i, _vp, _c.c_char_p, _i, _c.POINTER(_f), _i]),
    'ojf_net_set_arithmetic': (_i, [_i]),
    'ojf_net_get_arithmetic': (_i, [_vp]),
    'ojf_net_check': (_i, [_vp]),
    'ojf_guard_status': (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i)]),
    'ojf_guard_poll': (_i, []),
    'ojf_streams_overlap': (_i, [_vp, _vp]),
    'ojf_conv2d': (_i, [_vp, _i, _i, _vp, _i, _i, _c.POINTER(ConvLayer), _i, _i, _i, _vp]),
    'ojf_volume_fill_f16': (_i, [_vp, _sz, _f, _vp]),
    'ojf_volume_fill_u8': (_i, [_vp, _sz, _c.c_uint8, _vp]),
    'ojf_volume_filter': (_i, [_vp, _vp, _sz, _f, _f, _vp]),
    'ojf_volume_median5_u8': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'ojf_volume_evaluate': (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    'ojf_volume_confusion': (_i, [_vp, _vp, _vp, _sz, _i, _vp, _vp, _vp]),
    'ojf_segconv_create': (_i, [_c.POINTER(_vp), _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    'ojf_segdeconv_create': (_i, [_c.POINTER(_vp), _vp, _vp, _vp, _i, _i, _i]),
    'ojf_segconv_destroy': (None, [_vp]),
    'ojf_segconv_forward': (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'ojf_segconv_forward_batch': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'ojf_segconv_forward_group': (_i, [_i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'ojf_segconv_forward_multi': (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ojf_segconv_forward_group_batch': (_i, [_i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'ojf_train_packed_floats': (_sz, [_i, _i, _i]),
    'ojf_train_pack': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'ojf_train_conv': (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'ojf_train_avgpool3': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'ojf_train_partial_doubles': (_sz, [_i]),
    'ojf_train_channel_sums': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    # y, y_g0, out, out_g0, c_phys, c, h, w, gamma, beta, drop, act, scale, has_bn, training, momentum, eps, running_mean,
    # running_var, partial, mean, invstd, stream
    'ojf_train_bn_act': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    # y, y_g0, dout, dout_g0, dy, dy_g0, c_phys, c, h, w, mean, invstd, gamma, beta, drop, act, scale, has_bn, training, partial,
    # dgamma, dbeta, dbias, accumulate, stream
    'ojf_train_bn_act_bwd': (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _i, _vp,
                                  _vp, _vp, _vp, _i, _vp]),
    'ojf_train_wgrad_partial_floats': (_sz, [_i, _i, _i, _i, _i]),
    'ojf_train_wgrad': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'ojf_trainer_create': (_i, [_c.POINTER(_vp), _i, _i, _i, _i, _f, _i, _i]),
    'ojf_trainer_destroy': (None, [_vp]),
    'ojf_trainer_set_arithmetic': (_i, [_vp, _i]),
    'ojf_trainer_set_backward_arithmetic': (_i, [_vp, _i]),
    'ojf_trainer_layer_count': (_i, [_vp]),
    'ojf_trainer_launch_count': (_i, [_vp]),
    'ojf_trainer_set_graph': (_i, [_vp, _i]),
    'ojf_trainer_graph_replays': (_i, [_vp]),
    'ojf_trainer_forward': (_i, [_vp, _c.POINTER(TrainLayer), _i, _c.c_ulonglong, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ojf_trainer_backward': (_i, [_vp, _c.POINTER(TrainLayer), _i, _vp, _vp]),
    'ojf_train_fuse_output': (_i, [_vp, _vp, _vp, _vp, _i, _i, _c.c_longlong, _f, _vp, _vp]),
    'ojf_train_fuse_output_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _c.c_longlong, _f, _vp, _vp]),
    'ojf_train_loss_partial_doubles': (_sz, [_c.c_longlong]),
    'ojf_train_fusion_loss': (_i, [_vp, _vp, _c.c_longlong, _i, _f, _f, _f, _vp, _vp, _vp]),
    'ojf_train_fusion_loss_bwd': (_i, [_vp, _vp, _c.c_longlong, _i, _f, _f, _vp, _vp, _vp]),
    'ojf_seg_pack_input': (_i, [_vp, _i, _f, _i, _i, _vp, _i, _vp]),
    'ojf_seg_maxpool': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    'ojf_seg_maxpool_batch': (_i, [_i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    'ojf_seg_mean': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    'ojf_seg_broadcast': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    'ojf_seg_softmax_max': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    'ojf_seg_pool_fc': (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp]),
    'ojf_segconv_set_dropout': (_i, [_vp, _vp, _c.c_uint, _i]),
    'ojf_points_within': (_i, [_vp, _sz, _vp, _vp, _vp, _d, _i, _i, _i, _d, _vp, _vp, _vp]),
    'ojf_mesh_workspace_bytes': (_sz, [_i, _i, _i]),
    'ojf_mesh_extract': (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _d, _vp, _sz, _vp, _vp, _vp, _c.c_uint32, _vp, _vp]),
}

_LIB = None


def load():
    """Load libojf.so and bind every declared symbol.  Raises OjfError if the library is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise OjfError('libojf.so not found at {}: build it with `python -c "import __graft_entry__ as g; '
                       'g.build()"` or `make -C {}/csrc` (there is no CPU fallback)'.format(LIB_PATH, _HERE))
    # torch FIRST: it ships its own libamdhip64, and a process that has already loaded the system one through libojf.so ends up with two HIP
    # runtimes - the second to initialise sees no device ("no ROCm-capable device is detected": __graft_entry__.build() followed by smoke() in
    # one interpreter, tools/bench_with_lib.py).  With torch's copy loaded, libojf.so's dependency resolves to it by soname.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise OjfError('{} failed (rc={}): {}'.format(what, rc, load().ojf_last_error().decode()))


def require_gpu():
    lib = load()
    n = lib.ojf_device_count()
    if n <= 0:
        raise OjfError('no MI355X/HIP device visible: {} (the HIP path has no CPU fallback)'.format(
            lib.ojf_last_error().decode()))
    return n


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.flags.c_contiguous
        return t.ctypes.data
    assert t.is_contiguous(), 'tensor handed to libojf must be contiguous'
    return t.data_ptr()


def stream_ptr(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream


class TimingEvent:
    """HIP event for stage timing, created with ``hipEventDisableSystemFence``: a default event (torch.cuda.Event)
    makes the stream write back and invalidate caches at system scope when it is recorded - ~5 us of idle queue per
    record, four records per frame - which distorts what it measures.  Elapsed times only; not for host synchronisation
    of memory."""
    _hip = None
    FLAGS = 0x20000000  # hipEventDisableSystemFence (timing stays enabled)

    @classmethod
    def _rt(cls):
        if cls._hip is None:
            hip = _c.CDLL('libamdhip64.so')
            hip.hipEventCreateWithFlags.argtypes = [_c.POINTER(_vp), _c.c_uint]
            hip.hipEventRecord.argtypes = [_vp, _vp]
            hip.hipEventElapsedTime.argtypes = [_c.POINTER(_f), _vp, _vp]
            hip.hipEventDestroy.argtypes = [_vp]
            hip.hipEventSynchronize.argtypes = [_vp]
            cls._hip = hip
        return cls._hip

    def __init__(self):
        self._e = _vp()
        if self._rt().hipEventCreateWithFlags(_c.byref(self._e), self.FLAGS) != 0:
            raise OjfError('hipEventCreateWithFlags failed')

    def record(self, stream):
        if self._rt().hipEventRecord(self._e, _vp(stream)) != 0:
            raise OjfError('hipEventRecord failed')

    def elapsed_time(self, later):
        self._rt().hipEventSynchronize(later._e)
        ms = _f()
        if self._rt().hipEventElapsedTime(_c.byref(ms), self._e, later._e) != 0:
            raise OjfError('hipEventElapsedTime failed')
        return ms.value

    def __del__(self):
        if getattr(self, '_e', None) and self._hip is not None:
            self._hip.hipEventDestroy(self._e)
            self._e = None
