"""HIP training path of the fusion net: ``Pipeline.fuse_training``'s network forward / backward
(modules/pipeline.py:322, driven by train_fusion.py:166-189) on libojf kernels instead of torch autograd + MIOpen.

The unit of the reference's Sequentials (modules/model.py:4-52,115-141)

    Conv2d (1x1 / dilated 3x3) -> BatchNorm2d -> ReLU | LeakyReLU | Tanh -> Dropout2d

is ONE autograd node (``LayerUnit``): forward = device-side weight packing, fp32-MFMA convolution, batch statistics
(+ running-stat update), fused normalise / activation / channel dropout; backward = fused activation / BatchNorm
backward with its two reductions, the weight-gradient kernel, and backward-data as a convolution with the transposed,
tap-flipped weights (include/ojf.h ``ojf_train_*``).  Activations are "C4 planes" tensors ``[C/4, H, W, 4]`` (19
channels in a 20-wide slot, 114 in 116; padding channels are exactly zero), so a concatenation is a ``torch.cat`` along
dim 0.  Around the units: concatenations and the loss are torch on those tensors (autograd adds up the fan-out
gradients); the 3x3 average pools, the global-average branch's pooling and the backward of its broadcast are libojf
launches (``AvgPool3``, ``BroadcastPlanes``).  Parameter gradients do not travel through autograd's AccumulateGrad: the
kernels write / add them straight into ``p.grad`` (``_grad_target``).  Parameters, BatchNorm buffers and therefore
``state_dict`` / optimizer / checkpoint code are the module's own (``model.FusionNet_v3`` / ``_v2``): ``HipTrainNet``
only walks them.

There is no fallback inside this module: it needs libojf and a GPU.  ``Pipeline.fuse_training`` selects it with
``FUSION_MODEL.train_engine: hip`` (default) | ``torch``.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib

_ACT = {None: _lib.ACT_NONE, 'relu': _lib.ACT_RELU, 'leaky': _lib.ACT_LEAKY, 'tanh': _lib.ACT_TANH}


def _p(t):
    return None if t is None else t.data_ptr()


def _packed(lib, cache, w, bias, group, slot, c_in_phys, c_out_phys, transposed, st):
    """Fragment-layout copy of a weight tensor (ojf_train_pack).  With a cache (HipTrainNet) the packing launch is
    skipped while the parameter's version counter stands still - weights change once per accumulation window
    (train_fusion.py:186), not once per frame.  Keyed by storage address, layout and version."""
    OC, IC, k, _ = w.shape
    key = ver = None
    if cache is not None:
        key = (w.data_ptr(), _p(bias), group, slot, c_in_phys, c_out_phys, transposed)
        ver = (w._version, bias._version if bias is not None else -1)
        hit = cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2]
    rows, kch = (c_in_phys, c_out_phys) if transposed else (c_out_phys, c_in_phys)
    packed = torch.empty(lib.ojf_train_packed_floats(rows, kch, k), dtype=torch.float32, device=w.device)
    bias_packed = None
    if not transposed:
        n_ot = (((c_out_phys + 15) // 16) + 1) // 2 * 2
        bias_packed = torch.empty(n_ot * 16, dtype=torch.float32, device=w.device)
    _lib.check(lib.ojf_train_pack(w.data_ptr(), None if transposed else _p(bias), OC, IC, k, group, slot, c_in_phys, c_out_phys,
                                  int(transposed), packed.data_ptr(), _p(bias_packed), st), 'ojf_train_pack')
    if cache is not None:
        cache[key] = (ver, packed, bias_packed)
    return packed, bias_packed


def _grad_target(p, needed, capturing):
    """Where a parameter gradient of this backward pass goes: (tensor the kernel writes, accumulate flag, value returned
    to autograd).  With ``HipTrainNet(inplace_grads=True)`` (what Pipeline.fuse_training asks for) the kernels write /
    add a leaf parameter's gradient straight into ``p.grad`` and autograd gets None - the engine's AccumulateGrad would
    otherwise launch one add per parameter and frame (224 launches, 0.9 ms per 320x240 frame with gradients accumulated
    over 8 frames, train_fusion.py:174-189).  The sums are formed in the same order (grad = (g1 + g2) + ...).  Without
    the flag (``capturing`` then carries "take the ordinary route"), for non-leaf tensors, exotic ``.grad`` layouts and
    under graph capture the gradient is returned to autograd like from any other Function, so
    ``torch.autograd.grad(loss, net.parameters())`` works and nothing is mutated behind autograd's back."""
    if p is None or not needed:
        return None, 0, None
    if p.is_leaf and p.requires_grad and not capturing:
        if p.grad is None:
            p.grad = torch.empty_like(p, memory_format=torch.contiguous_format)
            return p.grad, 0, None
        if p.grad.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.device == p.device and p.grad.shape == p.shape:
            return p.grad, 1, None
    g = torch.empty_like(p, memory_format=torch.contiguous_format)
    return g, 0, g


class LayerUnit(torch.autograd.Function):
    """conv (+ bias) -> [BatchNorm2d] -> activation -> [Dropout2d scale per channel] on C4 planes."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, meta):
        lib = _lib.load()
        dev = x.device
        st = _lib.stream_ptr(dev)
        x = x.contiguous()
        c4_in, H, W, _ = x.shape
        OC, IC, k, _ = weight.shape
        c_in_phys, c_out_phys = 4 * c4_in, (OC + 3) // 4 * 4
        group, slot, dil = meta['group'], meta['slot'], meta['dil']
        if lib.ojf_train_packed_floats(c_out_phys, c_in_phys, k) == 0:
            raise _lib.OjfError('LayerUnit: unsupported layer shape %r on %d input planes' % (tuple(weight.shape), c4_in))
        w = weight.detach().contiguous()
        packed, bias_packed = _packed(lib, meta.get('cache'), w, bias, group, slot, c_in_phys, c_out_phys, False, st)
        y = torch.empty((c_out_phys // 4, H, W, 4), dtype=torch.float32, device=dev)
        _lib.check(lib.ojf_train_conv(x.data_ptr(), 0, c_in_phys, y.data_ptr(), 0, c_out_phys, packed.data_ptr(),
                                      bias_packed.data_ptr(), k, dil, H, W, st), 'ojf_train_conv')
        bn = meta['bn']
        mean = invstd = partial = None
        training = bool(meta['training'])
        momentum, eps = 0.0, 0.0
        if bn is not None:
            mean = torch.empty(c_out_phys, dtype=torch.float32, device=dev)
            invstd = torch.empty(c_out_phys, dtype=torch.float32, device=dev)
            partial = torch.empty(lib.ojf_train_partial_doubles(c_out_phys), dtype=torch.float64, device=dev)
            if bn.momentum is None:  # cumulative average: the factor 1 / num_batches_tracked lives on the device
                raise _lib.OjfError('HipTrainNet: BatchNorm2d(momentum=None) is not supported (use train_engine: torch)')
            momentum = float(bn.momentum)
            eps = float(bn.eps)
            if training:
                counters = meta.get('counters')
                if counters is not None:
                    counters.append(bn.num_batches_tracked)  # incremented together after the forward pass
                else:
                    bn.num_batches_tracked += 1
        drop = meta['drop']
        out = torch.empty_like(y)
        # statistics (batch or running), running-stat update, normalisation, activation and dropout scale: two launches
        _lib.check(lib.ojf_train_bn_act(y.data_ptr(), 0, out.data_ptr(), 0, c_out_phys, OC, H, W, _p(gamma), _p(beta), _p(drop),
                                        _ACT[meta['act']], float(meta['scale']), int(bn is not None), int(training), momentum, eps,
                                        _p(bn.running_mean if bn is not None else None), _p(bn.running_var if bn is not None else None),
                                        _p(partial), _p(mean), _p(invstd), st), 'ojf_train_bn_act')
        ctx.save_for_backward(x, y, w, gamma, beta, mean, invstd, drop)
        ctx.meta = dict(meta, training=training, has_bias=bias is not None)
        ctx.params = (weight, bias, gamma, beta)  # the tensors autograd would accumulate into (see _grad_target)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, y, w, gamma, beta, mean, invstd, drop = ctx.saved_tensors
        meta = ctx.meta
        dev = x.device
        st = _lib.stream_ptr(dev)
        dout = dout.contiguous()
        c4_in, H, W, _ = x.shape
        OC, IC, k, _ = w.shape
        c_in_phys, c_out_phys = 4 * c4_in, (OC + 3) // 4 * 4
        group, slot, dil = meta['group'], meta['slot'], meta['dil']
        has_bn = meta['bn'] is not None
        dy = torch.empty_like(y)
        partial = torch.empty(lib.ojf_train_partial_doubles(c_out_phys), dtype=torch.float64, device=dev)
        capturing = torch.cuda.is_current_stream_capturing() or not meta.get('inplace_grads', False)
        p_w, p_b, p_g, p_be = ctx.params
        gw, aw, rw = _grad_target(p_w, ctx.needs_input_grad[1], capturing)
        gb, ab, rb = _grad_target(p_b, p_b is not None and ctx.needs_input_grad[2], capturing)
        gg, ag, rg = _grad_target(p_g, p_g is not None and ctx.needs_input_grad[3], capturing)
        gbe, abe, rbe = _grad_target(p_be, p_be is not None and ctx.needs_input_grad[4], capturing)
        # gamma, beta and the bias share one launch and one accumulate flag: in a mixed state (one gradient fresh, another
        # accumulating) the accumulating ones go through autograd instead
        small = [[gb, ab, rb, p_b], [gg, ag, rg, p_g], [gbe, abe, rbe, p_be]]
        live = [t for t in small if t[0] is not None]
        acc_small = int(bool(live) and all(t[1] for t in live))
        if not acc_small:
            for t in live:
                if t[1]:
                    t[0] = t[2] = torch.empty_like(t[3], memory_format=torch.contiguous_format)
                    t[1] = 0
        (gb, ab, rb, _), (gg, ag, rg, _), (gbe, abe, rbe, _) = small
        _lib.check(lib.ojf_train_bn_act_bwd(y.data_ptr(), 0, dout.data_ptr(), 0, dy.data_ptr(), 0, c_out_phys, OC, H, W, _p(mean), _p(invstd),
                                            _p(gamma), _p(beta), _p(drop), _ACT[meta['act']], float(meta['scale']), int(has_bn),
                                            int(meta['training']), partial.data_ptr(), _p(gg), _p(gbe), _p(gb), acc_small, st),
                   'ojf_train_bn_act_bwd')
        if gw is not None:
            wpart = torch.empty(lib.ojf_train_wgrad_partial_floats(c_out_phys, c_in_phys, k, H, W), dtype=torch.float32, device=dev)
            _lib.check(lib.ojf_train_wgrad(x.data_ptr(), 0, c_in_phys, dy.data_ptr(), 0, c_out_phys, OC, IC, k, dil, group, slot, H, W,
                                           wpart.data_ptr(), gw.data_ptr(), int(aw), st), 'ojf_train_wgrad')
        dx = None
        if ctx.needs_input_grad[0]:
            # backward-data = convolution of dy with the transposed, tap-flipped weights
            packed, _ = _packed(lib, meta.get('cache'), w, None, group, slot, c_in_phys, c_out_phys, True, st)
            dx = torch.empty_like(x)
            _lib.check(lib.ojf_train_conv(dy.data_ptr(), 0, c_out_phys, dx.data_ptr(), 0, c_in_phys, packed.data_ptr(), None, k, dil, H, W, st),
                       'ojf_train_conv (backward-data)')
        return dx, rw, rb, rg, rbe, None


class BroadcastPlanes(torch.autograd.Function):
    """A per-channel vector [4 * o4] as constant C4 planes [o4, H, W, 4] (the up-sampled 1x1 map of the global-average
    branch).  Forward is a view; backward = per-channel sums of the gradient planes by a libojf launch (fp64 slab sums
    in fixed order) - torch's reduction of an expanded view over (H, W) with the 4-wide inner stride took 200 us."""

    @staticmethod
    def forward(ctx, vec, H, W):
        o4 = vec.shape[0] // 4
        return vec.view(o4, 1, 1, 4).expand(o4, H, W, 4)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        dout = dout.contiguous()
        o4, H, W, _ = dout.shape
        part = torch.empty(lib.ojf_train_partial_doubles(4 * o4), dtype=torch.float64, device=dout.device)
        _lib.check(lib.ojf_train_channel_sums(dout.data_ptr(), 0, 4 * o4, H, W, part.data_ptr(), _lib.stream_ptr(dout.device)),
                   'ojf_train_channel_sums')
        return part.view(-1, o4, 8)[:, :, :4].sum(0).reshape(-1).float(), None, None


class AvgPool3(torch.autograd.Function):
    """nn.AvgPool2d(3, stride 1, padding 1) (count_include_pad) on C4 planes; the operator is symmetric, so backward is
    the same launch on the gradient.  (torch's own pooling on the permuted planes view returned wrong input gradients
    on this ROCm build.)"""

    @staticmethod
    def _run(x):
        lib = _lib.load()
        x = x.contiguous()
        out = torch.empty_like(x)
        c4, H, W, _ = x.shape
        _lib.check(lib.ojf_train_avgpool3(x.data_ptr(), out.data_ptr(), 4 * c4, H, W, _lib.stream_ptr(x.device)), 'ojf_train_avgpool3')
        return out

    @staticmethod
    def forward(ctx, x):
        return AvgPool3._run(x)

    @staticmethod
    def backward(ctx, dout):
        return AvgPool3._run(dout)


def to_c4(t):
    """[1, C, H, W] -> C4 planes [ceil(C/4), H, W, 4] (padding channels zero)."""
    _, C, H, W = t.shape
    cp = (C + 3) // 4 * 4
    if cp != C:
        t = torch.cat([t, t.new_zeros(1, cp - C, H, W)], dim=1)
    return t.view(cp // 4, 4, H, W).permute(0, 2, 3, 1).contiguous()


def from_c4(t, C):
    """C4 planes -> [1, C, H, W]."""
    c4, H, W, _ = t.shape
    return t.permute(0, 3, 1, 2).reshape(1, 4 * c4, H, W)[:, :C]


class HipTrainNet:
    """Runs ``net`` (model.FusionNet_v3 / FusionNet_v2) through ``LayerUnit`` nodes.  ``net.training`` selects batch
    statistics + dropout (train) or running statistics, no dropout (eval), exactly like the module's own forward."""

    def __init__(self, net, graph=False, inplace_grads=False, executor=True, arithmetic='f16x3', backward_arithmetic=None, replay=None, overlap=False, overlap_thread=False):
        _lib.require_gpu()
        self.net = net
        self.inplace_grads = bool(inplace_grads)  # see _grad_target
        # executor: the whole forward / backward pass as two libojf calls (ojf_trainer_*, csrc/ojf_train_net.h: C++ layer
        # walk, grouped VortexPooling branches) behind ONE autograd node; False = one autograd node per layer unit (the
        # round-2 path, kept for A/B runs and for graph=True)
        self.executor = bool(executor)
        # Convolutions of the executor.  ``arithmetic``: 'f16x3' (split-fp16, the inference arithmetic) | 'f32' for the
        # forward pass.  ``backward_arithmetic`` (FUSION_MODEL.train_arithmetic_bwd; default = ``arithmetic``): under a
        # split-fp16 forward, backward-data and the weight gradients run in split-fp16 as well - every dy tensor is stored
        # under a power of two derived in the SAME pass from a guaranteed bound on its magnitude (csrc/ojf_net_train.h
        # BnActArgs::bnd), so no gradient can leave the fp16 range whatever the loss scale; 'f32' keeps the backward
        # convolutions on the fp32-input MFMA path.
        self.arithmetic = arithmetic
        self.backward_arithmetic = backward_arithmetic or arithmetic
        self._weights_sig = None
        self._weights_epoch = 0
        self._trainers = {}
        self._table = None
        self._gen = 0
        self._cache = {}      # packed weights by (address, layout) -> (version, tensors)
        self._counters = []   # num_batches_tracked of the BatchNorms that saw batch statistics in this forward
        self._rand = None     # one uniform draw per forward for all Dropout2d masks
        self._rand_at = 0
        self._masks = {}
        self.graph = bool(graph)  # capture forward / backward into device graphs (falls back to eager launches if a capture fails)
        # executor passes replayed as device graphs inside the library (ojf_trainer_set_graph): None = the library's default
        # (off; OJF_TRAIN_GRAPH=1 in the environment), True / False = set per trainer
        self.replay = replay
        self._graphs = {}
        # overlap (FUSION_MODEL.train_overlap; needs executor + inplace_grads): the backward pass of frame k runs on a stream of its
        # own (``gradient_stream``) beside the forward stage of frame k + 1 on the caller's stream - two executors per frame shape
        # take turns, so one frame's activations are read while the next frame's are written.  The caller's stream then does NOT
        # see the gradients: whatever reads or writes them (clipping, all-reduce, optimizer step, zero_grad) belongs inside
        # ``with net.gradients():``; a forward pass waits for that stream by itself when a weight changed, ``join_gradients()``
        # makes the caller's stream wait (checkpoints, validation, host reads of a gradient).
        self.overlap = bool(overlap) and self.executor and self.inplace_grads
        # overlap_thread (FUSION_MODEL.train_overlap_thread, default OFF): the ~170 launches of a steady-state backward pass - and what
        # ``gradient_work`` is handed - are ENQUEUED by a host thread of this object's own (ctypes releases the GIL inside the library
        # call), so the caller's thread goes on to the next frame's forward stage instead of spending that launch loop.  Bit for bit the
        # same results, and measured to change nothing (228.8 against 227.6 frames/s): the gradient thread does enqueue a backward pass in
        # 0.8 ms beside the caller, but the caller then waits that much longer for the next frame's valid-ray count - with the device
        # chains overlapped the step is bound by the kernels' summed time, not by the host.
        self.overlap_thread = self.overlap and bool(overlap_thread)
        self._pool = None
        self._jobs = []
        self._passes = 0
        self._grad_stream = None
        self._grad_tail = None
        self._epoch_joined = None
        self._unjoined_backward = False  # a backward pass went to the gradient stream and nothing has ordered the caller behind it yet

    # ---- one Sequential of conv/BN/act/dropout slots -> units ------------------------------------------------------
    def _unit(self, x, conv, bn, act, dropout, group, slot, scale=1.0):
        # batch vs running statistics and dropout follow the INDIVIDUAL modules' flags like nn.Sequential's forward does
        # (a BatchNorm frozen with bn.eval() inside a net in train() mode keeps its running statistics)
        training = bn.training if bn is not None else False
        drop = None
        if dropout is not None and dropout.training and dropout.p > 0:  # Dropout2d: whole channels, survivors scaled by 1 / keep
            n = conv.out_channels
            keep = 1.0 - dropout.p
            if keep not in self._masks:  # one compare / scale launch per distinct p and forward pass, sliced per layer
                self._masks[keep] = (self._rand < keep).float() / keep
            drop = self._masks[keep][self._rand_at:self._rand_at + n]
            self._rand_at += n
        meta = dict(group=group, slot=slot, dil=int(conv.dilation[0]), act=act, scale=scale, bn=bn, drop=drop, training=training,
                    inplace_grads=self.inplace_grads,
                    cache=None if torch.cuda.is_current_stream_capturing() else self._cache, counters=self._counters)
        return LayerUnit.apply(x, conv.weight, conv.bias, bn.weight if bn is not None else None, bn.bias if bn is not None else None, meta)

    def _sequential(self, x, seq, group, slot, scale=1.0):
        mods = list(seq)
        i = 0
        first = True
        while i < len(mods):
            conv = mods[i]
            assert isinstance(conv, nn.Conv2d), 'HipTrainNet: unexpected module order in a Sequential'
            i += 1
            bn = act = drop = None
            if i < len(mods) and isinstance(mods[i], nn.BatchNorm2d):
                bn = mods[i]; i += 1
            if i < len(mods) and isinstance(mods[i], (nn.ReLU, nn.LeakyReLU, nn.Tanh)):
                act = {nn.ReLU: 'relu', nn.LeakyReLU: 'leaky', nn.Tanh: 'tanh'}[type(mods[i])]; i += 1
            if i < len(mods) and isinstance(mods[i], nn.Dropout2d):
                drop = mods[i]; i += 1
            last = i >= len(mods)
            g, s = (group, slot) if first else (conv.in_channels, (conv.in_channels + 3) // 4 * 4)
            x = self._unit(x, conv, bn, act, drop, g, s, scale if last else 1.0)
            first = False
        return x

    def _dense(self, x, blocks, c):
        slot = (c + 3) // 4 * 4
        for blk in blocks:
            x = torch.cat([x, self._sequential(x, blk.block, c, slot)], dim=0)
        return x

    def _vortex(self, x, v, group, slot):
        """modules/model.py:143-161.  x: C4 planes of in_chs logical channels (`group`-wide tensors in `slot`-wide slots)."""
        c4, H, W, _ = x.shape
        dev = x.device
        # global-average branch: 1x1 map -> conv -> (bilinear up-sampling = broadcast) -> BatchNorm2d.  The BatchNorm
        # sees a constant map: statistics, running-stat update and gradients are those of any constant map, so a
        # 2-pixel stand-in gives them (batch variance 0 either way) without materialising the full-size tensor.
        idx = self._logical_index(group, slot, v.gave_pool[1].in_channels, dev)
        gc = v.gave_pool[1]
        bn = v.gave_pool[3]
        if bn.training:
            # batch statistics of a constant map: mean = the value, variance = 0 -> the output is beta, the gradients
            # towards g, gamma and x vanish; running_mean moves towards g, running_var towards 0 (unbiased estimate
            # of 0).  Nothing of this branch is differentiated, so the pooled input comes from a libojf launch (fp64
            # slab sums in fixed order) outside autograd instead of torch's strided mean and its broadcast backward.
            if bn.momentum is None:
                raise _lib.OjfError('HipTrainNet: BatchNorm2d(momentum=None) is not supported (use train_engine: torch)')
            m = bn.momentum
            with torch.no_grad():
                lib = _lib.load()
                xc = x.contiguous()
                part = torch.empty(lib.ojf_train_partial_doubles(4 * c4), dtype=torch.float64, device=dev)
                _lib.check(lib.ojf_train_channel_sums(xc.data_ptr(), 0, 4 * c4, H, W, part.data_ptr(), _lib.stream_ptr(dev)),
                           'ojf_train_channel_sums')
                pooled = (part.view(-1, c4, 8)[:, :, :4].sum(0).reshape(-1)[idx] / float(H * W)).float()
                g = torch.addmv(gc.bias, gc.weight.view(gc.out_channels, -1), pooled)  # the 1x1 conv on the 1x1 map
                bn.running_mean.mul_(1 - m).add_(g, alpha=m)
                bn.running_var.mul_(1 - m)
                zero = torch.zeros_like(g)
            # (gamma and the 1x1 conv's parameters still receive their - exactly zero - gradients, like under autograd)
            g = bn.bias + zero * bn.weight + (gc.weight.sum() + gc.bias.sum()) * 0.0
            self._counters.append(bn.num_batches_tracked)
        else:
            pooled = x.mean(dim=(1, 2)).reshape(-1)[idx]                       # [in_chs]
            g = torch.addmv(gc.bias, gc.weight.view(gc.out_channels, -1), pooled)
            g = (g - bn.running_mean) * torch.rsqrt(bn.running_var + bn.eps) * bn.weight + bn.bias
        out_c = g.shape[0]
        o4 = (out_c + 3) // 4
        gp = BroadcastPlanes.apply(torch.cat([g, g.new_zeros(4 * o4 - out_c)]), H, W)
        outs = [gp, self._sequential(x, v.branches[0], group, slot)]
        xp = x
        for i in (1, 2, 3):  # nn.AvgPool2d(3, 1, 1), count_include_pad
            xp = AvgPool3.apply(xp)
            outs.append(self._sequential(xp, v.branches[i], group, slot))
        cat = torch.cat(outs, dim=0)
        return self._sequential(cat, v.final, out_c, 4 * o4)

    def _logical_index(self, group, slot, n_logical, dev):
        key = (group, slot, n_logical, str(dev))
        cache = self.__dict__.setdefault('_idx', {})
        if key not in cache:
            j = torch.arange(n_logical, device=dev)
            cache[key] = (j // group) * slot + j % group
        return cache[key]

    def forward(self, x):
        """Same contract as the module's forward: dict of NCHW tensors -> [1, n_points, H, W] (times output_scale).
        With ``graph=True`` the ~1300 launches of a forward + backward pass are captured once per frame shape and mode
        into two device graphs and replayed; parameters, BatchNorm buffers and dropout randomness are read / written in
        place by the replays, so optimizer steps and checkpoints see the same tensors as ever.  Off by default: on ROCm
        7.2 the replay of that many kernel nodes is SLOWER than launching them (25.0 vs 17.8 ms per 320x240 frame)."""
        if self.graph and torch.is_grad_enabled():
            out = self._graphed(x)
            if out is not None:
                return out
        if self.executor and not self.graph:
            return self._forward_executor(x)
        return self._forward_impl(x)

    # ---- whole-net executor ---------------------------------------------------------------------------------------------
    def _layer_modules(self):
        """(conv, bn | None, dropout | None) per layer in ojf_net_create's order (model.fold_layers)."""
        from .model import FusionNet_v3

        def seq(mods):
            mods = list(mods)
            out = []
            for i, m in enumerate(mods):
                if isinstance(m, nn.Conv2d):
                    bn = drop = None
                    for nxt in mods[i + 1:]:
                        if isinstance(nxt, nn.Conv2d):
                            break
                        if isinstance(nxt, nn.BatchNorm2d):
                            bn = nxt
                        if isinstance(nxt, nn.Dropout2d):
                            drop = nxt
                    out.append((m, bn, drop))
            return out

        def vortex(v):
            out = [(v.gave_pool[1], v.gave_pool[3], None)]
            for br in v.branches:
                out += seq(br)
            return out + seq(v.final)
        net = self.net
        layers = []
        if isinstance(net, FusionNet_v3):
            for blk in net.block0:
                layers += seq(blk.block)
            layers += vortex(net.vortex0)
            if net.config.use_semantics:
                for blk in net.block2:
                    layers += seq(blk.block)
                layers += vortex(net.vortex2)
            layers += vortex(net.vortex3)
        else:
            for blk in net.block:
                layers += seq(blk.block)
            layers += vortex(net.vortex) + vortex(net.vortex_final)
        for p in net.pred:
            layers += seq(p.pred)
        return layers

    def _trainer(self, h, w, dev, slot=0):
        from .model import FusionNet_v3
        key = (h, w, str(dev)) if slot == 0 else (h, w, str(dev), slot)
        tr = self._trainers.get(key)
        if tr is None:
            lib = _lib.load()
            net = self.net
            handle = _lib._vp()
            with torch.cuda.device(dev):
                _lib.check(lib.ojf_trainer_create(_lib._c.byref(handle), 3 if isinstance(net, FusionNet_v3) else 2, net.n_points, net.gf,
                                                  int(bool(net.config.use_semantics)), float(net.scale), h, w), 'ojf_trainer_create')
            tr = self._trainers[key] = _TrainerHandle(handle)
            _lib.check(lib.ojf_trainer_set_arithmetic(handle, _lib.ARITHMETIC[self.arithmetic]), 'ojf_trainer_set_arithmetic')
            _lib.check(lib.ojf_trainer_set_backward_arithmetic(handle, _lib.ARITHMETIC[self.backward_arithmetic]),
                       'ojf_trainer_set_backward_arithmetic')
            if self.replay is not None:
                _lib.check(lib.ojf_trainer_set_graph(handle, int(bool(self.replay))), 'ojf_trainer_set_graph')
        return tr

    # ---- backward passes beside the next forward stage (overlap=True) -------------------------------------------------------
    def gradient_stream(self, dev):
        """The stream the executor's backward passes run on under ``overlap`` (None otherwise)."""
        if not self.overlap:
            return None
        if self._grad_stream is None or self._grad_stream.device != torch.device(dev):
            self._grad_stream = torch.cuda.Stream(device=dev)
        return self._grad_stream

    def _submit(self, fn):
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='ojf-gradients')  # ONE thread: jobs run in submission order
        fut = self._pool.submit(fn)
        self._jobs.append(fut)
        return fut

    def _drain(self):
        """The calling host thread waits until the gradient thread has enqueued everything it was handed; a job's exception surfaces here."""
        jobs, self._jobs = self._jobs, []
        err = None
        for f in jobs:
            try:
                f.result()
            except Exception as e:  # (every job is waited for: nothing of an earlier frame is left running behind the error)
                err = err or e
        if err is not None:
            raise err

    def gradient_work(self, fn, join=True):
        """Runs ``fn()`` - whatever reads or writes the parameter gradients or writes the parameters - behind the backward passes
        enqueued so far: on the gradient stream, from the gradient thread under ``overlap_thread`` (else like ``with gradients(): fn()``).
        ``join=False`` lets the caller go on before ``fn`` has been enqueued; it is only valid when ``fn`` writes NO parameter (gradient
        clipping between optimizer steps): a forward pass decides from the parameters' version counters whether it has to wait."""
        gs = self._grad_stream if self.overlap else None
        if gs is None or not self.overlap_thread:
            with self.gradients():
                fn()
            return
        ev = torch.cuda.current_stream(gs.device).record_event()
        self._unjoined_backward = False

        def job():
            with torch.cuda.stream(gs):
                gs.wait_event(ev)
                fn()
                self._grad_tail = gs.record_event()
        self._submit(job)
        if join:
            self._drain()

    def gradients(self):
        """Context for everything that reads or writes the parameter gradients or writes the parameters (clip_grad_norm_, the
        flat-gradient all-reduce, optimizer.step(), zero_grad()): under ``overlap`` it runs on the gradient stream, behind the backward
        passes enqueued so far and behind the caller's stream as of entry; otherwise it is the caller's stream unchanged."""
        return _GradientContext(self)

    def join_gradients(self, dev=None):
        """The caller's current stream waits for everything enqueued on the gradient stream so far (no-op without ``overlap``)."""
        self._drain()
        self._unjoined_backward = False
        if self._grad_tail is not None:
            torch.cuda.current_stream(dev if dev is not None else self._grad_stream.device).wait_event(self._grad_tail)

    def invalidate(self):
        """Forces the packed weight copies to be rebuilt by the next forward pass.  In-place updates (optimizer steps,
        ``load_state_dict``) and re-assigned storage are seen without it (per-tensor version counters and addresses);
        writes through ``p.data`` / ``tensor.data.copy_`` / ``dist.broadcast(p.data)`` bump no counter: call this after
        them."""
        self._weights_sig = None

    @property
    def replays(self):
        """Executor passes served by a device-graph replay so far (ojf_trainer_graph_replays), over all frame shapes."""
        lib = _lib.load()
        return sum(int(lib.ojf_trainer_graph_replays(t.handle)) for t in self._trainers.values())

    @property
    def launches(self):
        """Kernel launches of the most recent forward + backward pass of the executor (counted by the library)."""
        return max((t.fwd_launches + t.bwd_launches for t in self._trainers.values()), default=0)

    def _epoch(self, mods):
        """A number that changes whenever a convolution weight or bias may have changed (ojf_trainer_forward's
        ``weights_epoch``).  Per pass the (storage address, version counter) pairs of the cached tensor list are compared
        element-wise (no sum that could cancel): in-place updates (optimizer steps, ``load_state_dict``) bump the counters,
        ``p.data = ...`` / ``module.to()`` / ``.float()`` move the storage of the same Parameter object.  REPLACED Parameter
        objects (``load_state_dict(assign=True)``, a swapped sub-module) are seen when the list is collected again from the
        live modules, every 64th pass; ``invalidate()`` covers writes that bump nothing (``dist.broadcast(p.data)``)."""
        cache = self.__dict__.get('_epoch_cache')
        self._epoch_calls = self.__dict__.get('_epoch_calls', 0) + 1
        if cache is None or self._weights_sig is None or self._epoch_calls % 64 == 0:
            tensors = [t for c, _, _ in mods for t in (c.weight, c.bias) if t is not None]
            if cache is None or len(tensors) != len(cache) or any(a is not b for a, b in zip(tensors, cache)):
                self._weights_sig = None
            cache = self._epoch_cache = tensors
        sig = [(t.data_ptr(), t._version) for t in cache]
        if sig != self._weights_sig:
            self._weights_sig = sig
            self._weights_epoch += 1
        return self._weights_epoch

    def _forward_executor(self, x):
        net = self.net
        lib = _lib.load()
        v = x['tsdf_values']
        dev = v.device
        _, P, h, w = v.shape
        slot = self._passes % 2 if (self.overlap and torch.is_grad_enabled()) else 0
        self._passes += 1
        tr = self._trainer(h, w, dev, slot)
        if tr.bwd_job is not None:  # (its launches are being enqueued by the gradient thread: the table and the event below are its)
            job, tr.bwd_job = tr.bwd_job, None
            job.result()
        if tr.bwd_done is not None:  # this executor's activations are still being read by its last backward pass on the gradient stream
            torch.cuda.current_stream(dev).wait_event(tr.bwd_done)
            tr.bwd_done = None
        mods = self.__dict__.get('_mods')
        if mods is None:
            mods = self._mods = self._layer_modules()
            self._params = [p for p in net.parameters()]
            self._pindex = {id(p): i for i, p in enumerate(self._params)}
            assert lib.ojf_trainer_layer_count(tr.handle) == len(mods)
        table = tr.table
        if table is None:  # one table per trainer (= per frame shape), for its life: backward finds the gradient pointers in place
            table = tr.table = (_lib.TrainLayer * len(mods))()
        # Dropout2d: one uniform draw for every active layer of this pass, per-channel factors 0 or 1 / keep.  The factors live in
        # ONE persistent buffer per layer set (the table keeps its addresses): the same draw from the same generator as a fresh
        # tensor per pass, without a table update per pass.
        drops = [(i, d) for i, (c, b, d) in enumerate(mods) if d is not None and d.training and d.p > 0]
        drop_sig = tuple((i, d.p, mods[i][0].out_channels) for i, d in drops)
        scales = {}
        if drops:
            keeps = self.__dict__.setdefault('_keeps', {})  # (per executor: a backward pass reads the factors of ITS forward pass)
            keep = keeps.get(slot)
            if keep is None or keep[0] != drop_sig or keep[1].device != dev:
                kv = torch.cat([torch.full((mods[i][0].out_channels,), 1.0 - d.p) for i, d in drops]).to(dev)
                keep = keeps[slot] = (drop_sig, kv, torch.empty_like(kv), torch.empty_like(kv), torch.empty_like(kv, dtype=torch.bool))
            rand, factors = keep[2], keep[3]
            torch.rand(rand.shape, device=dev, out=rand)
            torch.lt(rand, keep[1], out=keep[4])
            factors.copy_(keep[4])                     # 0. / 1.
            factors.div_(keep[1])
            off = 0
            for i, d in drops:
                n = mods[i][0].out_channels
                scales[i] = factors[off:off + n]
                off += n
        epoch = self._epoch(mods)
        if self._grad_tail is not None and epoch != self._epoch_joined:
            if self._unjoined_backward:
                # a parameter was written while a backward pass on the gradient stream had neither been joined nor been followed by a
                # gradients() / gradient_work() section: that optimizer step ran on the caller's stream BESIDE the backward pass it needed
                self._epoch_joined = epoch
                raise _lib.OjfError('HipTrainNet: a parameter changed outside `with pipeline.gradients():` while train_overlap is on - the '
                                    'optimizer step (and gradient clipping, zero_grad) must run inside that context (or behind '
                                    'pipeline.join_gradients()): the backward pass runs on the gradient stream, not on the current one')
            # a weight changed since the last pass that looked (the optimizer step sits on the gradient stream): this pass - packing included -
            # goes behind it
            torch.cuda.current_stream(dev).wait_event(self._grad_tail)
        self._epoch_joined = epoch
        # The layer table (pointers, geometry, BatchNorm flags) changes only when a tensor moves or a module switches mode: per pass
        # the cached tensors' addresses and the mode flags are compared (~50 us); the walk over the module tree (~60 layers x a
        # dozen nn.Module attribute lookups and ctypes stores: ~0.4 ms of host time per frame, on a step whose host side paces
        # it on slower hosts) runs only when they differ, and every 64th pass (replaced Parameter objects).
        fc = tr.__dict__.get('fill_cache')
        tr.fill_calls = tr.__dict__.get('fill_calls', 0) + 1
        if (fc is not None and tr.fill_calls % 64 and fc['drop_sig'] == drop_sig
                and fc['flags'] == [m.training for m in fc['bns']] and fc['ptrs'] == [t.data_ptr() for t in fc['tensors']]
                and fc['drop_ptrs'] == [scales[i].data_ptr() for i, _ in drops]):
            counters = fc['counters']
        else:
            counters, tensors, bns = [], [], []
            for i, (conv, bn, drop) in enumerate(mods):
                e = table[i]
                e.weight, e.bias = conv.weight.data_ptr(), _p(conv.bias)
                tensors += [t for t in (conv.weight, conv.bias) if t is not None]
                e.out_channels, e.in_channels, e.ksize, e.dilation = conv.out_channels, conv.in_channels, conv.kernel_size[0], conv.dilation[0]
                if bn is not None:
                    if bn.momentum is None:
                        raise _lib.OjfError('HipTrainNet: BatchNorm2d(momentum=None) is not supported (use train_engine: torch)')
                    e.gamma, e.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                    e.running_mean, e.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                    e.bn_training, e.momentum, e.eps = int(bn.training), float(bn.momentum), float(bn.eps)
                    tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
                    bns.append(bn)
                    if bn.training:
                        counters.append(bn.num_batches_tracked)
                e.drop_scale = scales[i].data_ptr() if i in scales else None
            tr.fill_cache = dict(tensors=tensors, ptrs=[t.data_ptr() for t in tensors], bns=bns, flags=[m.training for m in bns],
                                 counters=counters, drop_sig=drop_sig, drop_ptrs=[scales[i].data_ptr() for i, _ in drops])
        ins = [x['tsdf_values'], x['tsdf_weights'], x['tsdf_frame']] + ([x['semantic_frame']] if net.config.use_semantics else [])
        ins = [t.contiguous().float() for t in ins]
        self._gen += 1
        state = dict(table=table, tr=tr, gen=self._gen, keep_alive=(scales, ins), epoch=epoch, dev=dev, shape=(1, P, h, w))
        est = _NetFn.apply(self, state, *(ins + self._params))
        if counters:
            torch._foreach_add_(counters, 1)  # nn.BatchNorm2d.num_batches_tracked, all at once
        return est

    def _graphed(self, x):
        net = self.net
        keys = ['tsdf_values', 'tsdf_weights', 'tsdf_frame'] + (['semantic_frame'] if net.config.use_semantics else [])
        sig = (net.training, tuple((k, tuple(x[k].shape)) for k in keys), str(x[keys[0]].device))
        g = self._graphs.get(sig)
        if g is None:
            g = self._graphs[sig] = _GraphedPass(self, {k: x[k] for k in keys})
        if not g.ok:
            return None
        return _GraphedFn.apply(g, *([x[k] for k in keys] + g.params))

    def _forward_impl(self, x):
        net = self.net
        from .model import FusionNet_v3
        self._counters = []
        if any(m.training and m.p > 0 for m in net.modules() if isinstance(m, nn.Dropout2d)):
            # every Dropout2d mask of this pass from one draw (sliced per layer)
            total = sum(m.out_channels for m in net.modules() if isinstance(m, nn.Conv2d))
            self._rand, self._rand_at, self._masks = torch.rand(total, device=x['tsdf_values'].device), 0, {}
        c = net.n_channels
        slot = (c + 3) // 4 * 4
        out_c = c * (net.gf + 1)
        out_slot = (out_c + 3) // 4 * 4
        if isinstance(net, FusionNet_v3):
            y = self._vortex(self._dense(to_c4(torch.cat([x['tsdf_values'], x['tsdf_weights'], x['tsdf_frame']], 1)), net.block0, c),
                             net.vortex0, c, slot)
            if net.config.use_semantics:
                s = to_c4(torch.cat([x['tsdf_values'], x['tsdf_weights'], x['semantic_frame']], 1))
                y = torch.cat([y, self._vortex(self._dense(s, net.block2, c), net.vortex2, c, slot)], dim=0)
            y = self._vortex(y, net.vortex3, out_c, out_slot)
        else:
            parts = [x['tsdf_values'], x['tsdf_weights'], x['tsdf_frame']]
            if net.config.use_semantics:
                parts.append(x['semantic_frame'])
            y = self._vortex(self._dense(to_c4(torch.cat(parts, 1)), net.block, c), net.vortex, c, slot)
            y = self._vortex(y, net.vortex_final, out_c, out_slot)
        preds = list(net.pred)
        for i, p in enumerate(preds):
            y = self._sequential(y, p.pred, p.pred[0].in_channels, y.shape[0] * 4, net.scale if i == len(preds) - 1 else 1.0)
        if self._counters:
            torch._foreach_add_(self._counters, 1)  # nn.BatchNorm2d.num_batches_tracked, all at once
        return from_c4(y, net.n_points)

    __call__ = forward


class FuseOutput(torch.autograd.Function):
    """modules/pipeline.py:104-127 as one launch each way (ojf_train_fuse_output): the weighted update
    (max(w, 0) v + clamp(est, +-init)) / (max(w, 0) + 1) on sample planes [1, P, n], gathered at the valid pixels into the
    API's rows [1, Nv, P].  Gradient flows to ``est`` only (values / weights come out of the volumes)."""

    @staticmethod
    def forward(ctx, est_pn, fv_pn, fw_pn, valid, init):
        lib = _lib.load()
        est_pn, fv_pn, fw_pn, valid = est_pn.contiguous(), fv_pn.contiguous(), fw_pn.contiguous(), valid.contiguous()
        _, P, n = est_pn.shape
        nv = valid.numel()
        rows = torch.empty((1, nv, P), dtype=torch.float32, device=est_pn.device)
        _lib.check(lib.ojf_train_fuse_output(est_pn.data_ptr(), fv_pn.data_ptr(), fw_pn.data_ptr(), valid.data_ptr(), n, P, nv, float(init),
                                             rows.data_ptr(), _lib.stream_ptr(est_pn.device)), 'ojf_train_fuse_output')
        ctx.save_for_backward(est_pn, fw_pn, valid)
        ctx.init = float(init)
        return rows

    @staticmethod
    def backward(ctx, d_rows):
        lib = _lib.load()
        est_pn, fw_pn, valid = ctx.saved_tensors
        _, P, n = est_pn.shape
        d_rows = d_rows.contiguous()
        d_est = torch.empty_like(est_pn)
        _lib.check(lib.ojf_train_fuse_output_bwd(d_rows.data_ptr(), est_pn.data_ptr(), fw_pn.data_ptr(), valid.data_ptr(), n, P, valid.numel(),
                                                 ctx.init, d_est.data_ptr(), _lib.stream_ptr(est_pn.device)), 'ojf_train_fuse_output_bwd')
        return d_est, None, None, None, None


class _GradientContext:
    def __init__(self, tn):
        self.tn, self.ctx = tn, None

    def __enter__(self):
        tn = self.tn
        st = tn._grad_stream if tn.overlap else None
        tn._drain()
        tn._unjoined_backward = False
        if st is not None:
            st.wait_stream(torch.cuda.current_stream(st.device))
            self.ctx = torch.cuda.stream(st)
            self.ctx.__enter__()
        return st

    def __exit__(self, *exc):
        if self.ctx is not None:
            st = self.tn._grad_stream
            self.tn._grad_tail = st.record_event()
            return self.ctx.__exit__(*exc)
        return False


class _TrainerHandle:
    fwd_launches = bwd_launches = 0
    bwd_done = None  # overlap: event behind this executor's last backward pass on the gradient stream
    bwd_job = None   # overlap_thread: the future of that pass's enqueue job

    def __init__(self, handle):
        self.handle = handle
        self.gen = 0  # generation of the forward pass whose activations the trainer holds
        self.table = None   # this trainer's layer table (ojf_train_layer[n]): pointers / flags of ITS last forward pass
        self.grad_sig = None  # the p.grad tensors whose addresses the table holds (steady state of in-place accumulation)

    def __del__(self):
        try:
            _lib.load().ojf_trainer_destroy(self.handle)
        except Exception:
            pass


class _NetFn(torch.autograd.Function):
    """The whole net as ONE autograd node on ojf_trainer_forward / ojf_trainer_backward.  The trainer keeps the
    activations of the LAST forward pass only: backward of an older pass raises instead of differentiating the wrong one."""

    @staticmethod
    def forward(ctx, tn, state, *tensors):
        lib = _lib.load()
        n_in = 4 if tn.net.config.use_semantics else 3
        ins = tensors[:n_in]
        est = torch.empty(state['shape'], dtype=torch.float32, device=state['dev'])
        tr = state['tr']
        _lib.check(lib.ojf_trainer_forward(tr.handle, state['table'], len(state['table']), state['epoch'], ins[0].data_ptr(), ins[1].data_ptr(),
                                           ins[2].data_ptr(), ins[3].data_ptr() if n_in == 4 else None, est.data_ptr(),
                                           _lib.stream_ptr(state['dev'])), 'ojf_trainer_forward')
        tr.fwd_launches = int(lib.ojf_trainer_launch_count(tr.handle))
        tr.gen = state['gen']
        ctx.tn, ctx.state, ctx.n_in = tn, state, n_in
        return est

    @staticmethod
    def backward(ctx, dest):
        lib = _lib.load()
        tn, state = ctx.tn, ctx.state
        tr, table = state['tr'], state['table']
        if tr.gen != state['gen']:
            raise _lib.OjfError('HipTrainNet: backward of a forward pass whose activations were overwritten by a later forward '
                                '(the executor keeps one pass; call backward before the next forward, or use executor=False)')
        needs = ctx.needs_input_grad[2 + ctx.n_in:]
        inplace = tn.inplace_grads and not torch.cuda.is_current_stream_capturing()
        out = [None] * len(tn._params)
        sig = tr.grad_sig
        steady = inplace and sig is not None and all(p.grad is g for p, g in zip(tn._params, sig))
        dest = dest.contiguous()

        def run_backward(threaded=False):
            gs = tn.gradient_stream(state['dev']) if inplace else None
            if gs is None:
                _lib.check(lib.ojf_trainer_backward(tr.handle, table, len(table), dest.data_ptr(), _lib.stream_ptr(state['dev'])), 'ojf_trainer_backward')
            else:  # overlap: behind everything enqueued on this stream so far (d_est, the gradient tensors), beside what comes next on it
                ev = torch.cuda.current_stream(state['dev']).record_event()

                def enqueue():
                    with torch.cuda.device(state['dev']):
                        gs.wait_event(ev)
                        dest.record_stream(gs)
                        _lib.check(lib.ojf_trainer_backward(tr.handle, table, len(table), dest.data_ptr(), gs.cuda_stream), 'ojf_trainer_backward')
                        tr.bwd_done = tn._grad_tail = gs.record_event()
                        tr.bwd_launches = int(lib.ojf_trainer_launch_count(tr.handle))
                tn._unjoined_backward = True
                if threaded:  # (steady state only: nothing touches this executor's table until its next forward pass has waited for the job)
                    tr.bwd_job = tn._submit(enqueue)
                else:
                    tn._drain()
                    enqueue()
                return
            tr.bwd_launches = int(lib.ojf_trainer_launch_count(tr.handle))

        if steady:  # every parameter still accumulates into the tensor whose address the table already holds
            run_backward(threaded=tn.overlap_thread)
            return (None, None) + (None,) * ctx.n_in + tuple(out)
        all_accumulating = inplace
        for i, (conv, bn, drop) in enumerate(tn._mods):
            e = table[i]
            ps = [(conv.weight, 'grad_weight'), (conv.bias, 'grad_bias')] + ([(bn.weight, 'grad_gamma'), (bn.bias, 'grad_beta')] if bn is not None else [])
            ps = [(p, f) for p, f in ps if p is not None]
            if inplace and all(p.is_leaf and p.requires_grad for p, _ in ps):
                fresh = [p.grad is None for p, _ in ps]
                ok = all(p.grad is None or (p.grad.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.device == p.device
                                            and p.grad.shape == p.shape) for p, _ in ps)
                if ok:
                    for (p, f), new in zip(ps, fresh):
                        if new:  # one accumulate flag per layer: in a mixed state the fresh tensors start from zero
                            p.grad = torch.empty_like(p, memory_format=torch.contiguous_format) if all(fresh) else \
                                torch.zeros_like(p, memory_format=torch.contiguous_format)
                        setattr(e, f, p.grad.data_ptr())
                    e.accumulate = 0 if all(fresh) else 1
                    all_accumulating = all_accumulating and not any(fresh)
                    continue
            e.accumulate = 0
            all_accumulating = False
            for p, f in ps:  # the ordinary route: gradients are returned to autograd
                g = torch.empty_like(p, memory_format=torch.contiguous_format)
                setattr(e, f, g.data_ptr())
                k = tn._pindex[id(p)]
                if needs[k]:
                    out[k] = g
                else:
                    state.setdefault('scratch', []).append(g)
            state.setdefault('scratch', []).extend(o for o in out if o is not None)
        run_backward()
        if tr.bwd_done is not None and any(o is not None for o in out):  # gradients handed to autograd: its stream must see them
            torch.cuda.current_stream(state['dev']).wait_event(tr.bwd_done)
        # steady state from the next pass on: the gradient tensors exist and the table says "accumulate" for every layer
        tr.grad_sig = [p.grad for p in tn._params] if inplace else None
        if tr.grad_sig is not None and not all_accumulating:
            for e in table:
                e.accumulate = 1
            if any(g is None for g in tr.grad_sig):
                tr.grad_sig = None
        return (None, None) + (None,) * ctx.n_in + tuple(out)


class _GraphedPass:
    """Forward and backward of a HipTrainNet for one (mode, frame shape) as two device graphs on static tensors (the
    recipe of torch.cuda.make_graphed_callables, for a callable that walks another module's parameters)."""

    def __init__(self, tn, sample):
        self.ok = False
        net = tn.net
        self.params = [p for p in net.parameters() if p.requires_grad]
        dev = self.params[0].device
        try:
            self.static_in = {k: torch.zeros_like(v, device=dev) for k, v in sample.items()}
            for k, v in sample.items():
                self.static_in[k].copy_(v)
            buffers = [b.clone() for b in net.buffers()]  # the warm-up passes must not count as training steps
            # ... nor leave anything in the parameters' .grad (with in-place gradients the warm-up's backward
            # wrote into p.grad - views of the flat all-reduce buffer in train_fusion - and the first optimizer step
            # saw it): the warm-up runs against detached .grad slots, the original tensors come back afterwards
            saved_grads = [p.grad for p in self.params]
            for p in self.params:
                p.grad = None
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):  # settles lazy allocations (zero-bias vector, rocBLAS handles) outside the capture
                    out = tn._forward_impl(self.static_in)
                    torch.autograd.grad(out, self.params, grad_outputs=torch.ones_like(out), allow_unused=True)
            torch.cuda.current_stream(dev).wait_stream(side)
            pool = torch.cuda.graph_pool_handle()
            self.fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.fwd, pool=pool):
                self.static_out = tn._forward_impl(self.static_in)
            self.static_dout = torch.zeros_like(self.static_out)
            self.bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.bwd, pool=pool):
                self.static_grads = torch.autograd.grad(self.static_out, self.params, grad_outputs=self.static_dout, allow_unused=True)
            with torch.no_grad():
                for b, saved in zip(net.buffers(), buffers):
                    b.copy_(saved)
            self.ok = True
        except Exception as e:  # capture is an optimisation only; the eager launches remain
            self.error = e
            torch.cuda.synchronize(dev)
        finally:
            if 'saved_grads' in locals():
                for p, gsaved in zip(self.params, saved_grads):
                    p.grad = gsaved


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, *tensors):
        n_in = len(g.static_in)
        for dst, src in zip(g.static_in.values(), tensors[:n_in]):
            dst.copy_(src)
        g.fwd.replay()
        ctx.g = g
        ctx.n_in = n_in
        return g.static_out.detach().clone()

    @staticmethod
    def backward(ctx, dout):
        g = ctx.g
        g.static_dout.copy_(dout)
        g.bwd.replay()
        return (None,) * (1 + ctx.n_in) + tuple(None if t is None else t.clone() for t in g.static_grads)
