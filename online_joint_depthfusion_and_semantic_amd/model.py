"""Fusion networks: parameter containers with the reference's state_dict key schema.

Counterpart of the reference's ``modules/model.py`` (FusionNet_v2 :164-216, FusionNet_v3 :219-283,
VortexPooling :100-161, Block :4-21, Pred :24-52; key schema in SURVEY.md Appendix C) so that
checkpoints written by ``train_fusion.py`` load unchanged (``model_state`` keys such as
``block0.3.block.4.weight`` or ``vortex3.branches.2.9.bias``).

Inference does NOT run these modules: ``fold_layers`` folds every BatchNorm into its convolution
and hands the result to the HIP executor (``FusionNetEngine`` -> ojf_net_* in include/ojf.h).
The torch ``forward`` exists for training (autograd) and as the shape reference.
"""
import numpy as np
import torch
from torch import nn


def _cbr(cin, cout, k=1, dilation=1, act='leaky', dropout=True, bn=True):
    """conv -> [BN] -> activation -> [Dropout2d]: the slot order the reference's Sequentials use."""
    pad = dilation * (k // 2)
    mods = [nn.Conv2d(cin, cout, kernel_size=k, dilation=dilation, padding=pad)]
    if bn:
        mods.append(nn.BatchNorm2d(cout))
    if act == 'leaky':
        mods.append(nn.LeakyReLU())
    elif act == 'relu':
        mods.append(nn.ReLU())
    elif act == 'tanh':
        mods.append(nn.Tanh())
    if dropout:
        mods.append(nn.Dropout2d(p=0.2))
    return mods


class Block(nn.Module):
    """Two 3x3 conv+BN+LeakyReLU+Dropout stages (model.py:4-21); keys ``block.{0,1,4,5}``."""

    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(*(_cbr(cin, cout, 3) + _cbr(cout, cout, 3)))

    def forward(self, x):
        return self.block(x)


class Pred(nn.Module):
    """1x1 stacks of the prediction head (model.py:24-52); the last one ends in Tanh."""

    def __init__(self, cin, cout, n_points=None):
        super().__init__()
        if n_points is None:
            mods = _cbr(cin, cout) + _cbr(cout, cout)
        else:
            mods = (_cbr(cin, cout) + _cbr(cout, cout, bn=False, dropout=False)
                    + _cbr(cout, n_points, act='tanh', bn=False, dropout=False))
        self.pred = nn.Sequential(*mods)

    def forward(self, x):
        return self.pred(x)


class _Broadcast(nn.Module):
    """Bilinear up-sampling of a 1x1 map (model.py:107-108) is a broadcast: same values, but its backward is a
    reduction instead of ``upsample_bilinear2d_backward``'s 76 800 x C atomic adds into one pixel (15 ms per
    VortexPooling at 320x240: two thirds of a training step).  No parameters: the state_dict is unchanged."""

    def __init__(self, size):
        super().__init__()
        self.size = tuple(size) if isinstance(size, (tuple, list)) else (size, size)

    def forward(self, x):
        if x.shape[-2:] != (1, 1):
            return nn.functional.interpolate(x, size=self.size, mode='bilinear', align_corners=True)
        return x.expand(-1, -1, self.size[0], self.size[1]).contiguous()


class VortexPooling(nn.Module):
    """Global-average branch + four dilated branches (rates 1,3,9,27) on successively 3x3-average-
    pooled inputs, concatenated and mixed by a 1x1 conv (model.py:100-161)."""
    rates = (1, 3, 9, 27)

    def __init__(self, in_chs, mid_chs, out_chs, feat_res):
        super().__init__()
        self.gave_pool = nn.Sequential(
            nn.AdaptiveAvgPool2d((1, 1)),
            nn.Conv2d(in_chs, out_chs, kernel_size=1),
            _Broadcast(feat_res),  # = nn.Upsample(size=feat_res, mode='bilinear', align_corners=True) of a 1x1 map
            nn.BatchNorm2d(out_chs))
        for i in (1, 2, 3):
            setattr(self, 'pool%d' % i, nn.AvgPool2d(kernel_size=3, stride=1, padding=1))
        self.branches = nn.ModuleList([
            nn.Sequential(*(_cbr(in_chs, mid_chs, 1, act='relu', dropout=False)
                            + _cbr(mid_chs, mid_chs, 3, r, act='relu', dropout=False)
                            + _cbr(mid_chs, mid_chs, 3, r, act='relu', dropout=False)
                            + _cbr(mid_chs, out_chs, 1, act='relu', dropout=False)))
            for r in self.rates])
        self.final = nn.Sequential(nn.Conv2d(5 * out_chs, out_chs, kernel_size=1),
                                   nn.BatchNorm2d(out_chs), nn.Dropout2d(p=0.2, inplace=True))

    def forward(self, x):
        outs = [self.gave_pool(x), self.branches[0](x)]
        xp = x
        for i in (1, 2, 3):
            xp = getattr(self, 'pool%d' % i)(xp)
            outs.append(self.branches[i](xp))
        return self.final(torch.cat(outs, dim=1))


def _dense(x, blocks):
    for blk in blocks:
        x = torch.cat([x, blk(x)], dim=1)
    return x


def _pred_stack(gf, c, n_points):
    return nn.Sequential(*[Pred((gf + 1 - i) * c, (gf - i) * c, n_points if i == gf - 1 else None)
                           for i in range(gf)])


class FusionNet_v3(nn.Module):
    """model.py:219-283.  ``config`` needs n_points, growth_factor, use_semantics, output_scale,
    resx, resy (Pipeline copies resx/resy from DATA, pipeline.py:24-25)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.scale = config.output_scale
        self.n_points = config.n_points
        self.n_channels = c = 2 * config.n_points + 1
        self.gf = gf = config.growth_factor - 1
        res = (config.resy, config.resx)
        pool_in = c * (gf + 1)
        self.block0 = nn.ModuleList([Block((i + 1) * c, c) for i in range(gf)])
        self.vortex0 = VortexPooling(pool_in, c, pool_in, res)
        heads = 1
        if config.use_semantics:
            heads = 2
            self.block2 = nn.ModuleList([Block((i + 1) * c, c) for i in range(gf)])
            self.vortex2 = VortexPooling(pool_in, c, pool_in, res)
        self.vortex3 = VortexPooling(heads * pool_in, c, pool_in, res)
        self.pred = _pred_stack(gf, c, config.n_points)

    def forward(self, x):
        y = self.vortex0(_dense(torch.cat([x['tsdf_values'], x['tsdf_weights'], x['tsdf_frame']], 1), self.block0))
        if self.config.use_semantics:
            s = torch.cat([x['tsdf_values'], x['tsdf_weights'], x['semantic_frame']], 1)
            y = torch.cat([y, self.vortex2(_dense(s, self.block2))], dim=1)
        return self.pred(self.vortex3(y)) * self.scale


class FusionNet_v2(nn.Module):
    """model.py:164-216: one head (semantic channel appended to the input), two VortexPoolings."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.scale = config.output_scale
        self.n_points = config.n_points
        self.n_channels = c = 2 * config.n_points + 1 + int(config.use_semantics)
        self.gf = gf = config.growth_factor - 1
        res = (config.resy, config.resx)
        pool_in = c * (gf + 1)
        self.block = nn.ModuleList([Block((i + 1) * c, c) for i in range(gf)])
        self.vortex = VortexPooling(pool_in, c, pool_in, res)
        self.vortex_final = VortexPooling(pool_in, c, pool_in, res)
        self.pred = _pred_stack(gf, c, config.n_points)

    def forward(self, x):
        parts = [x['tsdf_values'], x['tsdf_weights'], x['tsdf_frame']]
        if self.config.use_semantics:
            parts.append(x['semantic_frame'])
        y = _dense(torch.cat(parts, dim=1), self.block)
        return self.pred(self.vortex_final(self.vortex(y))) * self.scale


# ---- eval-mode folding for the HIP executor -----------------------------------------------------

def _fold(conv, bn=None):
    """(weight [co,ci,k,k] f32, bias [co] f32, ksize, dilation) with BN (running stats) folded in;
    the arithmetic is done in fp64 and rounded once."""
    w = conv.weight.detach().cpu().double()
    b = conv.bias.detach().cpu().double() if conv.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64)
    if bn is not None:
        s = bn.weight.detach().cpu().double() / torch.sqrt(bn.running_var.detach().cpu().double() + bn.eps)
        w = w * s.view(-1, 1, 1, 1)
        b = (b - bn.running_mean.detach().cpu().double()) * s + bn.bias.detach().cpu().double()
    return (np.ascontiguousarray(w.float().numpy()), np.ascontiguousarray(b.float().numpy()),
            int(conv.kernel_size[0]), int(conv.dilation[0]))


def _fold_sequential(seq):
    mods = list(seq)
    out = []
    for i, m in enumerate(mods):
        if isinstance(m, nn.Conv2d):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            out.append(_fold(m, nxt if isinstance(nxt, nn.BatchNorm2d) else None))
    return out


def _fold_vortex(v):
    # gave_pool: conv 1x1 on the 1x1 map, BN applied after the (constant) upsample
    out = [_fold(v.gave_pool[1], v.gave_pool[3])]
    for br in v.branches:
        out += _fold_sequential(br)
    out += _fold_sequential(v.final)
    return out


def fold_layers(net):
    """Canonical folded-layer order expected by ojf_net_create (DESIGN.md "net layer order"):
    v3: block0[i].{a,b} (10) | vortex0 (gave, 4 x {1x1, 3x3, 3x3, 1x1}, final = 18)
        | [block2 (10) | vortex2 (18)] | vortex3 (18) | pred (11)
    v2: block (10) | vortex (18) | vortex_final (18) | pred (11)."""
    layers = []
    if isinstance(net, FusionNet_v3):
        for blk in net.block0:
            layers += _fold_sequential(blk.block)
        layers += _fold_vortex(net.vortex0)
        if net.config.use_semantics:
            for blk in net.block2:
                layers += _fold_sequential(blk.block)
            layers += _fold_vortex(net.vortex2)
        layers += _fold_vortex(net.vortex3)
    elif isinstance(net, FusionNet_v2):
        for blk in net.block:
            layers += _fold_sequential(blk.block)
        layers += _fold_vortex(net.vortex)
        layers += _fold_vortex(net.vortex_final)
    else:
        raise TypeError('fold_layers: FusionNet_v2 or FusionNet_v3 expected')
    for p in net.pred:
        layers += _fold_sequential(p.pred)
    return layers
