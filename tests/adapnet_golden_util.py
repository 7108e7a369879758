"""Shared by tests/test_adapnet.py and tests/test_adapnet_engine_gpu.py: the seeded, signal-preserving initialisation of
tests/golden/make_golden_adapnet.py::randomise_net (one RNG stream consumed in module order - valid because the
package's module tree registers its modules in the reference's order, which the tests assert first)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'adapnet_net.npz')


def randomise_net(m, seed):
    g = torch.Generator().manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Conv2d):
            mod.weight.data.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / mod.weight[0].numel()) ** 0.5)
        elif isinstance(mod, torch.nn.ConvTranspose2d):
            fan = mod.weight.shape[0] * (mod.kernel_size[0] / mod.stride[0]) ** 2
            mod.weight.data.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and mod.bias is not None:
            mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) * 0.5 + 0.5)
            mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
            mod.running_mean.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.bias.shape, generator=g) + 0.5)
    for mod in m.modules():
        if hasattr(mod, 'bn3'):
            mod.bn3.weight.data.mul_(0.4)
        if hasattr(mod, 'dropout') and isinstance(mod.dropout, bool):
            mod.dropout = False
    return m.eval()


def golden_net(stage):
    """(package AdapNet initialised like the reference instance of the fixture, inputs, reference outputs)."""
    from online_joint_depthfusion_and_semantic_amd import adapnet
    from online_joint_depthfusion_and_semantic_amd.config import AttrDict
    g = np.load(GOLDEN)
    tag = 'stage%d.' % stage
    n_classes = int(g[tag + 'out0'].shape[1])
    net = adapnet.AdapNet(AttrDict(stage=stage, n_classes=n_classes))
    assert list(net.state_dict().keys()) == list(g[tag + 'keys'])  # same schema, same parameter order
    assert [type(x).__name__ for x in net.modules()] == list(g[tag + 'modules'])  # same module order (RNG stream)
    randomise_net(net, 20 + stage)
    ins = [torch.from_numpy(g[tag + 'in1'])] + ([torch.from_numpy(g[tag + 'in2'])] if stage != 1 else [])
    outs = [torch.from_numpy(g[tag + 'out%d' % i]) for i in range(3)]
    return net, ins, outs
