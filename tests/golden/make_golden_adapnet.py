#!/usr/bin/env python
"""Golden vectors for the AdapNet++ blocks that import from the reference (modules/adapnet.py with
torchvision.models stubbed): eASPP, SSMA, Decoder, BottleneckSSMA.  Build container only.
    python tests/golden/make_golden_adapnet.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
tv = types.ModuleType('torchvision')
tv.models = types.ModuleType('torchvision.models')
tv.models.resnet50 = lambda *a, **k: None
sys.modules.setdefault('torchvision', tv)
sys.modules.setdefault('torchvision.models', tv.models)
from modules import adapnet as ref  # noqa: E402

torch.set_num_threads(1)


def randomise(m, seed):
    torch.manual_seed(seed)
    for p in m.parameters():
        p.data.normal_(0, 0.1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
    return m.eval()


out = {}


def put(prefix, m):
    # weights are NOT stored (the decoder alone is 17 MB): tests rebuild them with the same seeded
    # randomise() on the package's modules, whose parameter registration order equals the reference's
    out[prefix + 'keys'] = np.array(list(m.state_dict().keys()))


with torch.no_grad():
    g = torch.Generator().manual_seed(1)
    m = randomise(ref.eASPP(64, 16, 32), 1)
    x = torch.randn(1, 64, 10, 14, generator=g)
    put('easpp.', m); out['easpp_in'] = x.numpy(); out['easpp_out'] = m(x).numpy()
    m = randomise(ref.SSMA(24, 6), 2)
    a, b = torch.randn(1, 24, 9, 11, generator=g), torch.randn(1, 24, 9, 11, generator=g)
    put('ssma.', m); out['ssma_in1'] = a.numpy(); out['ssma_in2'] = b.numpy(); out['ssma_out'] = m(a, b).numpy()
    m = randomise(ref.Decoder(7, True), 3)
    x, s1, s2 = torch.randn(1, 256, 3, 4, generator=g), torch.randn(1, 24, 6, 8, generator=g), torch.randn(1, 24, 12, 16, generator=g)
    put('dec.', m); out['dec_x'] = x.numpy(); out['dec_s1'] = s1.numpy(); out['dec_s2'] = s2.numpy()
    for i, o in enumerate(m(x, s1, s2)):
        out['dec_out%d' % i] = o.numpy()
    m = randomise(ref.BottleneckSSMA(64, 16, 1, 2, 16, drop_out=False), 4)
    x = torch.randn(1, 64, 8, 8, generator=g)
    put('bott.', m); out['bott_in'] = x.numpy(); out['bott_out'] = m(x).numpy()
    m.dropout = True
    torch.manual_seed(5)
    out['bott_out_dropout'] = m(x).numpy()
np.savez_compressed(os.path.join(HERE, 'adapnet_blocks.npz'), **out)
print('written', os.path.join(HERE, 'adapnet_blocks.npz'))
