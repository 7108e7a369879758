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


# ---- whole network: the reference's AdapNet / Encoder wiring (adapnet.py:87-149,356-415) around a ResNet-50 ---------
# torchvision is absent, so the backbone object handed to the reference's Encoder is the package's ResNet50 (public
# v1.5 layout, torchvision's attribute names).  What this pins: the unit swaps and stride edit of Encoder.__init__,
# Encoder / AdapNet / Decoder forward wiring, the state_dict key list and parameter order of the whole tree.  What it
# cannot pin: torchvision's own Bottleneck arithmetic (the "backbone unpinned" note in DESIGN.md).
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from online_joint_depthfusion_and_semantic_amd import adapnet as pkg  # noqa: E402
from types import SimpleNamespace  # noqa: E402

ref.resnet50 = lambda *a, **k: pkg.ResNet50()  # the name modules/adapnet.py:4 imported, used at :101


def randomise_net(m, seed):
    """Signal-preserving seeded initialisation, consuming ONE RNG stream in module order."""
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


net_out = {}
with torch.no_grad():
    g = torch.Generator().manual_seed(11)
    for stage, n_classes, h, w in ((2, 12, 32, 48), (1, 7, 16, 32)):
        m = randomise_net(ref.AdapNet(SimpleNamespace(stage=stage, n_classes=n_classes)), 20 + stage)
        tag = 'stage%d.' % stage
        net_out[tag + 'keys'] = np.array(list(m.state_dict().keys()))
        net_out[tag + 'modules'] = np.array([type(x).__name__ for x in m.modules()])
        a = torch.randn(1, 3, h, w, generator=g)
        b = torch.rand(1, 3, h, w, generator=g) * 3
        net_out[tag + 'in1'], net_out[tag + 'in2'] = a.numpy(), b.numpy()
        outs = m(a, b) if stage != 1 else m(a)
        for i, o in enumerate(outs):
            net_out[tag + 'out%d' % i] = o.numpy()
        print(tag, [tuple(o.shape) for o in outs], 'max |logit| %.3f' % float(outs[0].abs().max()))
np.savez_compressed(os.path.join(HERE, 'adapnet_net.npz'), **net_out)
print('written', os.path.join(HERE, 'adapnet_net.npz'))


# ---- semantic_strategy 'predict' end to end: the reference's Pipeline.fuse with its own AdapNet in front ------------
# (modules/pipeline.py:42-60,181-185: image / 255, depth replicated to three channels, softmax, max over classes).  The
# backbone is the same stand-in as above; dropout flags off (the reference leaves its ResNet dropout ON at inference,
# adapnet.py:80-82 - that randomness cannot be pinned); fusion-net weights = the seeded state of make_golden.py.
def predict_pipeline(h=64, w=96, grid=32, frames=2, n_classes=12):
    import importlib
    mg = importlib.import_module('make_golden')  # same directory; brings RefPipeline, DuckDatabase, seeded_state
    from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream, gt_volumes
    cfg = mg.ref_config(h, w, True, True)
    cfg.DATA.semantic_strategy = 'predict'
    cfg.SEMANTIC_2D_MODEL = mg.NS(stage=2, n_classes=n_classes)
    import modules.pipeline as ref_pipeline
    ref_pipeline.AdapNet = ref.AdapNet  # the class object whose resnet50 name was redirected to the stand-in above
    pipe = mg.RefPipeline(cfg)
    mg.seeded_state(pipe._fusion_network, 11)
    randomise_net(pipe._semantic_2d_network, 31)
    pipe.eval()
    small = np.load(os.path.join(HERE, 'pipeline_v3_sem_24x32_g32.npz'))
    for k, v in pipe._fusion_network.state_dict().items():
        assert np.array_equal(v.numpy(), small['state_' + k]), k
    st = SyntheticStream(h, w, grid, 20, n_classes=n_classes)
    gt, _ = gt_volumes(grid, n_classes=n_classes)
    db = mg.DuckDatabase(st, True, gt)
    s = st.scene
    out = {'keys': np.array(list(pipe._semantic_2d_network.state_dict().keys()))}
    with torch.no_grad():
        for i in range(frames):
            b = st.batch(i)
            pipe.device = torch.device('cpu')
            hist = pipe._segmentation({k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
            scores, ids = hist.max(dim=-1)
            out['f%d_seg_scores' % i], out['f%d_seg_ids' % i] = scores[0].numpy(), ids[0].numpy().astype(np.uint8)
            srt = torch.sort(hist[0], dim=-1, descending=True)[0]
            out['f%d_seg_margin' % i] = (srt[..., 0] - srt[..., 1]).numpy()  # top-1 minus top-2 probability
            pipe.fuse(b, db, torch.device('cpu'))
            for k, v in (('tsdf', db.scenes_est[s].volume), ('wgt', db.fusion_weights[s]), ('ids', db.ids_est[s].volume),
                         ('scores', db.scores[s].volume)):
                out['f%d_%s' % (i, k)] = v.numpy().copy()
    return out


if '--no-predict' not in sys.argv:
    sys.path.insert(0, HERE)
    np.savez_compressed(os.path.join(HERE, 'pipeline_predict_64x96_g32.npz'), **predict_pipeline())
    print('written', os.path.join(HERE, 'pipeline_predict_64x96_g32.npz'))


# ---- the same composition at BASELINE configs[2]'s own size: 320x240 -> 256^3, 30 classes, two frames -------------------
# Digest form (whole volumes would be 134 MB per frame): per frame the (score, id, margin) images of the 2-D network and
# sha256 of the weight volume; after the last frame the TSDF / id / score volumes at the touched voxels (C order of
# ``wgt > 0``) - an arg-max flip at a near-tie pixel may legitimately change a few ids, so those are kept as values.
def predict_pipeline_full_size(h=240, w=320, grid=256, frames=2, n_classes=30):
    import hashlib
    import importlib
    mg = importlib.import_module('make_golden')
    from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream, gt_volumes
    cfg = mg.ref_config(h, w, True, True)
    cfg.DATA.semantic_strategy = 'predict'
    cfg.SEMANTIC_2D_MODEL = mg.NS(stage=2, n_classes=n_classes)
    import modules.pipeline as ref_pipeline
    ref_pipeline.AdapNet = ref.AdapNet
    pipe = mg.RefPipeline(cfg)
    mg.seeded_state(pipe._fusion_network, 11)
    randomise_net(pipe._semantic_2d_network, 31)
    pipe.eval()
    small = np.load(os.path.join(HERE, 'pipeline_v3_sem_24x32_g32.npz'))
    for k, v in pipe._fusion_network.state_dict().items():
        assert np.array_equal(v.numpy(), small['state_' + k]), k
    st = SyntheticStream(h, w, grid, 20, n_classes=n_classes)
    gt, _ = gt_volumes(grid, n_classes=n_classes)
    db = mg.DuckDatabase(st, True, gt)
    s = st.scene
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    out = {}
    with torch.no_grad():
        for i in range(frames):
            b = st.batch(i)
            pipe.device = torch.device('cpu')
            hist = pipe._segmentation({k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
            scores, ids = hist.max(dim=-1)
            out['f%d_seg_scores' % i], out['f%d_seg_ids' % i] = scores[0].numpy(), ids[0].numpy().astype(np.uint8)
            srt = torch.sort(hist[0], dim=-1, descending=True)[0]
            out['f%d_seg_margin' % i] = (srt[..., 0] - srt[..., 1]).numpy().astype(np.float16)
            pipe.fuse(b, db, torch.device('cpu'))
            out['f%d_wgt_sha256' % i] = np.array(sha(db.fusion_weights[s].numpy()))
            out['f%d_touched' % i] = np.array(int((db.fusion_weights[s] > 0).sum()))
    touched = db.fusion_weights[s].numpy() > 0
    out['last_tsdf_touched'] = db.scenes_est[s].volume.numpy()[touched]
    out['last_ids_touched'] = db.ids_est[s].volume.numpy()[touched]
    out['last_scores_touched'] = db.scores[s].volume.numpy()[touched]
    return out


if '--no-predict' not in sys.argv:
    np.savez_compressed(os.path.join(HERE, 'pipeline_predict_240x320_g256.npz'), **predict_pipeline_full_size())
    print('written', os.path.join(HERE, 'pipeline_predict_240x320_g256.npz'))
