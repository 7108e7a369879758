"""AdapNet++ / SSMA 2-D semantic front-end (``semantic_strategy: predict``): counterpart of the
reference's ``modules/adapnet.py`` with the same module tree, hence the same ``state_dict`` keys, so
that the checkpoints its ``train_segmentation.py`` writes load unchanged.

torchvision is not available in this environment, so the ResNet-50 v1.5 backbone is defined here with
torchvision's key layout (``conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample},
fc``; reference call site adapnet.py:4,101-130).  Parity status: ``BottleneckSSMA``, ``eASPP``,
``Decoder`` and ``SSMA`` are pinned against the reference's own classes, and so is the whole network: the
reference's ``AdapNet`` built around this file's ``ResNet50`` (standing in for torchvision's class) gives the golden
outputs this tree reproduces bit for bit (tests/test_adapnet.py, tests/golden/make_golden_adapnet.py).  Only
torchvision's own ``Bottleneck`` arithmetic has no fixture.  This module tree is what trains and what
holds the weights; at inference ``Pipeline`` runs it through ``adapnet_engine.SegEngine`` - every convolution on
the SEGCONV MFMA kernels (csrc/ojf_seg.hip) - unless ``SEMANTIC_2D_MODEL.engine: torch`` asks for this forward.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    """ResNet v1.5 bottleneck (stride on the 3x3), torchvision key names."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idn)


class ResNet50(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(64, 3, 1)
        self.layer2 = self._stage(128, 4, 2)
        self.layer3 = self._stage(256, 6, 2)
        self.layer4 = self._stage(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)  # unused by AdapNet but part of the checkpoint keys

    def _stage(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class BottleneckSSMA(nn.Module):
    """Multi-scale residual unit (adapnet.py:12-84): 1x1 -> two parallel dilated 3x3 halves -> 1x1.

    Quirk kept: the reference builds a fresh ``nn.Dropout(p=0.5)`` inside ``forward`` (adapnet.py:80-82),
    which is always in training mode - units with ``dropout=True`` drop activations at inference too."""

    def __init__(self, in_channels, out_channels, r1, r2, d3, stride=1, downsample=None, copy_from=None, drop_out=True):
        super().__init__()
        self.dropout = drop_out
        half = int(d3 / 2)
        self.conv2a = nn.Conv2d(out_channels, half, 3, stride=1, dilation=r1, padding=r1, bias=False)
        self.bn2a = nn.BatchNorm2d(half)
        self.conv2b = nn.Conv2d(out_channels, half, 3, stride=1, dilation=r2, padding=r2, bias=False)
        self.bn2b = nn.BatchNorm2d(half)
        self.conv3 = nn.Conv2d(d3, in_channels, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(in_channels)
        if copy_from is None:
            self.conv1 = nn.Conv2d(in_channels, out_channels, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(out_channels)
        else:
            self.conv1, self.bn1 = copy_from.conv1, copy_from.bn1
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        a = self.relu(self.bn2a(self.conv2a(y)))
        b = self.relu(self.bn2b(self.conv2b(y)))
        y = self.bn3(self.conv3(torch.cat((a, b), dim=1)))
        y = self.relu(y + (x if self.downsample is None else self.downsample(x)))
        return F.dropout(y, p=0.5, training=True) if self.dropout else y


class Encoder(nn.Module):
    """ResNet-50 with multi-scale units swapped in (adapnet.py:87-149); returns (x, skip2, skip1)."""

    def __init__(self):
        super().__init__()
        self.enc_skip2_conv = nn.Conv2d(256, 24, 1)
        self.enc_skip2_conv_bn = nn.BatchNorm2d(24)
        self.enc_skip1_conv = nn.Conv2d(512, 24, 1)
        self.enc_skip1_conv_bn = nn.BatchNorm2d(24)
        nn.init.kaiming_uniform_(self.enc_skip2_conv.weight, nonlinearity='relu')
        nn.init.kaiming_uniform_(self.enc_skip1_conv.weight, nonlinearity='relu')
        r = self.res_n50_enc = ResNet50()
        r.layer2[-1] = BottleneckSSMA(512, 128, 1, 2, 64, copy_from=r.layer2[-1])
        for i, rate in enumerate((2, 16, 8, 4)):
            r.layer3[i + 2] = BottleneckSSMA(1024, 256, 1, rate, 256, copy_from=r.layer3[i + 2], drop_out=(i == 0))
        for i, rate in enumerate((4, 8, 16)):
            down = None
            if i == 0:
                down = r.layer4[0].downsample
                down[0].stride = (1, 1)  # layer4 keeps stride 16 (adapnet.py:124-126)
            r.layer4[i] = BottleneckSSMA(2048, 512, 2, rate, 512, downsample=down, copy_from=r.layer4[i])

    def forward(self, x):
        r = self.res_n50_enc
        x = r.maxpool(r.relu(r.bn1(r.conv1(x))))
        x = r.layer1(x)
        s2 = self.enc_skip2_conv_bn(self.enc_skip2_conv(x))
        x = r.layer2(x)
        s1 = self.enc_skip1_conv_bn(self.enc_skip1_conv(x))
        return r.layer4(r.layer3(x)), s2, s1


class eASPP(nn.Module):
    """Efficient atrous spatial pyramid pooling (adapnet.py:152-216), rates 3 / 6 / 12."""

    def __init__(self, in_chs, mid_chs, out_chs):
        super().__init__()
        self.branch1_conv = nn.Conv2d(in_chs, out_chs, 1)
        self.branch1_bn = nn.BatchNorm2d(out_chs)

        def cascade(rate):
            return nn.Sequential(
                nn.Conv2d(in_chs, mid_chs, 1), nn.BatchNorm2d(mid_chs), nn.ReLU(),
                nn.Conv2d(mid_chs, mid_chs, 3, dilation=rate, padding=rate), nn.BatchNorm2d(mid_chs), nn.ReLU(),
                nn.Conv2d(mid_chs, mid_chs, 3, dilation=rate, padding=rate), nn.BatchNorm2d(mid_chs), nn.ReLU(),
                nn.Conv2d(mid_chs, out_chs, 1), nn.BatchNorm2d(out_chs), nn.ReLU())
        self.branch234 = nn.ModuleList([cascade(r) for r in (3, 6, 12)])
        self.branch5_conv = nn.Conv2d(in_chs, out_chs, 1)
        self.branch5_bn = nn.BatchNorm2d(out_chs)  # present in checkpoints, unused (adapnet.py:209-210)
        self.eASPP_fin_conv = nn.Conv2d(out_chs * 5, out_chs, 1)
        self.eASPP_fin_bn = nn.BatchNorm2d(out_chs)

    def forward(self, x):
        outs = [torch.relu(self.branch1_bn(self.branch1_conv(x)))] + [b(x) for b in self.branch234]
        pooled = torch.relu(self.branch5_conv(F.adaptive_avg_pool2d(x, (1, 1))))
        outs.append(F.interpolate(pooled, size=outs[0].shape[2:], mode='bilinear', align_corners=True))
        return torch.relu(self.eASPP_fin_bn(self.eASPP_fin_conv(torch.cat(outs, 1))))


class Decoder(nn.Module):
    """Three-stage decoder with skip fusion and two auxiliary heads (adapnet.py:219-317)."""

    def __init__(self, C, fusion=False):
        super().__init__()
        self.n_classes, self.fusion = C, fusion
        self.deconv1 = nn.ConvTranspose2d(256, 256, 4, stride=2, padding=1)
        self.deconv1_bn = nn.BatchNorm2d(256)
        self.stage2 = nn.Sequential(
            nn.Conv2d(280, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU(),
            nn.Conv2d(256, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU(),
            nn.ConvTranspose2d(256, 256, 4, stride=2, padding=1), nn.BatchNorm2d(256))
        self.stage3 = nn.Sequential(
            nn.Conv2d(280, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU(),
            nn.Conv2d(256, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU(),
            nn.Conv2d(256, C, 1), nn.BatchNorm2d(C),
            nn.ConvTranspose2d(C, C, 8, stride=4, padding=2), nn.BatchNorm2d(C))
        self.aux_conv1 = nn.Conv2d(256, C, 1)
        self.aux_conv1_bn = nn.BatchNorm2d(C)
        self.aux_conv2 = nn.Conv2d(256, C, 1)
        self.aux_conv2_bn = nn.BatchNorm2d(C)
        self.fuse_conv1 = nn.Conv2d(256, 24, 1)
        self.fuse_conv1_bn = nn.BatchNorm2d(24)  # unused like in the reference (adapnet.py:315-316)
        self.fuse_conv2 = nn.Conv2d(256, 24, 1)
        self.fuse_conv2_bn = nn.BatchNorm2d(24)

    @staticmethod
    def _aux(x, conv, bn, scale):
        return F.interpolate(bn(conv(x)), scale_factor=scale, mode='bilinear', align_corners=True)

    def _skip(self, x, skip, conv):
        if not self.fusion:
            return skip
        return torch.relu(conv(F.adaptive_avg_pool2d(x, (1, 1)))) * skip

    def forward(self, x, skip1, skip2):
        x = torch.relu(self.deconv1_bn(self.deconv1(x)))
        y1 = self._aux(x, self.aux_conv1, self.aux_conv1_bn, 8)
        x = self.stage2(torch.cat((x, self._skip(x, skip1, self.fuse_conv1)), 1))
        y2 = self._aux(x, self.aux_conv2, self.aux_conv2_bn, 4)
        y3 = self.stage3(torch.cat((x, self._skip(x, skip2, self.fuse_conv2)), 1))
        return y1, y2, y3


class SSMA(nn.Module):
    """Self-supervised model adaptation fusion of two modalities (adapnet.py:320-354)."""

    def __init__(self, features, bottleneck):
        super().__init__()
        red, dbl = int(features / bottleneck), int(2 * features)
        self.link = nn.Sequential(nn.Conv2d(dbl, red, 3, padding=1), nn.ReLU(), nn.Conv2d(red, dbl, 3, padding=1), nn.Sigmoid())
        self.final_conv = nn.Sequential(nn.Conv2d(dbl, features, 3, padding=1), nn.BatchNorm2d(features))

    def forward(self, x1, x2):
        x = torch.cat((x1, x2), dim=1)
        return self.final_conv(x * self.link(x))


class AdapNet(nn.Module):
    """adapnet.py:356-415: stage 1 = one modality; otherwise two encoders fused by SSMA blocks.
    ``forward`` returns [res, aux1, aux2] like the reference."""

    def __init__(self, config):
        super().__init__()
        self.stage = config.stage
        self.n_classes = config.n_classes
        self.fusion = self.stage != 1
        if not self.fusion:
            self.encoder_mod1 = Encoder()
            self.eASPP = eASPP(2048, 64, 256)
        else:
            self.encoder_mod1 = Encoder()
            self.encoder_mod2 = Encoder()
            self.eASPP_mod1 = eASPP(2048, 64, 256)
            self.eASPP_mod2 = eASPP(2048, 64, 256)
            self.ssma_res = SSMA(256, 16)
            self.ssma_s1 = SSMA(24, 6)
            self.ssma_s2 = SSMA(24, 6)
        self.decoder = Decoder(self.n_classes, self.fusion)

    def no_resn50_dropout(self):
        self.encoder_mod1.res_n50_enc.layer3[2].dropout = False
        if self.fusion:
            self.encoder_mod2.res_n50_enc.layer3[2].dropout = False

    def forward(self, mod1, mod2=None):
        x, skip2, skip1 = self.encoder_mod1(mod1)
        if not self.fusion:
            x = self.eASPP(x)
        else:
            x2, m2_s2, m2_s1 = self.encoder_mod2(mod2)
            x, x2 = self.eASPP_mod1(x), self.eASPP_mod2(x2)
            skip2 = self.ssma_s2(skip2, m2_s2)
            skip1 = self.ssma_s1(skip1, m2_s1)
            x = self.ssma_res(x, x2)
        aux1, aux2, res = self.decoder(x, skip1, skip2)
        return [res, aux1, aux2]
