"""Frozen ImageNet classifiers for the ADiL CLIs and benchmarks, in plain torch.

torchvision is not available in this image and pretrained weights are a network fetch, so the architectures the
reference takes from torchvision (demo_dL_attack.py:41-53: resnet18, densenet121, googlenet, inception_v3, mobilenet_v2,
vgg11 — all six) and
the ones BASELINE.json names (ResNet-50, DenseNet-121, ViT-B/16) are defined here with torchvision-compatible
parameter names: a torchvision state_dict on local disk loads with `weights=path`.  Without weights the networks
are randomly initialised from a seed (synthetic throughput / plumbing runs).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


class Normalize(nn.Module):
    """(x - mean) / std per channel, the wrapper both reference CLIs put in front of the network
    (demo_dL_attack.py:16-25, main.py:15-24)."""

    def __init__(self, mean, std):
        super().__init__()
        self.register_buffer('mean', torch.tensor(mean, dtype=torch.float32))
        self.register_buffer('std', torch.tensor(std, dtype=torch.float32))

    def forward(self, input):
        mean = self.mean.reshape(1, 3, 1, 1).to(input.dtype)
        std = self.std.reshape(1, 3, 1, 1).to(input.dtype)
        return (input - mean) / std


# ----------------------------------------------------------------------------- ResNet
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inp, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inp, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make(self, block, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        blocks = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


# ----------------------------------------------------------------------------- DenseNet-121
class _DenseLayer(nn.Module):
    def __init__(self, inp, growth=32, bn_size=4):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(inp)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(inp, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, 1, 1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1)
        x = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(x)))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, n, inp, growth=32):
        super().__init__()
        for i in range(n):
            self.add_module(f'denselayer{i + 1}', _DenseLayer(inp + i * growth, growth))

    def forward(self, x):
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class DenseNet(nn.Module):
    def __init__(self, block_config=(6, 12, 24, 16), growth=32, init_features=64, num_classes=1000):
        super().__init__()
        feats = OrderedDict([('conv0', nn.Conv2d(3, init_features, 7, 2, 3, bias=False)),
                             ('norm0', nn.BatchNorm2d(init_features)), ('relu0', nn.ReLU(inplace=True)),
                             ('pool0', nn.MaxPool2d(3, 2, 1))])
        nf = init_features
        for i, n in enumerate(block_config):
            feats[f'denseblock{i + 1}'] = _DenseBlock(n, nf, growth)
            nf += n * growth
            if i != len(block_config) - 1:
                feats[f'transition{i + 1}'] = nn.Sequential(OrderedDict([
                    ('norm', nn.BatchNorm2d(nf)), ('relu', nn.ReLU(inplace=True)),
                    ('conv', nn.Conv2d(nf, nf // 2, 1, bias=False)), ('pool', nn.AvgPool2d(2, 2))]))
                nf //= 2
        feats['norm5'] = nn.BatchNorm2d(nf)
        self.features = nn.Sequential(feats)
        self.classifier = nn.Linear(nf, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        x = F.relu(self.features(x), inplace=True)
        return self.classifier(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


# ----------------------------------------------------------------------------- ViT-B/16
class _MLPBlock(nn.Sequential):
    def __init__(self, dim, hidden):
        super().__init__(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0), nn.Linear(hidden, dim), nn.Dropout(0.0))


class _EncoderBlock(nn.Module):
    def __init__(self, heads, dim, mlp_dim):
        super().__init__()
        self.heads = heads
        self.ln_1 = nn.LayerNorm(dim, eps=1e-6)
        self.self_attention = nn.MultiheadAttention(dim, heads, batch_first=True)   # parameter container (torchvision names)
        self.ln_2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _MLPBlock(dim, mlp_dim)

    def _attention(self, y):
        """Plain-matmul self attention with the MultiheadAttention parameters.  The fused SDPA / native-MHA kernels of
        this PyTorch-ROCm build raised a GPU memory access fault on ViT-B/16 (197 tokens, bf16), so they are not used."""
        att = self.self_attention
        b, t, dim = y.shape
        hd = dim // self.heads
        qkv = F.linear(y, att.in_proj_weight, att.in_proj_bias).reshape(b, t, 3, self.heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]                                   # (b, heads, t, hd)
        w = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        out = (w @ v).permute(0, 2, 1, 3).reshape(b, t, dim)
        return F.linear(out, att.out_proj.weight, att.out_proj.bias)

    def forward(self, x):
        x = x + self._attention(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class _Encoder(nn.Module):
    def __init__(self, seq, layers, heads, dim, mlp_dim):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.empty(1, seq, dim).normal_(std=0.02))
        self.layers = nn.Sequential(OrderedDict((f'encoder_layer_{i}', _EncoderBlock(heads, dim, mlp_dim))
                                                for i in range(layers)))
        self.ln = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, x):
        return self.ln(self.layers(x + self.pos_embedding))


class VisionTransformer(nn.Module):
    def __init__(self, image_size=224, patch=16, layers=12, heads=12, dim=768, mlp_dim=3072, num_classes=1000):
        super().__init__()
        self.patch, self.dim = patch, dim
        self.conv_proj = nn.Conv2d(3, dim, patch, patch)
        self.class_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.encoder = _Encoder((image_size // patch) ** 2 + 1, layers, heads, dim, mlp_dim)
        self.heads = nn.Sequential(OrderedDict([('head', nn.Linear(dim, num_classes))]))

    def _patch_embed(self, x):
        """conv_proj (kernel = stride = patch) as the GEMM it is: non-overlapping patches unfolded to rows, times the
        (dim, 3*patch*patch) view of the convolution weight.  Same parameters, same function; but forward and the INPUT
        gradient the attack needs are then plain library GEMMs, where the convolution's backward-data (3 output channels,
        16x16 / stride 16) sends MIOpen into a run-time kernel search + compilation: measured 72 s for the first backward
        at 256 images and > 120 s at 512 on a fresh MI355X box, which is what read as a "hang" in round 1."""
        n, c, hh, ww = x.shape
        p = self.patch
        gh, gw = hh // p, ww // p
        rows = x.reshape(n, c, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(n, gh * gw, c * p * p)
        return F.linear(rows, self.conv_proj.weight.reshape(self.dim, c * p * p), self.conv_proj.bias)

    def forward(self, x):
        n = x.shape[0]
        x = self._patch_embed(x)
        x = torch.cat([self.class_token.expand(n, -1, -1), x], dim=1)
        return self.heads(self.encoder(x)[:, 0])


# ----------------------------------------------------------------------------- MobileNetV2 / VGG11
class _ConvBNReLU6(nn.Sequential):
    def __init__(self, inp, out, k=3, stride=1, groups=1):
        super().__init__(nn.Conv2d(inp, out, k, stride, (k - 1) // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(out), nn.ReLU6(inplace=True))


class _InvertedResidual(nn.Module):
    def __init__(self, inp, out, stride, expand):
        super().__init__()
        hidden = int(round(inp * expand))
        self.use_res = stride == 1 and inp == out
        layers = []
        if expand != 1:
            layers.append(_ConvBNReLU6(inp, hidden, 1))
        layers += [_ConvBNReLU6(hidden, hidden, 3, stride, hidden), nn.Conv2d(hidden, out, 1, bias=False),
                   nn.BatchNorm2d(out)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res else self.conv(x)


class MobileNetV2(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        cfg = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
        feats = [_ConvBNReLU6(3, 32, 3, 2)]
        inp = 32
        for t, c, n, s in cfg:
            for i in range(n):
                feats.append(_InvertedResidual(inp, c, s if i == 0 else 1, t))
                inp = c
        feats.append(_ConvBNReLU6(inp, 1280, 1))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))

    def forward(self, x):
        x = F.adaptive_avg_pool2d(self.features(x), 1)
        return self.classifier(torch.flatten(x, 1))


class VGG11(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        layers, inp = [], 3
        for v in (64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'):
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(inp, v, 3, padding=1), nn.ReLU(inplace=True)]
                inp = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(7)
        self.classifier = nn.Sequential(nn.Linear(512 * 49, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096),
                                        nn.ReLU(True), nn.Dropout(), nn.Linear(4096, num_classes))

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


# ----------------------------------------------------------------------------- GoogLeNet / Inception-v3
class _ConvBN(nn.Module):
    """conv (no bias) -> BatchNorm(eps 1e-3) -> ReLU; parameter names `conv.*` / `bn.*` as in torchvision's BasicConv2d."""

    def __init__(self, inp, out, **kw):
        super().__init__()
        self.conv = nn.Conv2d(inp, out, bias=False, **kw)
        self.bn = nn.BatchNorm2d(out, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


def _seeded_init(net):
    """Variance-preserving random weights for the no-checkpoint runs (eval-mode BatchNorm with fresh statistics is the
    identity, so the published truncated-normal(0.1) initialisation would overflow bf16 after a few blocks)."""
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')


def _to_half_range(x):
    """The input re-scaling torchvision applies to its pretrained GoogLeNet / Inception-v3 (`transform_input=True`): the
    networks were trained on (x - 0.5) / 0.5, so ImageNet-normalised input is mapped back channel by channel."""
    c0 = x[:, 0:1] * (0.229 / 0.5) + (0.485 - 0.5) / 0.5
    c1 = x[:, 1:2] * (0.224 / 0.5) + (0.456 - 0.5) / 0.5
    c2 = x[:, 2:3] * (0.225 / 0.5) + (0.406 - 0.5) / 0.5
    return torch.cat((c0, c1, c2), 1)


class _GoogLeNetBlock(nn.Module):
    def __init__(self, inp, c1, c3r, c3, c5r, c5, pool):
        super().__init__()
        self.branch1 = _ConvBN(inp, c1, kernel_size=1)
        self.branch2 = nn.Sequential(_ConvBN(inp, c3r, kernel_size=1), _ConvBN(c3r, c3, kernel_size=3, padding=1))
        # a 3x3 convolution here too: torchvision's published weights were trained with it in place of the paper's 5x5
        self.branch3 = nn.Sequential(_ConvBN(inp, c5r, kernel_size=1), _ConvBN(c5r, c5, kernel_size=3, padding=1))
        self.branch4 = nn.Sequential(nn.MaxPool2d(3, stride=1, padding=1, ceil_mode=True), _ConvBN(inp, pool, kernel_size=1))

    def forward(self, x):
        return torch.cat([self.branch1(x), self.branch2(x), self.branch3(x), self.branch4(x)], 1)


class GoogLeNet(nn.Module):
    """GoogLeNet (Szegedy et al. 2015) with torchvision's module names; inference graph only (the two auxiliary heads
    exist for training and are dropped by torchvision's pretrained constructor as well)."""

    def __init__(self, num_classes=1000, transform_input=True):
        super().__init__()
        self.transform_input = transform_input
        self.conv1 = _ConvBN(3, 64, kernel_size=7, stride=2, padding=3)
        self.maxpool1 = nn.MaxPool2d(3, stride=2, ceil_mode=True)
        self.conv2 = _ConvBN(64, 64, kernel_size=1)
        self.conv3 = _ConvBN(64, 192, kernel_size=3, padding=1)
        self.maxpool2 = nn.MaxPool2d(3, stride=2, ceil_mode=True)
        self.inception3a = _GoogLeNetBlock(192, 64, 96, 128, 16, 32, 32)
        self.inception3b = _GoogLeNetBlock(256, 128, 128, 192, 32, 96, 64)
        self.maxpool3 = nn.MaxPool2d(3, stride=2, ceil_mode=True)
        self.inception4a = _GoogLeNetBlock(480, 192, 96, 208, 16, 48, 64)
        self.inception4b = _GoogLeNetBlock(512, 160, 112, 224, 24, 64, 64)
        self.inception4c = _GoogLeNetBlock(512, 128, 128, 256, 24, 64, 64)
        self.inception4d = _GoogLeNetBlock(512, 112, 144, 288, 32, 64, 64)
        self.inception4e = _GoogLeNetBlock(528, 256, 160, 320, 32, 128, 128)
        self.maxpool4 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        self.inception5a = _GoogLeNetBlock(832, 256, 160, 320, 32, 128, 128)
        self.inception5b = _GoogLeNetBlock(832, 384, 192, 384, 48, 128, 128)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.dropout = nn.Dropout(0.2)
        self.fc = nn.Linear(1024, num_classes)
        _seeded_init(self)

    def forward(self, x):
        if self.transform_input:
            x = _to_half_range(x)
        x = self.maxpool2(self.conv3(self.conv2(self.maxpool1(self.conv1(x)))))
        x = self.maxpool3(self.inception3b(self.inception3a(x)))
        x = self.inception4e(self.inception4d(self.inception4c(self.inception4b(self.inception4a(x)))))
        x = self.inception5b(self.inception5a(self.maxpool4(x)))
        return self.fc(self.dropout(torch.flatten(self.avgpool(x), 1)))


class _InceptionA(nn.Module):
    def __init__(self, inp, pool_features):
        super().__init__()
        self.branch1x1 = _ConvBN(inp, 64, kernel_size=1)
        self.branch5x5_1 = _ConvBN(inp, 48, kernel_size=1)
        self.branch5x5_2 = _ConvBN(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = _ConvBN(inp, 64, kernel_size=1)
        self.branch3x3dbl_2 = _ConvBN(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = _ConvBN(96, 96, kernel_size=3, padding=1)
        self.branch_pool = _ConvBN(inp, pool_features, kernel_size=1)

    def forward(self, x):
        return torch.cat([self.branch1x1(x), self.branch5x5_2(self.branch5x5_1(x)),
                          self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                          self.branch_pool(F.avg_pool2d(x, 3, stride=1, padding=1))], 1)


class _InceptionB(nn.Module):
    def __init__(self, inp):
        super().__init__()
        self.branch3x3 = _ConvBN(inp, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = _ConvBN(inp, 64, kernel_size=1)
        self.branch3x3dbl_2 = _ConvBN(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = _ConvBN(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        return torch.cat([self.branch3x3(x), self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                          F.max_pool2d(x, 3, stride=2)], 1)


class _InceptionC(nn.Module):
    def __init__(self, inp, c7):
        super().__init__()
        self.branch1x1 = _ConvBN(inp, 192, kernel_size=1)
        self.branch7x7_1 = _ConvBN(inp, c7, kernel_size=1)
        self.branch7x7_2 = _ConvBN(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = _ConvBN(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = _ConvBN(inp, c7, kernel_size=1)
        self.branch7x7dbl_2 = _ConvBN(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = _ConvBN(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = _ConvBN(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = _ConvBN(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = _ConvBN(inp, 192, kernel_size=1)

    def forward(self, x):
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))))
        return torch.cat([self.branch1x1(x), b7, bd, self.branch_pool(F.avg_pool2d(x, 3, stride=1, padding=1))], 1)


class _InceptionD(nn.Module):
    def __init__(self, inp):
        super().__init__()
        self.branch3x3_1 = _ConvBN(inp, 192, kernel_size=1)
        self.branch3x3_2 = _ConvBN(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = _ConvBN(inp, 192, kernel_size=1)
        self.branch7x7x3_2 = _ConvBN(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = _ConvBN(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = _ConvBN(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        return torch.cat([self.branch3x3_2(self.branch3x3_1(x)), b7, F.max_pool2d(x, 3, stride=2)], 1)


class _InceptionE(nn.Module):
    def __init__(self, inp):
        super().__init__()
        self.branch1x1 = _ConvBN(inp, 320, kernel_size=1)
        self.branch3x3_1 = _ConvBN(inp, 384, kernel_size=1)
        self.branch3x3_2a = _ConvBN(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = _ConvBN(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = _ConvBN(inp, 448, kernel_size=1)
        self.branch3x3dbl_2 = _ConvBN(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = _ConvBN(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = _ConvBN(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = _ConvBN(inp, 192, kernel_size=1)

    def forward(self, x):
        b3 = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(b3), self.branch3x3_2b(b3)], 1)
        bd = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(bd), self.branch3x3dbl_3b(bd)], 1)
        return torch.cat([self.branch1x1(x), b3, bd, self.branch_pool(F.avg_pool2d(x, 3, stride=1, padding=1))], 1)


class InceptionV3(nn.Module):
    """Inception-v3 (Szegedy et al. 2016) with torchvision's module names; inference graph only (no AuxLogits).  Fully
    convolutional up to the adaptive pool, so the reference's 224x224 crops run (DS_ImageNet.py:14-18) as well as 299."""

    def __init__(self, num_classes=1000, transform_input=True):
        super().__init__()
        self.transform_input = transform_input
        self.Conv2d_1a_3x3 = _ConvBN(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = _ConvBN(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = _ConvBN(32, 64, kernel_size=3, padding=1)
        self.maxpool1 = nn.MaxPool2d(kernel_size=3, stride=2)
        self.Conv2d_3b_1x1 = _ConvBN(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = _ConvBN(80, 192, kernel_size=3)
        self.maxpool2 = nn.MaxPool2d(kernel_size=3, stride=2)
        self.Mixed_5b = _InceptionA(192, 32)
        self.Mixed_5c = _InceptionA(256, 64)
        self.Mixed_5d = _InceptionA(288, 64)
        self.Mixed_6a = _InceptionB(288)
        self.Mixed_6b = _InceptionC(768, 128)
        self.Mixed_6c = _InceptionC(768, 160)
        self.Mixed_6d = _InceptionC(768, 160)
        self.Mixed_6e = _InceptionC(768, 192)
        self.Mixed_7a = _InceptionD(768)
        self.Mixed_7b = _InceptionE(1280)
        self.Mixed_7c = _InceptionE(2048)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.dropout = nn.Dropout(0.5)
        self.fc = nn.Linear(2048, num_classes)
        _seeded_init(self)

    def forward(self, x):
        if self.transform_input:
            x = _to_half_range(x)
        x = self.maxpool1(self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x))))
        x = self.maxpool2(self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x)))
        x = self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(x)))
        x = self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(self.Mixed_6a(x)))))
        x = self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x)))
        return self.fc(self.dropout(torch.flatten(self.avgpool(x), 1)))


# auxiliary training heads present in torchvision checkpoints; the inference graphs above do not have them
_TRAINING_ONLY_PREFIXES = ('aux1.', 'aux2.', 'AuxLogits.')


# ----------------------------------------------------------------------------- registry
_BUILDERS = {
    'resnet18': lambda nc: ResNet(BasicBlock, [2, 2, 2, 2], nc),
    'resnet50': lambda nc: ResNet(Bottleneck, [3, 4, 6, 3], nc),
    'densenet121': lambda nc: DenseNet(num_classes=nc),
    'vit_b_16': lambda nc: VisionTransformer(num_classes=nc),
    'mobilenet_v2': lambda nc: MobileNetV2(nc),
    'vgg11': lambda nc: VGG11(nc),
    'googlenet': lambda nc: GoogLeNet(nc),
    'inception_v3': lambda nc: InceptionV3(nc),
}
# names accepted by the reference CLIs (demo_dL_attack.py:41-53) and the BASELINE.json config names
ALIASES = {'resnet': 'resnet18', 'densenet': 'densenet121', 'mobilenet': 'mobilenet_v2', 'vgg': 'vgg11',
           'vit': 'vit_b_16', 'vit-b/16': 'vit_b_16', 'resnet-50': 'resnet50', 'densenet-121': 'densenet121',
           'inception': 'inception_v3'}


def canonical_name(name: str) -> str:
    key = name.lower()
    key = ALIASES.get(key, key)
    if key not in _BUILDERS:
        raise ValueError(f"unknown model {name!r}; available: {sorted(_BUILDERS) + sorted(ALIASES)}")
    return key


# ----------------------------------------------------------------------------- frozen-classifier rewrites
def fold_batchnorm_(module: nn.Module) -> nn.Module:
    """Fold every eval-mode BatchNorm2d that directly follows a Conv2d into that convolution (exact in real
    arithmetic: w' = w * gamma/sqrt(var+eps), b' = beta + (b - mean) * gamma/sqrt(var+eps)) and replace the BN by
    Identity.  Valid for the networks of this file, whose forward applies a conv and the BatchNorm registered right
    after it back to back.  The classifier is frozen, so this removes one elementwise pass over every activation
    in the forward AND in the input-gradient backward (measured on MI355X, ResNet-50 B=512 bf16: 116 -> 66 ms)."""
    for child in module.children():
        fold_batchnorm_(child)
    names = list(module._modules.keys())
    for a, b in zip(names, names[1:]):
        conv, bn = module._modules[a], module._modules[b]
        if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d)) or bn.training:
            continue
        with torch.no_grad():
            scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            bias = bn.bias.double() - bn.running_mean.double() * scale
            if conv.bias is not None:
                bias = bias + conv.bias.double() * scale
            fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                              conv.dilation, conv.groups, bias=True, padding_mode=conv.padding_mode)
            fused.weight.copy_((conv.weight.double() * scale.view(-1, 1, 1, 1)).float())
            fused.bias.copy_(bias.float())
        fused = fused.to(device=conv.weight.device, dtype=conv.weight.dtype)
        for p in fused.parameters():
            p.requires_grad_(False)
        module._modules[a] = fused.eval()
        module._modules[b] = nn.Identity()
    return module


class ChannelPaddedConv(nn.Module):
    """First convolution with its 3 input channels zero-padded to `width`: MIOpen's backward-data kernels for a
    3-channel input are pathologically slow (measured: 7x7/2 conv1 of ResNet-50, B=512 bf16 NHWC, input gradient
    22.2 ms at C_in=3 vs 4.7 ms at C_in=8).  The extra channels carry zero weights and zero inputs."""

    def __init__(self, conv: nn.Conv2d, width: int = 8):
        super().__init__()
        self.in_channels, self.width = conv.in_channels, width
        wide = nn.Conv2d(width, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation,
                         conv.groups, bias=conv.bias is not None)
        with torch.no_grad():
            wide.weight.zero_()
            wide.weight[:, :conv.in_channels].copy_(conv.weight)
            if conv.bias is not None:
                wide.bias.copy_(conv.bias)
        self.conv = wide.to(device=conv.weight.device, dtype=conv.weight.dtype)
        for p in self.conv.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        fmt = torch.channels_last if self.conv.weight.is_contiguous(memory_format=torch.channels_last) \
            and not self.conv.weight.is_contiguous() else torch.contiguous_format
        xp = torch.empty((x.shape[0], self.width, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device,
                         memory_format=fmt)
        xp[:, self.in_channels:].zero_()
        xp[:, :self.in_channels] = x
        return self.conv(xp)


def pad_first_conv_(net: nn.Module, width: int = 8) -> nn.Module:
    """Wrap the network's first 3-channel Conv2d (groups == 1) in a ChannelPaddedConv."""
    for parent in net.modules():
        for name, child in parent._modules.items():
            if isinstance(child, nn.Conv2d) and child.in_channels == 3 and child.groups == 1:
                parent._modules[name] = ChannelPaddedConv(child, width)
                return net
    return net


def _bn_affine(bn: nn.BatchNorm2d):
    """Eval-mode BatchNorm as a per-channel affine map: y = x * scale + shift (fp32 buffers)."""
    with torch.no_grad():
        scale = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)).float()
        shift = (bn.bias.double() - bn.running_mean.double() * scale.double()).float()
    return scale.contiguous(), shift.contiguous()


class _Fp32Tables(nn.Module):
    """The per-channel epilogue tables `scale` / `shift` are derived in fp64 and kept in fp32 whatever dtype the
    network is cast to: `.to(torch.bfloat16)` would otherwise round them to 8 mantissa bits (and a later `.float()`
    cannot bring the bits back).  They follow device moves only."""

    def _apply(self, fn, recurse=True):
        keep = {n: getattr(self, n) for n in ('scale', 'shift')}
        super()._apply(fn, recurse)
        for n, t in keep.items():
            setattr(self, n, t.to(device=fn(t).device))
        return self


class _ConvAffine(_Fp32Tables):
    """conv (weights untouched) followed by the fused eval-BatchNorm [+ residual] [+ ReLU] epilogue kernel."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, relu: bool):
        super().__init__()
        self.conv, self.relu = conv, relu
        scale, shift = _bn_affine(bn)
        self.register_buffer('scale', scale)
        self.register_buffer('shift', shift)
        # a 1x1 / stride-1 convolution of a channels_last tensor IS a row-major GEMM (B*H*W x Cin) @ (Cin x Cout)
        one_by_one = (conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None)
        self.pointwise = one_by_one and conv.stride == (1, 1)
        self.pointwise_s2 = one_by_one and conv.stride == (2, 2)         # the ResNet downsample convolutions
        if one_by_one:          # (Cin, Cout) copy for the input-gradient kernel; 2-D, so channels_last leaves it alone
            self.register_buffer('wt2d', conv.weight.detach().reshape(conv.out_channels, conv.in_channels).t().contiguous())
        self.dense3x3 = (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
                         and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
                         and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0)
        if self.dense3x3:
            from . import ops
            wp_fwd, wp_bwd = ops.pack_conv3x3_weights(conv.weight)
            self.register_buffer('wp_fwd', wp_fwd)
            self.register_buffer('wp_bwd', wp_bwd)

    def raw_conv(self, x):
        """The convolution alone (no BatchNorm / ReLU): the hand-written 3x3 kernel where it applies, else the library."""
        if self.dense3x3 and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] <= 63:   # kernel limit: W <= 63
            from . import ops
            return ops.conv3x3(x, self.wp_fwd, self.wp_bwd)
        return self.conv(x)

    def _conv(self, x):
        if self.pointwise and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last):
            # hipBLASLt instead of MIOpen: no zero-fill launch in front of the kernel (MIOpen's implicit-GEMM solvers
            # clear their output first: 6 ms of a 55 ms step at B=512) and a faster kernel on most ResNet-50 shapes
            b, cin, h, w = x.shape
            wt = self.conv.weight.reshape(self.conv.out_channels, cin).t()            # (Cin, Cout) view
            y2 = torch.mm(x.permute(0, 2, 3, 1).reshape(b * h * w, cin), wt)
            return y2.reshape(b, h, w, -1).permute(0, 3, 1, 2)                         # channels_last storage
        return self.raw_conv(x)

    def _fusable(self, x) -> bool:
        return (x.is_cuda and x.dtype == torch.bfloat16 and self.conv.in_channels % 64 == 0
                and self.conv.out_channels % 64 == 0 and self.conv.out_channels <= 2048)

    def fused_pointwise(self, x) -> bool:
        return self.pointwise and self._fusable(x)

    def fused_pointwise_s2(self, x) -> bool:
        return self.pointwise_s2 and self._fusable(x)

    def forward(self, x, res=None, outputs=1, pre=None, strided_view=False):
        """outputs = 2 / 3: the result twice (+ its [:, :, ::2, ::2] view); pre=(scale, shift) of the previous layer
        means x is that layer's RAW convolution output; strided_view: x is a [:, :, ::2, ::2] view for a stride-2 1x1
        convolution (see ops.PointwiseConvFunction).  All only with the fused kernels."""
        from . import ops
        if self.fused_pointwise(x) or (strided_view and self.fused_pointwise_s2(x)):
            w2d = self.conv.weight.reshape(self.conv.out_channels, self.conv.in_channels)
            if w2d.is_contiguous():                                      # (Cout, Cin): true for either memory format
                return ops.pointwise_conv_affine(x, w2d, self.wt2d, self.scale, self.shift, res=res, relu=self.relu,
                                                 outputs=outputs, pre=pre, stride2=strided_view)
        if pre is not None or strided_view:
            raise RuntimeError("pre= / strided_view need the fused pointwise kernels")
        y = ops.affine_act(self._conv(x), self.scale.float(), self.shift.float(), res=res, relu=self.relu)
        return y if outputs == 1 else ((y, y) if outputs == 2 else (y, y, y[:, :, ::2, ::2]))


class _FusedResBlock(nn.Module):
    def __init__(self, block):
        super().__init__()
        self.bottleneck = isinstance(block, Bottleneck)
        self.c1 = _ConvAffine(block.conv1, block.bn1, True)
        if self.bottleneck:
            self.c2 = _ConvAffine(block.conv2, block.bn2, True)
            self.c3 = _ConvAffine(block.conv3, block.bn3, True)          # + residual, then ReLU
        else:
            self.c2 = _ConvAffine(block.conv2, block.bn2, True)          # + residual, then ReLU
        self.down = None if block.downsample is None else _ConvAffine(block.downsample[0], block.downsample[1], False)
        self.emit_sub = False            # set by FusedResNet: the NEXT block has a stride-2 pointwise downsample

    def forward(self, x):
        """x is a tensor, a (main, skip) pair or a (main, skip, skip[:, :, ::2, ::2]) triple of the same activation (see
        _ConvAffine.forward): the pair keeps the two gradients of the residual join apart until the producing kernel's
        backward adds them; the third member feeds a stride-2 downsample convolution without a copy."""
        xm, xs, xsub = (x + (None,))[:3] if isinstance(x, tuple) else (x, x, None)
        if self.down is None:
            idt = xs
        elif xsub is not None and self.down.fused_pointwise_s2(xsub):
            idt = self.down(xsub, strided_view=True)
        else:
            idt = self.down(xs)
        out = self.c1(xm)
        if self.bottleneck:
            n_out = 3 if self.emit_sub else 2
            raw = self.c2.raw_conv(out)                                  # 3x3 convolution, no epilogue pass:
            if (self.c2.relu and self.c3.fused_pointwise(raw) and self.c3.conv.in_channels <= 512
                    and raw.is_contiguous(memory_format=torch.channels_last)):
                return self.c3(raw, res=idt, outputs=n_out, pre=(self.c2.scale, self.c2.shift))   # bn2+ReLU run inside c3
            from . import ops
            out = ops.affine_act(raw, self.c2.scale.float(), self.c2.shift.float(), relu=self.c2.relu)
            return self.c3(out, res=idt, outputs=n_out)
        return self.c2(out, res=idt)


class _FusedStem(_Fp32Tables):
    """Normalize -> conv1 7x7/2 -> bn1(eval) -> ReLU -> maxpool as the hand-written stem kernels (`ops.resnet_stem`):
    consumes the attack's (B,3,H,W) fp32/bf16 tensor directly and returns bf16 channels_last activations; its
    backward produces dLoss/dx in the layout `adil_grad` reads.  bf16 networks on the GPU only."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, mean, std):
        super().__init__()
        from . import ops
        if conv.bias is not None or conv.stride != (2, 2) or conv.padding != (3, 3):
            raise ValueError("stem kernels implement the torchvision ResNet conv1 (7x7, stride 2, padding 3, no bias)")
        w_fwd, w_bwd = ops.pack_stem_weights(conv.weight)
        scale, shift = _bn_affine(bn)
        self.register_buffer('w_fwd', w_fwd)
        self.register_buffer('w_bwd', w_bwd)
        self.register_buffer('scale', scale)
        self.register_buffer('shift', shift)
        self.mean = [float(m) for m in mean]
        self.inv_std = [1.0 / float(s) for s in std]

    def forward(self, x):
        from . import ops
        return ops.resnet_stem(x, self.w_fwd, self.w_bwd, self.scale, self.shift, self.mean, self.inv_std)


class _Fp32Head(nn.Module):
    """Global average pool + the last linear layer in fp32, whatever dtype the network is cast to: the weights keep their
    fp32 bits (like _Fp32Tables), the pooled features are widened before the layer, the logits come out fp32.
    Why (round 4, VERDICT r3 #1c): a bf16 head rounds logits of magnitude 16-64 to steps of 0.125-0.25, which is the scale
    on which the attack's argmax decisions ("fooled") and its margin loss are taken; the head is 2 MFLOP per image next to
    the backbone's 4 GFLOP, so keeping it fp32 costs nothing measurable.  The backbone stays bf16."""

    def __init__(self, fc: nn.Linear):
        super().__init__()
        self.register_buffer('weight', fc.weight.detach().float().clone())
        self.register_buffer('bias', fc.bias.detach().float().clone() if fc.bias is not None else torch.zeros(fc.out_features))

    def _apply(self, fn, recurse=True):
        keep = {n: getattr(self, n) for n in ('weight', 'bias')}
        super()._apply(fn, recurse)
        for n, t in keep.items():
            setattr(self, n, t.to(device=fn(t).device))
        return self

    def forward(self, x):
        return F.linear(x.float().mean(dim=(2, 3)), self.weight, self.bias)


class FusedResNet(nn.Module):
    """A frozen ResNet whose BatchNorm(eval) / residual add / ReLU run as ONE elementwise kernel per convolution
    (`ops.affine_act`, forward and input-gradient backward) instead of 2-3 separate PyTorch kernels.  Convolution
    weights are untouched; the function is the original network's (up to one rounding per activation).  GPU only.
    With `normalize=(mean, std)` the input normalisation and the whole first stage run in the stem kernels.
    `head_fp32`: global pooling + the last linear layer in fp32 with fp32 logits (_Fp32Head) — True: always;
    "inference": only while `precise_head(True)` is in force, which the DDrague inference solver switches on around its
    classifier calls (engine.precise_head): the logits' precision matters where the attack works next to the decision
    boundary — at inference on held-out images (+1.5 pp ASR over five paired dictionaries, most on the bad ones, at no
    cost) — while dictionaries LEARNED against the fp32 head came out consistently a little worse than those learned
    against the bf16 head (profiles/r04_asr_gap.md)."""

    def __init__(self, net: ResNet, normalize=None, head_fp32=False):
        super().__init__()
        if head_fp32 not in (False, True, "inference"):
            raise ValueError("head_fp32 must be False, True or 'inference'")
        self.head32 = _Fp32Head(net.fc) if head_fp32 else None
        self.head32_on = head_fp32 is True
        if normalize is not None:
            self.fstem = _FusedStem(net.conv1, net.bn1, *normalize)
            self.stem, self.maxpool = None, None
        else:
            self.fstem = None
            self.stem = _ConvAffine(net.conv1, net.bn1, True)
            self.maxpool = net.maxpool
        self.layers = nn.Sequential(*[_FusedResBlock(b) for layer in (net.layer1, net.layer2, net.layer3, net.layer4)
                                      for b in layer])
        blocks = list(self.layers)
        for cur, nxt in zip(blocks[:-1], blocks[1:]):
            cur.emit_sub = bool(cur.bottleneck and nxt.down is not None and nxt.down.pointwise_s2)
        self.avgpool, self.fc = net.avgpool, net.fc

    def forward(self, x):
        x = self.fstem(x) if self.fstem is not None else self.maxpool(self.stem(x))
        x = self.layers(x)
        if isinstance(x, tuple):
            x = x[0]
        if self.head32 is not None and self.head32_on:
            return self.head32(x)
        return self.fc(torch.flatten(self.avgpool(x), 1))

    def precise_head(self, enabled: bool) -> bool:
        """Switch the fp32 head on / off (networks built with head_fp32="inference"); returns the previous state."""
        prev = self.head32_on
        if self.head32 is not None:
            self.head32_on = bool(enabled)
        return prev


@torch.no_grad()
def fit_centroid_head(model: nn.Module, images: torch.Tensor, labels: torch.Tensor, classes: int, device,
                      target_margin: float = 10.0, chunk: int = 64):
    """Overwrite the last linear layer of `model` (Sequential(Normalize, net) from build_classifier, fp32, plain modules)
    with the nearest-centroid classifier of its own penultimate features on (images, labels):
    logit_c = s * (z . m_c - |m_c|^2 / 2), z = standardised feature, m_c = class mean; s scales the median clean margin to
    `target_margin`.  The other output classes get weight 0 and a large negative bias.  Closed form, seeded by its inputs,
    a few seconds: it gives a random-weight backbone a head that separates a structured synthetic dataset
    (imagenet_loading.SyntheticImageNet(structured=True)) with margins far above bf16 rounding, which is what the offline
    demo and the ASR-parity tests need (there are no pretrained weights on a box without network).
    Returns (clean top-2 margins, predicted labels) of the fitted network on the images."""
    net = model[-1]
    fc = net.fc if hasattr(net, "fc") else (net.classifier if hasattr(net, "classifier") else net.heads.head)
    if isinstance(fc, nn.Sequential):
        fc = fc[-1]
    if not isinstance(fc, nn.Linear):
        raise TypeError("fit_centroid_head: the network's last layer must be a Linear")
    feats = []
    hook = fc.register_forward_pre_hook(lambda m, a: feats.append(a[0].detach().double().cpu()))
    for part in images.split(chunk):
        model(part.to(device))
    hook.remove()
    f = torch.cat(feats)
    mu, sd = f.mean(0), f.std(0) + 1e-6 * f.std(0).max()
    z = (f - mu) / sd
    means = torch.stack([z[labels == c].mean(0) for c in range(classes)])           # (C, F)
    logits = z @ means.t() - 0.5 * (means * means).sum(1)
    top2 = logits.topk(2, dim=1).values
    s = target_margin / float((top2[:, 0] - top2[:, 1]).median())
    w = s * means / sd                                                                # (C, F) on raw features
    b = s * (-(means * (mu / sd)).sum(1) - 0.5 * (means * means).sum(1))
    fc.weight.zero_()
    fc.bias.fill_(-1.0e4)
    fc.weight[:classes] = w.to(fc.weight)
    fc.bias[:classes] = b.to(fc.bias)
    out = torch.cat([model(part.to(device)).double().cpu() for part in images.split(chunk)])
    top2 = out.topk(2, dim=1).values
    return top2[:, 0] - top2[:, 1], out.argmax(1)


def build_classifier(name: str, num_classes: int = 1000, seed: int = 0, weights: Optional[str] = None,
                     device=None, dtype: torch.dtype = torch.float32, channels_last: bool = False,
                     fold_bn: bool = False, pad_input_channels: int = 0, fuse_bn_act: bool = False,
                     fuse_stem: bool = False, head_fp32=False) -> nn.Module:
    """Sequential(Normalize, net), eval mode, parameters frozen — the object both CLIs hand to ADIL.
    fold_bn / pad_input_channels / fuse_bn_act / fuse_stem apply the function-preserving rewrites above (off by
    default); fuse_bn_act (ResNets, GPU only) supersedes fold_bn; fuse_stem (with fuse_bn_act, bf16 only) moves the
    normalisation and the first stage into the stem kernels (the Sequential then holds the network alone); head_fp32
    (with fuse_bn_act) keeps global pooling + the last linear layer in fp32 under a bf16 cast (fp32 logits): True = always,
    "inference" = only inside engine.precise_head (the DDrague inference solver), see FusedResNet."""
    if head_fp32 and not fuse_bn_act:
        raise ValueError("head_fp32 is a switch of the FusedResNet path (fuse_bn_act=True)")
    key = canonical_name(name)
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        net = _BUILDERS[key](num_classes)
    if weights is not None:
        state = torch.load(weights, map_location='cpu')
        net.load_state_dict({k: v for k, v in state.items() if not k.startswith(_TRAINING_ONLY_PREFIXES)})
    net.eval()
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    stem_fused = False
    if fuse_bn_act and isinstance(net, ResNet):
        stem_fused = bool(fuse_stem)
        if stem_fused and dtype != torch.bfloat16:
            raise ValueError("fuse_stem needs a bfloat16 network (the stem kernels produce bf16 activations)")
        net = FusedResNet(net, normalize=(mean, std) if stem_fused else None, head_fp32=head_fp32)
    elif fold_bn:
        fold_batchnorm_(net)
    model = nn.Sequential(net) if stem_fused else nn.Sequential(Normalize(mean=mean, std=std), net)
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    model = model.to(device=device, dtype=dtype)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    if pad_input_channels and not stem_fused and not isinstance(net, VisionTransformer):   # ViT: patch embedding is a GEMM
        pad_first_conv_(net, pad_input_channels)
    return model
