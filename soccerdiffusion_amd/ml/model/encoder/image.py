"""Image context path (reference: soccer_diffusion/ml/model/encoder/image.py:11-174).

Per frame a ResNet-18/50 turns an image into one token; a BaseEncoder with 8 heads
(image.py:116) lets the frames' tokens talk to each other.  The sequence transformer runs
on this package's HIP kernels.  The ResNet backbone is SURVEY §8 f2 "next" scope and NOT
hand-written: it is the torchvision architecture (absent offline, so restated here layer for
layer with torchvision's ``state_dict`` keys) built from ``torch.nn`` layers, i.e. MIOpen
convolutions through PyTorch-ROCm.  Pretrained ImageNet weights cannot be fetched offline
(``weights=...DEFAULT`` in the reference): load them from a reference checkpoint.
Parity of the backbone is unpinned (no torchvision here, no reference fixture).
"""

from __future__ import annotations

from enum import Enum

import torch
from torch import nn

from .encoders import BaseEncoder


class ImageEncoderType(Enum):
    RESNET18 = "resnet18"
    RESNET50 = "resnet50"
    SWIN_TRANSFORMER_TINY = "swin_transformer_tiny"
    SWIN_TRANSFORMER_SMALL = "swin_transformer_small"


class SequenceEncoderType(Enum):
    TRANSFORMER = "transformer"
    NONE = "none"


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, planes, 3, stride, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, 1, 1), nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != planes:
            self.downsample = nn.Sequential(_conv(cin, planes, 1, stride), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        return self.relu(self.bn2(self.conv2(out)) + idt)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, planes, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride, 1), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(_conv(cin, planes * 4, 1, stride), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        return self.relu(self.bn3(self.conv3(out)) + idt)


class _ResNet(nn.Module):
    """torchvision.models.resnet.ResNet, attribute for attribute (conv1, bn1, layer1-4, avgpool, fc)."""

    def __init__(self, block, layers):
        super().__init__()
        self.conv1, self.bn1 = _conv(3, 64, 7, 2, 3), nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers), start=1):
            blocks = []
            for j in range(n):
                blocks.append(block(cin, planes, (1 if i == 1 else 2) if j == 0 else 1))
                cin = planes * block.expansion
            setattr(self, f"layer{i}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class AbstractImageEncoder(nn.Module):
    encoder: nn.Module

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, F, 3, R, R) -> one token per frame (B, F, hidden_dim)."""
        tokens = self.encoder(x.reshape(-1, *x.shape[2:]))
        return tokens.view(x.shape[0], x.shape[1], -1)


class ResNetImageEncoder(AbstractImageEncoder):
    def __init__(self, resnet_type: ImageEncoderType, hidden_dim: int, use_final_avgpool: bool, resolution: int):
        super().__init__()
        if resnet_type == ImageEncoderType.RESNET18:
            self.encoder = _ResNet(_BasicBlock, (2, 2, 2, 2))
        elif resnet_type == ImageEncoderType.RESNET50:
            self.encoder = _ResNet(_Bottleneck, (3, 4, 6, 3))
        else:
            raise ValueError(f"Invalid ResNet type: {resnet_type}")
        feat = self.encoder.fc.in_features
        if use_final_avgpool:
            self.encoder.fc = nn.Linear(feat, hidden_dim)
        else:  # a 1x1 conv to 32 channels in place of the pooling, then one linear layer over the map (image.py:69-73)
            self.encoder.avgpool = nn.Conv2d(feat, 32, 1)
            self.encoder.fc = nn.Linear(self.calculate_output_size(resolution) ** 2 * 32, hidden_dim)

    @staticmethod
    def calculate_output_size(resolution: int) -> int:
        resolution = (resolution - 7 + 2 * 3) // 2 + 1   # conv1
        resolution = (resolution - 3 + 2 * 1) // 2 + 1   # maxpool
        return resolution // 2 // 2 // 2                 # layer2-4


class TransformerImageSequenceEncoder(nn.Module):
    def __init__(self, image_encoder: AbstractImageEncoder, hidden_dim: int, num_layers: int, max_seq_len: int):
        super().__init__()
        self.image_encoder = image_encoder
        self.transformer_encoder = BaseEncoder(input_dim=hidden_dim, patch_size=1, hidden_dim=hidden_dim,
                                               num_layers=num_layers, num_heads=8, max_seq_len=max_seq_len)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.transformer_encoder(self.image_encoder(x).contiguous())


def image_encoder_factory(encoder_type: ImageEncoderType, hidden_dim: int, use_final_avgpool: bool, resolution: int):
    if encoder_type in (ImageEncoderType.RESNET18, ImageEncoderType.RESNET50):
        return ResNetImageEncoder(encoder_type, hidden_dim, use_final_avgpool, resolution)
    raise NotImplementedError("the Swin image encoders are not available (torchvision is absent; SURVEY §8 f2)")


def image_sequence_encoder_factory(encoder_type: SequenceEncoderType, image_encoder_type: ImageEncoderType, hidden_dim: int,
                                   num_layers: int, max_seq_len: int, use_final_avgpool: bool, resolution: int):
    image_encoder = image_encoder_factory(image_encoder_type, hidden_dim, use_final_avgpool, resolution)
    if encoder_type == SequenceEncoderType.TRANSFORMER:
        return TransformerImageSequenceEncoder(image_encoder, hidden_dim, num_layers, max_seq_len)
    if encoder_type == SequenceEncoderType.NONE:
        return image_encoder
    raise ValueError(f"Invalid sequence encoder type: {encoder_type}")
