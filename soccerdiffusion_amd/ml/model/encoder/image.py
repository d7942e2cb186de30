"""Image context path (reference: soccer_diffusion/ml/model/encoder/image.py).

Only the enums of the config schema live here in this round; the torchvision backbones are
SURVEY.md §8(f2) "next" scope (third-party ResNet/Swin weights, parity unpinned)."""

from __future__ import annotations

from enum import Enum


class ImageEncoderType(Enum):
    RESNET18 = "resnet18"
    RESNET50 = "resnet50"
    SWIN_TRANSFORMER_TINY = "swin_transformer_tiny"
    SWIN_TRANSFORMER_SMALL = "swin_transformer_small"


class SequenceEncoderType(Enum):
    TRANSFORMER = "transformer"
    NONE = "none"


def image_sequence_encoder_factory(*args, **kwargs):
    raise NotImplementedError(
        "use_images=True is not available yet: the ResNet/Swin image backbone is outside this round's hot-path "
        "scope (SURVEY.md §8 f2); build the model with use_images=False")
