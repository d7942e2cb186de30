"""Parameter containers that reproduce the reference checkpoint's key names (SURVEY.md
App. C) without instantiating ``nn.Transformer*``: the weights are plain ``nn.Parameter``s
consumed by the HIP kernels through ``sd_layer_weights``."""

from __future__ import annotations

import math

import torch
from torch import nn


def _linear_init(weight: torch.Tensor, bias: torch.Tensor | None) -> None:
    # nn.Linear / nn.Conv1d default: kaiming_uniform(a=sqrt(5)) and U(+-1/sqrt(fan_in))
    nn.init.kaiming_uniform_(weight, a=math.sqrt(5))
    if bias is not None:
        fan_in = weight[0].numel()
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        nn.init.uniform_(bias, -bound, bound)


class LinearParams(nn.Module):
    """``weight`` (out, in...) and ``bias`` (out) with nn.Linear's init law."""

    def __init__(self, *weight_shape: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*weight_shape))
        self.bias = nn.Parameter(torch.empty(weight_shape[0]))
        _linear_init(self.weight, self.bias)


class NormParams(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class AttentionParams(nn.Module):
    """Keys of nn.MultiheadAttention: in_proj_weight (3d,d), in_proj_bias, out_proj.{weight,bias}."""

    def __init__(self, d: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = LinearParams(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class LayerParams(nn.Module):
    """One pre-norm layer with dim_feedforward = d; ``cross`` adds multihead_attn + norm3."""

    def __init__(self, d: int, cross: bool):
        super().__init__()
        self.self_attn = AttentionParams(d)
        if cross:
            self.multihead_attn = AttentionParams(d)
        self.linear1 = LinearParams(d, d)
        self.linear2 = LinearParams(d, d)
        self.norm1 = NormParams(d)
        self.norm2 = NormParams(d)
        if cross:
            self.norm3 = NormParams(d)


class LayerStack(nn.Module):
    """``layers.{i}.…`` like nn.TransformerDecoder / nn.TransformerEncoder."""

    def __init__(self, d: int, num_layers: int, cross: bool):
        super().__init__()
        self.layers = nn.ModuleList([LayerParams(d, cross) for _ in range(num_layers)])


class PackedWeightsMixin:
    """Caches the C-ABI weight descriptor; rebuilt only when a parameter's storage moved
    (``.to(device)``); in-place updates (optimizer steps, ``load_state_dict``) keep pointers."""

    _packed = None
    _packed_sig = None

    def _signature(self):
        return tuple(p.data_ptr() for p in self.parameters())

    def _packed_weights(self, builder):
        sig = self._signature()
        if self._packed is None or sig != self._packed_sig:
            self._packed = builder()
            self._packed_sig = sig
        return self._packed
