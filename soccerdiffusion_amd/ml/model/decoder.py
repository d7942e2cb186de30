"""DiffusionActionGenerator: the denoiser (reference: soccer_diffusion/ml/model/decoder.py:6-54).

Same constructor, same ``forward(x, context)``, same state_dict keys
(``embedding``, ``transformer_decoder.layers.N.*``, ``fc_out``); the forward is one call
into ``sd_denoiser_forward``."""

from __future__ import annotations

import torch
from torch import nn

from ... import ops
from ._params import LayerStack, LinearParams, PackedWeightsMixin
from .misc import PositionalEncoding


class DiffusionActionGenerator(PackedWeightsMixin, nn.Module):
    def __init__(self, num_joints: int, hidden_dim: int, num_layers: int, num_heads: int, max_seq_len: int):
        super().__init__()
        if hidden_dim not in (64, 128, 256, 512):
            raise ValueError("hidden_dim must be one of 64, 128, 256, 512 for the gfx950 kernels")
        self.num_joints, self.hidden_dim, self.num_heads, self.max_seq_len = num_joints, hidden_dim, num_heads, max_seq_len
        self.embedding = LinearParams(hidden_dim, num_joints)
        self.positional_encoding = PositionalEncoding(hidden_dim, max_seq_len)
        self.transformer_decoder = LayerStack(hidden_dim, num_layers, cross=True)
        self.fc_out = LinearParams(num_joints, hidden_dim)
        from ...training import Dropout

        # torch's default: the reference builds nn.TransformerDecoderLayer without a dropout argument (decoder.py:26-33)
        self.dropout = Dropout(p=0.1)

    def packed(self):
        def build():
            sd = {k: v for k, v in self.state_dict(keep_vars=True).items()}
            return ops.pack_denoiser(sd, self.embedding.weight.device, prefix="", heads=self.num_heads,
                                     max_len=self.max_seq_len)

        return self._packed_weights(build)

    def forward(self, x: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        """x (B, T, J) noisy actions; context (B, M, d) memory tokens -> predicted noise (B, T, J)."""
        tape = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if tape or (self.training and self.dropout.p > 0.0):   # dropout follows module.training, not the grad mode (torch)
            from ...training import denoiser_forward_autograd  # training kernels (+ backward)

            return denoiser_forward_autograd(self, x, context)
        return ops.denoiser_forward(self.packed(), x.contiguous(), context.contiguous())
