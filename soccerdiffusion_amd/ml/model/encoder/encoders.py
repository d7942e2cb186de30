"""Context encoders (reference: soccer_diffusion/ml/model/encoder/{base,joint,imu,game_state}.py).

All three sequence encoders are the same network — Conv1d(kernel = stride = patch) patch
embedding, positional table, pre-norm encoder layers — run by ``sd_encoder_forward``."""

from __future__ import annotations

from enum import Enum

import torch
from torch import nn

from .... import ops
from .._params import LayerStack, LinearParams, PackedWeightsMixin
from ..misc import PositionalEncoding

ROBOT_STATES = 4  # len(RobotState): PLAYING, POSITIONING, STOPPED, UNKNOWN (reference dataset/models.py:13-25)


class BaseEncoder(PackedWeightsMixin, nn.Module):
    def __init__(self, input_dim: int, patch_size: int, hidden_dim: int, num_layers: int, num_heads: int, max_seq_len: int):
        super().__init__()
        self.input_dim, self.patch_size, self.hidden_dim = input_dim, patch_size, hidden_dim
        self.num_heads, self.max_seq_len = num_heads, max_seq_len
        self.embedding = LinearParams(hidden_dim, input_dim, patch_size)  # Conv1d weight (d, C, p)
        self.positional_encoding = PositionalEncoding(hidden_dim, max_seq_len)
        self.transformer_encoder = LayerStack(hidden_dim, num_layers, cross=False)
        from ....training import Dropout

        self.dropout = Dropout(p=0.1)  # torch's default (the reference's nn.TransformerEncoderLayer, encoder/base.py:29-40)

    def packed(self):
        def build():
            sd = {"enc." + k: v for k, v in self.state_dict(keep_vars=True).items()}
            return ops.pack_encoder(sd, self.embedding.weight.device, "enc.", heads=self.num_heads, max_len=self.max_seq_len)

        return self._packed_weights(build)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, S, input_dim) -> context tokens (B, S // patch, hidden_dim)."""
        tape = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if tape or (self.training and self.dropout.p > 0.0):
            from ....training import encoder_forward_autograd

            return encoder_forward_autograd(self, x)
        return ops.encoder_forward(self.packed(), x.contiguous())


class JointEncoder(BaseEncoder):
    def __init__(self, num_joints: int, patch_size: int, hidden_dim: int, num_layers: int, num_heads: int, max_seq_len: int):
        super().__init__(num_joints, patch_size, hidden_dim, num_layers, num_heads, max_seq_len)


class IMUEncoder(BaseEncoder):
    class OrientationEmbeddingMethod(Enum):
        QUATERNION = "quaternion"
        FIVE_DIM = "five_dim"  # axis + (sin, cos) of the angle

    def __init__(self, orientation_embedding_method, patch_size: int, hidden_dim: int, num_layers: int, num_heads: int,
                 max_seq_len: int):
        method = IMUEncoder.OrientationEmbeddingMethod(orientation_embedding_method)
        features = {IMUEncoder.OrientationEmbeddingMethod.QUATERNION: 4, IMUEncoder.OrientationEmbeddingMethod.FIVE_DIM: 5}[method]
        super().__init__(features, patch_size, hidden_dim, num_layers, num_heads, max_seq_len)
        self.orientation_embedding_method = method


class _EmbeddingTable(nn.Module):
    def __init__(self, n: int, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, d))


class GameStateEncoder(nn.Module):
    """``embedding.weight`` (4, d) gathered by state index -> (B, 1, d)."""

    def __init__(self, hidden_dim: int):
        super().__init__()
        self.embedding = _EmbeddingTable(ROBOT_STATES, hidden_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and self.embedding.weight.requires_grad:
            return self.embedding.weight[x].unsqueeze(1)  # index_select autograd (plumbing; 1 row per sample)
        return ops.game_state_embed(x.contiguous(), self.embedding.weight.detach())
