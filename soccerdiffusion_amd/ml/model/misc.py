"""StepToken and PositionalEncoding with the reference's interface
(reference: soccer_diffusion/ml/model/misc.py:6-65)."""

from __future__ import annotations

import torch
from torch import nn

from ... import ops


class StepToken(nn.Module):
    """Diffusion-step token: [sin(t f) | cos(t f) | learned half] -> (B, 1, dim).
    Parameter ``token`` (1, dim//2) as in the reference (misc.py:23)."""

    def __init__(self, dim: int):
        super().__init__()
        if dim % 4 != 0 or dim // 4 < 2:
            raise ValueError("StepToken needs dim divisible by 4 and >= 8")
        self.dim = dim
        self.token = nn.Parameter(torch.randn(1, dim // 2))
        self.register_buffer("_freq", ops.step_frequencies(dim), persistent=False)

    def forward(self, steps: torch.Tensor) -> torch.Tensor:
        return ops.step_token(steps.contiguous(), self._freq, self.token.detach())

    def table(self, timesteps, device) -> torch.Tensor:
        """Tokens of a whole timestep schedule at once: (n_steps, dim)."""
        t = torch.as_tensor(list(timesteps), dtype=torch.int64, device=device)
        return self.forward(t).reshape(len(timesteps), self.dim)


class PositionalEncoding(nn.Module):
    """Fixed sin/cos table in a NON-persistent buffer ``pe`` (1, max_len, d_model), so
    checkpoints carry no table (reference misc.py:51).  The HIP kernels add the table
    inside the embedding kernel; ``forward`` exists for API parity."""

    def __init__(self, d_model: int, max_len: int):
        super().__init__()
        self.register_buffer("pe", ops.positional_table(d_model, max_len).unsqueeze(0), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x + self.pe[:, : x.size(1)]
