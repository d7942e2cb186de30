"""Deterministic synthetic weights with the reference checkpoint's keys and shapes
(SURVEY.md App. C).  Used by bench.py, smoke() and the tests on the GPU box, where the
reference (and therefore its init law) is absent.  Parity depends on the loaded values
only, not on how they were drawn (SURVEY.md §8(d) "Synthetic inputs")."""

from __future__ import annotations

import math
from typing import Mapping

import torch

Tensor = torch.Tensor


def synthetic_state_dict(
    d: int,
    J: int,
    L: int,
    seed: int = 0,
    encoders: Mapping[str, tuple[int, int, int]] | None = None,
    game_state: bool = False,
) -> dict[str, Tensor]:
    """Counter-seeded weights with the reference's checkpoint keys and shapes (App. C).

    ``encoders`` maps an encoder prefix (``action_history_encoder`` ...) to
    ``(input_dim, patch, num_layers)``."""
    g = torch.Generator().manual_seed(seed)

    def mat(*shape):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return (torch.rand(*shape, generator=g) * 2 - 1) / math.sqrt(fan_in)

    def vec(n, scale=0.1):
        return (torch.rand(n, generator=g) * 2 - 1) * scale

    sd: dict[str, Tensor] = {}

    def attn(pre, name):
        sd[f"{pre}{name}.in_proj_weight"] = mat(3 * d, d)
        sd[f"{pre}{name}.in_proj_bias"] = vec(3 * d)
        sd[f"{pre}{name}.out_proj.weight"] = mat(d, d)
        sd[f"{pre}{name}.out_proj.bias"] = vec(d)

    def ffn_norms(pre, norms):
        for nm in ("linear1", "linear2"):
            sd[f"{pre}{nm}.weight"] = mat(d, d)
            sd[f"{pre}{nm}.bias"] = vec(d)
        for nm in norms:
            sd[f"{pre}{nm}.weight"] = 1.0 + vec(d)
            sd[f"{pre}{nm}.bias"] = vec(d)

    sd["mean"] = vec(J, 1.0)
    sd["std"] = 0.5 + torch.rand(J, generator=g)
    sd["step_encoding.token"] = torch.randn(1, d // 2, generator=g)
    for name, (C, p, n_layers) in (encoders or {}).items():
        sd[f"{name}.embedding.weight"] = mat(d, C, p)
        sd[f"{name}.embedding.bias"] = vec(d)
        for l in range(n_layers):
            pre = f"{name}.transformer_encoder.layers.{l}."
            attn(pre, "self_attn")
            ffn_norms(pre, ("norm1", "norm2"))
    if game_state:
        sd["game_state_encoder.embedding.weight"] = torch.randn(4, d, generator=g)
    p0 = "diffusion_action_generator."
    sd[p0 + "embedding.weight"] = mat(d, J)
    sd[p0 + "embedding.bias"] = vec(d)
    for l in range(L):
        pre = f"{p0}transformer_decoder.layers.{l}."
        attn(pre, "self_attn")
        attn(pre, "multihead_attn")
        ffn_norms(pre, ("norm1", "norm2", "norm3"))
    sd[p0 + "fc_out.weight"] = mat(J, d)
    sd[p0 + "fc_out.bias"] = vec(J)
    return sd
