"""CPU oracle for the SoccerDiffusion denoiser hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-tensor restatement (stock ``torch`` CPU ops on explicit weight
tensors, no ``nn.Transformer*``) of what the reference computes on the path named in
BASELINE.json.  It is the parity checker and the ``cpu_baseline`` leg of ``bench.py``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` may
import it; the product package ``soccerdiffusion_amd`` never does.

Pinning: ``tests/golden/*.pt`` were produced by importing the reference's own model
modules in the build container (``tools/gen_golden.py``); ``tests/test_oracle_golden.py``
checks every function here against them.  The DDIM arithmetic (``oracle/ddim_ref.py``) is
the exception: it lives in diffusers==0.31.0, which is absent -> parity unpinned there.

Every function cites the reference lines it follows (paths relative to the reference
checkout).  All functions take a flat ``state_dict``-style mapping whose keys are the
reference checkpoint's keys, so the same weights feed the oracle and
the HIP path.
"""

from __future__ import annotations

import math
from typing import Mapping, Sequence

import torch

Tensor = torch.Tensor
NUM_HEADS = 4  # hard-coded by the reference: soccer_diffusion/ml/model/model.py:57,71,85,115
LN_EPS = 1e-5  # torch.nn.LayerNorm default used by nn.TransformerDecoderLayer


def _w(sd: Mapping[str, Tensor], key: str, dtype: torch.dtype) -> Tensor:
    return sd[key].to(device="cpu", dtype=dtype)  # no detach: train_loss_and_grads differentiates through


# --------------------------------------------------------------------------------------
# misc.py
# --------------------------------------------------------------------------------------
def step_token(steps: Tensor, token: Tensor, dim: int) -> Tensor:
    """StepToken.forward — soccer_diffusion/ml/model/misc.py:25-35.

    The frequency table is built in fp32 exactly like the reference
    (``arange(int64) * -float64(ln 1e4) / (n-1)`` -> fp32 tensor ``exp``) and multiplied by
    the step in the step's own dtype promotion (int64 -> fp32)."""
    n = dim // 4
    freq = torch.exp(torch.arange(n) * -math.log(10000) / (n - 1))  # fp32, misc.py:32
    ang = steps[:, None] * freq[None, :]  # misc.py:33
    tok = token.to(ang.dtype).expand(steps.size(0), dim // 2)
    return torch.cat((ang.sin(), ang.cos(), tok), dim=-1).unsqueeze(1)  # misc.py:34


def positional_table(d_model: int, max_len: int) -> Tensor:
    """PositionalEncoding.__init__ — soccer_diffusion/ml/model/misc.py:43-56 (fp32 table)."""
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


# --------------------------------------------------------------------------------------
# building blocks (torch nn.TransformerDecoderLayer, norm_first branch)
# --------------------------------------------------------------------------------------
def layer_norm(v: Tensor, g: Tensor, b: Tensor) -> Tensor:
    mu = v.mean(dim=-1, keepdim=True)
    var = ((v - mu) ** 2).mean(dim=-1, keepdim=True)  # biased
    return (v - mu) / torch.sqrt(var + LN_EPS) * g + b


def gelu_erf(u: Tensor) -> Tensor:
    return 0.5 * u * (1.0 + torch.erf(u / math.sqrt(2.0)))


# Dropout sites of one layer, in torch's forward order (nn.TransformerDecoderLayer._sa_block / _mha_block / _ff_block
# and nn.MultiheadAttention): probabilities of the self-attention, its out-projection (dropout1), probabilities of the
# cross-attention, its out-projection (dropout2), the activation (dropout) and linear2 (dropout3; dropout2 in an encoder
# layer).  The reference trains with torch's default p = 0.1 at all of them (decoder.py:26-33, encoder/base.py:29-40).
# ``masks`` is None (p = 0: the golden-vector path) or a callable (kind, shape) -> multiplier tensor (0 or 1/(1-p)) that
# the caller builds - the tests pass the very masks the HIP kernels regenerate from their Philox counter.
SITE_SA_PROBS, SITE_SA_OUT, SITE_CA_PROBS, SITE_CA_OUT, SITE_FFN_ACT, SITE_FFN_OUT = range(6)


def _drop(v: Tensor, masks, kind: int) -> Tensor:
    if masks is None:
        return v
    m = masks(kind, tuple(v.shape))
    return v if m is None else v * m.to(v.dtype)


def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, masks=None, kind: int = SITE_SA_PROBS) -> Tensor:
    """softmax(q k^T / sqrt(hd)) v per head, unmasked.  q (B,T,d); k,v (B,S,d).  Dropout acts on the probabilities
    (B, heads, T, S) after the softmax, as in torch's scaled-dot-product path."""
    B, T, d = q.shape
    S = k.shape[1]
    hd = d // heads
    qh = q.view(B, T, heads, hd).transpose(1, 2)
    kh = k.view(B, S, heads, hd).transpose(1, 2)
    vh = v.view(B, S, heads, hd).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    p = _drop(p, masks, kind)
    return (p @ vh).transpose(1, 2).reshape(B, T, d)


def self_attn_block(sd, pre: str, h: Tensor, heads: int, dt, masks=None) -> Tensor:
    d = h.shape[-1]
    n = layer_norm(h, _w(sd, pre + "norm1.weight", dt), _w(sd, pre + "norm1.bias", dt))
    qkv = n @ _w(sd, pre + "self_attn.in_proj_weight", dt).T + _w(sd, pre + "self_attn.in_proj_bias", dt)
    a = attention(qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :], heads, masks, SITE_SA_PROBS)
    return h + _drop(a @ _w(sd, pre + "self_attn.out_proj.weight", dt).T + _w(sd, pre + "self_attn.out_proj.bias", dt), masks, SITE_SA_OUT)


def cross_attn_block(sd, pre: str, h: Tensor, memory: Tensor, heads: int, dt, masks=None) -> Tensor:
    d = h.shape[-1]
    w = _w(sd, pre + "multihead_attn.in_proj_weight", dt)
    b = _w(sd, pre + "multihead_attn.in_proj_bias", dt)
    n = layer_norm(h, _w(sd, pre + "norm2.weight", dt), _w(sd, pre + "norm2.bias", dt))
    q = n @ w[:d].T + b[:d]
    k = memory @ w[d : 2 * d].T + b[d : 2 * d]  # memory is NOT normalised
    v = memory @ w[2 * d :].T + b[2 * d :]
    a = attention(q, k, v, heads, masks, SITE_CA_PROBS)
    return h + _drop(a @ _w(sd, pre + "multihead_attn.out_proj.weight", dt).T + _w(sd, pre + "multihead_attn.out_proj.bias", dt),
                     masks, SITE_CA_OUT)


def ffn_block(sd, pre: str, h: Tensor, norm: str, dt, masks=None) -> Tensor:
    n = layer_norm(h, _w(sd, pre + norm + ".weight", dt), _w(sd, pre + norm + ".bias", dt))
    u = _drop(gelu_erf(n @ _w(sd, pre + "linear1.weight", dt).T + _w(sd, pre + "linear1.bias", dt)), masks, SITE_FFN_ACT)
    return h + _drop(u @ _w(sd, pre + "linear2.weight", dt).T + _w(sd, pre + "linear2.bias", dt), masks, SITE_FFN_OUT)


def _num_layers(sd: Mapping[str, Tensor], stem: str) -> int:
    n = 0
    while f"{stem}{n}.norm1.weight" in sd:
        n += 1
    return n


# --------------------------------------------------------------------------------------
# decoder.py
# --------------------------------------------------------------------------------------
def denoiser_forward(
    sd: Mapping[str, Tensor],
    x: Tensor,
    memory: Tensor,
    prefix: str = "diffusion_action_generator.",
    dtype: torch.dtype = torch.float32,
    heads: int = NUM_HEADS,
    dropout_masks=None,
) -> Tensor:
    """DiffusionActionGenerator.forward — soccer_diffusion/ml/model/decoder.py:38-54.

    x (B,T,J) noisy trajectory, memory (B,M,d) context tokens incl. the step token.
    ``dropout_masks``: None, or a callable (layer, kind, shape) -> multiplier tensor (training-mode forward)."""
    dt = dtype
    x = x.to("cpu", dt)
    memory = memory.to("cpu", dt)
    w_emb = _w(sd, prefix + "embedding.weight", dt)
    d = w_emb.shape[0]
    T = x.shape[1]
    h = x @ w_emb.T + _w(sd, prefix + "embedding.bias", dt)  # decoder.py:48
    h = h + positional_table(d, T).to(dt)  # decoder.py:50 (fp32 table, misc.py:65)
    stem = prefix + "transformer_decoder.layers."
    for l in range(_num_layers(sd, stem)):  # decoder.py:52
        pre = f"{stem}{l}."
        mk = None if dropout_masks is None else (lambda kind, shape, l=l: dropout_masks(l, kind, shape))
        h = self_attn_block(sd, pre, h, heads, dt, mk)
        h = cross_attn_block(sd, pre, h, memory, heads, dt, mk)
        h = ffn_block(sd, pre, h, "norm3", dt, mk)
    return h @ _w(sd, prefix + "fc_out.weight", dt).T + _w(sd, prefix + "fc_out.bias", dt)  # decoder.py:54


# --------------------------------------------------------------------------------------
# encoder/base.py, encoder/game_state.py
# --------------------------------------------------------------------------------------
def encoder_forward(
    sd: Mapping[str, Tensor], x: Tensor, prefix: str, dtype: torch.dtype = torch.float32, heads: int = NUM_HEADS
) -> Tensor:
    """BaseEncoder.forward — soccer_diffusion/ml/model/encoder/base.py:41-53.

    Conv1d(kernel=stride=patch) on (B,C,S) is the GEMM (B*S/p, C*p) x (C*p, d) with
    W_conv (d,C,p) flattened over (c,k)."""
    dt = dtype
    x = x.to("cpu", dt)
    w = _w(sd, prefix + "embedding.weight", dt)  # (d, C, p)
    d, C, p = w.shape
    B, S, _ = x.shape
    n = S // p
    patches = x[:, : n * p].reshape(B, n, p, C).permute(0, 1, 3, 2).reshape(B, n, C * p)  # (c,k) order
    h = patches @ w.reshape(d, C * p).T + _w(sd, prefix + "embedding.bias", dt)  # base.py:49
    h = h + positional_table(d, n).to(dt)  # base.py:51
    stem = prefix + "transformer_encoder.layers."
    for l in range(_num_layers(sd, stem)):  # base.py:53
        pre = f"{stem}{l}."
        h = self_attn_block(sd, pre, h, heads, dt)
        h = ffn_block(sd, pre, h, "norm2", dt)
    return h


def game_state_forward(sd: Mapping[str, Tensor], idx: Tensor, dtype: torch.dtype = torch.float32) -> Tensor:
    """GameStateEncoder.forward — soccer_diffusion/ml/model/encoder/game_state.py:19-27."""
    return _w(sd, "game_state_encoder.embedding.weight", dtype)[idx.cpu().long()].unsqueeze(1)


# --------------------------------------------------------------------------------------
# model.py
# --------------------------------------------------------------------------------------
def encode_input_data(sd: Mapping[str, Tensor], input_data: Mapping[str, Tensor], dtype=torch.float32) -> list[Tensor]:
    """End2EndDiffusionTransformer.encode_input_data — soccer_diffusion/ml/model/model.py:123-148.

    Enabled encoders are detected from the keys present in ``sd`` (image path excluded)."""
    ctx = []
    if "action_history_encoder.embedding.weight" in sd:
        ctx.append(encoder_forward(sd, input_data["joint_command_history"], "action_history_encoder.", dtype))
    if "imu_encoder.embedding.weight" in sd:
        ctx.append(encoder_forward(sd, input_data["rotation"], "imu_encoder.", dtype))
    if "joint_states_encoder.embedding.weight" in sd:
        ctx.append(encoder_forward(sd, input_data["joint_state"], "joint_states_encoder.", dtype))
    if "game_state_encoder.embedding.weight" in sd:
        ctx.append(game_state_forward(sd, input_data["game_state"], dtype))
    return ctx


def forward_with_context(
    sd: Mapping[str, Tensor], context: Sequence[Tensor], noisy: Tensor, step: Tensor, dtype: torch.dtype = torch.float32,
    dropout_masks=None,
) -> Tensor:
    """End2EndDiffusionTransformer.forward_with_context — soccer_diffusion/ml/model/model.py:159-179."""
    d = sd["diffusion_action_generator.embedding.weight"].shape[0]
    tok = step_token(step.cpu(), sd["step_encoding.token"].cpu().float(), d)  # model.py:173
    mem = torch.cat([c.to("cpu", dtype) for c in context] + [tok.to(dtype)], dim=1)  # model.py:176
    return denoiser_forward(sd, noisy, mem, dtype=dtype, dropout_masks=dropout_masks)  # model.py:179


def forward(sd, input_data, noisy, step, dtype=torch.float32) -> Tensor:
    """End2EndDiffusionTransformer.forward — soccer_diffusion/ml/model/model.py:150-157."""
    return forward_with_context(sd, encode_input_data(sd, input_data, dtype), noisy, step, dtype)


def train_loss_and_grads(sd, noisy, step, noise, context=None, input_data=None, dtype=torch.float32, dropout_masks=None):
    """Loss and per-parameter gradients of one reference training step:
    ``mse_loss(model(...), noise)`` + ``backward`` — soccer_diffusion/ml/training/train.py:221-238; dropout p = 0
    unless ``dropout_masks`` (decoder sites only, see denoiser_forward) hands in the masks to apply.
    Exactly one of ``context`` (decoder-pretraining path, train.py:221-224) and
    ``input_data`` (full model, train.py:226) is given.  Returns (pred, loss, grads)."""
    leaf = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k not in ("mean", "std")) for k, v in sd.items()}
    with torch.enable_grad():
        if context is not None:
            pred = forward_with_context(leaf, context, noisy, step, dtype, dropout_masks=dropout_masks)
        else:
            pred = forward(leaf, input_data, noisy, step, dtype)
        loss = torch.nn.functional.mse_loss(pred, noise.to(dtype))
        loss.backward()
    grads = {k: v.grad for k, v in leaf.items() if v.grad is not None}
    return pred.detach(), loss.detach(), grads


def normalize(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """Normalizer.normalize — soccer_diffusion/dataset/pytorch.py:410-411."""
    return (x - mean) / std


def denormalize(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """Normalizer.denormalize — soccer_diffusion/dataset/pytorch.py:413-414."""
    return x * std + mean


# Deterministic synthetic weights are DATA shared with bench.py/smoke; the generator lives
# in the product package so that the product never imports oracle/.
from soccerdiffusion_amd.synthetic import synthetic_state_dict  # noqa: E402,F401
