"""Tensor-level wrappers over the C ABI (include/soccerdiffusion_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation is a
HIP kernel behind ``libsoccerdiffusion_hip.so``.  All functions require contiguous fp32
CUDA(HIP) tensors and raise otherwise — there is no CPU path in this package.
"""

from __future__ import annotations

import ctypes as C
import math
from typing import Mapping, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import DenoiserWeights, EncoderWeights, LayerWeights, check

Tensor = torch.Tensor
NUM_TRAIN_TIMESTEPS = 1000


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: Tensor, name: str, dtype=torch.float32) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the MI355X (cuda) device, got {t.device}; "
                           "soccerdiffusion_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    return t


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# --------------------------------------------------------------------------------------
# host-side tables, built exactly as the reference builds them (fp32 CPU ops)
# --------------------------------------------------------------------------------------
def positional_table(d_model: int, max_len: int) -> Tensor:
    """The ``pe`` buffer of PositionalEncoding (reference ml/model/misc.py:43-56), CPU fp32."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def step_frequencies(dim: int) -> Tensor:
    """Frequency table of StepToken (reference ml/model/misc.py:31-32), CPU fp32."""
    half_dim = dim // 4
    return torch.exp(torch.arange(half_dim) * -math.log(10000) / (half_dim - 1))


def alphas_cumprod(num_train_timesteps: int = NUM_TRAIN_TIMESTEPS) -> Tensor:
    """squaredcos_cap_v2 schedule of the scheduler the reference constructs at
    ml/training/train.py:185 (diffusers DDIMScheduler defaults)."""
    def alpha_bar(s):
        return math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2

    n = num_train_timesteps
    betas = torch.tensor([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), 0.999) for i in range(n)],
                         dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(num_inference_steps: int, num_train_timesteps: int = NUM_TRAIN_TIMESTEPS) -> list[int]:
    """``set_timesteps`` (leading spacing, offset 0): e.g. 50 -> 980, 960, ..., 0."""
    ratio = num_train_timesteps // num_inference_steps
    return [int(round(i * ratio)) for i in range(num_inference_steps)][::-1]


def ddim_coefficients(timesteps: Sequence[int], acp: Tensor, num_inference_steps: int,
                      num_train_timesteps: int = NUM_TRAIN_TIMESTEPS) -> np.ndarray:
    """Per step (sqrt a_t, sqrt(1-a_t), sqrt a_prev, sqrt(1-a_prev)) as fp32 (eta = 0,
    final_alpha_cumprod = 1).  Computed with fp32 tensor ops like the scheduler does."""
    acp = acp.detach().cpu().float()
    ratio = num_train_timesteps // num_inference_steps
    out = np.zeros((len(timesteps), 4), dtype=np.float32)
    one = torch.tensor(1.0)
    for i, t in enumerate(timesteps):
        prev = t - ratio
        a_t = acp[t]
        a_p = acp[prev] if prev >= 0 else one
        out[i] = [float(a_t.sqrt()), float((1 - a_t).sqrt()), float(a_p.sqrt()), float((1 - a_p).sqrt())]
    return out


# --------------------------------------------------------------------------------------
# weight descriptors
# --------------------------------------------------------------------------------------
_DEC_KEYS = {
    "sa_in_w": "self_attn.in_proj_weight", "sa_in_b": "self_attn.in_proj_bias",
    "sa_out_w": "self_attn.out_proj.weight", "sa_out_b": "self_attn.out_proj.bias",
    "ca_in_w": "multihead_attn.in_proj_weight", "ca_in_b": "multihead_attn.in_proj_bias",
    "ca_out_w": "multihead_attn.out_proj.weight", "ca_out_b": "multihead_attn.out_proj.bias",
    "lin1_w": "linear1.weight", "lin1_b": "linear1.bias", "lin2_w": "linear2.weight", "lin2_b": "linear2.bias",
    "n1_w": "norm1.weight", "n1_b": "norm1.bias", "n2_w": "norm2.weight", "n2_b": "norm2.bias",
    "n3_w": "norm3.weight", "n3_b": "norm3.bias",
}


class _Packed:
    """Keeps the ctypes structs and every tensor they point at alive together."""

    def __init__(self):
        self.keep: list[Tensor] = []
        self.layers = None
        self.struct = None

    def dev(self, t: Tensor, device) -> Tensor:
        t = t.detach()
        if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(device=device, dtype=torch.float32).contiguous()
        self.keep.append(t)
        return t

    def fill_layers(self, sd: Mapping[str, Tensor], stem: str, n_layers: int, device, decoder: bool):
        arr = (LayerWeights * n_layers)()
        for l in range(n_layers):
            for field, key in _DEC_KEYS.items():
                full = f"{stem}{l}.{key}"
                if full in sd:
                    setattr(arr[l], field, self.dev(sd[full], device).data_ptr())
                elif decoder or not (field.startswith("ca_") or field.startswith("n3_")):
                    raise KeyError(f"missing weight {full}")
        self.layers = arr


def _count_layers(sd: Mapping[str, Tensor], stem: str) -> int:
    n = 0
    while f"{stem}{n}.norm1.weight" in sd:
        n += 1
    return n


def pack_denoiser(sd: Mapping[str, Tensor], device, prefix: str = "diffusion_action_generator.",
                  heads: int = 4, max_len: Optional[int] = None) -> _Packed:
    """Builds the ``sd_denoiser_weights`` descriptor from checkpoint-keyed tensors
    (zero-copy for tensors already on ``device``)."""
    device = torch.device(device)
    p = _Packed()
    emb_w = p.dev(sd[prefix + "embedding.weight"], device)
    d, J = emb_w.shape
    stem = prefix + "transformer_decoder.layers."
    L = _count_layers(sd, stem)
    p.fill_layers(sd, stem, L, device, decoder=True)
    T_max = int(max_len) if max_len is not None else 512
    pe = p.dev(positional_table(d, T_max), device)
    w = DenoiserWeights()
    w.d, w.J, w.L, w.heads = d, J, L, heads
    w.emb_w = emb_w.data_ptr()
    w.emb_b = p.dev(sd[prefix + "embedding.bias"], device).data_ptr()
    w.out_w = p.dev(sd[prefix + "fc_out.weight"], device).data_ptr()
    w.out_b = p.dev(sd[prefix + "fc_out.bias"], device).data_ptr()
    w.pe = pe.data_ptr()
    w.T_max = T_max
    w.layers = C.cast(p.layers, C.POINTER(LayerWeights))
    p.struct = w
    p.d, p.J, p.L, p.heads, p.T_max = d, J, L, heads, T_max
    return p


def pack_encoder(sd: Mapping[str, Tensor], device, prefix: str, heads: int = 4, max_len: Optional[int] = None) -> _Packed:
    """Builds the ``sd_encoder_weights`` descriptor for a BaseEncoder (``prefix`` ends with '.')."""
    device = torch.device(device)
    p = _Packed()
    emb_w = p.dev(sd[prefix + "embedding.weight"], device)
    d, Cin, patch = emb_w.shape
    stem = prefix + "transformer_encoder.layers."
    L = _count_layers(sd, stem)
    p.fill_layers(sd, stem, L, device, decoder=False)
    S_max = int(max_len) if max_len is not None else 512
    pe = p.dev(positional_table(d, S_max), device)
    w = EncoderWeights()
    w.d, w.C, w.p, w.L, w.heads, w.S_max = d, Cin, patch, L, heads, S_max
    w.emb_w = emb_w.data_ptr()
    w.emb_b = p.dev(sd[prefix + "embedding.bias"], device).data_ptr()
    w.pe = pe.data_ptr()
    w.layers = C.cast(p.layers, C.POINTER(LayerWeights))
    p.struct = w
    p.d, p.C, p.p, p.L, p.heads, p.S_max = d, Cin, patch, L, heads, S_max
    return p


_ws_cache: dict = {}

# Derived copies of the weights (split fp16 planes, folded BatchNorm vectors, the loop-form sampler's prepared workspace) are keyed on
# the tensors' version counters - which do NOT move when FusedAdamW updates its flat buffer through a native kernel on raw pointers
# (training.py) or when a captured training graph is replayed.  Every such update bumps this counter, and every cache key includes it.
_weights_generation = 0


def weights_generation() -> int:
    return _weights_generation


def bump_weights_generation() -> None:
    global _weights_generation
    _weights_generation += 1



def workspace(n_floats: int, device) -> Tensor:
    """Grow-only scratch buffer per (device, stream)."""
    key = (torch.device(device), _stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < n_floats:
        buf = torch.empty(int(n_floats), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


# --------------------------------------------------------------------------------------
# calls
# --------------------------------------------------------------------------------------
def denoiser_forward(packed: _Packed, x: Tensor, memory: Tensor) -> Tensor:
    """DiffusionActionGenerator.forward (reference ml/model/decoder.py:38-54)."""
    lib = _lib.load()
    _req(x, "x"); _req(memory, "memory")
    B, T, J = x.shape
    Bm, M, d = memory.shape
    if Bm != B or d != packed.d or J != packed.J:
        raise ValueError(f"shape mismatch: x {tuple(x.shape)}, memory {tuple(memory.shape)}, model d={packed.d} J={packed.J}")
    out = torch.empty_like(x)
    ws = workspace(lib.sd_workspace_floats(B, T, M, d, packed.L, 0), x.device)
    check(lib.sd_denoiser_forward(C.byref(packed.struct), x.data_ptr(), memory.data_ptr(), out.data_ptr(),
                                  ws.data_ptr(), B, T, M, _stream()), "sd_denoiser_forward")
    return out


def encoder_forward(packed: _Packed, x: Tensor) -> Tensor:
    """BaseEncoder.forward (reference ml/model/encoder/base.py:41-53)."""
    lib = _lib.load()
    _req(x, "x")
    B, S, Cin = x.shape
    if Cin != packed.C:
        raise ValueError(f"encoder expects {packed.C} input features, got {Cin}")
    n = S // packed.p
    out = torch.empty(B, n, packed.d, dtype=torch.float32, device=x.device)
    ws = workspace(lib.sd_workspace_floats(B, n, 1, packed.d, 1, 0), x.device)
    check(lib.sd_encoder_forward(C.byref(packed.struct), x.data_ptr(), out.data_ptr(), ws.data_ptr(), B, S, _stream()),
          "sd_encoder_forward")
    return out


def step_token(steps: Tensor, freq: Tensor, token: Tensor, out: Optional[Tensor] = None, row_stride: Optional[int] = None) -> Tensor:
    """StepToken.forward (reference ml/model/misc.py:25-35) -> (B, 1, d)."""
    lib = _lib.load()
    if steps.dtype not in (torch.int64, torch.float32):
        steps = steps.to(torch.int64 if not steps.is_floating_point() else torch.float32)
    _req(steps, "steps", steps.dtype); _req(freq, "freq"); _req(token, "token")
    B = steps.shape[0]
    d = token.numel() * 2
    if out is None:
        out = torch.empty(B, 1, d, dtype=torch.float32, device=steps.device)
        row_stride = d
    check(lib.sd_step_token(steps.data_ptr(), int(steps.dtype == torch.int64), freq.data_ptr(), token.data_ptr(),
                            out.data_ptr(), int(row_stride), B, d, _stream()), "sd_step_token")
    return out


def game_state_embed(idx: Tensor, table: Tensor) -> Tensor:
    """GameStateEncoder.forward (reference ml/model/encoder/game_state.py:19-27) -> (B, 1, d)."""
    lib = _lib.load()
    _req(idx, "game_state", torch.int64); _req(table, "embedding.weight")
    B = idx.shape[0]
    n, d = table.shape
    out = torch.empty(B, 1, d, dtype=torch.float32, device=idx.device)
    check(lib.sd_game_state_embed(idx.data_ptr(), table.data_ptr(), out.data_ptr(), d, B, d, n, _stream()),
          "sd_game_state_embed")
    return out


def normalize(x: Tensor, mean: Tensor, std: Tensor, inverse: bool = False) -> Tensor:
    """Normalizer.normalize / denormalize (reference dataset/pytorch.py:410-414), last dim = joints."""
    lib = _lib.load()
    _req(x, "x"); _req(mean, "mean"); _req(std, "std")
    out = torch.empty_like(x)
    check(lib.sd_normalize(x.data_ptr(), mean.data_ptr(), std.data_ptr(), out.data_ptr(), x.numel(), x.shape[-1],
                           int(inverse), _stream()), "sd_normalize")
    return out


def ddim_add_noise(x0: Tensor, noise: Tensor, t: Tensor, acp: Tensor) -> Tensor:
    """scheduler.add_noise (reference call site ml/training/train.py:218)."""
    lib = _lib.load()
    _req(x0, "x0"); _req(noise, "noise"); _req(t, "timesteps", torch.int64); _req(acp, "alphas_cumprod")
    out = torch.empty_like(x0)
    B = x0.shape[0]
    check(lib.sd_ddim_add_noise(x0.data_ptr(), noise.data_ptr(), t.data_ptr(), acp.data_ptr(), out.data_ptr(), B,
                                x0.numel() // B, _stream()), "sd_ddim_add_noise")
    return out


def ddim_step(eps: Tensor, x: Tensor, coef4) -> Tensor:
    """scheduler.step(...).prev_sample (reference call site ml/inference/plot.py:131)."""
    lib = _lib.load()
    _req(eps, "eps"); _req(x, "x")
    out = torch.empty_like(x)
    c = [float(v) for v in coef4]
    check(lib.sd_ddim_step(eps.data_ptr(), x.data_ptr(), out.data_ptr(), c[0], c[1], c[2], c[3], x.numel(), _stream()),
          "sd_ddim_step")
    return out


STATUS_NONFINITE = 1  # SD_STATUS_NONFINITE
STATUS_SHARP_LOGITS = 2  # SD_STATUS_SHARP_LOGITS (sampler mode 4: a self-attention logit beyond the validated range)


class GraphedSampler:
    """The whole rollout (n_steps x (L x 2 + 3) kernel launches of ``sd_ddim_sample``) captured
    once into a hipGraph and replayed: removes the per-launch host cost, which dominates at
    small batch (the robot's B = 1, 30-step rollout).  Static shapes; the inputs are copied
    into the captured buffers before every replay.  ``status`` (one int32 on the device) is
    the range-guard word of ``sd_ddim_sample_ex``, rewritten by every replay."""

    def __init__(self, packed: _Packed, B: int, T: int, Mc: int, step_tokens: Tensor, coef: np.ndarray, max_mode: int = -1):
        dev = step_tokens.device
        self.packed, self.coef = packed, np.ascontiguousarray(coef, dtype=np.float32)
        self.tokens = step_tokens.contiguous()
        self.max_mode = int(max_mode)
        self.x = torch.zeros(B, T, packed.J, dtype=torch.float32, device=dev)
        self.ctx = torch.zeros(B, Mc, packed.d, dtype=torch.float32, device=dev) if Mc > 0 else None
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        lib = _lib.load()
        self.ws = torch.empty(lib.sd_workspace_floats(B, T, max(Mc, 1), packed.d, packed.L, len(self.coef)),
                              dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._run()  # warm-up outside capture: lazy module load and LDS attributes happen here
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._run()

    def _run(self):
        B, T, _ = self.x.shape
        check(_lib.load().sd_ddim_sample_ex(C.byref(self.packed.struct), _ptr(self.ctx), self.tokens.data_ptr(),
                                            self.coef.ctypes.data_as(_lib.c_float_p), self.x.data_ptr(), None,
                                            self.ws.data_ptr(), B, T, 0 if self.ctx is None else self.ctx.shape[1],
                                            len(self.coef), self.status.data_ptr(), self.max_mode, _stream()),
              "sd_ddim_sample_ex")

    def replay_into(self, ctx: Optional[Tensor], x_T: Tensor) -> Tensor:
        """Copies the inputs into the captured buffers, replays, and returns the captured x buffer itself
        (overwritten by the next replay)."""
        self.x.copy_(x_T)
        if self.ctx is not None:
            self.ctx.copy_(ctx)
        self.graph.replay()
        return self.x

    def __call__(self, ctx: Optional[Tensor], x_T: Tensor) -> Tensor:
        return self.replay_into(ctx, x_T).clone()


def ddim_sample(packed: _Packed, ctx: Optional[Tensor], step_tokens: Tensor, coef: np.ndarray, x_T: Tensor,
                trace: bool = False, inplace: bool = False, status: Optional[Tensor] = None, max_mode: int = -1,
                eps_trace: bool = False):
    """The reference's sampling loop (ml/inference/plot.py:122-131, ml/training/distill.py:179-189)
    as ONE native call.  Returns the sample, or (sample, per-step trace) when ``trace``; with ``eps_trace`` the
    noise prediction of every step (n_steps, B, T, J) - the value of ``forward_with_context`` inside the loop - is
    appended to the returned tuple.
    ``status`` (int32 tensor of one element on the device) receives the range-guard word of
    ``sd_ddim_sample_ex`` - not read here, so the call stays asynchronous; ``max_mode`` caps the kernel selection
    (see ``ddim_sample_guarded``)."""
    lib = _lib.load()
    _req(x_T, "x_T"); _req(step_tokens, "step_tokens")
    B, T, J = x_T.shape
    n_steps = step_tokens.shape[0]
    Mc = 0
    if ctx is not None:
        _req(ctx, "context")
        Mc = ctx.shape[1]
        if ctx.shape[0] != B or ctx.shape[2] != packed.d:
            raise ValueError("context shape mismatch")
    coef = np.ascontiguousarray(coef, dtype=np.float32)
    if coef.shape != (n_steps, 4):
        raise ValueError("coef must be (n_steps, 4)")
    if status is not None:
        _req(status, "status", torch.int32)
    x = x_T if inplace else x_T.clone()
    tr = torch.empty(n_steps, B, T, J, dtype=torch.float32, device=x.device) if trace else None
    et = torch.empty(n_steps, B, T, J, dtype=torch.float32, device=x.device) if eps_trace else None
    ws = workspace(lib.sd_workspace_floats(B, T, max(Mc, 1), packed.d, packed.L, n_steps), x.device)
    check(lib.sd_ddim_sample_eps(C.byref(packed.struct), _ptr(ctx), step_tokens.data_ptr(),
                                 coef.ctypes.data_as(_lib.c_float_p), x.data_ptr(), _ptr(tr), _ptr(et), ws.data_ptr(),
                                 B, T, Mc, n_steps, _ptr(status), int(max_mode), _stream()), "sd_ddim_sample_eps")
    out = (x,) + ((tr,) if trace else ()) + ((et,) if eps_trace else ())
    return out if len(out) > 1 else x


def default_sampler_cap() -> int:
    """The highest sampler mode a call runs unless told otherwise: 3 - three fp16 products at every site, valid for any weights.
    Mode 4 (two products at the Q | K | V projection, guarded by SD_STATUS_SHARP_LOGITS) is opt-in: ``max_mode=4`` at the call
    (``End2EndDiffusionTransformer.sample(..., max_mode=4)``) or ``SD_SAMPLER_MODE=4`` in the environment."""
    import os

    try:
        cap = int(os.environ.get("SD_SAMPLER_MODE", "3"))
    except ValueError:
        cap = 3
    return cap if 0 <= cap <= 4 else 3


def sampler_cap(packed: _Packed, max_mode: Optional[int] = None) -> int:
    """``max_mode`` of a call: the caller's (or the default), lowered to 3 once these weights tripped mode 4's guard."""
    cap = default_sampler_cap() if max_mode is None else int(max_mode)
    return min(cap, getattr(packed, "sampler_cap", 4))


def ddim_sample_guarded(packed: _Packed, ctx: Optional[Tensor], step_tokens: Tensor, coef: np.ndarray, x_T: Tensor,
                        trace: bool = False, max_mode: Optional[int] = None):
    """``ddim_sample`` with the range guard read back (one host synchronisation): when the split-fp16 kernels of
    sampler modes 2 .. 4 were driven out of their operand range (|8 v| >= 65520 for a LayerNorm / attention / GELU output -
    e.g. a checkpoint with LayerNorm weights in the thousands), the rollout is repeated on the exact-fp32 MFMA kernels
    (``max_mode`` 1), which have no such limit.  Raises if that result is not finite either (non-finite inputs).
    ``max_mode``: None = ``default_sampler_cap()`` (3).  With 4, mode 4's own guard (SD_STATUS_SHARP_LOGITS: a self-attention logit
    beyond the range its two-product Q | K | V site is validated on) repeats the rollout on mode 3 and pins ``packed.sampler_cap`` there."""
    import warnings

    status = torch.zeros(1, dtype=torch.int32, device=x_T.device)
    cap = sampler_cap(packed, max_mode)
    out = ddim_sample(packed, ctx, step_tokens, coef, x_T, trace=trace, status=status, max_mode=cap)
    word = int(status.item())
    if word == 0:
        return out
    if word & STATUS_SHARP_LOGITS:
        # sampler mode 4 met a self-attention logit beyond the range its two-product Q | K | V projection is validated on
        # (SD_SHARP_LOGIT_LIMIT): the same rollout with three fp16 products at every site (mode 3).  Sharpness is a property
        # of the checkpoint, so later calls with these weights start there.
        packed.sampler_cap = 3
        if not word & STATUS_NONFINITE:
            out = ddim_sample(packed, ctx, step_tokens, coef, x_T, trace=trace, status=status, max_mode=3)
            if int(status.item()) == 0:
                return out
    mode = _lib.load().sd_sampler_mode(packed.d, packed.heads, x_T.shape[1], 0 if ctx is None else ctx.shape[1], packed.J)
    if mode < 2 and _chain16_possible(packed):
        mode = 2   # the unfused row chains run on the split-fp16 pipe as well
    if mode >= 2:
        warnings.warn("sd_ddim_sample: the split-fp16 kernels left their operand range (non-finite sample); "
                      "repeating the rollout on the fp32-MFMA kernels", RuntimeWarning, stacklevel=2)
        out = ddim_sample(packed, ctx, step_tokens, coef, x_T, trace=trace, status=status, max_mode=1)
        if int(status.item()) == 0:
            return out
    raise FloatingPointError("sd_ddim_sample produced non-finite values on the fp32 kernels too: the inputs or the weights are not finite")


PREPARE_WEIGHTS, PREPARE_CONTEXT = 1, 2   # SD_PREPARE_*
E_UNSUPPORTED = -4   # SD_E_UNSUPPORTED


class LoopSampler:
    """``forward_with_context`` inside the reference's own denoising loop (soccer_diffusion/ml/inference/plot.py:122-131,
    ml/training/distill.py:179-189, ml/inference/ros.py:301-310) on the trajectory kernels of sampler mode 3: one launch of
    ``traj_step_kernel`` per call (plus the step tokens' K / V and fold).  What does not depend on x or the step - the split
    weight planes, the context's K / V folded with Wq / Woc - is prepared into a workspace this object owns and reused until the
    weights (``weights_key``) or the context tensors (identity and version counters; held here so that their addresses cannot be
    recycled under the cache) change.  ``supported`` is False where the shape does not take these kernels."""

    def __init__(self, packed: _Packed, B: int, T: int, Mc: int, n_tok: int, device, max_mode: int = 3):
        lib = _lib.load()
        self.shape = (B, T, Mc, n_tok)
        self.max_mode = int(max_mode)
        self.supported = lib.sd_sampler_mode(packed.d, packed.heads, T, Mc, packed.J) >= 3 and packed.L <= 8
        self.ws = (torch.empty(lib.sd_workspace_floats(B, T, max(Mc, 1), packed.d, packed.L, n_tok), dtype=torch.float32, device=device)
                   if self.supported else None)
        self.status = torch.zeros(1, dtype=torch.int32, device=device) if self.supported and self.max_mode == 4 else None
        self.weights_key = None
        self.context: list = []
        self.versions: tuple = ()
        self.prepares = 0   # (tests: how many times the context was folded)

    def _context_hit(self, context) -> bool:
        return (len(context) == len(self.context) and all(a is b for a, b in zip(context, self.context))
                and tuple(c._version for c in context) == self.versions)

    def eps(self, packed: _Packed, context, tokens: Tensor, x: Tensor, weights_key) -> Optional[Tensor]:
        """Noise prediction (B, T, J) or None when the library declines the shape (the caller falls back)."""
        lib = _lib.load()
        B, T, Mc, n_tok = self.shape
        _req(x, "x"); _req(tokens, "step tokens")
        what = 0
        if weights_key != self.weights_key:
            what = PREPARE_WEIGHTS | PREPARE_CONTEXT   # the fold multiplies the context's K / V with Wq / Woc
        elif not self._context_hit(context):
            what = PREPARE_CONTEXT
        if what:
            ctx = None
            if Mc > 0:
                ctx = (context[0] if len(context) == 1 else torch.cat(list(context), dim=1)).contiguous()
                _req(ctx, "context")
            rc = lib.sd_sampler_prepare(C.byref(packed.struct), _ptr(ctx), self.ws.data_ptr(), B, T, Mc, n_tok, what, self.max_mode, _stream())
            if rc == E_UNSUPPORTED:
                self.supported = False
                return None
            check(rc, "sd_sampler_prepare")
            self.weights_key, self.context, self.versions = weights_key, list(context), tuple(c._version for c in context)
            self.prepares += 1
        eps = torch.empty_like(x)
        rc = lib.sd_sampler_eps(C.byref(packed.struct), tokens.data_ptr(), x.data_ptr(), eps.data_ptr(), self.ws.data_ptr(), B, T, Mc, n_tok,
                                _ptr(self.status), self.max_mode, _stream())
        if rc == E_UNSUPPORTED:
            self.supported = False
            return None
        check(rc, "sd_sampler_eps")
        return eps


def _chain16_possible(packed: _Packed) -> bool:
    return packed.d in (128, 256, 512) and packed.J % 4 == 0


# ---- single ops (unit parity tests) ----------------------------------------------------
def linear(A: Tensor, W: Tensor, bias: Optional[Tensor] = None, ln: Optional[tuple] = None, act: str = "none",
           res: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    _req(A, "A"); _req(W, "W")
    R, d = A.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(R, N, dtype=torch.float32, device=A.device)
    check(lib.sd_op_linear(A.data_ptr(), W.data_ptr(), _ptr(bias), _ptr(ln[0]) if ln else None,
                           _ptr(ln[1]) if ln else None, _ptr(res), out.data_ptr(), R, N, d,
                           {"none": 0, "gelu": 1}[act], _stream()), "sd_op_linear")
    return out


def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, extra: Optional[tuple] = None) -> Tensor:
    """q (B,Tq,d), k/v (B,S,d) contiguous; extra = (k_row (d,), v_row (d,)) shared by the batch."""
    lib = _lib.load()
    _req(q, "q"); _req(k, "k"); _req(v, "v")
    B, Tq, d = q.shape
    S = k.shape[1]
    out = torch.empty_like(q)
    check(lib.sd_op_attention(q.data_ptr(), d, k.data_ptr(), v.data_ptr(), d, _ptr(extra[0]) if extra else None,
                              _ptr(extra[1]) if extra else None, out.data_ptr(), d, B, Tq, S, d, heads, _stream()),
          "sd_op_attention")
    return out


def patch_embed(x: Tensor, w: Tensor, b: Tensor, pe: Tensor) -> Tensor:
    lib = _lib.load()
    _req(x, "x"); _req(w, "w"); _req(b, "b"); _req(pe, "pe")
    B, S, Cin = x.shape
    if w.dim() == 2:
        d, p = w.shape[0], 1
    else:
        d, _, p = w.shape
    out = torch.empty(B, S // p, d, dtype=torch.float32, device=x.device)
    check(lib.sd_op_patch_embed(x.data_ptr(), w.data_ptr(), b.data_ptr(), pe.data_ptr(), out.data_ptr(), B, S, Cin, p, d,
                                _stream()), "sd_op_patch_embed")
    return out


def fc_out(h: Tensor, W: Tensor, b: Tensor, x_io: Optional[Tensor] = None, coef4=None, want_eps: bool = True):
    lib = _lib.load()
    _req(h, "h"); _req(W, "W"); _req(b, "b")
    R, d = h.shape
    J = W.shape[0]
    eps = torch.empty(R, J, dtype=torch.float32, device=h.device) if want_eps else None
    cbuf = None
    if coef4 is not None:
        cbuf = (C.c_float * 4)(*[float(v) for v in coef4])
    check(lib.sd_op_fc_out(h.data_ptr(), W.data_ptr(), b.data_ptr(), _ptr(eps), _ptr(x_io),
                           C.cast(cbuf, _lib.c_float_p) if cbuf is not None else None, R, d, J, _stream()), "sd_op_fc_out")
    return eps


# ---- training ops (backward of the blocks, loss, optimizer) ------------------------------
def _rows(t: Tensor, name: str):
    """(data_ptr, row stride) of a tensor viewed as rows of its last dim; the last dim must
    be contiguous and all leading dims must collapse to one uniform row stride (true for
    column slices of a packed [.., 3d] buffer)."""
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected a float32 tensor on the MI355X; soccerdiffusion_amd has no CPU path")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: last dimension must be contiguous")
    ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
    for i in range(t.dim() - 2):
        if t.shape[i] != 1 and t.stride(i) != t.stride(i + 1) * t.shape[i + 1]:
            raise ValueError(f"{name}: rows are not uniformly strided")
    return t.data_ptr(), int(ld)


def linear_strided(A: Tensor, W: Tensor, bias: Optional[Tensor] = None, res: Optional[Tensor] = None,
                   out: Optional[Tensor] = None) -> Tensor:
    """out[R,N] = A W^T + bias (+res) where A may be a column slice (row stride > d)."""
    lib = _lib.load()
    ap, lda = _rows(A, "A")
    _req(W, "W")
    R = A.numel() // A.shape[-1]
    d = A.shape[-1]
    N = W.shape[0]
    if out is None:
        out = torch.empty(R, N, dtype=torch.float32, device=A.device)
    check(lib.sd_op_linear_strided(ap, lda, W.data_ptr(), _ptr(bias), None, None, _ptr(res), out.data_ptr(), R, N, d, 0,
                                   _stream()), "sd_op_linear_strided")
    return out


def pack_weight_blocks(src: Tensor, src_off: Tensor, n_blocks: int, d: int, dst: Tensor, transposed: bool = False) -> None:
    """Split ``n_blocks`` d x d fp32 blocks (block b at ``src`` + ``src_off[b]`` floats; device int64 offsets) - or their
    transposes - into the fp16 hi | lo fragment planes of ``linear_packed`` (2 d^2 halfs per block, consecutive in ``dst``),
    one launch."""
    lib = _lib.load()
    _req(src, "src")
    if src_off.dtype != torch.int64 or not src_off.is_cuda or dst.dtype != torch.float16 or dst.numel() < 2 * d * d * n_blocks:
        raise ValueError("pack_weight_blocks: src_off must be a device int64 tensor and dst a float16 tensor of 2 d^2 n_blocks")
    check(lib.sd_pack_weight_blocks(src.data_ptr(), src_off.data_ptr(), n_blocks, dst.data_ptr(), d, int(bool(transposed)), _stream()),
          "sd_pack_weight_blocks")


def linear_packed(A: Tensor, wpk: int, N: int, bias: Optional[Tensor] = None, ln: Optional[tuple] = None,
                  res: Optional[Tensor] = None, drop=None, out: Optional[Tensor] = None) -> Tensor:
    """out[R,N] = [res +] [dropout](LN?(A) W^T + bias) with ``wpk`` = the ADDRESS of the split planes of W's N / d blocks
    (pack_weight_blocks); A may be a column slice."""
    lib = _lib.load()
    ap, lda = _rows(A, "A")
    d = A.shape[-1]
    R = A.numel() // d
    if out is None:
        out = torch.empty(R, N, dtype=torch.float32, device=A.device)
    p, seed, site = _drop(drop)
    check(lib.sd_op_linear_packed(ap, lda, wpk, _ptr(bias), _ptr(ln[0]) if ln else None, _ptr(ln[1]) if ln else None, _ptr(res),
                                  out.data_ptr(), R, N, d, p, seed, site, _stream()), "sd_op_linear_packed")
    return out


def _addr(t) -> Optional[int]:
    """Address of a tensor / an int address / None, for the pointer fields of the chain argument structs."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("chain operands must be contiguous float32 tensors on the MI355X; soccerdiffusion_amd has no CPU path")
    return t.data_ptr()


def train_fwd_chain(R: int, d: int, h_in: Tensor, *, a=None, wo=None, bo=None, h_out=None, ln=None, n_out=None, w1=None, b1=None,
                    pre=None, u=None, w2=None, b2=None, h2_out=None, nln=None, nn_out=None, wn=None, bn=None, y_out=None,
                    n_next: int = 0, p: float = 0.0, seed: int = 0, sites=(0, 0, 0), amax=(None, None, None, None)) -> None:
    """One launch of sd_train_fwd_chain (include/soccerdiffusion_hip.h).  Weights (wo, w1, w2, wn) are ADDRESSES of split
    planes; ``ln`` / ``nln`` are (weight, bias) pairs; ``sites`` = (out-projection, GELU, FFN output) dropout sites;
    ``amax`` = addresses of the abs-max words of (a, n_out, u, nn_out) or None."""
    lib = _lib.load()
    args = _lib.TrainFwdChainArgs(
        R=R, d=d, n_next=n_next, a=_addr(a), wo=wo, bo=_addr(bo), h_in=_addr(h_in), h_out=_addr(h_out),
        ln_w=_addr(ln[0]) if ln else None, ln_b=_addr(ln[1]) if ln else None, n_out=_addr(n_out), w1=w1, b1=_addr(b1), pre=_addr(pre),
        u=_addr(u), w2=w2, b2=_addr(b2), h2_out=_addr(h2_out), nln_w=_addr(nln[0]) if nln else None,
        nln_b=_addr(nln[1]) if nln else None, nn_out=_addr(nn_out), wn=wn, bn=_addr(bn), y_out=_addr(y_out), p=float(p),
        seed=int(seed) & 0xFFFFFFFFFFFFFFFF, site_out=int(sites[0]), site_act=int(sites[1]), site_ffn=int(sites[2]),
        amax_a=amax[0], amax_n=amax[1], amax_u=amax[2], amax_nn=amax[3], amax_h2=amax[4] if len(amax) > 4 else None)
    check(lib.sd_train_fwd_chain(C.byref(args), _stream()), "sd_train_fwd_chain")


def pack_weight_traj(W: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """(N, 256) weight rows -> the split fp16 planes sd_train_layer_fwd streams (16 x 16 x 32 fragment order of csrc/sd_traj.h, scale 2^8)."""
    lib = _lib.load()
    _req(W, "W")
    N, K = W.shape
    if out is None:
        out = torch.empty(lib.sd_pack_weight_traj_halfs(N, K), dtype=torch.float16, device=W.device)
    check(lib.sd_pack_weight_traj(W.data_ptr(), N, K, out.data_ptr(), _stream()), "sd_pack_weight_traj")
    return out


def pack_weight_traj_multi(base: Tensor, src: Tensor, rows: Tensor, dst: Tensor, max_rows: int, planes: Tensor) -> None:
    """All registered weight slices of a flat parameter buffer -> their trajectory-kernel planes, one launch (device index arrays)."""
    check(_lib.load().sd_pack_weight_traj_multi(base.data_ptr(), src.data_ptr(), rows.data_ptr(), dst.data_ptr(), src.numel(), int(max_rows),
                                                planes.data_ptr(), _stream()), "sd_pack_weight_traj_multi")


def train_head_fwd(x: Tensor, w_emb: int, b_emb: Tensor, pe: Tensor, ln, w_qkv: int, b_qkv: Tensor, amax_n1: Optional[int]):
    """One launch of sd_train_head_fwd: (h0, n1, qkv) of the decoder stack's entry; w_emb / w_qkv are ADDRESSES of planes."""
    _req(x, "x")
    B, T, J = x.shape
    h0 = torch.empty(B, T, 256, dtype=torch.float32, device=x.device)
    n1 = torch.empty(B * T, 256, dtype=torch.float32, device=x.device)
    qkv = torch.empty(B, T, 768, dtype=torch.float32, device=x.device)
    check(_lib.load().sd_train_head_fwd(x.data_ptr(), w_emb, _addr(b_emb), _addr(pe), h0.data_ptr(), _addr(ln[0]), _addr(ln[1]), n1.data_ptr(), w_qkv,
                                        _addr(b_qkv), qkv.data_ptr(), amax_n1, B, T, J, _stream()), "sd_train_head_fwd")
    return h0, n1, qkv


def train_layer_fwd_ok(d: int, heads: int, T: int, M: int) -> bool:
    return bool(_lib.load().sd_train_layer_fwd_ok(d, heads, T, M))


def train_layer_fwd(B: int, T: int, M: int, heads: int, *, tensors: dict, weights: dict, p: float = 0.0, seed: int = 0, sites=(0,) * 6,
                    amax=(None,) * 7) -> None:
    """One launch of sd_train_layer_fwd: ``tensors`` / ``weights`` map the field names of sd_train_layer_fwd_args to tensors (weights
    w_*: ADDRESSES of planes from pack_weight_traj); ``sites`` = dropout sites (self-attention probabilities, its out-projection, cross-
    attention probabilities, its out-projection, GELU, FFN output); ``amax`` = addresses of the abs-max words of (a_sa, n2, a_ca, nf, u,
    nn1, h3) or None."""
    lib = _lib.load()
    kw = {k: (_addr(v) if isinstance(v, Tensor) else v) for k, v in {**tensors, **weights}.items()}
    args = _lib.TrainLayerFwdArgs(B=B, T=T, M=M, d=256, heads=heads, p=float(p), seed=int(seed) & 0xFFFFFFFFFFFFFFFF,
                                  site_sa_probs=int(sites[0]), site_sa_out=int(sites[1]), site_ca_probs=int(sites[2]), site_ca_out=int(sites[3]),
                                  site_act=int(sites[4]), site_ffn=int(sites[5]), amax_a_sa=amax[0], amax_n2=amax[1], amax_a_ca=amax[2],
                                  amax_nf=amax[3], amax_u=amax[4], amax_nn=amax[5], amax_out=amax[6], **kw)
    check(lib.sd_train_layer_fwd(C.byref(args), _stream()), "sd_train_layer_fwd")


def train_bwd_chain(R: int, d: int, dy: Tensor, wt: int, dx: Tensor, *, passes: int = 1, dym=None, pre=None, dpre=None, wt1=None,
                    x=None, ln_w=None, dres=None, dg=None, db=None, p: float = 0.0, seed: int = 0, sites=(0, 0),
                    amax=(None, None)) -> None:
    """One launch of sd_train_bwd_chain.  ``dy`` (R, passes * d) may be a row-strided view; ``wt`` / ``wt1`` are ADDRESSES of
    the split planes of the transposed blocks; ``sites`` = (mask of dy, mask after the GELU); ``amax`` = addresses of the
    abs-max words of (masked dy, dpre) or None."""
    lib = _lib.load()
    dyp, ldy = _rows(dy, "dy")
    args = _lib.TrainBwdChainArgs(
        R=R, d=d, passes=passes, ldy=ldy, dy=dyp, dym=_addr(dym), wt=wt, pre=_addr(pre), dpre=_addr(dpre), wt1=wt1, x=_addr(x),
        ln_w=_addr(ln_w), dres=_addr(dres), dg=_addr(dg), db=_addr(db), dx=_addr(dx), p=float(p), seed=int(seed) & 0xFFFFFFFFFFFFFFFF,
        site_in=int(sites[0]), site_act=int(sites[1]), amax_dy=amax[0], amax_dpre=amax[1], amax_dx=amax[2] if len(amax) > 2 else None)
    check(lib.sd_train_bwd_chain(C.byref(args), _stream()), "sd_train_bwd_chain")


def absmax(x: Tensor, amax: int) -> None:
    """Max the bits of max |x| into the SD_AMAX_WORDS words at address ``amax`` (zeroed by the caller); x may be row-strided."""
    lib = _lib.load()
    xp, ld = _rows(x, "x")
    width = x.shape[-1]
    check(lib.sd_op_absmax(xp, x.numel() // width, width, ld, amax, _stream()), "sd_op_absmax")


def gemm_tn_grouped(problems) -> None:
    """dW += dY^T X (db += column sums of dY) for every (dY, X, dW, db, amax_dy, amax_x) in ``problems`` in one launch per 8.
    dY / X may be row-strided views; ``amax_*`` are ADDRESSES of device words with the bits of max |dY| / max |X|."""
    lib = _lib.load()
    arr = (_lib.GemmTnProblem * len(problems))()
    for q, (dY, X, dW, db, ay, ax) in zip(arr, problems):
        yp, ldy = _rows(dY, "dY"); xp, ldx = _rows(X, "X")
        _req(dW, "dW")
        q.dY, q.X, q.dW, q.db, q.amax_dy, q.amax_x = yp, xp, dW.data_ptr(), _ptr(db), ay, ax
        q.R, q.N, q.K = dY.numel() // dY.shape[-1], dY.shape[-1], X.shape[-1]
        q.ldy, q.ldx, q.ldw = ldy, ldx, dW.stride(0)
    check(lib.sd_gemm_tn_grouped(arr, len(problems), _stream()), "sd_gemm_tn_grouped")


# ---- dropout (training; one Philox mask function shared by every kernel, include/soccerdiffusion_hip.h) ----------------
def _drop(drop) -> tuple:
    """(p, seed, site) -> ctypes-ready triple; None = no dropout."""
    if drop is None:
        return 0.0, 0, 0
    p, seed, site = drop
    return float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(site) & 0xFFFFFFFFFFFFFFFF


def dropout(x: Tensor, drop, out: Optional[Tensor] = None) -> Tensor:
    """x o mask over the last dim as the mask width (rows = everything else)."""
    lib = _lib.load()
    _req(x, "x")
    width = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    p, seed, site = _drop(drop)
    check(lib.sd_op_dropout(x.data_ptr(), out.data_ptr(), x.numel() // width, width, p, seed, site, _stream()), "sd_op_dropout")
    return out


def dropout_mask(rows: int, width: int, drop, device) -> Tensor:
    """The (rows, width) multiplier tensor (0 or 1/(1-p)) the kernels apply for this (p, seed, site)."""
    lib = _lib.load()
    mask = torch.empty(rows, width, dtype=torch.float32, device=device)
    p, seed, site = _drop(drop)
    check(lib.sd_op_dropout_mask(mask.data_ptr(), rows, width, p, seed, site, _stream()), "sd_op_dropout_mask")
    return mask


def gelu_dropout_fwd(pre: Tensor, drop) -> Tensor:
    lib = _lib.load()
    _req(pre, "pre")
    out = torch.empty_like(pre)
    width = pre.shape[-1]
    p, seed, site = _drop(drop)
    check(lib.sd_op_gelu_dropout_fwd(pre.data_ptr(), out.data_ptr(), pre.numel() // width, width, p, seed, site, _stream()),
          "sd_op_gelu_dropout_fwd")
    return out


def gelu_dropout_bwd(dy: Tensor, pre: Tensor, drop) -> Tensor:
    lib = _lib.load()
    _req(dy, "dy"); _req(pre, "pre")
    out = torch.empty_like(pre)
    width = pre.shape[-1]
    p, seed, site = _drop(drop)
    check(lib.sd_op_gelu_dropout_bwd(dy.data_ptr(), pre.data_ptr(), out.data_ptr(), pre.numel() // width, width, p, seed, site,
                                     _stream()), "sd_op_gelu_dropout_bwd")
    return out


def linear_dropout(A: Tensor, W: Tensor, bias: Optional[Tensor], res: Tensor, drop, out: Optional[Tensor] = None) -> Tensor:
    """out = res + dropout(A W^T + bias): the fused epilogue of the split-fp16 panel GEMM."""
    lib = _lib.load()
    _req(A, "A"); _req(W, "W"); _req(res, "res")
    R, d = A.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(R, N, dtype=torch.float32, device=A.device)
    p, seed, site = _drop(drop)
    check(lib.sd_op_linear_dropout(A.data_ptr(), d, W.data_ptr(), _ptr(bias), res.data_ptr(), out.data_ptr(), R, N, d, p, seed, site,
                                   _stream()), "sd_op_linear_dropout")
    return out


def attention_lse(q: Tensor, k: Tensor, v: Tensor, heads: int, drop=None):
    """Forward attention that also returns lse2 (B, heads, Tq).  q (B,Tq,d), k/v (B,S,d) may be
    column-slice views of packed buffers.  ``drop`` = (p, seed, site): dropout on the probabilities."""
    lib = _lib.load()
    qp, ldq = _rows(q, "q"); kp, ldk = _rows(k, "k"); vp, ldv = _rows(v, "v")
    if ldk != ldv:
        raise ValueError("k and v must share a row stride")
    B, Tq, d = q.shape
    S = k.shape[1]
    out = torch.empty(B, Tq, d, dtype=torch.float32, device=q.device)
    lse = torch.empty(B, heads, Tq, dtype=torch.float32, device=q.device)
    p, seed, site = _drop(drop)
    check(lib.sd_op_attention_lse_dropout(qp, ldq, kp, vp, ldk, out.data_ptr(), d, lse.data_ptr(), B, Tq, S, d, heads, p, seed, site,
                                          _stream()), "sd_op_attention_lse")
    return out, lse


def attention_bwd(q: Tensor, k: Tensor, v: Tensor, o: Tensor, dO: Tensor, lse: Tensor, dq: Tensor, dk: Tensor, dv: Tensor,
                  heads: int, drop=None) -> None:
    lib = _lib.load()
    qp, ldq = _rows(q, "q"); kp, ldk = _rows(k, "k"); vp, ldv = _rows(v, "v")
    op, ldo = _rows(o, "o"); dop, lddo = _rows(dO, "dO")
    dqp, lddq = _rows(dq, "dq"); dkp, lddk = _rows(dk, "dk"); dvp, lddv = _rows(dv, "dv")
    if ldk != ldv or lddk != lddv:
        raise ValueError("k/v (and dk/dv) must share a row stride")
    B, Tq, d = q.shape
    S = k.shape[1]
    p, seed, site = _drop(drop)
    check(lib.sd_op_attention_bwd_dropout(qp, ldq, kp, vp, ldk, op, ldo, dop, lddo, lse.data_ptr(), dqp, lddq, dkp, dvp, lddk,
                                          B, Tq, S, d, heads, p, seed, site, _stream()), "sd_op_attention_bwd")


def gemm_tn(dY: Tensor, X: Tensor, dW: Tensor, db: Optional[Tensor] = None) -> None:
    """dW[N,K] += dY[R,N]^T X[R,K]; db[N] += colsum(dY).  dY / X may be column slices."""
    lib = _lib.load()
    yp, ldy = _rows(dY, "dY"); xp, ldx = _rows(X, "X")
    _req(dW, "dW")
    N, K = dY.shape[-1], X.shape[-1]
    R = dY.numel() // N
    if X.numel() // K != R or tuple(dW.shape) != (N, K):
        raise ValueError("gemm_tn: shape mismatch")
    check(lib.sd_op_gemm_tn(yp, ldy, xp, ldx, dW.data_ptr(), K, _ptr(db), R, N, K, _stream()), "sd_op_gemm_tn")


def layernorm_fwd(x: Tensor, g: Tensor, b: Tensor, want_stats: bool = True):
    lib = _lib.load()
    _req(x, "x")
    d = x.shape[-1]
    R = x.numel() // d
    y = torch.empty_like(x)
    mean = torch.empty(R, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(R, dtype=torch.float32, device=x.device) if want_stats else None
    check(lib.sd_op_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), _ptr(mean), _ptr(rstd), R, d,
                                  _stream()), "sd_op_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, g: Tensor, dres: Optional[Tensor] = None):
    lib = _lib.load()
    _req(dy, "dy"); _req(x, "x")
    d = x.shape[-1]
    R = x.numel() // d
    dx = torch.empty_like(x)
    dg = torch.zeros(d, dtype=torch.float32, device=x.device)
    db = torch.zeros(d, dtype=torch.float32, device=x.device)
    check(lib.sd_op_layernorm_bwd(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), _ptr(dres),
                                  dx.data_ptr(), dg.data_ptr(), db.data_ptr(), R, d, _stream()), "sd_op_layernorm_bwd")
    return dx, dg, db


def layernorm_bwd_into(dy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, g: Tensor, dg: Tensor, db: Tensor,
                       dres: Optional[Tensor] = None) -> Tensor:
    """LayerNorm backward that ACCUMULATES the affine gradients into existing dg / db buffers; ``dres`` (same shape as x)
    is added to dx inside the kernel (the residual branch's gradient)."""
    lib = _lib.load()
    _req(dy, "dy"); _req(x, "x")
    if dres is not None:
        _req(dres, "dres")
    d = x.shape[-1]
    R = x.numel() // d
    dx = torch.empty_like(x)
    check(lib.sd_op_layernorm_bwd(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), _ptr(dres),
                                  dx.data_ptr(), dg.data_ptr(), db.data_ptr(), R, d, _stream()), "sd_op_layernorm_bwd")
    return dx


def gelu_fwd(pre: Tensor) -> Tensor:
    lib = _lib.load()
    _req(pre, "pre")
    out = torch.empty_like(pre)
    check(lib.sd_op_gelu_fwd(pre.data_ptr(), out.data_ptr(), pre.numel(), _stream()), "sd_op_gelu_fwd")
    return out


def gelu_bwd(dy: Tensor, pre: Tensor) -> Tensor:
    lib = _lib.load()
    _req(dy, "dy"); _req(pre, "pre")
    out = torch.empty_like(pre)
    check(lib.sd_op_gelu_bwd(dy.data_ptr(), pre.data_ptr(), out.data_ptr(), pre.numel(), _stream()), "sd_op_gelu_bwd")
    return out


def colsum(src: Tensor, out: Tensor) -> None:
    """out[c] += sum_r src[r, c] for a (possibly column-sliced) 2-D / 3-D src."""
    lib = _lib.load()
    sp, ld = _rows(src, "src")
    width = src.shape[-1]
    check(lib.sd_op_colsum(sp, ld, src.numel() // width, width, out.data_ptr(), _stream()), "sd_op_colsum")


def small_k_matmul(A: Tensor, Bm: Tensor) -> Tensor:
    lib = _lib.load()
    _req(A, "A"); _req(Bm, "B")
    K, N = Bm.shape
    R = A.numel() // K
    out = torch.empty(R, N, dtype=torch.float32, device=A.device)
    check(lib.sd_op_small_k_matmul(A.data_ptr(), Bm.data_ptr(), out.data_ptr(), R, K, N, _stream()), "sd_op_small_k_matmul")
    return out


def mse_loss(pred: Tensor, target: Tensor, want_grad: bool = True):
    """(loss (1,), grad or None): F.mse_loss mean reduction and 2 (pred - target) / n."""
    lib = _lib.load()
    _req(pred, "pred"); _req(target, "target")
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    scratch = torch.empty(256, dtype=torch.float64, device=pred.device)
    check(lib.sd_mse_loss(pred.data_ptr(), target.data_ptr(), loss.data_ptr(), _ptr(grad), scratch.data_ptr(), pred.numel(),
                          _stream()), "sd_mse_loss")
    return loss, grad


def adamw_step_dev(p: Tensor, g: Tensor, m: Tensor, v: Tensor, hyper7: Tensor) -> None:
    """AdamW update with its seven scalars in device memory (graph-capturable; see adamw_hyper)."""
    lib = _lib.load()
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (hyper7, "hyper7")):
        _req(t, n)
    if hyper7.numel() < 7:
        raise ValueError("hyper7 needs 7 floats")
    check(lib.sd_adamw_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), hyper7.data_ptr(), _stream()),
          "sd_adamw_step_dev")


def adamw_hyper(lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int, out: Tensor) -> None:
    """Fills ``out[:7]`` (a CPU float32 tensor, e.g. pinned) with the scalars sd_adamw_step derives for update ``step``."""
    if out.is_cuda or out.dtype != torch.float32 or out.numel() < 7 or not out.is_contiguous():
        raise ValueError("out must be a contiguous CPU float32 tensor with >= 7 elements")
    check(_lib.load().sd_adamw_hyper(lr, beta1, beta2, eps, weight_decay, step, C.cast(out.data_ptr(), _lib.c_float_p)), "sd_adamw_hyper")


def set_dropout_epoch(word: Optional[Tensor]) -> None:
    """Process-wide: the uint32 device word every dropout kernel adds to its Philox key at run time (None = off)."""
    if word is not None and (not word.is_cuda or word.element_size() != 4 or word.numel() < 1):
        raise ValueError("the epoch word must be a 4-byte element on the device")
    check(_lib.load().sd_set_dropout_epoch(None if word is None else word.data_ptr()), "sd_set_dropout_epoch")


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, beta1: float, beta2: float, eps: float,
               weight_decay: float, step: int) -> None:
    lib = _lib.load()
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, n)
    check(lib.sd_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                            weight_decay, step, _stream()), "sd_adamw_step")


# ---- image path: ResNet basic-block convolution (csrc/sd_conv.hip) -------------------------------------
class PackedConv3x3:
    """A 3 x 3 (or 1 x 1) convolution weight (Cout, Cin, k, k) in the fragment order of ``sd_conv3x3_bn_act`` / ``sd_conv_s2_bn_act`` plus
    the power-of-two scale the fp16 planes carry; repacked when the weight's version counter moves."""

    def __init__(self, weight: Tensor):
        lib = _lib.load()
        _req(weight, "weight")
        Cout, Cin, kh, kw = weight.shape
        if kh != kw or kh not in (1, 3) or Cout % 64 or Cin % 64:
            raise ValueError("3 x 3 or 1 x 1 kernels with channel counts that are multiples of 64")
        self.Cout, self.Cin, self.ksize = Cout, Cin, kh
        self.planes = torch.empty(lib.sd_conv_packed_halfs(Cout, Cin, kh), dtype=torch.float16, device=weight.device)
        self.scale = torch.empty(1, dtype=torch.float32, device=weight.device)
        self._word = torch.zeros(1, dtype=torch.int32, device=weight.device)
        self.version = None
        self.refresh(weight)

    def refresh(self, weight: Tensor) -> "PackedConv3x3":
        if self.version != (weight._version, _weights_generation):
            w = weight.detach().contiguous()
            check(_lib.load().sd_conv_pack(w.data_ptr(), self.Cout, self.Cin, self.ksize, self.planes.data_ptr(), self.scale.data_ptr(),
                                           self._word.data_ptr(), _stream()), "sd_conv_pack")
            self.version = (weight._version, _weights_generation)
        return self


def absmax_word(x: Tensor, word: Optional[Tensor] = None, zero: bool = True) -> Tensor:
    """The bits of max |x| in one int32 device word (the activation scale ``conv3x3_bn_act`` derives for its fp16 planes)."""
    _req(x, "x")
    if word is None:
        word = torch.zeros(1, dtype=torch.int32, device=x.device)
    elif zero:
        word.zero_()
    check(_lib.load().sd_absmax_word(x.data_ptr(), x.numel(), word.data_ptr(), _stream()), "sd_absmax_word")
    return word


def conv3x3_bn_act(x: Tensor, x_amax: Tensor, w: PackedConv3x3, bn_scale: Tensor, bn_shift: Tensor, res: Optional[Tensor] = None,
                   relu: bool = True, y_amax: Optional[Tensor] = None, zero_amax: bool = True) -> Tensor:
    """relu(conv(x) * bn_scale + bn_shift (+ res)) on NHWC fp32 tensors, stride 1: 3 x 3 / padding 1 or 1 x 1 (``w.ksize``) - torchvision
    BasicBlock's / Bottleneck's conv / bn / relu in inference mode (reference: soccer_diffusion/ml/model/encoder/image.py:55-83).  ``x_amax``: word from
    ``absmax_word`` or the ``y_amax`` of the launch that produced x; ``y_amax`` (zeroed here unless ``zero_amax`` is False: the caller
    zeroed it, e.g. all words of a forward in one fill) receives max |y|."""
    _req(x, "x"); _req(bn_scale, "bn_scale"); _req(bn_shift, "bn_shift")
    N, H, W, Cin = x.shape
    if Cin != w.Cin:
        raise ValueError("channel mismatch")
    y = torch.empty(N, H, W, w.Cout, dtype=torch.float32, device=x.device)
    if res is not None:
        _req(res, "res")
        if res.shape != y.shape:
            raise ValueError("residual shape mismatch")
    if y_amax is not None and zero_amax:
        y_amax.zero_()
    fn = _lib.load().sd_conv3x3_bn_act if w.ksize == 3 else _lib.load().sd_conv1x1_bn_act   # (a 1 x 1 weight: Bottleneck projections)
    check(fn(x.data_ptr(), w.planes.data_ptr(), w.scale.data_ptr(), x_amax.data_ptr(), bn_scale.data_ptr(),
             bn_shift.data_ptr(), _ptr(res), y.data_ptr(), _ptr(y_amax), N, H, W, Cin, w.Cout, int(relu), _stream()),
          "sd_conv3x3_bn_act" if w.ksize == 3 else "sd_conv1x1_bn_act")
    return y


def conv_s2_bn_act(x: Tensor, x_amax: Tensor, w: PackedConv3x3, bn_scale: Tensor, bn_shift: Tensor, relu: bool = True,
                   y_amax: Optional[Tensor] = None, zero_amax: bool = True) -> Tensor:
    """act(conv(x; stride 2) * bn_scale + bn_shift) on NHWC fp32 tensors: the 3 x 3 / padding 1 convolution that opens ResNet layers 2 - 4 or
    their 1 x 1 shortcut (``w.ksize``), inference BatchNorm folded (reference: torchvision BasicBlock via
    soccer_diffusion/ml/model/encoder/image.py:55-83).  Same conventions as ``conv3x3_bn_act``."""
    _req(x, "x"); _req(bn_scale, "bn_scale"); _req(bn_shift, "bn_shift")
    N, H, W, Cin = x.shape
    if Cin != w.Cin:
        raise ValueError("channel mismatch")
    y = torch.empty(N, (H + 1) // 2, (W + 1) // 2, w.Cout, dtype=torch.float32, device=x.device)
    if y_amax is not None and zero_amax:
        y_amax.zero_()
    check(_lib.load().sd_conv_s2_bn_act(x.data_ptr(), w.planes.data_ptr(), w.scale.data_ptr(), x_amax.data_ptr(), bn_scale.data_ptr(),
                                        bn_shift.data_ptr(), y.data_ptr(), _ptr(y_amax), N, H, W, Cin, w.Cout, w.ksize, int(relu), _stream()),
          "sd_conv_s2_bn_act")
    return y


class PackedStem:
    """ResNet's first convolution (64, 3, 7, 7) in the fragment order of ``sd_stem_conv_bn_relu_pool``; repacked when the weight moves."""

    def __init__(self, weight: Tensor):
        lib = _lib.load()
        _req(weight, "weight")
        if tuple(weight.shape) != (64, 3, 7, 7):
            raise ValueError("the ResNet stem convolution is (64, 3, 7, 7)")
        self.planes = torch.empty(lib.sd_stem_packed_halfs(), dtype=torch.float16, device=weight.device)
        self.scale = torch.empty(1, dtype=torch.float32, device=weight.device)
        self._word = torch.zeros(1, dtype=torch.int32, device=weight.device)
        self.version = None
        self.refresh(weight)

    def refresh(self, weight: Tensor) -> "PackedStem":
        if self.version != (weight._version, _weights_generation):
            w = weight.detach().contiguous()
            check(_lib.load().sd_stem_pack(w.data_ptr(), self.planes.data_ptr(), self.scale.data_ptr(), self._word.data_ptr(), _stream()),
                  "sd_stem_pack")
            self.version = (weight._version, _weights_generation)
        return self


def stem_conv_bn_relu_pool(x: Tensor, x_amax: Tensor, w: PackedStem, bn_scale: Tensor, bn_shift: Tensor,
                           y_amax: Optional[Tensor] = None, zero_amax: bool = True) -> Tensor:
    """maxpool3x3/s2/p1(relu(conv7x7/s2/p3(x) * bn_scale + bn_shift)): NCHW frames (N, 3, H, W) -> NHWC map (N, Hp, Wp, 64) in one
    launch - torchvision ResNet's conv1 / bn1 / relu / maxpool in inference mode (reference: soccer_diffusion/ml/model/encoder/image.py:55-83)."""
    _req(x, "x"); _req(bn_scale, "bn_scale"); _req(bn_shift, "bn_shift")
    N, C3, H, W = x.shape
    if C3 != 3:
        raise ValueError("the stem takes 3-channel frames")
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(N, (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1, 64, dtype=torch.float32, device=x.device)
    if y_amax is not None and zero_amax:
        y_amax.zero_()
    check(_lib.load().sd_stem_conv_bn_relu_pool(x.data_ptr(), w.planes.data_ptr(), w.scale.data_ptr(), x_amax.data_ptr(), bn_scale.data_ptr(),
                                                bn_shift.data_ptr(), y.data_ptr(), _ptr(y_amax), N, H, W, _stream()),
          "sd_stem_conv_bn_relu_pool")
    return y
