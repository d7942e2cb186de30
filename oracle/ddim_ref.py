"""CPU oracle for the DDIM scheduler arithmetic.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference takes this arithmetic from the third-party dependency
``diffusers==0.31.0`` (reference ``poetry.lock:447-448``), which is not vendored under the
reference checkout and is not installed in the build container; no reference test holds
a known-answer vector for it.  This file restates the published DDIM algorithm
(Song et al. 2021, eq. 12 with eta = 0) with the constructor arguments the reference
passes at ``soccer_diffusion/ml/training/train.py:185-186`` and the documented diffusers
defaults.  The only anchors are the reference's call sites:

* ``DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)``  train.py:185
* ``scheduler.add_noise(x0, noise, t)``                                      train.py:218
* ``scheduler.set_timesteps(n)`` / ``scheduler.timesteps``                   plot.py:73,124
* ``scheduler.step(noise_pred, t, x).prev_sample``                           plot.py:131
"""

from __future__ import annotations

import math

import torch

Tensor = torch.Tensor


def alphas_cumprod(num_train_timesteps: int = 1000, max_beta: float = 0.999) -> Tensor:
    """``betas_for_alpha_bar`` (cosine / squaredcos_cap_v2): betas in python float64 ->
    fp32 tensor; cumulative product in fp32."""

    def alpha_bar(s: float) -> float:
        return math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2

    n = num_train_timesteps
    betas = [min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)]
    return torch.cumprod(1.0 - torch.tensor(betas, dtype=torch.float32), dim=0)


def timesteps(num_inference_steps: int, num_train_timesteps: int = 1000) -> Tensor:
    """``set_timesteps`` with timestep_spacing="leading", steps_offset=0."""
    ratio = num_train_timesteps // num_inference_steps
    return (torch.arange(num_inference_steps) * ratio).round().flip(0).to(torch.int64)


def add_noise(x0: Tensor, noise: Tensor, t: Tensor, acp: Tensor) -> Tensor:
    """x_t = sqrt(acp[t]) x0 + sqrt(1-acp[t]) eps, per-sample t broadcast over (T, J)."""
    a = acp.to(x0.dtype)[t.long()]
    sa = a.sqrt().view(-1, *([1] * (x0.dim() - 1)))
    sb = (1 - a).sqrt().view(-1, *([1] * (x0.dim() - 1)))
    return sa * x0 + sb * noise


def step_coefficients(t: int, num_inference_steps: int, acp: Tensor, num_train_timesteps: int = 1000):
    """Scalars of one eta=0 step: (a_t, a_prev) with final_alpha_cumprod = 1.0."""
    prev = t - num_train_timesteps // num_inference_steps
    a_t = acp[t]
    a_p = acp[prev] if prev >= 0 else torch.tensor(1.0, dtype=acp.dtype)
    return a_t, a_p


def step(eps_hat: Tensor, t: int, x: Tensor, num_inference_steps: int, acp: Tensor) -> Tensor:
    """prev_sample of ``DDIMScheduler.step`` (epsilon prediction, eta=0, no clipping)."""
    a_t, a_p = step_coefficients(int(t), num_inference_steps, acp)
    a_t = a_t.to(x.dtype)
    a_p = a_p.to(x.dtype)
    x0_hat = (x - (1 - a_t).sqrt() * eps_hat) / a_t.sqrt()
    return a_p.sqrt() * x0_hat + (1 - a_p).sqrt() * eps_hat


def sample(denoise, x_T: Tensor, num_inference_steps: int, acp: Tensor | None = None) -> list[Tensor]:
    """The reference's sampling loop (``plot.py:122-131``, ``distill.py:179-189``):
    ``denoise(x, t)`` -> eps_hat.  Returns x after every step (last = sample)."""
    acp = alphas_cumprod() if acp is None else acp
    x = x_T
    out = []
    for t in timesteps(num_inference_steps).tolist():
        x = step(denoise(x, t), t, x, num_inference_steps, acp)
        out.append(x)
    return out
