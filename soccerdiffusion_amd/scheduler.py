"""DDIMScheduler: the subset of ``diffusers.DDIMScheduler`` (0.31.0) the reference uses,
restated natively (reference call sites: ml/training/train.py:185-186,218;
ml/inference/plot.py:71-73,124-131; ml/training/distill.py:151-152,179-189).

Constructed exactly like the reference does — ``DDIMScheduler(beta_schedule=
"squaredcos_cap_v2", clip_sample=False)`` — everything else is the diffusers default
(1000 train steps, epsilon prediction, leading spacing, eta = 0, final alpha 1).
diffusers itself is not available offline: scheduler parity is UNPINNED."""

from __future__ import annotations

from types import SimpleNamespace

import torch

from . import ops


class DDIMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_schedule: str = "squaredcos_cap_v2",
                 clip_sample: bool = False, prediction_type: str = "epsilon"):
        if beta_schedule != "squaredcos_cap_v2" or clip_sample or prediction_type != "epsilon":
            raise NotImplementedError("only the configuration the reference uses is implemented: "
                                      "beta_schedule='squaredcos_cap_v2', clip_sample=False, epsilon prediction")
        self.config = {"num_train_timesteps": num_train_timesteps, "beta_schedule": beta_schedule,
                       "clip_sample": clip_sample, "prediction_type": prediction_type}
        self._table_len = num_train_timesteps
        self.alphas_cumprod = ops.alphas_cumprod(num_train_timesteps)  # CPU fp32, like diffusers
        self._acp_dev: dict = {}
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def _check_config(self):
        # the reference overwrites config["num_train_timesteps"] after construction (train.py:186);
        # every shipped YAML uses 1000 = the table length.  Any other value is undefined there.
        if self.config["num_train_timesteps"] != self._table_len:
            raise ValueError("train_denoising_timesteps must equal the scheduler table length (1000)")

    def _acp(self, device) -> torch.Tensor:
        key = torch.device(device)
        if key not in self._acp_dev:
            self._acp_dev[key] = self.alphas_cumprod.to(key)
        return self._acp_dev[key]

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        self._check_config()
        return ops.ddim_add_noise(original_samples.contiguous(), noise.contiguous(),
                                  timesteps.to(device=original_samples.device, dtype=torch.int64).contiguous(),
                                  self._acp(original_samples.device))

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        self._check_config()
        if num_inference_steps > self._table_len:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.tensor(ops.ddim_timesteps(num_inference_steps, self._table_len), dtype=torch.int64,
                                      device=device)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor):
        if self.num_inference_steps is None:
            raise ValueError("call set_timesteps before step")
        t = int(timestep)
        coef = ops.ddim_coefficients([t], self.alphas_cumprod, self.num_inference_steps, self._table_len)[0]
        return SimpleNamespace(prev_sample=ops.ddim_step(model_output.contiguous(), sample.contiguous(), coef))
