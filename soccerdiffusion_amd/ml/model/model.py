"""End2EndDiffusionTransformer — the boundary class of the hot path
(reference: soccer_diffusion/ml/model/model.py:16-179).

Same keyword constructor, same three methods, same buffers (``mean``, ``std``) and the
same ``state_dict`` keys, so reference checkpoints load unchanged.  Extra (not in the
reference class): ``sample`` runs the whole DDIM loop natively."""

from __future__ import annotations

import weakref
from typing import Optional, Sequence

import torch
from torch import nn

from ... import ops
from .decoder import DiffusionActionGenerator
from .encoder.encoders import GameStateEncoder, IMUEncoder, JointEncoder
from .encoder.image import ImageEncoderType, SequenceEncoderType, image_sequence_encoder_factory
from .misc import StepToken

NUM_HEADS = 4  # fixed by the reference (model.py:57,71,85,115)

# per-model caches (captured rollout graphs, the loop form's prepared workspaces): outside the module, keyed weakly by it, so that
# copy.deepcopy / pickling of a model neither carries nor shares them
_CACHES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _model_cache(model: nn.Module, name: str) -> dict:
    per = _CACHES.get(model)
    if per is None:
        per = _CACHES[model] = {}
    return per.setdefault(name, {})


class End2EndDiffusionTransformer(nn.Module):
    def __init__(
        self,
        num_joints: int,
        hidden_dim: int,
        use_action_history: bool,
        num_action_history_encoder_layers: int,
        max_action_context_length: int,
        encoder_patch_size: int,
        use_imu: bool,
        imu_orientation_embedding_method,
        num_imu_encoder_layers: int,
        imu_context_length: int,
        use_joint_states: bool,
        joint_state_encoder_layers: int,
        joint_state_context_length: int,
        use_images: bool,
        image_encoder_type: ImageEncoderType,
        image_sequence_encoder_type: SequenceEncoderType,
        num_image_sequence_encoder_layers: int,
        image_context_length: int,
        image_use_final_avgpool: bool,
        image_resolution: int,
        use_gamestate: bool,
        num_decoder_layers: int,
        trajectory_prediction_length: int,
    ):
        super().__init__()
        d = hidden_dim
        self.hidden_dim, self.num_joints = d, num_joints
        self.trajectory_prediction_length = trajectory_prediction_length
        self.step_encoding = StepToken(d)

        def seq(kind, enabled, *args):
            return kind(*args) if enabled else None

        self.action_history_encoder = seq(JointEncoder, use_action_history, num_joints, encoder_patch_size, d,
                                          num_action_history_encoder_layers, NUM_HEADS, max_action_context_length)
        self.imu_encoder = seq(IMUEncoder, use_imu, imu_orientation_embedding_method, encoder_patch_size, d,
                               num_imu_encoder_layers, NUM_HEADS, imu_context_length)
        self.joint_states_encoder = seq(JointEncoder, use_joint_states, num_joints, encoder_patch_size, d,
                                        joint_state_encoder_layers, NUM_HEADS, joint_state_context_length)
        self.image_sequence_encoder = (
            image_sequence_encoder_factory(
                encoder_type=image_sequence_encoder_type, image_encoder_type=image_encoder_type, hidden_dim=d,
                num_layers=num_image_sequence_encoder_layers, max_seq_len=image_context_length,
                use_final_avgpool=image_use_final_avgpool, resolution=image_resolution)
            if use_images else None
        )
        self.game_state_encoder = GameStateEncoder(d) if use_gamestate else None
        self.diffusion_action_generator = DiffusionActionGenerator(
            num_joints=num_joints, hidden_dim=d, num_layers=num_decoder_layers, num_heads=NUM_HEADS,
            max_seq_len=trajectory_prediction_length)
        self.register_buffer("mean", torch.zeros(num_joints))
        self.register_buffer("std", torch.ones(num_joints))

    # ---- reference API -----------------------------------------------------------------
    def encode_input_data(self, input_data: dict[str, torch.Tensor]) -> list[torch.Tensor]:
        """Context tokens per enabled modality, in the reference's fixed order
        [action history, IMU, joint state, images, game state] (model.py:135-144)."""
        pairs = ((self.action_history_encoder, "joint_command_history"), (self.imu_encoder, "rotation"),
                 (self.joint_states_encoder, "joint_state"), (self.image_sequence_encoder, "image_data"),
                 (self.game_state_encoder, "game_state"))
        return [enc(input_data[key]) for enc, key in pairs if enc is not None]

    def forward(self, input_data: dict[str, torch.Tensor], noisy_action_predictions: torch.Tensor,
                step: torch.Tensor) -> torch.Tensor:
        return self.forward_with_context(self.encode_input_data(input_data), noisy_action_predictions, step)

    def forward_with_context(self, context: Sequence[torch.Tensor], noisy_action_predictions: torch.Tensor,
                             step: torch.Tensor) -> torch.Tensor:
        """Predicted noise for x_t given precomputed context tokens; the step token is the
        LAST memory row (model.py:176).  ``step`` is int64 or float, shape (B,) (or (1,) at B=1)."""
        x = noisy_action_predictions
        B = x.shape[0]
        if step.dim() == 0:
            step = step.reshape(1)
        if step.shape[0] != B and step.shape[0] != 1:
            raise RuntimeError(f"step has {step.shape[0]} entries for a batch of {B}")  # torch.cat would fail too
        step = step.to(x.device)
        eps = self._loop_form_eps(context, x, step)
        if eps is not None:
            return eps
        if step.shape[0] != B:
            step = step.expand(B)
        memory = self._assemble_memory(context, step, B, x.device)
        return self.diffusion_action_generator(x, memory)

    def _loop_form_eps(self, context, x: torch.Tensor, step: torch.Tensor) -> Optional[torch.Tensor]:
        """Inference calls (no tape, no live dropout) on shapes the trajectory kernels take run ``ops.LoopSampler``: the reference's
        loops call this method once per DDIM step with the same context (plot.py:122-131, distill.py:179-189, ros.py:301-310), so the
        weights' split planes and the context's folded K / V are prepared on the first call and reused by the following ones.
        Returns None where that route does not apply (training, a differentiable input, CPU tensors, other shapes)."""
        import os

        dag = self.diffusion_action_generator
        if (not x.is_cuda or x.dtype != torch.float32 or x.dim() != 3 or (self.training and dag.dropout.p > 0.0)
                or os.environ.get("SD_LOOP_FORM", "1") == "0"):
            return None
        if torch.is_grad_enabled() and (x.requires_grad or any(c.requires_grad for c in context)
                                        or any(p.requires_grad for p in dag.parameters()) or self.step_encoding.token.requires_grad):
            return None
        if any((not c.is_cuda) or c.dtype != torch.float32 or c.dim() != 3 or c.shape[0] != x.shape[0] for c in context):
            return None
        B, T, _ = x.shape
        Mc = sum(int(c.shape[1]) for c in context)
        n_tok = 1 if step.shape[0] == 1 else B
        packed = dag.packed()
        cap = min(ops.sampler_cap(packed), 3)   # (mode 4 needs its status word read back: a host synchronisation per call)
        key = (B, T, Mc, n_tok, x.device, cap)
        cache = _model_cache(self, "loop")
        ls = cache.get(key)
        if ls is None:
            if len(cache) >= 4:
                cache.clear()   # a prepared workspace is pinned memory: a handful of shapes at a time
            ls = cache[key] = ops.LoopSampler(packed, B, T, Mc, n_tok, x.device, max_mode=cap)
        if not ls.supported:
            return None
        if step.dtype not in (torch.int64, torch.float32):
            step = step.to(torch.float32 if step.is_floating_point() else torch.int64)
        tokens = ops.step_token(step.contiguous(), self.step_encoding._freq, self.step_encoding.token.detach()).view(n_tok, self.hidden_dim)
        # (dag.packed() rebuilds its descriptor when a parameter's storage moved: its identity stands for the pointers)
        wkey = (id(packed), tuple(p._version for p in dag.parameters()), ops.weights_generation())
        return ls.eps(packed, list(context), tokens, x.contiguous(), wkey)

    # ---- extras ------------------------------------------------------------------------
    def dropout_stacks(self):
        """The transformer stacks that carry dropout (decoder + every enabled context encoder), in a fixed order."""
        from .encoder.encoders import BaseEncoder

        stacks = [self.diffusion_action_generator]
        for m in self.modules():
            if isinstance(m, BaseEncoder):
                stacks.append(m)
        return stacks

    def set_dropout(self, p: float, seed: Optional[int] = None) -> "End2EndDiffusionTransformer":
        """Dropout probability of every transformer layer (live in ``train()`` mode only).  The reference never sets it,
        i.e. trains with torch's default 0.1 - also the default here; 0 selects the parity path (golden gradients).
        ``seed`` keys the Philox masks (default: ``torch.initial_seed()`` at first use; give every data-parallel rank
        its own).  The ResNet / Swin image backbones (torch.nn modules) are not affected."""
        if not 0.0 <= p < 1.0:
            raise ValueError("dropout probability must be in [0, 1)")
        for i, m in enumerate(self.dropout_stacks()):
            m.dropout.p = float(p)
            m.dropout.salt = i
            if seed is not None:
                m.dropout.seed = int(seed)
        return self

    def _assemble_memory(self, context, step, B, device) -> torch.Tensor:
        grad_path = torch.is_grad_enabled() and any(c.requires_grad for c in context)
        if grad_path or (torch.is_grad_enabled() and self.step_encoding.token.requires_grad):
            from ...training import step_token_autograd

            return torch.cat(list(context) + [step_token_autograd(self.step_encoding, step)], dim=1)
        d = self.hidden_dim
        M = sum(int(c.shape[1]) for c in context) + 1
        memory = torch.empty(B, M, d, dtype=torch.float32, device=device)
        at = 0
        for c in context:
            n = c.shape[1]
            memory[:, at : at + n].copy_(c)
            at += n
        step = step.contiguous()
        if step.dtype not in (torch.int64, torch.float32):
            step = step.to(torch.float32 if step.is_floating_point() else torch.int64)
        # the kernel writes sample b's token straight into row M-1 of its memory block
        ops.step_token(step, self.step_encoding._freq, self.step_encoding.token.detach(),
                       out=memory.view(-1)[(M - 1) * d :], row_stride=M * d)
        return memory

    @torch.no_grad()
    def sample(self, context: Sequence[torch.Tensor], x_T: torch.Tensor, num_inference_steps: int,
               return_trace: bool = False, alphas_cumprod: Optional[torch.Tensor] = None, use_graph: bool = False,
               with_dropout: bool = False, max_mode: Optional[int] = None):
        """The reference's denoising loop (plot.py:122-131 / distill.py:179-189 / ros.py:301-310)
        as one native call: n x (denoiser forward + DDIM update) with the context K/V cached.
        ``use_graph`` replays the rollout from a hipGraph captured for this (B, T, M, n) shape.
        ``max_mode``: the highest sampler mode (``sd_sampler_mode``) the call may run; None = ``ops.default_sampler_cap()`` = 3 (three fp16
        products at every site: valid for any weights); 4 opts in to the guarded two-product Q | K | V site (falls back to 3 by itself).
        The native rollout has no dropout (inference).  ``with_dropout=True`` on a model in ``train()`` mode instead
        steps through ``forward_with_context`` + the scheduler update like the reference's loop does, dropout live in
        every call - what distill.py's teacher, which is never put into eval mode, actually computes (distill.py:127-189)."""
        if with_dropout and self.training and self.diffusion_action_generator.dropout.p > 0.0:
            from ...scheduler import DDIMScheduler

            sch = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
            sch.set_timesteps(num_inference_steps)
            x, trace = x_T, []
            for t in sch.timesteps.tolist():
                eps = self.forward_with_context(context, x, torch.full((x.shape[0],), t, dtype=torch.int64, device=x.device))
                x = sch.step(eps, t, x).prev_sample
                if return_trace:
                    trace.append(x)
            return (x, torch.stack(trace)) if return_trace else x
        ts = ops.ddim_timesteps(num_inference_steps)
        acp = ops.alphas_cumprod() if alphas_cumprod is None else alphas_cumprod
        coef = ops.ddim_coefficients(ts, acp, num_inference_steps)
        ctx = torch.cat(list(context), dim=1).contiguous() if len(context) else None
        packed = self.diffusion_action_generator.packed()
        if use_graph and not return_trace:
            B, T, _ = x_T.shape
            Mc = 0 if ctx is None else ctx.shape[1]
            cap = ops.sampler_cap(packed, max_mode)   # as ops.ddim_sample_guarded
            key = (B, T, Mc, num_inference_steps, x_T.device, self.diffusion_action_generator._signature(),
                   self.step_encoding.token._version, cap)
            cache = _model_cache(self, "graphs")
            if key not in cache:
                cache.clear()  # one shape at a time: a graph pins its workspace
                cache[key] = ops.GraphedSampler(packed, B, T, Mc, self.step_encoding.table(ts, x_T.device), coef, max_mode=cap)
            out = cache[key](ctx, x_T)
            word = int(cache[key].status.item())
            if word == 0:
                return out
            if word & ops.STATUS_SHARP_LOGITS:
                packed.sampler_cap = 3   # pinned before the eager rerun: it starts on mode 3, not on mode 4 again
            # range guard tripped (ops.ddim_sample_guarded): fall through to the guarded eager path
        tokens = self.step_encoding.table(ts, x_T.device)
        return ops.ddim_sample_guarded(packed, ctx, tokens, coef, x_T.contiguous(), trace=return_trace, max_mode=max_mode)
