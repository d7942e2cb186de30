"""The robot's rollout (B = 1, horizon 10, 30 DDIM steps, d = 256, L = 4, 10 context rows) ten times: for a rocprofv3 kernel table
(what does the once-per-rollout preparation cost beside the 30 step launches?).  usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/exp/robot_rollout.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import denoiser_ref as ref  # noqa: E402  (weights only: the synthetic state dict)
from soccerdiffusion_amd import ops  # noqa: E402

d, L, T, Mc, J, n = 256, 4, 10, 10, 20, 30
sd = ref.synthetic_state_dict(d, J, L, seed=3)
packed = ops.pack_denoiser(sd, "cuda", max_len=T)
ts = ops.ddim_timesteps(n)
coef = ops.ddim_coefficients(ts, ops.alphas_cumprod(), n)
toks = ops.step_token(torch.tensor(ts).cuda(), ops.step_frequencies(d).cuda(), sd["step_encoding.token"].cuda()).reshape(n, d)
x = torch.randn(1, T, J, device="cuda")
ctx = torch.randn(1, Mc, d, device="cuda")
for _ in range(10):
    ops.ddim_sample(packed, ctx, toks, coef, x)
torch.cuda.synchronize()
