"""Sampler throughput vs rollout batch (is a MALL-resident working set worth the emptier launches?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from soccerdiffusion_amd import ops
from soccerdiffusion_amd.synthetic import synthetic_state_dict

D, L, T, J, MC = bench.D, bench.L, bench.T, bench.J, bench.MC
sd = synthetic_state_dict(D, J, L, seed=0)
packed = ops.pack_denoiser(sd, "cuda", max_len=T)
ts = ops.ddim_timesteps(50)
coef = ops.ddim_coefficients(ts, ops.alphas_cumprod(), 50)
toks = ops.step_token(torch.tensor(ts).cuda(), ops.step_frequencies(D).cuda(), sd["step_encoding.token"].cuda()).reshape(50, D)
for B in (256, 328, 512, 656, 1024, 2048, 4096, 8192):
    x = torch.randn(B, T, J, device="cuda")
    c = torch.randn(B, MC, D, device="cuda")
    ops.ddim_sample(packed, c, toks, coef, x); torch.cuda.synchronize()
    n = max(1, 4096 // B)
    t0 = time.perf_counter()
    for _ in range(n):
        ops.ddim_sample(packed, c, toks, coef, x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B}: {dt * 1e3:.1f} ms per rollout -> {B / dt:.0f} traj/s", flush=True)
