"""Experiment: one B=4096 rollout vs N concurrent B/N rollouts on separate streams (tail/phase overlap)."""
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
B = 4096
for nsplit in (1, 2, 4, 1, 2):
    Bs = B // nsplit
    xs = [torch.randn(Bs, T, J, device="cuda") for _ in range(nsplit)]
    cs = [torch.randn(Bs, MC, D, device="cuda") for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    def run():
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                ops.ddim_sample(packed, cs[i], toks, coef, xs[i])
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"{nsplit} stream(s) x B={Bs}: {dt * 1e3:.1f} ms per {B} trajectories -> {B / dt:.0f} traj/s", flush=True)
