"""Rollout latency of sampler modes 3 and 2 at small batches (the robot runs B = 1): where does one-workgroup-per-trajectory lose?
Run once on the GPU box: python tools/exp/small_batch_modes.py"""
import os, subprocess, sys, time

if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import bench
    from soccerdiffusion_amd import ops
    from soccerdiffusion_amd.synthetic import synthetic_state_dict
    D, L, T, J, MC = bench.D, bench.L, bench.T, bench.J, bench.MC
    sd = synthetic_state_dict(D, J, L, seed=0)
    packed = ops.pack_denoiser(sd, "cuda", max_len=T)
    ts = ops.ddim_timesteps(50)
    coef = ops.ddim_coefficients(ts, ops.alphas_cumprod(), 50)
    toks = ops.step_token(torch.tensor(ts).cuda(), ops.step_frequencies(D).cuda(), sd["step_encoding.token"].cuda()).reshape(50, D)
    for B in (1, 4, 16, 32, 64, 96, 128, 192, 256, 384, 512):
        x = torch.randn(B, T, J, device="cuda")
        c = torch.randn(B, MC, D, device="cuda")
        for _ in range(2):
            ops.ddim_sample(packed, c, toks, coef, x)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            ops.ddim_sample(packed, c, toks, coef, x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{sys.argv[1]} B={B}: {dt * 1e3:.2f} ms per rollout -> {B / dt:.0f} traj/s", flush=True)
else:
    for mode, env in (("mode3", {}), ("mode2", {"SD_SAMPLER_TRAJ": "0"})):
        subprocess.run([sys.executable, __file__, mode], env={**os.environ, **env}, check=True)
