#!/usr/bin/env python3
"""Rollout time of the denoiser at the DECODER shapes of the reference's shipped configs (ml/training/config/*.yaml: hidden_dim, layers,
memory rows = context tokens + step token, horizon 10; 30 DDIM steps as ros.py:50 / distill_teacher_inference_steps) at 20 and 22 joints,
per kernel selection: sampler mode 3 (the trajectory kernels: sd_traj.h for hidden_dim 256 up to 64 memory rows, sd_trajg.hip for
everything else) against the older row-panel / unfused-chain kernels (max_mode 2), eager and as a replayed hipGraph, plus the reference's
own loop form (forward_with_context + scheduler.step per step) through the boundary class.  One JSON line per (config, batch).
usage: python tools/bench_yaml_shapes.py [--quick]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccerdiffusion_amd import _lib, ops  # noqa: E402
from soccerdiffusion_amd.scheduler import DDIMScheduler  # noqa: E402
from soccerdiffusion_amd.synthetic import synthetic_state_dict  # noqa: E402

CONFIGS = {   # name: (hidden_dim, decoder layers, context rows)
    "default": (128, 4, 311), "decoder_only": (256, 4, 0), "larger_model": (512, 8, 311), "sim_scratch": (256, 6, 50),
    # not shipped, same family: BASELINE's horizon at the other widths
    "d128_T100": (128, 4, 10), "d512_T48": (512, 4, 10),
}
T_BY = {"d128_T100": 100, "d512_T48": 48}
quick = "--quick" in sys.argv
n_steps = 30


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


def loop_model(d, J, L, T, sd):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_gpu_loop_form import _model

    m, _ = _model(d, J, L, T)
    m.load_state_dict(sd)
    return m


for name, (d, L, Mc) in CONFIGS.items():
    T = T_BY.get(name, 10)
    for J in ((20,) if quick or name in T_BY else (20, 22)):
        sd = synthetic_state_dict(d, J, L, seed=3)
        packed = ops.pack_denoiser(sd, "cuda", max_len=T)
        ts = ops.ddim_timesteps(n_steps)
        coef = ops.ddim_coefficients(ts, ops.alphas_cumprod(), n_steps)
        toks = ops.step_token(torch.tensor(ts).cuda(), ops.step_frequencies(d).cuda(), sd["step_encoding.token"].cuda()).reshape(n_steps, d)
        model = loop_model(d, J, L, T, sd)
        sched = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
        sched.set_timesteps(n_steps)
        for B in ((1, 64) if quick else (1, 16, 64, 1024)):
            if name in T_BY and B != 1024 and not quick:
                continue
            x = torch.randn(B, T, J, device="cuda")
            ctx = torch.randn(B, Mc, d, device="cuda") if Mc else None
            rec = {"config": name, "hidden_dim": d, "layers": L, "memory_rows": Mc + 1, "T": T, "J": J, "B": B, "steps": n_steps,
                   "sd_sampler_mode": _lib.load().sd_sampler_mode(d, 4, T, Mc, J)}
            n = 3 if B >= 1024 else 10
            for mode in (2, 3):
                rec[f"eager_max_mode_{mode}_ms"] = timed(lambda: ops.ddim_sample(packed, ctx, toks, coef, x, max_mode=mode), n)
                gs = ops.GraphedSampler(packed, B, T, Mc, toks, coef, max_mode=mode)
                rec[f"hipgraph_max_mode_{mode}_ms"] = timed(lambda: gs.replay_into(ctx, x), n)
                del gs
            cl = [ctx] if Mc else []

            def loop():
                traj = x
                with torch.no_grad():
                    for t in sched.timesteps:
                        eps = model.forward_with_context(cl, traj, torch.full((B,), int(t), device="cuda"))
                        traj = sched.step(eps, t, traj).prev_sample
                return traj

            rec["loop_form_ms"] = timed(loop, n)
            os.environ["SD_LOOP_FORM"] = "0"
            rec["loop_form_row_panel_kernels_ms"] = timed(loop, n)   # what the same Python loop cost before: sd_denoiser_forward per call
            del os.environ["SD_LOOP_FORM"]
            rec["trajectories_per_s_mode3"] = round(B / (min(rec["eager_max_mode_3_ms"], rec["hipgraph_max_mode_3_ms"]) * 1e-3), 1)
            print(json.dumps(rec), flush=True)
