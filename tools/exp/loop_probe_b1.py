"""The reference's loop form at the robot's B = 1 (ros.py:301-310): ms per 30-step rollout on the shipped decoder shapes, next to model.sample()."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_loop_form import _model
from soccerdiffusion_amd.scheduler import DDIMScheduler
for name, d, L, Mc, T in [("default.yaml", 128, 4, 311, 10), ("larger_model.yaml", 512, 8, 311, 10), ("decoder_only.yaml", 256, 4, 0, 10), ("sim_scratch.yaml", 256, 6, 50, 10)]:
    m, _ = _model(d, 20, L, T)
    sched = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False); sched.set_timesteps(30)
    x = torch.randn(1, T, 20, device="cuda"); ctx = [torch.randn(1, Mc, d, device="cuda")] if Mc else []
    def loop():
        traj = x
        with torch.no_grad():
            for t in sched.timesteps:
                eps = m.forward_with_context(ctx, traj, torch.full((1,), int(t), device="cuda"))
                traj = sched.step(eps, t, traj).prev_sample
        return traj
    def timed(f, n=10):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("%-18s loop form %.2f ms, model.sample %.2f ms per 30-step rollout at B = 1" % (name, timed(loop), timed(lambda: m.sample(ctx, x, 30))), flush=True)
