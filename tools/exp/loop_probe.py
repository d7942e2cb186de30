"""Where the loop form's per-call time goes at B = 256: CPU time per call (no sync), GPU time per rollout, and a cProfile of one rollout."""
import cProfile, os, pstats, sys, time, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from soccerdiffusion_amd import cli
from soccerdiffusion_amd.scheduler import DDIMScheduler
from soccerdiffusion_amd.synthetic import synthetic_state_dict
dev = torch.device("cuda", 0)
model = cli.build_model(bench.C2_PARAMS).to(dev).eval()
sched = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False); sched.set_timesteps(50)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B, 100, 20, device=dev); ctx = [torch.randn(B, 10, 256, device=dev)]
def loop():
    traj = x
    with torch.no_grad():
        for t in sched.timesteps:
            eps = model.forward_with_context(ctx, traj, torch.full((B,), int(t), device=dev))
            traj = sched.step(eps, t, traj).prev_sample
    return traj
for _ in range(3): loop()
torch.cuda.synchronize()
t0 = time.perf_counter(); loop(); t_cpu = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("one rollout: CPU returns after %.2f ms, GPU done after %.2f ms" % (t_cpu * 1e3, t_all * 1e3))
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
