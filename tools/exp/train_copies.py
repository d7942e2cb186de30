"""Which Python lines issue the device-to-device copies / fills / adds of one eager training step (bench's C2 leg)?  torch.profiler with stacks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from torch.profiler import profile, ProfilerActivity
leg = bench.TrainLeg(torch.device("cuda", 0), 0, 1, 256, 100, dropout=0.1, graph=False)
for _ in range(3): leg.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    leg.step()
    torch.cuda.synchronize()
import collections
c = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::clone", "aten::contiguous", "aten::cat", "aten::zeros", "aten::zeros_like", "aten::to", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "soccerdiffusion_amd" in s or "bench.py" in s][:2]
        c[(ev.name, tuple(st), str(ev.input_shapes)[:60])] += 1
for (name, st, shp), n in c.most_common(60):
    print(n, name, shp, " <- ", " | ".join(s.split("/")[-1] for s in st))
