"""Where do the device-to-device memcpys and fill kernels of a C2 training step come from?  torch.profiler with python stacks over two
eager steps (bench.TrainLeg without the hipGraph).  usage (GPU box): python tools/exp/train_copies.py"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
leg = bench.TrainLeg(dev, 0, 1, bench.TRAIN_B, 40, graph=False)
for _ in range(3):
    leg.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        leg.step()
    torch.cuda.synchronize()
by = collections.Counter()
for e in prof.events():
    n = e.name
    if n.startswith(("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::cat", "aten::add_", "aten::zeros", "aten::empty_like")):
        st = [s for s in (e.stack or []) if "soccerdiffusion_amd" in s or "bench.py" in s]
        by[(n, st[0] if st else "?")] += 1
for (n, s), c in sorted(by.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c / 2:6.1f} per step  {n:22s} {s}")
