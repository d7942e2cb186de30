"""model.sample() at the robot's B = 1 on default.yaml's decoder shape: 20 rollouts for a kernel table."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_loop_form import _model
d, L, Mc, T = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 4, 311, 10)))
m, _ = _model(d, 20, L, T)
x = torch.randn(1, T, 20, device="cuda"); ctx = [torch.randn(1, Mc, d, device="cuda")] if Mc else []
for _ in range(20): m.sample(ctx, x, 30)
torch.cuda.synchronize()
