"""Times sd_op_gemm_tn at the training shape (R = 25 600, N = K = 256).  SD_GEMM_TN=quartet|regs|f32 selects the other kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import ops
R, N, K = 25600, 256, 256
g = torch.Generator(device="cuda").manual_seed(0)
dY = torch.randn(R, N, device="cuda", generator=g); X = torch.randn(R, K, device="cuda", generator=g)
dW = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
for _ in range(3): ops.gemm_tn(dY, X, dW, db)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): ops.gemm_tn(dY, X, dW, db)
b.record(); torch.cuda.synchronize()
print("gemm_tn us", round(a.elapsed_time(b) / 50 * 1e3, 1), "variant", os.environ.get("SD_GEMM_TN"), flush=True)
