"""Times sd_gemm_tn_grouped on the six weight gradients of one decoder layer at the training shape (R = 25 600, d = 256):
in_proj (N = 768), out_proj, q, cross out_proj, linear1, linear2 - 36 tiles of 128 x 128, 28 800 slab-tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import ops
R, d = 25600, 256
g = torch.Generator(device="cuda").manual_seed(0)
keep, probs = [], []
for N in (3 * d, d, d, d, d, d):
    dY = torch.randn(R, N, device="cuda", generator=g) * 1e-3; X = torch.randn(R, d, device="cuda", generator=g)
    dW = torch.zeros(N, d, device="cuda"); db = torch.zeros(N, device="cuda")
    ay = torch.zeros(64, dtype=torch.int32, device="cuda"); ax = torch.zeros(64, dtype=torch.int32, device="cuda")
    ay[0] = dY.abs().max().reshape(1).view(torch.int32)[0]; ax[0] = X.abs().max().reshape(1).view(torch.int32)[0]
    keep += [dY, X, ay, ax]
    probs.append((dY, X, dW, db, ay.data_ptr(), ax.data_ptr()))
for _ in range(3): ops.gemm_tn_grouped(probs)
torch.cuda.synchronize()
want = (probs[1][0].double().t() @ probs[1][1].double()) * 3
print("rel err", float((probs[1][2].double() - want).norm() / want.norm()))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30): ops.gemm_tn_grouped(probs)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 30 * 1e3
flops = 2.0 * R * d * (8 * d)
print("gemm_tn_grouped us", round(us, 1), " algorithmic TFLOP/s", round(flops / us / 1e6, 1), " x3 MFMA TFLOP/s", round(3 * flops / us / 1e6, 1), flush=True)
