"""Time the fused stem BatchNorm + ReLU + max-pool (sd_bn_relu_pool_fwd / _bwd) at the stem's shape of 160 frames of 480 x 640."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import conv_training as ct
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
N, H, W, C = 160, 240, 320, 64
y = torch.randn(N, H, W, C, device="cuda"); g = torch.rand(C, device="cuda") + 0.5; b = torch.randn(C, device="cuda")
p, word, idx, mean, rstd = ct.bn_relu_pool_fwd(y, g, b, None, None, 1e-5, 0.1)
dp = torch.randn_like(p)
print("forward %.3f ms, backward %.3f ms" % (t(lambda: ct.bn_relu_pool_fwd(y, g, b, None, None, 1e-5, 0.1)) * 1e3, t(lambda: ct.bn_relu_pool_bwd(dp, idx, y, mean, rstd, g)) * 1e3))
