"""Time sd_stem_wgrad and sd_stem_conv_raw at 160 frames of 480 x 640."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import conv_training as ct, ops
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
x = torch.rand(160, 3, 480, 640, device="cuda"); dy = torch.randn(160, 240, 320, 64, device="cuda")
xa, ya = ops.absmax_word(x), ops.absmax_word(dy)
print("stem weight gradient %.3f ms" % (t(lambda: ct.stem_wgrad(dy, x, ya, xa)) * 1e3))
