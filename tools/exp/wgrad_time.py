"""Time sd_conv_wgrad on ResNet-18 layer-1's shape (160 x 120 x 160 x 64 -> 64, 3 x 3): SD_W3_ABL selects diagnostic ablations."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import conv_training as ct, ops
N, H, W, C = 160, 120, 160, 64
x = torch.rand(N, H, W, C, device="cuda"); dy = torch.randn(N, H, W, C, device="cuda")
xa, ya = ops.absmax_word(x), ops.absmax_word(dy)
for _ in range(3): ct.conv_wgrad(dy, x, (C, C, 3, 3), 1, ya, xa)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ct.conv_wgrad(dy, x, (C, C, 3, 3), 1, ya, xa)
torch.cuda.synchronize()
print("SD_W3_ABL=%s: %.3f ms per call" % (os.environ.get("SD_W3_ABL", "0"), (time.perf_counter() - t0) / 10 * 1e3))
