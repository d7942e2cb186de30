"""Time sd_conv_wgrad on every convolution shape of ResNet-18 at 160 frames of 480 x 640 (weight-gradient launches of one backward)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import conv_training as ct, ops
N = 160
SHAPES = [  # H, W (input), Cin, Cout, k, stride, count per backward
    (120, 160, 64, 64, 3, 1, 4), (120, 160, 64, 128, 3, 2, 1), (120, 160, 64, 128, 1, 2, 1), (60, 80, 128, 128, 3, 1, 3),
    (60, 80, 128, 256, 3, 2, 1), (60, 80, 128, 256, 1, 2, 1), (30, 40, 256, 256, 3, 1, 3), (30, 40, 256, 512, 3, 2, 1),
    (30, 40, 256, 512, 1, 2, 1), (15, 20, 512, 512, 3, 1, 3)]
tot = 0.0
for H, W, Ci, Co, k, s, cnt in SHAPES:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.rand(N, H, W, Ci, device="cuda"); dy = torch.randn(N, Ho, Wo, Co, device="cuda")
    xa, ya = ops.absmax_word(x), ops.absmax_word(dy)
    f = lambda: ct.conv_wgrad(dy, x, (Co, Ci, k, k), s, ya, xa)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    fl = 2.0 * N * Ho * Wo * Ci * Co * k * k
    print("%3dx%3d %3d->%3d k%d s%d: %.3f ms  %.0f TFLOP/s (x%d)" % (H, W, Ci, Co, k, s, t * 1e3, fl / t / 1e12, cnt), flush=True)
    tot += t * cnt
    del x, dy
print("weight gradients of one backward: %.2f ms" % (tot * 1e3))
