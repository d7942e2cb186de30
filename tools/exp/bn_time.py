"""Time sd_bn_train_fwd / sd_bn_train_bwd on ResNet-18's BatchNorm shapes at 160 frames of 480 x 640 and print the bytes they move per second."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import conv_training as ct

def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

for (H, W, C) in [(240, 320, 64), (120, 160, 64), (60, 80, 128), (30, 40, 256), (15, 20, 512)]:
    N = 160
    y = torch.randn(N, H, W, C, device="cuda"); res = torch.randn_like(y); dz = torch.randn_like(y)
    g = torch.rand(C, device="cuda") + 0.5; b = torch.randn(C, device="cuda")
    gb = y.numel() * 4 / 1e9
    z, _, mean, rstd = ct.bn_train_fwd(y, g, b, res, None, None, 1e-5, 0.1, True)
    tf = t(lambda: ct.bn_train_fwd(y, g, b, res, None, None, 1e-5, 0.1, True))
    tb = t(lambda: ct.bn_train_bwd(dz, z, y, mean, rstd, g, True, True))
    z2, _, mean2, rstd2 = ct.bn_train_fwd(y, g, b, None, None, None, 1e-5, 0.1, True)
    tn = t(lambda: ct.bn_train_bwd(dz, None, y, mean2, rstd2, g, True, False, b))
    tz = t(lambda: ct.bn_train_bwd(dz, z2, y, mean2, rstd2, g, True, False))
    # forward: statistics read y; apply reads y, res, writes z.  backward: reduce reads dz, z, y; apply reads dz, z, y, writes dy, dres
    print("%3dx%3dx%4d (%.2f GB): fwd %.3f ms = %.2f TB/s of 4 tensors; bwd %.3f ms = %.2f TB/s of 8 tensors" %
          (H, W, C, gb, tf * 1e3, 4 * gb / tf / 1e3, tb * 1e3, 8 * gb / tb / 1e3), flush=True)
    print("      no residual operand: bwd reading z %.3f ms, recomputing the mask from y %.3f ms" % (tz * 1e3, tn * 1e3), flush=True)
    del y, res, dz, z, z2
