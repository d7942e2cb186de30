"""Randomised soak of the image-training kernels against torch CPU fp64: weight gradients (all kernel paths), the transposed stride-2 data
gradients, the fused BatchNorm + ReLU + max-pool forward / backward.  Usage (GPU box): python tools/exp/soak_conv_train.py [cases] [seed]"""
import os, sys, random
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import conv_training as ct, ops

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

def soak(cases, seed, verbose=True):
  rng = random.Random(seed)
  worst = {"wgrad": 0.0, "dgrad": 0.0, "pool_fwd": 0.0, "pool_bwd": 0.0}
  for case in range(cases):
      N, H, W = rng.randint(1, 3), rng.randint(1, 44), rng.randint(1, 70)
      Cin, Cout = rng.choice([64, 128, 256]), rng.choice([64, 128, 256])
      k, stride = rng.choice([1, 3, 3]), rng.choice([1, 2])
      if stride == 2 and Cout % 128:
          Cout = 128
      g = torch.Generator().manual_seed(case)
      h = torch.randn(N, H, W, Cin, generator=g) * rng.choice([0.01, 1.0, 30.0])
      w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
      hd = h.double().permute(0, 3, 1, 2).requires_grad_()
      wd = w.double().requires_grad_()
      out = F.conv2d(hd, wd, stride=stride, padding=k // 2)
      dy = torch.randn(out.shape, generator=g, dtype=torch.float64) * rng.choice([1e-3, 1.0, 100.0])
      out.backward(dy)
      dyg = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
      hg = h.cuda()
      ya, xa = ops.absmax_word(dyg), ops.absmax_word(hg)
      e1 = rel(ct.conv_wgrad(dyg, hg, (Cout, Cin, k, k), stride, ya, xa), wd.grad)
      e2 = rel(ct.conv_wgrad(dyg, hg, (Cout, Cin, k, k), stride), wd.grad)
      _, bwd = ct.PackedPair().get(w.cuda())
      if stride == 2:
          dh = ct.convt3x3_s2(dyg, ya, bwd, H, W) if k == 3 else ct.convt1x1_s2(dyg, ya, bwd, H, W)
      else:
          dh = ct.conv_raw(dyg, ya, bwd, 1)
      e3 = rel(dh.permute(0, 3, 1, 2), hd.grad)
      worst["wgrad"] = max(worst["wgrad"], e1, e2)
      worst["dgrad"] = max(worst["dgrad"], e3)
      assert e1 < 2e-5 and e2 < 2e-5 and e3 < 5e-6, (case, N, H, W, Cin, Cout, k, stride, e1, e2, e3)
      # fused BatchNorm + ReLU + max-pool on a map of the same size
      C = rng.choice([64, 128])
      y = torch.randn(N, H, W, C, generator=g) * 1.3 + 0.2
      gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5
      if N * H * W < 2:
          continue   # (batch statistics of a single value)
      yd = y.double().permute(0, 3, 1, 2).requires_grad_()
      gd, bd = gamma.double().requires_grad_(), beta.double().requires_grad_()
      ref = F.max_pool2d(F.batch_norm(yd, None, None, gd, bd, training=True, eps=1e-5).relu(), 3, 2, 1)
      dp = torch.randn(ref.shape, generator=g, dtype=torch.float64)
      ref.backward(dp)
      p, word, idx, mean, rstd = ct.bn_relu_pool_fwd(y.cuda(), gamma.cuda(), beta.cuda(), None, None, 1e-5, 0.1)
      e4 = rel(p.permute(0, 3, 1, 2), ref)
      dyb, _, dgm, dbt = ct.bn_relu_pool_bwd(dp.float().permute(0, 2, 3, 1).contiguous().cuda(), idx, y.cuda(), mean, rstd, gamma.cuda())
      e5 = max(rel(dyb.permute(0, 3, 1, 2), yd.grad), rel(dgm, gd.grad), rel(dbt, bd.grad))
      worst["pool_fwd"] = max(worst["pool_fwd"], e4)
      worst["pool_bwd"] = max(worst["pool_bwd"], e5)
      assert e4 < 5e-6 and e5 < 5e-5, (case, N, H, W, C, e4, e5)
      if verbose and case % 10 == 9:
          print("case", case + 1, "worst so far", {k_: "%.1e" % v for k_, v in worst.items()}, flush=True)
  return worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    w = soak(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("SOAK PASS:", n, "cases, worst relative errors", {k_: "%.2e" % v for k_, v in w.items()})
