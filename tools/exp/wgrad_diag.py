"""Diagnostic: conv_wgrad / dgrad error on the tensors of a real ResNet-18 backward at 1 x 480 x 640 (vs fp64 on the same tensors)."""
import os, sys, copy
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import conv_training as ct
from soccerdiffusion_amd.ml.model.encoder.image import _BasicBlock, _ResNet

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())

torch.manual_seed(1)
net = _ResNet(_BasicBlock, [2, 2, 2, 2]); net.fc = torch.nn.Linear(512, 32)
gpu = net.cuda().train()
x = torch.rand(1, 3, 480, 640, generator=torch.Generator().manual_seed(2)).cuda().requires_grad_()
wr = torch.randn(1, 32, generator=torch.Generator().manual_seed(3)).cuda()
rec = []
orig_w, orig_raw = ct.conv_wgrad, ct.conv_raw
def wg(dy, h, shape, stride, *words):
    dw = orig_w(dy, h, shape, stride, *words)
    rec.append(("wgrad", dy.detach().clone(), h.detach().clone(), shape, stride, dw.detach().clone()))
    return dw
ct.conv_wgrad = wg
(gpu(x) * wr).sum().backward()
ct.conv_wgrad = orig_w
for kind, dy, h, shape, stride, dw in rec:
    Cout, Cin, k, _ = shape
    if h.shape[1] * h.shape[2] < 4000 and Cin > 64: continue
    hd = h.double().cpu().permute(0, 3, 1, 2)
    dyd = dy.double().cpu().permute(0, 3, 1, 2)
    want = torch.nn.grad.conv2d_weight(hd, shape, dyd, stride=stride, padding=k // 2)
    amax = float(dy.abs().max()); med = float(dy.abs().median()); 
    print(shape, "stride", stride, "hw", tuple(h.shape[1:3]), "wgrad err %.2e" % rel(dw, want), "dy max/median %.1e" % (amax / max(med, 1e-30)),
          "h max/median %.1e" % (float(h.abs().max()) / max(float(h.abs().median()), 1e-30)), flush=True)
