#!/usr/bin/env python3
"""BASELINE configs[4]'s per-GPU share, inference forward of the image backbone (16 trajectories x 10 frames of 480 x 640): per basic-block
shape the hand-written sd_conv3x3_bn_act (csrc/sd_conv.hip) beside torch.nn's conv + BatchNorm(eval) + ReLU (MIOpen), the stem
(sd_stem_conv_bn_relu_pool) and the stage entries (sd_conv_s2_bn_act) likewise, then the whole ResNet-18 forward on both routes.  usage (GPU box): python tools/bench_conv.py [frames=160] > profiles/r04_c5_conv_forward.json"""
import json
import os
import sys
import time

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccerdiffusion_amd import ops  # noqa: E402
from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, image_encoder_factory  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev = torch.device("cuda", 0)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


out = {"workload": f"ResNet-18 basic-block 3x3 convolutions at {N} frames of 480 x 640 (BASELINE configs[4] per-GPU share: 16 x 10 frames), inference",
       "shapes": []}
for C, H, W in ((64, 120, 160), (128, 60, 80), (256, 30, 40), (512, 15, 20)):
    g = torch.Generator(device=dev).manual_seed(C)
    x = torch.rand(N, C, H, W, device=dev, generator=g)
    w = torch.randn(C, C, 3, 3, device=dev, generator=g) * (2.0 / (9 * C)) ** 0.5
    bn = torch.nn.BatchNorm2d(C).to(dev).eval()
    xh = x.permute(0, 2, 3, 1).contiguous()
    pk = ops.PackedConv3x3(w)
    xa = ops.absmax_word(xh)
    s, t = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ya = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.no_grad():
        t_hip = timed(lambda: ops.conv3x3_bn_act(xh, xa, pk, s, t, res=xh, relu=True, y_amax=ya))
        t_lib = timed(lambda: F.relu(bn(F.conv2d(x, w, padding=1)) + x))
    flops = 2.0 * 9 * C * C * H * W * N
    out["shapes"].append({"channels": C, "map": [H, W], "hip_ms": round(t_hip * 1e3, 3), "miopen_conv_bn_relu_ms": round(t_lib * 1e3, 3),
                          "hip_algorithmic_tflops": round(flops / t_hip / 1e12, 1), "miopen_algorithmic_tflops": round(flops / t_lib / 1e12, 1),
                          "hip_frac_of_fp16_mfma_peak": round(flops / t_hip / 1e12 / 2516.8, 4)})
    del x, xh
# the stem (conv 7x7 s2 + BN + ReLU + max-pool 3x3 s2) and the stage entries (3x3 s2 and their 1x1 s2 shortcut)
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand(N, 3, 480, 640, device=dev, generator=g)
w = torch.randn(64, 3, 7, 7, device=dev, generator=g) * (2.0 / (49 * 64)) ** 0.5
bn = torch.nn.BatchNorm2d(64).to(dev).eval()
pk, xa, ya = ops.PackedStem(w), ops.absmax_word(x), torch.zeros(1, dtype=torch.int32, device=dev)
s, t = torch.ones(64, device=dev), torch.zeros(64, device=dev)
xcl = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    t_hip = timed(lambda: ops.stem_conv_bn_relu_pool(x, xa, pk, s, t, y_amax=ya))
    t_lib = timed(lambda: F.max_pool2d(F.relu(bn(F.conv2d(x, w, stride=2, padding=3))), 3, 2, 1))
    t_lib_cl = timed(lambda: F.max_pool2d(F.relu(bn(F.conv2d(xcl, w, stride=2, padding=3))), 3, 2, 1))
out["stem"] = {"hip_ms": round(t_hip * 1e3, 3), "miopen_conv_bn_relu_pool_ms": round(t_lib * 1e3, 3),
               "miopen_channels_last_ms": round(t_lib_cl * 1e3, 3),
               "hip_algorithmic_tflops": round(2.0 * 147 * 64 * 240 * 320 * N / t_hip / 1e12, 1)}
del x, xcl
out["stage_entries"] = []
for C, H, W in ((64, 120, 160), (128, 60, 80), (256, 30, 40)):
    g = torch.Generator(device=dev).manual_seed(C + 1)
    x = torch.rand(N, C, H, W, device=dev, generator=g)
    xh = x.permute(0, 2, 3, 1).contiguous()
    xa = ops.absmax_word(xh)
    bn = torch.nn.BatchNorm2d(2 * C).to(dev).eval()
    s, t = torch.ones(2 * C, device=dev), torch.zeros(2 * C, device=dev)
    rec = {"channels": [C, 2 * C], "map": [H, W]}
    for k in (3, 1):
        w = torch.randn(2 * C, C, k, k, device=dev, generator=g) * (2.0 / (k * k * 2 * C)) ** 0.5
        pk = ops.PackedConv3x3(w)
        with torch.no_grad():
            rec[f"hip_{k}x{k}_ms"] = round(timed(lambda: ops.conv_s2_bn_act(xh, xa, pk, s, t, relu=k == 3)) * 1e3, 3)
            rec[f"miopen_{k}x{k}_ms"] = round(timed(lambda: bn(F.conv2d(x, w, stride=2, padding=k // 2))) * 1e3, 3)
    out["stage_entries"].append(rec)
    del x, xh
torch.manual_seed(0)
enc = image_encoder_factory(ImageEncoderType.RESNET18, 256, True, 480).to(dev).eval()
frames = torch.rand(16, N // 16, 3, 480, 640, device=dev)
with torch.no_grad():
    t_hip = timed(lambda: enc(frames), 3)
    a = enc(frames)
    os.environ["SD_CONV"] = "torch"
    t_lib = timed(lambda: enc(frames), 3)
    b = enc(frames)
    del os.environ["SD_CONV"]
out["backbone_forward"] = {"frames": N, "hip_ms": round(t_hip * 1e3, 1), "all_miopen_ms": round(t_lib * 1e3, 1),
                           "frames_per_s_hip": round(N / t_hip, 1), "frames_per_s_miopen": round(N / t_lib, 1),
                           "max_rel_diff_of_tokens": float((a - b).abs().max() / b.abs().max())}
# the reference's other ResNet option (Bottleneck blocks: 1 x 1 / 3 x 3 / 1 x 1 on sd_conv1x1_bn_act, sd_conv3x3_bn_act, sd_conv_s2_bn_act)
del enc, a, b
torch.manual_seed(0)
enc = image_encoder_factory(ImageEncoderType.RESNET50, 256, True, 480).to(dev).eval()
with torch.no_grad():
    t_hip = timed(lambda: enc(frames), 3)
    a = enc(frames)
    os.environ["SD_CONV"] = "torch"
    t_lib = timed(lambda: enc(frames), 3)
    b = enc(frames)
    del os.environ["SD_CONV"]
out["resnet50_forward"] = {"frames": N, "hip_ms": round(t_hip * 1e3, 1), "all_miopen_ms": round(t_lib * 1e3, 1),
                           "frames_per_s_hip": round(N / t_hip, 1), "frames_per_s_miopen": round(N / t_lib, 1),
                           "max_rel_diff_of_tokens": float((a - b).abs().max() / b.abs().max())}
print(json.dumps(out, indent=1))
