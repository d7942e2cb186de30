"""ResNet-18 inference forward on 160 frames of 480 x 640 (configs[4]'s per-GPU share), the route with sd_conv3x3_bn_act: for a rocprofv3 kernel table.
usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/exp/backbone_fwd.py"""
import os, sys
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, image_encoder_factory
torch.manual_seed(0)
enc = image_encoder_factory(ImageEncoderType.RESNET18, 256, True, 480).cuda().eval()
frames = torch.rand(16, 10, 3, 480, 640, device="cuda")
with torch.no_grad():
    for _ in range(4):
        enc(frames)
torch.cuda.synchronize()
