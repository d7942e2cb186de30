"""SURVEY 8 row f2, first hand-written kernel of the image backbone: the 3 x 3 stride-1 convolution + inference BatchNorm (+ residual)
+ ReLU of ResNet-18's basic blocks (reference: torchvision BasicBlock as configured by soccer_diffusion/ml/model/encoder/image.py:55-83)
through the C ABI (sd_conv3x3_bn_act, csrc/sd_conv.hip) against the same operation in torch on the CPU in fp32 / fp64.
torchvision itself is absent offline: parity of the restated ARCHITECTURE stays unpinned; what is pinned here is
the arithmetic of the block."""

import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from soccerdiffusion_amd import ops as o

    return o


def _case(C_in, C_out, N, H, W, seed, act_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C_in, H, W, generator=g).abs() * act_scale          # post-ReLU activations
    w = torch.randn(C_out, C_in, 3, 3, generator=g) * (2.0 / (9 * C_out)) ** 0.5   # kaiming, fan_out
    gamma = 0.5 + torch.rand(C_out, generator=g)
    beta = 0.2 * torch.randn(C_out, generator=g)
    mean = 0.3 * torch.randn(C_out, generator=g)
    var = 0.5 + torch.rand(C_out, generator=g)
    res = torch.randn(N, C_out, H, W, generator=g).abs() * act_scale
    return x, w, gamma, beta, mean, var, res


def _torch_block(x, w, gamma, beta, mean, var, res, relu, dtype):
    y = F.conv2d(x.to(dtype), w.to(dtype), padding=1)
    y = F.batch_norm(y, mean.to(dtype), var.to(dtype), gamma.to(dtype), beta.to(dtype), training=False, eps=1e-5)
    if res is not None:
        y = y + res.to(dtype)
    return F.relu(y) if relu else y


# the four basic-block shapes of ResNet-18 on a 480 x 640 frame (120 x 160 after the stem) + ragged maps and a non-square channel pair
@pytest.mark.parametrize("C_in,C_out,N,H,W", [(64, 64, 2, 120, 160), (128, 128, 2, 60, 80), (256, 256, 2, 30, 40), (512, 512, 3, 15, 20),
                                             (64, 128, 1, 13, 23), (128, 64, 2, 8, 16), (64, 64, 1, 1, 1), (192, 64, 1, 9, 17)])
@pytest.mark.parametrize("with_res,relu", [(True, True), (False, True), (False, False)])
def test_conv3x3_bn_act_matches_torch_cpu(ops, C_in, C_out, N, H, W, with_res, relu):
    x, w, gamma, beta, mean, var, res = _case(C_in, C_out, N, H, W, seed=C_in + H)
    want64 = _torch_block(x, w, gamma, beta, mean, var, res if with_res else None, relu, torch.float64)
    want32 = _torch_block(x, w, gamma, beta, mean, var, res if with_res else None, relu, torch.float32)
    dev = "cuda"
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    s = (gamma * torch.rsqrt(var + 1e-5)).to(dev)
    t = (beta - mean * gamma * torch.rsqrt(var + 1e-5)).to(dev)
    pk = ops.PackedConv3x3(w.to(dev))
    xa = ops.absmax_word(xh)
    assert xa.view(torch.float32).item() == float(x.abs().max())
    ya = torch.zeros(1, dtype=torch.int32, device=dev)
    rh = res.permute(0, 2, 3, 1).contiguous().to(dev) if with_res else None
    y = ops.conv3x3_bn_act(xh, xa, pk, s, t, res=rh, relu=relu, y_amax=ya)
    got = y.permute(0, 3, 1, 2).cpu()
    e = rel_err(got, want64)
    e32 = rel_err(want32, want64)
    assert e < 2e-6, (e, e32)                     # fp32-grade: three fp16 products on 22-bit operands, fp32 accumulation
    assert e < 8 * e32 + 2e-7, (e, e32)            # and at the level of torch's own fp32 convolution
    assert abs(ya.view(torch.float32).item() - float(got.abs().max())) <= 1e-6 * float(got.abs().max())


# the stage entries of ResNet-18 on a 480 x 640 frame (3 x 3 stride 2 and the 1 x 1 stride-2 shortcut), odd maps, a ragged tile
@pytest.mark.parametrize("C_in,C_out,N,H,W", [(64, 128, 2, 120, 160), (128, 256, 2, 60, 80), (256, 512, 2, 30, 40), (64, 128, 1, 15, 21), (128, 128, 1, 7, 9),
                                             (64, 256, 1, 1, 1)])
@pytest.mark.parametrize("ksize,relu", [(3, True), (1, False)])
def test_conv_stride2_bn_act_matches_torch_cpu(ops, C_in, C_out, N, H, W, ksize, relu):
    g = torch.Generator().manual_seed(C_in + H + ksize)
    x = torch.randn(N, C_in, H, W, generator=g).abs()
    w = torch.randn(C_out, C_in, ksize, ksize, generator=g) * (2.0 / (ksize * ksize * C_out)) ** 0.5
    gamma, beta = 0.5 + torch.rand(C_out, generator=g), 0.2 * torch.randn(C_out, generator=g)
    mean, var = 0.3 * torch.randn(C_out, generator=g), 0.5 + torch.rand(C_out, generator=g)

    def block(dtype):
        y = F.conv2d(x.to(dtype), w.to(dtype), stride=2, padding=ksize // 2)
        y = F.batch_norm(y, mean.to(dtype), var.to(dtype), gamma.to(dtype), beta.to(dtype), training=False, eps=1e-5)
        return F.relu(y) if relu else y

    want64, want32 = block(torch.float64), block(torch.float32)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    inv = torch.rsqrt(var + 1e-5)
    ya = torch.zeros(1, dtype=torch.int32, device="cuda")
    y = ops.conv_s2_bn_act(xh, ops.absmax_word(xh), ops.PackedConv3x3(w.cuda()), (gamma * inv).cuda(), (beta - mean * gamma * inv).cuda(), relu=relu, y_amax=ya)
    got = y.permute(0, 3, 1, 2).cpu()
    assert got.shape == want64.shape
    e, e32 = rel_err(got, want64), rel_err(want32, want64)
    assert e < 2e-6 and e < 8 * e32 + 2e-7, (e, e32)
    assert abs(ya.view(torch.float32).item() - float(got.abs().max())) <= 1e-6 * float(got.abs().max())


def test_conv3x3_large_and_tiny_activations(ops):
    """One scale per tensor from its abs-max word: activations in the hundreds (an untrained backbone's deep layers) and of 1e-3."""
    for scale in (300.0, 1e-3):
        x, w, gamma, beta, mean, var, res = _case(64, 64, 1, 20, 24, seed=5, act_scale=scale)
        want = _torch_block(x, w, gamma, beta, mean * scale, var * scale * scale, None, True, torch.float64)
        xh = x.permute(0, 2, 3, 1).contiguous().cuda()
        inv = torch.rsqrt(var * scale * scale + 1e-5)
        y = ops.conv3x3_bn_act(xh, ops.absmax_word(xh), ops.PackedConv3x3(w.cuda()), (gamma * inv).cuda(),
                               (beta - mean * scale * gamma * inv).cuda(), relu=True)
        assert rel_err(y.permute(0, 3, 1, 2), want) < 2e-6


def test_packed_weight_follows_updates(ops):
    w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
    pk = ops.PackedConv3x3(w)
    before = pk.planes.clone()
    with torch.no_grad():
        w.mul_(2.0)
    pk.refresh(w)
    assert not torch.equal(before, pk.planes) or float(pk.scale) != 0.0
    x = torch.rand(1, 5, 7, 64, device="cuda")
    one, zero = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    y = ops.conv3x3_bn_act(x, ops.absmax_word(x), pk, one, zero, relu=False)
    want = F.conv2d(x.permute(0, 3, 1, 2).cpu().double(), w.detach().cpu().double(), padding=1)
    assert rel_err(y.permute(0, 3, 1, 2), want) < 2e-6


@pytest.mark.parametrize("avgpool,HW", [(True, (96, 128)), (False, (64, 64))])
def test_resnet18_inference_runs_the_hip_blocks_and_matches_cpu_fp32(avgpool, HW):
    """The module-level route (ml/model/encoder/image.py): in eval mode without a tape the basic blocks go through
    sd_conv3x3_bn_act (13 of the 20 convolutions) and sd_conv_s2_bn_act (the three stage entries and their shortcuts), the stem through
    sd_stem_conv_bn_relu_pool; tokens equal the same torch.nn modules on the CPU in fp32 (1e-4) and the
    all-MIOpen route on the GPU."""
    import copy

    from soccerdiffusion_amd import ops as o
    from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, image_encoder_factory

    H, W = HW
    torch.manual_seed(0)
    enc = image_encoder_factory(ImageEncoderType.RESNET18, 64, avgpool, H)
    # non-trivial running statistics (a fresh BatchNorm has mean 0 / var 1)
    enc.train()
    with torch.no_grad():
        enc(torch.rand(2, 2, 3, H, W))
    enc.eval()
    x = torch.rand(2, 3, 3, H, W, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = enc(x)
    g = copy.deepcopy(enc).cuda().eval()
    calls, calls2, calls3 = [], [], []
    orig, orig2, orig3 = o.conv3x3_bn_act, o.conv_s2_bn_act, o.stem_conv_bn_relu_pool
    o.stem_conv_bn_relu_pool = lambda *a, **k: (calls3.append(1), orig3(*a, **k))[1]
    o.conv3x3_bn_act = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    o.conv_s2_bn_act = lambda *a, **k: (calls2.append(1), orig2(*a, **k))[1]
    try:
        with torch.no_grad():
            got = g(x.cuda())
    finally:
        o.conv3x3_bn_act, o.conv_s2_bn_act, o.stem_conv_bn_relu_pool = orig, orig2, orig3
    assert len(calls3) == 1                 # conv1 / bn1 / relu / maxpool: one launch
    assert len(calls) == 13, len(calls)     # layer1: 4, layers 2-4: 3 each (their first conv strides)
    assert len(calls2) == 6, len(calls2)    # ... which, with its 1 x 1 shortcut, runs on the stride-2 kernel: no library convolution is left
    assert got.shape == (2, 3, 64)
    assert float((got.cpu() - want).abs().max() / want.abs().max()) < 1e-4
    os.environ["SD_CONV"] = "torch"
    try:
        with torch.no_grad():
            lib = g(x.cuda())
    finally:
        del os.environ["SD_CONV"]
    assert float((got - lib).abs().max() / lib.abs().max()) < 1e-4
    # with a tape (training / fine-tuning) the library path runs: gradients flow
    xg = x.cuda().requires_grad_(True)
    g(xg).square().sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


def test_eval_forward_sees_fused_adamw_updates():
    """FusedAdamW re-points every parameter into one flat buffer and updates it with a native kernel on raw pointers: no tensor version
    counter moves.  The packed planes / folded BatchNorm vectors of the inference route are keyed on ops.weights_generation()
    as well, so an eval forward after an optimizer step encodes frames with the UPDATED weights - equal to the torch.nn route."""
    import copy

    from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, image_encoder_factory
    from soccerdiffusion_amd.training import FusedAdamW

    torch.manual_seed(3)
    enc = image_encoder_factory(ImageEncoderType.RESNET18, 64, True, 64).cuda()
    x = torch.rand(1, 2, 3, 64, 64, device="cuda")
    enc.eval()
    with torch.no_grad():
        before = enc(x)
    assert copy.deepcopy(enc) is not None   # (the derived planes live outside the modules: nothing un-copyable inside)
    opt = FusedAdamW(enc.parameters(), lr=5e-2)
    for p in enc.parameters():
        p.grad.normal_()
    opt.step()
    with torch.no_grad():
        after = enc(x)
    os.environ["SD_CONV"] = "torch"
    try:
        with torch.no_grad():
            lib = enc(x)
    finally:
        del os.environ["SD_CONV"]
    assert float((after - before).abs().max()) > 1e-3            # the update is visible ...
    assert float((after - lib).abs().max() / lib.abs().max()) < 1e-4   # ... and it is the updated weights' result


# the stem on a 480 x 640 frame, small / odd / ragged frames (conv and pool borders inside one tile), a single pixel
@pytest.mark.parametrize("N,H,W", [(2, 480, 640), (3, 96, 128), (1, 37, 53), (2, 64, 64), (1, 7, 9), (1, 1, 1), (1, 30, 200)])
def test_stem_conv_bn_relu_pool_matches_torch_cpu(ops, N, H, W):
    g = torch.Generator().manual_seed(H + W)
    x = torch.rand(N, 3, H, W, generator=g) * 2.0 - 0.7                     # normalised frames: both signs
    w = torch.randn(64, 3, 7, 7, generator=g) * (2.0 / (49 * 64)) ** 0.5
    gamma, beta = 0.5 + torch.rand(64, generator=g), 0.2 * torch.randn(64, generator=g)
    mean, var = 0.3 * torch.randn(64, generator=g), 0.5 + torch.rand(64, generator=g)

    def stem(dtype):
        y = F.conv2d(x.to(dtype), w.to(dtype), stride=2, padding=3)
        y = F.batch_norm(y, mean.to(dtype), var.to(dtype), gamma.to(dtype), beta.to(dtype), training=False, eps=1e-5)
        return F.max_pool2d(F.relu(y), 3, 2, 1)

    want64, want32 = stem(torch.float64), stem(torch.float32)
    xd = x.cuda()
    inv = torch.rsqrt(var + 1e-5)
    ya = torch.zeros(1, dtype=torch.int32, device="cuda")
    xa = ops.absmax_word(xd)
    assert xa.view(torch.float32).item() == float(x.abs().max())
    y = ops.stem_conv_bn_relu_pool(xd, xa, ops.PackedStem(w.cuda()), (gamma * inv).cuda(), (beta - mean * gamma * inv).cuda(), y_amax=ya)
    got = y.permute(0, 3, 1, 2).cpu()
    assert got.shape == want64.shape
    e, e32 = rel_err(got, want64), rel_err(want32, want64)
    assert e < 2e-6 and e < 8 * e32 + 2e-7, (e, e32)
    assert abs(ya.view(torch.float32).item() - float(got.abs().max())) <= 1e-6 * float(got.abs().max())


# ResNet-50's bottleneck projections on a 480 x 640 frame (64 -> 256 at 120 x 160 ... 2048 -> 512 at 15 x 20), ragged maps, one pixel
@pytest.mark.parametrize("C_in,C_out,N,H,W", [(64, 256, 2, 120, 160), (256, 64, 2, 120, 160), (512, 128, 2, 60, 80), (1024, 256, 2, 30, 40),
                                             (2048, 512, 2, 15, 20), (512, 2048, 1, 15, 20), (64, 64, 1, 13, 23), (128, 192, 1, 1, 1)])
@pytest.mark.parametrize("with_res,relu", [(True, True), (False, False)])
def test_conv1x1_bn_act_matches_torch_cpu(ops, C_in, C_out, N, H, W, with_res, relu):
    g = torch.Generator().manual_seed(C_in + C_out + H)
    x = torch.randn(N, C_in, H, W, generator=g).abs()
    w = torch.randn(C_out, C_in, 1, 1, generator=g) * (2.0 / C_out) ** 0.5
    gamma, beta = 0.5 + torch.rand(C_out, generator=g), 0.2 * torch.randn(C_out, generator=g)
    mean, var = 0.3 * torch.randn(C_out, generator=g), 0.5 + torch.rand(C_out, generator=g)
    res = torch.randn(N, C_out, H, W, generator=g).abs()

    def block(dtype):
        y = F.batch_norm(F.conv2d(x.to(dtype), w.to(dtype)), mean.to(dtype), var.to(dtype), gamma.to(dtype), beta.to(dtype), training=False, eps=1e-5)
        if with_res:
            y = y + res.to(dtype)
        return F.relu(y) if relu else y

    want64, want32 = block(torch.float64), block(torch.float32)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    inv = torch.rsqrt(var + 1e-5)
    ya = torch.zeros(1, dtype=torch.int32, device="cuda")
    y = ops.conv3x3_bn_act(xh, ops.absmax_word(xh), ops.PackedConv3x3(w.cuda()), (gamma * inv).cuda(), (beta - mean * gamma * inv).cuda(),
                           res=res.permute(0, 2, 3, 1).contiguous().cuda() if with_res else None, relu=relu, y_amax=ya)
    got = y.permute(0, 3, 1, 2).cpu()
    e, e32 = rel_err(got, want64), rel_err(want32, want64)
    assert e < 2e-6 and e < 8 * e32 + 2e-7, (e, e32)
    assert abs(ya.view(torch.float32).item() - float(got.abs().max())) <= 1e-6 * float(got.abs().max())


def test_resnet50_inference_runs_the_hip_kernels_and_matches_cpu_fp32():
    """The reference's other ResNet option (image_encoder_type "resnet50", ml/model/encoder/image.py:62-66): torchvision Bottleneck blocks
    (1 x 1, 3 x 3 with the stride, 1 x 1 x 4; 1 x 1 shortcuts) on sd_conv3x3_bn_act / sd_conv1x1_bn_act / sd_conv_s2_bn_act - 52 convolutions
    + the stem, none on the library - against the same modules on the CPU in fp32 and the all-MIOpen route."""
    import copy

    from soccerdiffusion_amd import ops as o
    from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, image_encoder_factory

    torch.manual_seed(0)
    enc = image_encoder_factory(ImageEncoderType.RESNET50, 64, True, 64)
    enc.train()
    with torch.no_grad():
        enc(torch.rand(2, 2, 3, 64, 96))
    enc.eval()
    x = torch.rand(2, 2, 3, 64, 96, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = enc(x)
    g = copy.deepcopy(enc).cuda().eval()
    calls = {"s1": 0, "s2": 0, "stem": 0}
    orig = (o.conv3x3_bn_act, o.conv_s2_bn_act, o.stem_conv_bn_relu_pool)
    o.conv3x3_bn_act = lambda *a, **k: (calls.__setitem__("s1", calls["s1"] + 1), orig[0](*a, **k))[1]
    o.conv_s2_bn_act = lambda *a, **k: (calls.__setitem__("s2", calls["s2"] + 1), orig[1](*a, **k))[1]
    o.stem_conv_bn_relu_pool = lambda *a, **k: (calls.__setitem__("stem", calls["stem"] + 1), orig[2](*a, **k))[1]
    try:
        with torch.no_grad():
            got = g(x.cuda())
    finally:
        o.conv3x3_bn_act, o.conv_s2_bn_act, o.stem_conv_bn_relu_pool = orig
    # 16 blocks x 3 convolutions + 4 shortcuts = 52: stride 2 in conv2 and the shortcut of layers 2 - 4's first blocks (6)
    assert calls == {"s1": 46, "s2": 6, "stem": 1}, calls
    assert got.shape == (2, 2, 64)
    assert float((got.cpu() - want).abs().max() / want.abs().max()) < 1e-4
    os.environ["SD_CONV"] = "torch"
    try:
        with torch.no_grad():
            lib = g(x.cuda())
    finally:
        del os.environ["SD_CONV"]
    assert float((got - lib).abs().max() / lib.abs().max()) < 1e-4


def test_misaligned_vectors_are_refused_not_dereferenced(ops):
    """The epilogues use 16-byte accesses on y / res / the BatchNorm vectors: a misaligned pointer must come back as SD_E_BADARG from the
    C ABI, not reach the kernel."""
    x = torch.rand(1, 4, 4, 64, device="cuda")
    pk = ops.PackedConv3x3(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
    buf = torch.ones(65, device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv3x3_bn_act(x, ops.absmax_word(x), pk, buf[1:], torch.zeros(64, device="cuda"))
    pk2 = ops.PackedConv3x3(torch.randn(128, 64, 3, 3, device="cuda") * 0.05)
    buf2 = torch.ones(129, device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv_s2_bn_act(x, ops.absmax_word(x), pk2, torch.ones(128, device="cuda"), buf2[1:])
