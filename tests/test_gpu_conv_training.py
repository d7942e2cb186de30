"""SURVEY 8 row f2, TRAINING path of the image backbone (csrc/sd_conv_train.hip, soccerdiffusion_amd/conv_training.py): training-mode
BatchNorm forward / backward, the convolution weight gradient and the data gradient (the forward kernels on flipped, transposed weights;
3 x 3 / stride 2: the transposed convolution by parity classes) against torch's CPU operators in fp64 - per op, per conv + BN unit, and for EVERY parameter of ResNet-18
through the module (reference: torchvision BasicBlock under autograd as soccer_diffusion/ml/model/encoder/image.py:55-83 builds it and
ml/training/train.py:226-240 trains it)."""

import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W,C,relu,with_res", [(2, 9, 11, 64, True, True), (3, 8, 8, 128, False, False), (1, 5, 7, 512, True, False),
                                                    (2, 30, 40, 256, True, True)])
def test_bn_train_forward_and_backward_match_torch_fp64(N, H, W, C, relu, with_res):
    from soccerdiffusion_amd import conv_training as ct

    g = torch.Generator().manual_seed(C + H)
    y = (torch.randn(N, H, W, C, generator=g) * 2 + 3.0)   # a mean well away from zero: the shifted sums must not cancel
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    res = torch.randn(N, H, W, C, generator=g) if with_res else None
    dz = torch.randn(N, H, W, C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    # torch reference in fp64 (NCHW)
    yd = y.double().permute(0, 3, 1, 2).requires_grad_()
    gd, bd = gamma.double().requires_grad_(), beta.double().requires_grad_()
    rd = res.double().permute(0, 3, 1, 2).requires_grad_() if with_res else None
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    out = F.batch_norm(yd, rm_ref, rv_ref, gd, bd, training=True, momentum=0.1, eps=1e-5)
    if with_res:
        out = out + rd
    if relu:
        out = out.relu()
    out.backward(dz.double().permute(0, 3, 1, 2))
    # ours
    rm_g, rv_g = rm.cuda(), rv.cuda()
    z, word, mean, rstd = ct.bn_train_fwd(y.cuda(), gamma.cuda(), beta.cuda(), res.cuda() if with_res else None, rm_g, rv_g, 1e-5, 0.1, relu)
    assert rel_err(z.permute(0, 3, 1, 2), out) < 2e-6
    assert rel_err(rm_g, rm_ref) < 1e-6 and rel_err(rv_g, rv_ref) < 1e-6
    assert abs(float(torch.tensor([int(word.item())], dtype=torch.int32).view(torch.float32)) - float(z.abs().max())) < 1e-6 * float(z.abs().max())
    dy, dword, dgamma, dbeta, dres = ct.bn_train_bwd(dz.cuda(), z if relu else None, y.cuda(), mean, rstd, gamma.cuda(), relu, with_res)
    assert rel_err(dy.permute(0, 3, 1, 2), yd.grad) < 1e-5
    assert rel_err(dgamma, gd.grad) < 1e-5 and rel_err(dbeta, bd.grad) < 1e-5
    if with_res:
        assert rel_err(dres.permute(0, 3, 1, 2), rd.grad) < 1e-6
    if relu and not with_res:
        # without a residual operand the backward may recompute the ReLU mask from y (z = None: one tensor less to read): the SAME mask, bit for bit
        dy2, dword2, dgamma2, dbeta2, _ = ct.bn_train_bwd(dz.cuda(), None, y.cuda(), mean, rstd, gamma.cuda(), True, False, beta.cuda())
        assert torch.equal(dy2, dy) and torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta) and int(dword2.item()) == int(dword.item())


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride", [(2, 9, 11, 64, 64, 3, 1), (1, 17, 33, 64, 128, 3, 2), (2, 8, 8, 128, 64, 1, 1), (1, 10, 12, 64, 128, 1, 2),
                                                     (1, 7, 70, 128, 256, 3, 2), (3, 6, 5, 256, 64, 3, 1)])
def test_conv_weight_and_data_gradients_match_torch_fp64(N, H, W, Cin, Cout, k, stride):
    from soccerdiffusion_amd import conv_training as ct
    from soccerdiffusion_amd import ops

    g = torch.Generator().manual_seed(Cin + Cout + k + stride)
    h = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    hd = h.double().permute(0, 3, 1, 2).requires_grad_()
    wd = w.double().requires_grad_()
    out = F.conv2d(hd, wd, stride=stride, padding=k // 2)
    dy = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dy)
    dy_nhwc = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    dw = ct.conv_wgrad(dy_nhwc, h.cuda(), (Cout, Cin, k, k), stride)
    assert rel_err(dw, wd.grad) < 1e-5
    from soccerdiffusion_amd import ops as o
    dw2 = ct.conv_wgrad(dy_nhwc, h.cuda(), (Cout, Cin, k, k), stride, o.absmax_word(dy_nhwc), o.absmax_word(h.cuda()))   # one scale per operand
    assert rel_err(dw2, wd.grad) < 1e-5
    # forward and data gradient through the packed pair (the data gradient = forward kernel on flipped, transposed weights)
    pair = ct.PackedPair()
    fwd, bwd = pair.get(w.cuda())
    y = ct.conv_raw(h.cuda(), ops.absmax_word(h.cuda()), fwd, stride)
    assert rel_err(y.permute(0, 3, 1, 2), out) < 2e-6
    d = dy_nhwc
    if stride == 2:
        d = torch.zeros(N, H, W, Cout, device="cuda")
        d[:, ::2, ::2] = dy_nhwc
    dh = ct.conv_raw(d, ops.absmax_word(d), bwd, 1)
    assert rel_err(dh.permute(0, 3, 1, 2), hd.grad) < 2e-6
    if stride == 2 and k == 3:   # what ConvBNUnit.backward runs: the transposed convolution without the zero-dilated tensor
        dh2 = ct.convt3x3_s2(dy_nhwc, ops.absmax_word(dy_nhwc), bwd, H, W)
        assert rel_err(dh2.permute(0, 3, 1, 2), hd.grad) < 2e-6
        assert rel_err(dh2, dh) < 1e-6
    if stride == 2 and k == 1:   # the shortcut's data gradient: one parity class written, zeros elsewhere
        dh2 = ct.convt1x1_s2(dy_nhwc, ops.absmax_word(dy_nhwc), bwd, H, W)
        assert rel_err(dh2.permute(0, 3, 1, 2), hd.grad) < 2e-6


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(1, 1, 1, 64, 64), (2, 2, 3, 64, 128), (1, 16, 32, 128, 64), (1, 17, 33, 64, 64), (2, 31, 20, 64, 128),
                                            (1, 120, 160, 64, 128)])
def test_transposed_stride2_convolution_matches_torch_fp64(N, H, W, Cin, Cout):
    """sd_convt3x3_s2 = the input gradient of conv2d(3 x 3, stride 2, padding 1): even / odd maps, maps smaller than a tile, one pixel, the
    layer-2 entry's own shape."""
    from soccerdiffusion_amd import conv_training as ct
    from soccerdiffusion_amd import ops

    g = torch.Generator().manual_seed(H * 7 + W)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    dy = torch.randn(N, Cout, Ho, Wo, generator=g, dtype=torch.float64)
    want = torch.nn.grad.conv2d_input((N, Cin, H, W), w.double(), dy, stride=2, padding=1)
    _, bwd = ct.PackedPair().get(w.cuda())
    dy_nhwc = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    got = ct.convt3x3_s2(dy_nhwc, ops.absmax_word(dy_nhwc), bwd, H, W)
    assert tuple(got.shape) == (N, H, W, Cin)
    assert rel_err(got.permute(0, 3, 1, 2), want) < 2e-6


@pytest.mark.parametrize("shape", [(2, 3, 96, 128), (1, 3, 480, 640)])
def test_every_resnet18_parameter_gradient_matches_torch_cpu_fp64(shape):
    """gradients of every ResNet-18 parameter (and of the input frames) against the same modules on the CPU in fp64,
    train() mode: batch statistics, running statistics and num_batches_tracked updated as torch does."""
    import copy

    from soccerdiffusion_amd import conv_training as ct
    from soccerdiffusion_amd.ml.model.encoder.image import _BasicBlock, _ResNet

    torch.manual_seed(1)
    net = _ResNet(_BasicBlock, [2, 2, 2, 2])
    net.fc = torch.nn.Linear(512, 32)
    with torch.no_grad():   # non-trivial BatchNorm affine parameters and statistics
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    ref = copy.deepcopy(net).double().train()
    gpu = copy.deepcopy(net).cuda().train()
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(2))
    xr = x.double().requires_grad_()
    wr = torch.randn(shape[0], 32, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    out_ref = ref(xr)
    (out_ref * wr).sum().backward()
    calls, wg_rec, bn_rec = [], [], []
    orig, orig_wg, orig_bn, orig_bnf = ct.ConvBNUnit.apply, ct.conv_wgrad, ct.bn_train_bwd, ct.bn_train_fwd
    z_of = {}   # forward outputs by the address of y: the backward of a unit without residual recomputes its ReLU mask instead of reading z

    def bnf_spy(y, *a):
        out = orig_bnf(y, *a)
        z_of[y.data_ptr()] = out[0]
        return out

    def wgrad_spy(dy, h, wshape, stride, *words):
        dw = orig_wg(dy, h, wshape, stride, *words)
        wg_rec.append((dy, h, wshape, stride, dw))
        return dw

    def bn_spy(dz, z, y, mean, rstd, gamma, relu, want_dres, beta=None):
        res = orig_bn(dz, z, y, mean, rstd, gamma, relu, want_dres, beta)
        bn_rec.append((dz, z if z is not None else z_of[y.data_ptr()], y, mean, rstd, gamma, relu, res))
        return res

    ct.ConvBNUnit.apply = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    ct.conv_wgrad, ct.bn_train_bwd, ct.bn_train_fwd = wgrad_spy, bn_spy, bnf_spy
    try:
        xg = x.cuda().requires_grad_()
        out = gpu(xg)
        (out * wr.float().cuda()).sum().backward()
    finally:
        ct.ConvBNUnit.apply, ct.conv_wgrad, ct.bn_train_bwd, ct.bn_train_fwd = orig, orig_wg, orig_bn, orig_bnf
    # every weight-gradient and BatchNorm-backward launch of this backward against fp64 ON THE SAME TENSORS (the kernels' own error, free of
    # what fp32 rounding upstream does to ReLU masks and max-pool winners)
    assert len(wg_rec) == 19 and len(bn_rec) == 19   # (the stem's BatchNorm + ReLU + max-pool run on the fused sd_bn_relu_pool_* kernels: own test below)
    for dy, h, wshape, stride, dw in wg_rec:
        want = torch.nn.grad.conv2d_weight(h.double().cpu().permute(0, 3, 1, 2), wshape, dy.double().cpu().permute(0, 3, 1, 2), stride=stride,
                                           padding=wshape[2] // 2)
        assert rel_err(dw, want) < 1e-5, (wshape, stride, tuple(h.shape))
    for dz, z, y, mean, rstd, gamma, relu, (dy, _w, dgamma, dbeta, dres) in bn_rec:
        g = dz.double().cpu() * ((z.double().cpu() > 0) if relu else 1.0)
        xh = (y.double().cpu() - mean.double().cpu()) * rstd.double().cpu()
        n = g.numel() // g.shape[-1]
        s1, s2 = g.reshape(n, -1).sum(0), (g * xh).reshape(n, -1).sum(0)
        want = gamma.double().cpu() * rstd.double().cpu() * (g - s1 / n - xh * s2 / n)
        assert rel_err(dy, want) < 1e-5 and rel_err(dgamma, s2) < 1e-5 and rel_err(dbeta, s1) < 1e-5
    assert len(calls) == 19   # 16 block convolutions + 3 shortcuts: every unit behind the stem ran on this package's kernels
    assert rel_err(out, out_ref) < 1e-5
    pr, pg = dict(ref.named_parameters()), dict(gpu.named_parameters())
    errs = {name: rel_err(pg[name].grad, p.grad) for name, p in pr.items()}
    errs["input frames"] = rel_err(xg.grad, xr.grad)
    # the same modules on the GPU's torch.nn route (MIOpen) against the same fp64 reference: how well conditioned this backward is in fp32 at all
    lib = copy.deepcopy(net).cuda().train()
    os.environ["SD_CONV"] = "torch"
    try:
        xl = x.cuda().requires_grad_()
        (lib(xl) * wr.float().cuda()).sum().backward()
    finally:
        del os.environ["SD_CONV"]
    pl = dict(lib.named_parameters())
    lerr = {name: rel_err(pl[name].grad, p.grad) for name, p in pr.items()}
    lerr["input frames"] = rel_err(xl.grad, xr.grad)
    print("worst gradient errors:", sorted(errs.items(), key=lambda kv: -kv[1])[:6])
    print("torch.nn route       :", sorted(lerr.items(), key=lambda kv: -kv[1])[:6])
    # End to end an fp32 backward of this depth is only as good as its conditioning: at 480 x 640 a handful of ReLU masks / max-pool winners
    # differ between ANY fp32 forward and the fp64 one and move the early layers' gradients by ~ 5e-3 on both routes alike.  Bar: 2e-5 where
    # the problem is well conditioned (the small frame), never worse than 1.5 x the library route's worst error on the same problem.
    bar = max(2e-5, 1.5 * max(lerr.values()))
    for name, e in errs.items():
        assert e < bar, (name, e, lerr[name], bar)
    # the same backward with frames that need no gradient (what training does): the stem runs on StemPoolUnit (sd_stem_conv_raw / sd_bn_relu_pool_* / sd_stem_wgrad)
    gpu2 = copy.deepcopy(net).cuda().train()
    stem_calls = []
    orig_stem = ct.StemPoolUnit.apply
    ct.StemPoolUnit.apply = lambda *a, **k: (stem_calls.append(1), orig_stem(*a, **k))[1]
    try:
        (gpu2(x.cuda()) * wr.float().cuda()).sum().backward()
    finally:
        ct.StemPoolUnit.apply = orig_stem
    assert len(stem_calls) == 1
    for name, p in gpu2.named_parameters():
        assert rel_err(p.grad, pr[name].grad) < bar, name
    # running statistics and counters after ONE training forward
    bg = dict(gpu.named_buffers())
    for name, b in ref.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(bg[name]) == int(b) == 1, name
        else:
            assert rel_err(bg[name], b) < 1e-5, name


@pytest.mark.parametrize("N,Hc,Wc,C", [(2, 48, 64, 64), (1, 31, 39, 64), (3, 7, 9, 64), (2, 1, 1, 64), (1, 2, 5, 128), (1, 240, 320, 64)])
def test_fused_bn_relu_maxpool_matches_torch_fp64(N, Hc, Wc, C):
    """The stem's bn1 -> relu -> maxpool(3, 2, 1) (torchvision ResNet.forward) as sd_bn_relu_pool_fwd / _bwd: pooled map, running statistics, dy,
    dgamma, dbeta against torch in fp64; even / odd maps, the borders of the pool's padding, a single pixel, the stem's own shape."""
    from soccerdiffusion_amd import conv_training as ct

    g = torch.Generator().manual_seed(N * 100 + Hc + Wc)
    y = torch.randn(N, Hc, Wc, C, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    yd = y.double().permute(0, 3, 1, 2).requires_grad_()
    gd, bd = gamma.double().requires_grad_(), beta.double().requires_grad_()
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    out = F.max_pool2d(F.batch_norm(yd, rm_ref, rv_ref, gd, bd, training=True, momentum=0.1, eps=1e-5).relu(), 3, 2, 1)
    dp = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dp)
    rm_g, rv_g = rm.cuda(), rv.cuda()
    p, word, idx, mean, rstd = ct.bn_relu_pool_fwd(y.cuda(), gamma.cuda(), beta.cuda(), rm_g, rv_g, 1e-5, 0.1)
    assert tuple(p.shape) == (N, (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1, C)
    assert rel_err(p.permute(0, 3, 1, 2), out) < 2e-6
    assert rel_err(rm_g, rm_ref) < 1e-6 and rel_err(rv_g, rv_ref) < 1e-6
    pmax = float(p.abs().max())
    assert abs(float(torch.tensor([int(word.item())], dtype=torch.int32).view(torch.float32)) - pmax) <= 1e-6 * pmax
    dy, dword, dgamma, dbeta = ct.bn_relu_pool_bwd(dp.float().permute(0, 2, 3, 1).contiguous().cuda(), idx, y.cuda(), mean, rstd, gamma.cuda())
    # (a window whose two largest values differ by less than fp32 rounding could pick another winner than fp64: not with these sizes and seeds)
    assert rel_err(dy.permute(0, 3, 1, 2), yd.grad) < 1e-5
    assert rel_err(dgamma, gd.grad) < 1e-5 and rel_err(dbeta, bd.grad) < 1e-5
    assert abs(float(torch.tensor([int(dword.item())], dtype=torch.int32).view(torch.float32)) - float(dy.abs().max())) <= 1e-6 * float(dy.abs().max())


@pytest.mark.parametrize("N,H,W", [(2, 96, 128), (1, 61, 77), (3, 32, 40)])
def test_stem_raw_convolution_and_weight_gradient_match_torch_fp64(N, H, W):
    from soccerdiffusion_amd import conv_training as ct
    from soccerdiffusion_amd import ops

    g = torch.Generator().manual_seed(H + W)
    x = torch.rand(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    wd = w.double().requires_grad_()
    out = F.conv2d(x.double(), wd, stride=2, padding=3)
    dy = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dy)
    xg = x.cuda()
    xa = ops.absmax_word(xg)
    y = ct.stem_conv_raw(xg, xa, ops.PackedStem(w.cuda()))
    assert rel_err(y.permute(0, 3, 1, 2), out) < 2e-6
    dyg = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    dw = ct.stem_wgrad(dyg, xg, ops.absmax_word(dyg), xa)
    assert rel_err(dw, wd.grad) < 1e-5


def test_resnet50_bottleneck_gradients_match_torch_cpu_fp64():
    """The reference's other ResNet option (image_encoder_type "resnet50", ml/model/encoder/image.py:62-66): torchvision Bottleneck blocks under
    autograd - 1 x 1 / 3 x 3 (stride 1 and 2) / 1 x 1 units and 1 x 1 shortcuts with either stride - on ConvBNUnit: every parameter gradient
    against the same modules on the CPU in fp64."""
    import copy

    from soccerdiffusion_amd import conv_training as ct
    from soccerdiffusion_amd.ml.model.encoder.image import _Bottleneck, _ResNet

    torch.manual_seed(4)
    net = _ResNet(_Bottleneck, [2, 1, 1, 1])   # (every block kind of [3, 4, 6, 3] at a quarter of the depth)
    net.fc = torch.nn.Linear(2048, 16)
    ref = copy.deepcopy(net).double().train()
    gpu = copy.deepcopy(net).cuda().train()
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(5))
    wr = torch.randn(2, 16, generator=torch.Generator().manual_seed(6), dtype=torch.float64)
    out_ref = ref(x.double())
    (out_ref * wr).sum().backward()
    calls = []
    orig = ct.ConvBNUnit.apply
    ct.ConvBNUnit.apply = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        out = gpu(x.cuda())
        (out * wr.float().cuda()).sum().backward()
    finally:
        ct.ConvBNUnit.apply = orig
    assert len(calls) == 5 * 3 + 4   # 5 blocks x 3 units + 4 shortcuts
    assert rel_err(out, out_ref) < 1e-5
    pr = dict(ref.named_parameters())
    errs = {name: rel_err(p.grad, pr[name].grad) for name, p in gpu.named_parameters()}
    # End to end, an fp32 backward is only as good as its conditioning: ONE ReLU-mask element of layer1.0's output sits within fp32 rounding of
    # zero on this input and flips against the fp64 forward (tools/exp/r50_diag.py: 1 mismatch of 196 608) - a 100 % change of the gradient
    # at that pixel, which moves layer1.0's and the stem's parameter gradients by ~ 3e-3 while every block behind it stays at 6e-6.  So: the
    # bulk must be tight, nothing may be far off, and the per-kernel accuracy is asserted on identical tensors elsewhere in this file.
    vals = sorted(errs.values())
    assert vals[len(vals) // 2] < 2e-5, vals[len(vals) // 2]
    assert sum(v < 5e-5 for v in vals) >= 0.7 * len(vals), sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    assert vals[-1] < 2e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    late = [e for name, e in errs.items() if name.startswith(("layer2", "layer3", "layer4", "fc"))]
    assert max(late) < 5e-5   # behind the flip's reach the kernels' own accuracy shows


def test_training_step_of_the_image_conditioned_model_uses_the_unit_kernels():
    """Through the boundary class: a train_step of the image-conditioned model (configs[4]'s structure at a small size) runs the backbone's
    blocks on ConvBNUnit and moves every backbone parameter; SD_CONV=torch gives the same loss on the torch.nn route."""
    from test_gpu_image_path import _model

    from soccerdiffusion_amd import conv_training as ct
    from soccerdiffusion_amd.scheduler import DDIMScheduler
    from soccerdiffusion_amd.training import FusedAdamW, train_step

    dev = torch.device("cuda:0")
    losses = {}
    for route in ("hip", "torch"):
        torch.manual_seed(0)
        m = _model(dev).train().set_dropout(0.0)
        opt = FusedAdamW(m.parameters(), lr=1e-3)
        sch = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
        B, Fr, R = 4, 3, 64
        g = torch.Generator().manual_seed(3)
        data = {"joint_command_history": torch.randn(B, 20, 20, generator=g).to(dev), "image_data": torch.rand(B, Fr, 3, R, R, generator=g).to(dev),
                "game_state": torch.randint(0, 4, (B,), generator=g).to(dev)}
        target = torch.randn(B, 12, 20, generator=g).to(dev)
        gen = torch.Generator(device=dev).manual_seed(4)
        calls = []
        orig = ct.ConvBNUnit.apply
        ct.ConvBNUnit.apply = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        if route == "torch":
            os.environ["SD_CONV"] = "torch"
        try:
            losses[route] = [float(train_step(m, opt, None, sch, target, input_data=data, generator=gen)) for _ in range(2)]
        finally:
            ct.ConvBNUnit.apply = orig
            os.environ.pop("SD_CONV", None)
        assert len(calls) == (2 * 19 if route == "hip" else 0)
    assert abs(losses["hip"][0] - losses["torch"][0]) < 1e-4 * abs(losses["torch"][0])
    assert abs(losses["hip"][1] - losses["torch"][1]) < 5e-3 * abs(losses["torch"][1])   # after one update of both replicas


def test_randomised_soak_of_the_image_training_kernels():
    """30 random shapes (maps of 1 .. 44 x 1 .. 70 pixels, 64 - 256 channels, 1 x 1 / 3 x 3, stride 1 / 2, activations of 0.01 .. 30, gradients of
    1e-3 .. 100): the weight gradient on both scale paths, the data gradient (transposed convolutions for stride 2) and the fused BatchNorm + ReLU +
    max-pool forward / backward against torch CPU fp64 (tools/exp/soak_conv_train.py holds the bars)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("soak_conv_train", os.path.join(os.path.dirname(__file__), "..", "tools", "exp", "soak_conv_train.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    worst = mod.soak(30, 7, verbose=False)
    assert max(worst.values()) < 5e-6, worst
