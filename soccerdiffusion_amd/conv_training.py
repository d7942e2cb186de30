"""Training path of the image backbone on this package's kernels: one ``torch.autograd.Function`` per
convolution + BatchNorm (+ residual) (+ ReLU) unit of a torchvision BasicBlock / Bottleneck, NHWC tensors.

Reference: the backbone is trained with every step (soccer_diffusion/ml/training/train.py:226-240 calls ``model(batch, ...)`` ->
ml/model/encoder/image.py:38-52 -> torchvision ResNet under autograd, BatchNorm2d in training mode).

forward   y = conv(h)                      sd_conv3x3_bn_act / sd_conv1x1_bn_act / sd_conv_s2_bn_act with an identity epilogue (csrc/sd_conv.hip)
          z = relu?(BN_train(y) (+ res))   sd_bn_train_fwd: batch statistics, running statistics as torch updates them, abs-max word of z
backward  dy, dgamma, dbeta (, dres)       sd_bn_train_bwd
          dW                               sd_conv_wgrad (csrc/sd_conv_train.hip)
          dh                               the forward convolution kernel on the flipped, transposed weights; 3 x 3 / stride 2: sd_convt3x3_s2 (the
                                           transposed convolution by output parity classes); the 1 x 1 / stride-2 shortcut: sd_convt1x1_s2
stem      p = maxpool(relu(BN_train(conv7x7(x))))   sd_stem_conv_raw, sd_bn_relu_pool_fwd (BatchNorm + ReLU + max-pool in one pass, relu(BN(.))
                                           never written), sd_bn_relu_pool_bwd, sd_stem_wgrad  (StemPoolUnit / BNPoolUnit)
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, ops
from ._lib import check

Tensor = torch.Tensor
_const_cache: dict = {}


def _ones_zeros(n: int, device):
    key = (n, torch.device(device))
    hit = _const_cache.get(key)
    if hit is None:
        hit = _const_cache[key] = (torch.ones(n, dtype=torch.float32, device=device), torch.zeros(n, dtype=torch.float32, device=device))
    return hit


def conv_raw(h: Tensor, amax: Tensor, pk: "ops.PackedConv3x3", stride: int, add: Optional[Tensor] = None) -> Tensor:
    """The bare convolution (k = pk.ksize, padding k // 2, ``stride`` 1 or 2) of an NHWC tensor: the inference kernels with scale 1, shift 0, no ReLU
    (+ ``add`` in the epilogue, stride 1: a data gradient joins the residual branch's gradient there)."""
    one, zero = _ones_zeros(pk.Cout, h.device)
    if stride == 1:
        return ops.conv3x3_bn_act(h, amax, pk, one, zero, res=add, relu=False)
    if add is not None:
        raise ValueError("conv_raw: add needs stride 1")
    return ops.conv_s2_bn_act(h, amax, pk, one, zero, relu=False)


def convt3x3_s2(dy: Tensor, dy_amax: Tensor, pk: "ops.PackedConv3x3", H: int, W: int, add: Optional[Tensor] = None) -> Tensor:
    """Data gradient of a 3 x 3 / stride-2 / padding-1 convolution with input (N, H, W, pk.Cout): dy (N, (H + 1) // 2, (W + 1) // 2, pk.Cin), pk = the
    flipped, transposed weights - the stride-1 convolution of the zero-dilated dy without the zeros (sd_convt3x3_s2: four parity classes)."""
    ops._req(dy, "dy")
    N, Ho, Wo, Cin = dy.shape
    if Cin != pk.Cin or pk.ksize != 3 or Ho != (H + 1) // 2 or Wo != (W + 1) // 2:
        raise ValueError("convt3x3_s2: shape mismatch")
    one, zero = _ones_zeros(pk.Cout, dy.device)
    dx = torch.empty(N, H, W, pk.Cout, dtype=torch.float32, device=dy.device)
    if add is not None:
        ops._req(add, "add")
        if add.shape != dx.shape:
            raise ValueError("convt3x3_s2: add shape mismatch")
    check(_lib.load().sd_convt3x3_s2(dy.data_ptr(), pk.planes.data_ptr(), pk.scale.data_ptr(), dy_amax.data_ptr(), one.data_ptr(), zero.data_ptr(),
                                     ops._ptr(add), dx.data_ptr(), None, N, H, W, Cin, pk.Cout, ops._stream()), "sd_convt3x3_s2")
    return dx


def convt1x1_s2(dy: Tensor, dy_amax: Tensor, pk: "ops.PackedConv3x3", H: int, W: int) -> Tensor:
    """Data gradient of a 1 x 1 / stride-2 convolution (the stage entries' shortcut): dx[:, ::2, ::2] = dy . w^T, zero elsewhere."""
    ops._req(dy, "dy")
    N, Ho, Wo, Cin = dy.shape
    if Cin != pk.Cin or pk.ksize != 1 or Ho != (H + 1) // 2 or Wo != (W + 1) // 2:
        raise ValueError("convt1x1_s2: shape mismatch")
    one, zero = _ones_zeros(pk.Cout, dy.device)
    dx = torch.zeros(N, H, W, pk.Cout, dtype=torch.float32, device=dy.device)
    check(_lib.load().sd_convt1x1_s2(dy.data_ptr(), pk.planes.data_ptr(), pk.scale.data_ptr(), dy_amax.data_ptr(), one.data_ptr(), zero.data_ptr(),
                                     dx.data_ptr(), N, H, W, Cin, pk.Cout, ops._stream()), "sd_convt1x1_s2")
    return dx


def bn_train_fwd(y: Tensor, gamma: Tensor, beta: Tensor, res: Optional[Tensor], running_mean: Optional[Tensor], running_var: Optional[Tensor],
                 eps: float, momentum: float, relu: bool):
    """-> (z, z_amax word, mean, rstd) for the NHWC tensor y; running statistics updated in place."""
    lib = _lib.load()
    Cn = y.shape[-1]
    npix = y.numel() // Cn
    z = torch.empty_like(y)
    mean = torch.empty(Cn, dtype=torch.float32, device=y.device)
    rstd = torch.empty_like(mean)
    acc = torch.empty(2 * Cn, dtype=torch.float64, device=y.device)
    scratch = torch.empty(lib.sd_bn_scratch_floats(npix, Cn), dtype=torch.float32, device=y.device)
    word = torch.zeros(1, dtype=torch.int32, device=y.device)
    check(lib.sd_bn_train_fwd(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ops._ptr(res), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                              ops._ptr(running_mean), ops._ptr(running_var), acc.data_ptr(), scratch.data_ptr(), word.data_ptr(), npix, Cn, float(eps), float(momentum),
                              int(relu), ops._stream()), "sd_bn_train_fwd")
    return z, word, mean, rstd


def bn_train_bwd(dz: Tensor, z: Optional[Tensor], y: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, relu: bool, want_dres: bool,
                 beta: Optional[Tensor] = None):
    """-> (dy, dy_amax word, dgamma, dbeta, dres or None).  ``z`` None with ``relu``: a unit WITHOUT residual operand, the ReLU mask is recomputed
    from y with ``beta`` (the kernels read one tensor less)."""
    lib = _lib.load()
    Cn = y.shape[-1]
    npix = y.numel() // Cn
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    dgamma = torch.empty(Cn, dtype=torch.float32, device=y.device)
    dbeta = torch.empty_like(dgamma)
    acc = torch.empty(2 * Cn, dtype=torch.float64, device=y.device)
    scratch = torch.empty(lib.sd_bn_scratch_floats(npix, Cn), dtype=torch.float32, device=y.device)
    word = torch.zeros(1, dtype=torch.int32, device=y.device)
    check(lib.sd_bn_train_bwd(dz.data_ptr(), ops._ptr(z), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), ops._ptr(beta), dy.data_ptr(), ops._ptr(dres),
                              dgamma.data_ptr(), dbeta.data_ptr(), acc.data_ptr(), scratch.data_ptr(), word.data_ptr(), npix, Cn, int(relu), ops._stream()), "sd_bn_train_bwd")
    return dy, word, dgamma, dbeta, dres


def bn_relu_pool_fwd(y: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor], eps: float, momentum: float):
    """max_pool2d(relu(BatchNorm_train(y)), 3, 2, 1) of the NHWC tensor y in one pass -> (p, p_amax word, idx, mean, rstd); relu(BN(y)) is never
    written.  idx: one byte per pooled element (the winner's position in its window) for the backward."""
    lib = _lib.load()
    N, Hc, Wc, Cn = y.shape
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    p = torch.empty(N, Hp, Wp, Cn, dtype=torch.float32, device=y.device)
    idx = torch.empty(N, Hp, Wp, Cn // 4, dtype=torch.int32, device=y.device)
    mean = torch.empty(Cn, dtype=torch.float32, device=y.device)
    rstd = torch.empty_like(mean)
    acc = torch.empty(2 * Cn, dtype=torch.float64, device=y.device)
    scratch = torch.empty(lib.sd_bn_scratch_floats(N * Hc * Wc, Cn), dtype=torch.float32, device=y.device)
    word = torch.zeros(1, dtype=torch.int32, device=y.device)
    check(lib.sd_bn_relu_pool_fwd(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), p.data_ptr(), idx.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                  ops._ptr(running_mean), ops._ptr(running_var), acc.data_ptr(), scratch.data_ptr(), word.data_ptr(), N, Hc, Wc, Cn,
                                  float(eps), float(momentum), ops._stream()), "sd_bn_relu_pool_fwd")
    return p, word, idx, mean, rstd


def bn_relu_pool_bwd(dp: Tensor, idx: Tensor, y: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor):
    """-> (dy, dy_amax word, dgamma, dbeta): the max-pool's, the ReLU's and the BatchNorm's backward in two launches over y (no dz tensor)."""
    lib = _lib.load()
    N, Hc, Wc, Cn = y.shape
    dy = torch.empty_like(y)
    dgamma = torch.empty(Cn, dtype=torch.float32, device=y.device)
    dbeta = torch.empty_like(dgamma)
    acc = torch.empty(2 * Cn, dtype=torch.float64, device=y.device)
    scratch = torch.empty(lib.sd_bn_scratch_floats(N * Hc * Wc, Cn), dtype=torch.float32, device=y.device)
    word = torch.zeros(1, dtype=torch.int32, device=y.device)
    check(lib.sd_bn_relu_pool_bwd(dp.data_ptr(), idx.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dy.data_ptr(),
                                  dgamma.data_ptr(), dbeta.data_ptr(), acc.data_ptr(), scratch.data_ptr(), word.data_ptr(), N, Hc, Wc, Cn, ops._stream()),
          "sd_bn_relu_pool_bwd")
    return dy, word, dgamma, dbeta


def conv_wgrad(dy: Tensor, h: Tensor, weight_shape, stride: int, dy_amax: Optional[Tensor] = None, h_amax: Optional[Tensor] = None) -> Tensor:
    """dW (Cout, Cin, k, k) of the k x k / padding k // 2 / ``stride`` convolution with input h (N,H,W,Cin) and output gradient dy (N,Ho,Wo,Cout).
    With both abs-max words the kernel uses one fp16 scale per operand; without, block floating point per 32 pixels."""
    lib = _lib.load()
    Cout, Cin, k, _ = weight_shape
    N, H, W, _ = h.shape
    dw = torch.zeros(Cout, Cin, k, k, dtype=torch.float32, device=h.device)
    both = dy_amax is not None and h_amax is not None
    n_scratch = lib.sd_conv_wgrad_scratch_floats(N, H, W, Cin, Cout, k, stride) if both else 0
    scratch = torch.empty(n_scratch, dtype=torch.float32, device=h.device) if n_scratch else None
    check(lib.sd_conv_wgrad(dy.data_ptr(), h.data_ptr(), ops._ptr(dy_amax) if both else None, ops._ptr(h_amax) if both else None, dw.data_ptr(),
                            ops._ptr(scratch), N, H, W, Cin, Cout, k, stride, ops._stream()), "sd_conv_wgrad")
    return dw


class PackedPair:
    """The forward planes of a convolution weight and the planes of its flipped transpose (the data gradient's weight), repacked together when
    the weight changes (version counter or ``ops.weights_generation()``)."""

    def __init__(self):
        self.fwd = self.bwd = None
        self.key = None

    def get(self, weight: Tensor):
        key = (weight._version, ops.weights_generation(), weight.data_ptr())
        if self.key != key or self.fwd.planes.device != weight.device:
            w = weight.detach()
            wt = w.flip(2, 3).transpose(0, 1).contiguous()
            self.fwd, self.bwd = ops.PackedConv3x3(w.contiguous()), ops.PackedConv3x3(wt)
            self.key = key
        return self.fwd, self.bwd


class ConvBNUnit(torch.autograd.Function):
    """z, z_amax = relu?(BatchNorm_train(conv(h)) (+ res)) on NHWC tensors.  With ``pass_input`` the unit also returns h itself (an alias): a block
    hands THAT to its residual branch (the identity or the 1 x 1 shortcut), so the branch's gradient arrives in this unit's backward and is added in
    the data-gradient kernel's epilogue instead of by a separate element-wise launch over the whole tensor."""

    @staticmethod
    def forward(ctx, h, amax, weight, gamma, beta, res, pair: PackedPair, running_mean, running_var, stride: int, relu: bool, eps: float, momentum: float,
                pass_input: bool = False):
        hc = h.contiguous()
        fwd, bwd = pair.get(weight)
        y = conv_raw(hc, amax, fwd, stride)
        z, word, mean, rstd = bn_train_fwd(y, gamma.detach(), beta.detach(), res, running_mean, running_var, eps, momentum, relu)
        # the ReLU mask of a unit without residual operand is recomputed from y in the backward (z is not read there)
        ctx.save_for_backward(hc, y, z if relu and res is not None else None, mean, rstd, gamma, amax, beta)
        ctx.cfg = (bwd, tuple(weight.shape), stride, relu, res is not None)
        ctx.mark_non_differentiable(word)
        ctx.set_materialize_grads(False)
        if pass_input:
            return z, word, h
        return z, word

    @staticmethod
    def backward(ctx, dz, _dword=None, dpass=None):
        h, y, z, mean, rstd, gamma, h_amax, beta = ctx.saved_tensors
        bwd, wshape, stride, relu, has_res = ctx.cfg
        if dz is None:   # only the passed-through input was used downstream
            return dpass, None, None, None, None, None, None, None, None, None, None, None, None, None
        dy, word, dgamma, dbeta, dres = bn_train_bwd(dz.contiguous(), z, y, mean, rstd, gamma.detach(), relu, has_res and ctx.needs_input_grad[5],
                                                     beta.detach())
        dW = conv_wgrad(dy, h, wshape, stride, word, h_amax) if ctx.needs_input_grad[2] else None
        dh = None
        if ctx.needs_input_grad[0]:
            add = None if dpass is None else dpass.contiguous()
            if stride == 2 and wshape[2] == 3:   # the transposed convolution by parity classes (no zero-dilated tensor)
                dh = convt3x3_s2(dy, word, bwd, h.shape[1], h.shape[2], add)
            elif stride == 2:                    # the 1 x 1 shortcut: one parity class, the rest of dh is zero
                dh = convt1x1_s2(dy, word, bwd, h.shape[1], h.shape[2])
                if add is not None:
                    dh = dh + add
            else:
                dh = conv_raw(dy, word, bwd, 1, add)
        return dh, None, dW, dgamma, dbeta, dres, None, None, None, None, None, None, None, None


def stem_conv_raw(x: Tensor, x_amax: Tensor, pk: "ops.PackedStem") -> Tensor:
    """The bare stem convolution (7 x 7, stride 2, padding 3, 3 -> 64) of NCHW frames -> NHWC map (N, Hc, Wc, 64): the inference stem kernel without
    its BatchNorm / ReLU / max-pool epilogue."""
    ops._req(x, "x")
    N, _, H, W = x.shape
    y = torch.empty(N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64, dtype=torch.float32, device=x.device)
    check(_lib.load().sd_stem_conv_raw(x.data_ptr(), pk.planes.data_ptr(), pk.scale.data_ptr(), x_amax.data_ptr(), y.data_ptr(), N, H, W, ops._stream()),
          "sd_stem_conv_raw")
    return y


def stem_wgrad(dy: Tensor, x: Tensor, dy_amax: Tensor, x_amax: Tensor) -> Tensor:
    """dW (64, 3, 7, 7) of the stem convolution from dy (N, Hc, Wc, 64) NHWC and the frames x (N, 3, H, W)."""
    lib = _lib.load()
    N, _, H, W = x.shape
    dw = torch.empty(64, 3, 7, 7, dtype=torch.float32, device=x.device)
    scratch = torch.empty(lib.sd_stem_wgrad_scratch_floats(N, H, W), dtype=torch.float32, device=x.device)
    check(lib.sd_stem_wgrad(dy.data_ptr(), x.data_ptr(), dy_amax.data_ptr(), x_amax.data_ptr(), dw.data_ptr(), scratch.data_ptr(), N, H, W, ops._stream()),
          "sd_stem_wgrad")
    return dw


class StemPoolUnit(torch.autograd.Function):
    """p, p_amax = maxpool3x3/s2(relu(BatchNorm_train(conv7x7/s2(x)))): NCHW frames -> pooled NHWC map (torchvision ResNet's whole stem).  The
    frames receive no gradient; relu(BN(.)) (3 GB at 160 frames of 480 x 640) is never written, in either direction."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, pk: "ops.PackedStem", running_mean, running_var, eps: float, momentum: float):
        x = x.contiguous()
        x_amax = ops.absmax_word(x)
        y = stem_conv_raw(x, x_amax, pk.refresh(weight))
        p, word, idx, mean, rstd = bn_relu_pool_fwd(y, gamma.detach(), beta.detach(), running_mean, running_var, eps, momentum)
        ctx.save_for_backward(x, x_amax, y, idx, mean, rstd, gamma)
        ctx.mark_non_differentiable(word)
        return p, word

    @staticmethod
    def backward(ctx, dp, _dword):
        x, x_amax, y, idx, mean, rstd, gamma = ctx.saved_tensors
        dy, word, dgamma, dbeta = bn_relu_pool_bwd(dp.contiguous(), idx, y, mean, rstd, gamma.detach())
        dW = stem_wgrad(dy, x, word, x_amax) if ctx.needs_input_grad[1] else None
        return None, dW, dgamma, dbeta, None, None, None, None, None


class BNPoolUnit(torch.autograd.Function):
    """p, p_amax = maxpool3x3/s2(relu(BatchNorm_train(y))) of an NHWC tensor (behind torch's stem convolution when the frames need a gradient)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps: float, momentum: float):
        y = y.contiguous()
        p, word, idx, mean, rstd = bn_relu_pool_fwd(y, gamma.detach(), beta.detach(), running_mean, running_var, eps, momentum)
        ctx.save_for_backward(y, idx, mean, rstd, gamma)
        ctx.mark_non_differentiable(word)
        return p, word

    @staticmethod
    def backward(ctx, dp, _dword):
        y, idx, mean, rstd, gamma = ctx.saved_tensors
        dy, _word, dgamma, dbeta = bn_relu_pool_bwd(dp.contiguous(), idx, y, mean, rstd, gamma.detach())
        return dy, dgamma, dbeta, None, None, None, None


def _bn_args(bn: torch.nn.BatchNorm2d):
    momentum = 0.1 if bn.momentum is None else bn.momentum
    track = bn.track_running_stats and bn.running_mean is not None
    return momentum, track


def stem_pool_unit(x: Tensor, conv: torch.nn.Conv2d, bn: torch.nn.BatchNorm2d, pk: "ops.PackedStem"):
    momentum, track = _bn_args(bn)
    p, word = StemPoolUnit.apply(x, conv.weight, bn.weight, bn.bias, pk, bn.running_mean if track else None, bn.running_var if track else None, bn.eps, momentum)
    if track and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return p, word


def bn_pool_unit(y: Tensor, bn: torch.nn.BatchNorm2d):
    momentum, track = _bn_args(bn)
    p, word = BNPoolUnit.apply(y, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None, bn.eps, momentum)
    if track and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return p, word


def unit(h: Tensor, amax: Tensor, conv: torch.nn.Conv2d, bn: torch.nn.BatchNorm2d, res: Optional[Tensor], relu: bool, pair: PackedPair,
         pass_input: bool = False):
    """One conv + BatchNorm(train) (+ res) (+ ReLU) unit of a torchvision block on NHWC tensors; bumps ``num_batches_tracked`` like torch.
    ``pass_input``: -> (z, word, h) with h an alias of the input for the block's residual branch (see ConvBNUnit)."""
    momentum = 0.1 if bn.momentum is None else bn.momentum
    track = bn.track_running_stats and bn.running_mean is not None
    out = ConvBNUnit.apply(h, amax, conv.weight, bn.weight, bn.bias, res, pair, bn.running_mean if track else None, bn.running_var if track else None,
                           conv.stride[0], relu, bn.eps, momentum, pass_input)
    if track and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return out


def supported(conv: torch.nn.Conv2d) -> bool:
    k, s = conv.kernel_size[0], conv.stride[0]
    ok = (k in (1, 3) and s in (1, 2) and conv.padding[0] == k // 2 and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0
          and conv.bias is None and conv.groups == 1 and conv.dilation[0] == 1)
    return ok and (s == 1 or conv.out_channels % 128 == 0)
