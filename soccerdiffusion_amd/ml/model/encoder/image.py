"""Image context path (reference: soccer_diffusion/ml/model/encoder/image.py:11-174).

Per frame a ResNet-18/50 turns an image into one token; a BaseEncoder with 8 heads
(image.py:116) lets the frames' tokens talk to each other.  The sequence transformer runs
on this package's HIP kernels.  The ResNet backbone is the torchvision
architecture - absent offline, so restated here layer for layer with torchvision's
``state_dict`` keys; architecture parity is unpinned (no torchvision, no reference fixture).
INFERENCE (eval mode, no autograd tape, fp32 CUDA tensors): the whole forward runs on the
hand-written kernels of csrc/sd_conv.hip on NHWC tensors - the stem (conv 7 x 7 / BatchNorm /
ReLU / max-pool in one launch, ``sd_stem_conv_bn_relu_pool``), every 3 x 3 stride-1 block
convolution (``sd_conv3x3_bn_act``), the stride-2 stage entries and 1 x 1 shortcuts
(``sd_conv_s2_bn_act`` / ``sd_conv1x1_bn_act``), BatchNorm folded, residual and ReLU in the
epilogues (``SD_CONV=torch`` in the environment keeps ``torch.nn``).
TRAINING (a tape, or ``train()`` mode): see ``_ResNet.forward``.
Pretrained ImageNet weights cannot be fetched offline (``weights=...DEFAULT`` in the
reference): load them from a reference checkpoint.  The Swin-T/S encoders are restated with
torch ops (shifted-window attention); every shipped YAML uses resnet18.
"""

from __future__ import annotations

from enum import Enum

import os
import weakref

import torch
from torch import nn

from .... import ops
from .encoders import BaseEncoder


class ImageEncoderType(Enum):
    RESNET18 = "resnet18"
    RESNET50 = "resnet50"
    SWIN_TRANSFORMER_TINY = "swin_transformer_tiny"
    SWIN_TRANSFORMER_SMALL = "swin_transformer_small"


class SequenceEncoderType(Enum):
    TRANSFORMER = "transformer"
    NONE = "none"


# Derived tensors of a module (packed weight planes, folded BatchNorm vectors) live OUTSIDE the module - keyed weakly by it - so that
# ``copy.deepcopy`` / pickling of a model neither carries nor shares them.
_DERIVED: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _derived(mod: nn.Module) -> dict:
    d = _DERIVED.get(mod)
    if d is None:
        d = _DERIVED[mod] = {}
    return d


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, planes, 3, stride, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, 1, 1), nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != planes:
            self.downsample = nn.Sequential(_conv(cin, planes, 1, stride), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        return self.relu(self.bn2(self.conv2(out)) + idt)

    @staticmethod
    def _bn_fold(bn: nn.BatchNorm2d):
        """Inference BatchNorm as y = x * s + t per channel; cached on the module until one of its four tensors changes (four tiny
        launches per convolution otherwise - a third of the robot's 10-frame forward)."""
        key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.device, bn.weight.data_ptr(),
               ops.weights_generation())
        store = _derived(bn)
        hit = store.get("fold")
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        s = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
        t = (bn.bias.detach() - bn.running_mean * s).contiguous()
        s = s.contiguous()
        store["fold"] = (key, s, t)
        return s, t

    def _packed(self, name: str, conv: nn.Conv2d) -> "ops.PackedConv3x3":
        store = _derived(conv)
        pk = store.get("planes")
        if pk is None or pk.planes.device != conv.weight.device:
            pk = store["planes"] = ops.PackedConv3x3(conv.weight)
        return pk.refresh(conv.weight)

    def forward_nhwc(self, h: torch.Tensor, amax: torch.Tensor, words: torch.Tensor):
        """Inference on the hand-written kernels: h (N, H, W, C) fp32 NHWC with its abs-max word -> (h', its abs-max word): conv1 / bn1 /
        relu (stride 1: sd_conv3x3_bn_act; a stage entry: sd_conv_s2_bn_act, as is its 1 x 1 shortcut), conv2 / bn2 / + identity / relu.
        ``words``: two zeroed int32 words for the abs-max of conv1's and conv2's outputs."""
        a1, a2 = words[0:1], words[1:2]
        if self.downsample is None:
            out = ops.conv3x3_bn_act(h, amax, self._packed("_pk1", self.conv1), *self._bn_fold(self.bn1), relu=True, y_amax=a1, zero_amax=False)
            idt = h
        else:   # the stage entry: 3 x 3 stride-2 conv1 and the 1 x 1 stride-2 shortcut (sd_conv_s2_bn_act)
            out = ops.conv_s2_bn_act(h, amax, self._packed("_pk1", self.conv1), *self._bn_fold(self.bn1), relu=True, y_amax=a1, zero_amax=False)
            idt = ops.conv_s2_bn_act(h, amax, self._packed("_pkd", self.downsample[0]), *self._bn_fold(self.downsample[1]), relu=False)
        y = ops.conv3x3_bn_act(out, a1, self._packed("_pk2", self.conv2), *self._bn_fold(self.bn2), res=idt, relu=True, y_amax=a2, zero_amax=False)
        return y, a2

    def forward_train_nhwc(self, h: torch.Tensor, amax: torch.Tensor):
        """Training (BatchNorm on batch statistics, autograd) on the hand-written kernels: h (N, H, W, C) NHWC with its abs-max word ->
        (h', its abs-max word).  Three conv + BN units: the 1 x 1 shortcut where the block has one, conv1, conv2 (+ identity)."""
        # conv1's unit hands its input on (an alias): the residual branch hangs off THAT, so its gradient reaches conv1's backward and is added in
        # the data-gradient kernel's epilogue (no separate element-wise launch at the join)
        out, a1, hp = _train_unit(h, amax, self.conv1, self.bn1, None, True, pass_input=True)
        idt = hp if self.downsample is None else _train_unit(hp, amax, self.downsample[0], self.downsample[1], None, False)[0]
        return _train_unit(out, a1, self.conv2, self.bn2, idt, True)


def _train_unit(h, amax, conv, bn, res, relu, pass_input=False):
    """conv + BatchNorm(training) (+ res) (+ ReLU) of a torchvision block under autograd on this package's kernels (conv_training.py)."""
    from .... import conv_training as ct

    store = _derived(conv)
    pair = store.get("pair")
    if pair is None:
        pair = store["pair"] = ct.PackedPair()
    return ct.unit(h, amax, conv, bn, res, relu, pair, pass_input)


def _train_ok(block) -> bool:
    from .... import conv_training as ct

    convs = [m for m in block.modules() if isinstance(m, nn.Conv2d)]
    return all(ct.supported(c) for c in convs)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, planes, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride, 1), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(_conv(cin, planes * 4, 1, stride), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        return self.relu(self.bn3(self.conv3(out)) + idt)

    _packed = _BasicBlock._packed

    def forward_nhwc(self, h: torch.Tensor, amax: torch.Tensor, words: torch.Tensor):
        """Inference on the hand-written kernels (as _BasicBlock.forward_nhwc): conv1 1 x 1 / bn1 / relu, conv2 3 x 3 with the block's stride /
        bn2 / relu, conv3 1 x 1 / bn3 / + identity (a 1 x 1 shortcut with the stride where the shapes differ) / relu.  ``words``: three
        zeroed int32 words for the abs-max of the three outputs."""
        fold = _BasicBlock._bn_fold
        a1, a2, a3 = words[0:1], words[1:2], words[2:3]
        out = ops.conv3x3_bn_act(h, amax, self._packed("_pk1", self.conv1), *fold(self.bn1), relu=True, y_amax=a1, zero_amax=False)
        stride = self.conv2.stride[0]
        conv2 = ops.conv3x3_bn_act if stride == 1 else ops.conv_s2_bn_act
        out = conv2(out, a1, self._packed("_pk2", self.conv2), *fold(self.bn2), relu=True, y_amax=a2, zero_amax=False)
        if self.downsample is None:
            idt = h
        elif stride == 1:
            idt = ops.conv3x3_bn_act(h, amax, self._packed("_pkd", self.downsample[0]), *fold(self.downsample[1]), relu=False)
        else:
            idt = ops.conv_s2_bn_act(h, amax, self._packed("_pkd", self.downsample[0]), *fold(self.downsample[1]), relu=False)
        y = ops.conv3x3_bn_act(out, a2, self._packed("_pk3", self.conv3), *fold(self.bn3), res=idt, relu=True, y_amax=a3, zero_amax=False)
        return y, a3

    def forward_train_nhwc(self, h: torch.Tensor, amax: torch.Tensor):
        """As _BasicBlock.forward_train_nhwc: conv1 1 x 1, conv2 3 x 3 (the block's stride), conv3 1 x 1 (+ identity / 1 x 1 shortcut)."""
        out, a1, hp = _train_unit(h, amax, self.conv1, self.bn1, None, True, pass_input=True)   # (the residual branch: as _BasicBlock)
        idt = hp if self.downsample is None else _train_unit(hp, amax, self.downsample[0], self.downsample[1], None, False)[0]
        out, a2 = _train_unit(out, a1, self.conv2, self.bn2, None, True)
        return _train_unit(out, a2, self.conv3, self.bn3, idt, True)


class _ResNet(nn.Module):
    """torchvision.models.resnet.ResNet, attribute for attribute (conv1, bn1, layer1-4, avgpool, fc)."""

    def __init__(self, block, layers):
        super().__init__()
        self.conv1, self.bn1 = _conv(3, 64, 7, 2, 3), nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers), start=1):
            blocks = []
            for j in range(n):
                blocks.append(block(cin, planes, (1 if i == 1 else 2) if j == 0 else 1))
                cin = planes * block.expansion
            setattr(self, f"layer{i}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _hip_inference(self, x: torch.Tensor) -> bool:
        # fp32 only (a half / bf16 model or an autocast input keeps the torch.nn route)
        return (x.is_cuda and x.dtype == torch.float32 and self.conv1.weight.dtype == torch.float32 and not self.training
                and not torch.is_grad_enabled() and os.environ.get("SD_CONV", "hip") != "torch")

    def _hip_training(self, x: torch.Tensor) -> bool:
        """train() mode (BatchNorm on batch statistics) on fp32 CUDA tensors: the blocks run conv_training.ConvBNUnit.  (eval() mode WITH a tape -
        fine-tuning on frozen statistics - keeps torch.nn, as does SD_CONV=torch.)"""
        if not (x.is_cuda and x.dtype == torch.float32 and self.conv1.weight.dtype == torch.float32 and self.training
                and os.environ.get("SD_CONV", "hip") != "torch"):
            return False
        ok = self.__dict__.get("_train_ok")
        if ok is None:
            ok = self.__dict__["_train_ok"] = all(_train_ok(b) for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer)
        return ok

    def forward(self, x):
        if self._hip_training(x):
            # Training: every convolution and BatchNorm of the backbone on this package's kernels (convolution forward / data gradient / weight
            # gradient, training-mode BatchNorm forward / backward, the stem's max-pool) behind one autograd.Function per conv + BatchNorm
            # unit, on NHWC tensors; the head (avgpool + fc) is torch.
            from .... import conv_training as ct

            # the stem: BatchNorm (batch statistics) + ReLU + max-pool are ONE forward launch and two backward launches (sd_bn_relu_pool_*):
            # relu(bn1(.)) - 3 GB at 160 frames of 480 x 640 - is never written
            if x.requires_grad or os.environ.get("SD_STEM", "hip") == "torch":   # (a differentiable input: torch's convolution has the data gradient)
                y = self.conv1(x).permute(0, 2, 3, 1).contiguous()
                h, amax = ct.bn_pool_unit(y, self.bn1)
            else:   # the stem kernel's bare convolution, and in the backward the stem's own weight-gradient kernel
                store = _derived(self.conv1)
                pk = store.get("planes")
                if pk is None or pk.planes.device != x.device:
                    pk = store["planes"] = ops.PackedStem(self.conv1.weight)
                h, amax = ct.stem_pool_unit(x, self.conv1, self.bn1, pk)
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                for blk in layer:
                    h, amax = blk.forward_train_nhwc(h, amax)
            x = h.permute(0, 3, 1, 2)
            return self.fc(torch.flatten(self.avgpool(x), 1))
        if self._hip_inference(x):
            # the whole inference forward on the hand-written kernels: the stem (conv1 / bn1 / relu / maxpool) in one launch from the NCHW
            # frames to an NHWC map, the basic blocks on NHWC tensors, back to an NCHW view for the head
            x = x.contiguous()
            store = _derived(self.conv1)
            pk = store.get("planes")
            if pk is None or pk.planes.device != x.device:
                pk = store["planes"] = ops.PackedStem(self.conv1.weight)
            # the abs-max words of the forward's activation tensors (frames, stem, up to three per block): one fill, allocated per call
            # (forwards on different streams / threads must not share them; the caching allocator hands the same block back)
            words = torch.zeros(64, dtype=torch.int32, device=x.device)
            amax = words[1:2]
            h = ops.stem_conv_bn_relu_pool(x, ops.absmax_word(x, words[0:1], zero=False), pk.refresh(self.conv1.weight), *_BasicBlock._bn_fold(self.bn1),
                                           y_amax=amax, zero_amax=False)
            at = 2
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                for blk in layer:
                    h, amax = blk.forward_nhwc(h, amax, words[at:at + 3])
                    at += 3
            x = h.permute(0, 3, 1, 2)
        else:
            x = self.layer4(self.layer3(self.layer2(self.layer1(self.maxpool(self.relu(self.bn1(self.conv1(x))))))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


# ---- torchvision.models.swin_transformer (V1), attribute for attribute ------------------------------
class _SwinMLP(nn.Sequential):          # torchvision.ops.MLP: Linear, GELU, Dropout, Linear, Dropout -> keys mlp.0.*, mlp.3.*
    def __init__(self, dim: int):
        super().__init__(nn.Linear(dim, 4 * dim), nn.GELU(), nn.Dropout(0.0), nn.Linear(4 * dim, dim), nn.Dropout(0.0))


class _ShiftedWindowAttention(nn.Module):
    def __init__(self, dim: int, window: int, shift: int, heads: int):
        super().__init__()
        self.window, self.shift, self.heads = window, shift, heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window - 1) ** 2, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        coords = torch.stack(torch.meshgrid(torch.arange(window), torch.arange(window), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window - 1
        rel[:, :, 1] += window - 1
        rel[:, :, 0] *= 2 * window - 1
        self.register_buffer("relative_position_index", rel.sum(-1).flatten())

    def forward(self, x: torch.Tensor) -> torch.Tensor:   # (B, H, W, C)
        B, H, W, C = x.shape
        w = self.window
        pad_r, pad_b = (w - W % w) % w, (w - H % w) % w
        x = nn.functional.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        _, pH, pW, _ = x.shape
        sh = [0 if w >= pH else self.shift, 0 if w >= pW else self.shift]   # no shift when the window covers the map
        if sum(sh) > 0:
            x = torch.roll(x, shifts=(-sh[0], -sh[1]), dims=(1, 2))
        nW = (pH // w) * (pW // w)
        x = x.view(B, pH // w, w, pW // w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, w * w, C)
        qkv = self.qkv(x).reshape(x.shape[0], x.shape[1], 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (C // self.heads) ** -0.5, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index].view(w * w, w * w, -1).permute(2, 0, 1)
        attn = attn + bias.unsqueeze(0)
        if sum(sh) > 0:   # tokens that came from different sides of the roll must not see each other
            mask = x.new_zeros((pH, pW))
            hs = ((0, -w), (-w, -sh[0]), (-sh[0], None))
            ws = ((0, -w), (-w, -sh[1]), (-sh[1], None))
            count = 0
            for h0, h1 in hs:
                for w0, w1 in ws:
                    mask[h0:h1, w0:w1] = count
                    count += 1
            mask = mask.view(pH // w, w, pW // w, w).permute(0, 2, 1, 3).reshape(nW, w * w)
            mask = mask.unsqueeze(1) - mask.unsqueeze(2)
            mask = mask.masked_fill(mask != 0, -100.0).masked_fill(mask == 0, 0.0)
            attn = attn.view(B, nW, self.heads, w * w, w * w) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.heads, w * w, w * w)
        x = (attn.softmax(dim=-1) @ v).transpose(1, 2).reshape(B * nW, w * w, C)
        x = self.proj(x)
        x = x.view(B, pH // w, pW // w, w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B, pH, pW, C)
        if sum(sh) > 0:
            x = torch.roll(x, shifts=(sh[0], sh[1]), dims=(1, 2))
        return x[:, :H, :W, :].contiguous()


class _StochasticDepth(nn.Module):       # torchvision.ops.StochasticDepth(p, "row")
    def __init__(self, p: float):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = 1.0 - self.p
        noise = torch.empty([x.shape[0]] + [1] * (x.dim() - 1), dtype=x.dtype, device=x.device).bernoulli_(keep)
        return x * noise.div_(keep)


class _SwinBlock(nn.Module):
    def __init__(self, dim, heads, window, shift, sd_prob):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn = _ShiftedWindowAttention(dim, window, shift, heads)
        self.stochastic_depth = _StochasticDepth(sd_prob)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.mlp = _SwinMLP(dim)

    def forward(self, x):
        x = x + self.stochastic_depth(self.attn(self.norm1(x)))
        return x + self.stochastic_depth(self.mlp(self.norm2(x)))


class _PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim, eps=1e-5)

    def forward(self, x):   # (B, H, W, C) -> (B, H/2, W/2, 2C)
        H, W = x.shape[1:3]
        x = nn.functional.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[..., 0::2, 0::2, :], x[..., 1::2, 0::2, :], x[..., 0::2, 1::2, :], x[..., 1::2, 1::2, :]], -1)
        return self.reduction(self.norm(x))


class _Permute(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.dims = dims

    def forward(self, x):
        return x.permute(*self.dims)


class _SwinTransformer(nn.Module):
    """torchvision.models.swin_t / swin_s (patch 4, embed 96, heads 3-6-12-24, window 7), same ``state_dict`` keys."""

    def __init__(self, depths, sd_prob):
        super().__init__()
        dim, heads, window = 96, (3, 6, 12, 24), 7
        layers = [nn.Sequential(nn.Conv2d(3, dim, kernel_size=4, stride=4), _Permute([0, 2, 3, 1]), nn.LayerNorm(dim, eps=1e-5))]
        total, idx = sum(depths), 0
        for stage, depth in enumerate(depths):
            d = dim * 2 ** stage
            blocks = []
            for i in range(depth):
                blocks.append(_SwinBlock(d, heads[stage], window, 0 if i % 2 == 0 else window // 2, sd_prob * idx / (total - 1.0)))
                idx += 1
            layers.append(nn.Sequential(*blocks))
            if stage < len(depths) - 1:
                layers.append(_PatchMerging(d))
        self.features = nn.Sequential(*layers)
        num_features = dim * 2 ** (len(depths) - 1)
        self.norm = nn.LayerNorm(num_features, eps=1e-5)
        self.permute = _Permute([0, 3, 1, 2])
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.flatten = nn.Flatten(1)
        self.head = nn.Linear(num_features, 1000)
        for m in self.modules():   # torchvision: every Linear trunc_normal(0.02), zero bias
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        return self.head(self.flatten(self.avgpool(self.permute(self.norm(self.features(x))))))


class AbstractImageEncoder(nn.Module):
    encoder: nn.Module

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, F, 3, R, R) -> one token per frame (B, F, hidden_dim)."""
        tokens = self.encoder(x.reshape(-1, *x.shape[2:]))
        return tokens.view(x.shape[0], x.shape[1], -1)


class ResNetImageEncoder(AbstractImageEncoder):
    def __init__(self, resnet_type: ImageEncoderType, hidden_dim: int, use_final_avgpool: bool, resolution: int):
        super().__init__()
        if resnet_type == ImageEncoderType.RESNET18:
            self.encoder = _ResNet(_BasicBlock, (2, 2, 2, 2))
        elif resnet_type == ImageEncoderType.RESNET50:
            self.encoder = _ResNet(_Bottleneck, (3, 4, 6, 3))
        else:
            raise ValueError(f"Invalid ResNet type: {resnet_type}")
        feat = self.encoder.fc.in_features
        if use_final_avgpool:
            self.encoder.fc = nn.Linear(feat, hidden_dim)
        else:  # a 1x1 conv to 32 channels in place of the pooling, then one linear layer over the map (image.py:69-73)
            self.encoder.avgpool = nn.Conv2d(feat, 32, 1)
            self.encoder.fc = nn.Linear(self.calculate_output_size(resolution) ** 2 * 32, hidden_dim)

    @staticmethod
    def calculate_output_size(resolution: int) -> int:
        resolution = (resolution - 7 + 2 * 3) // 2 + 1   # conv1
        resolution = (resolution - 3 + 2 * 1) // 2 + 1   # maxpool
        return resolution // 2 // 2 // 2                 # layer2-4


class TransformerImageSequenceEncoder(nn.Module):
    def __init__(self, image_encoder: AbstractImageEncoder, hidden_dim: int, num_layers: int, max_seq_len: int):
        super().__init__()
        self.image_encoder = image_encoder
        self.transformer_encoder = BaseEncoder(input_dim=hidden_dim, patch_size=1, hidden_dim=hidden_dim,
                                               num_layers=num_layers, num_heads=8, max_seq_len=max_seq_len)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.transformer_encoder(self.image_encoder(x).contiguous())


class SwinTransformerImageEncoder(AbstractImageEncoder):
    def __init__(self, swin_type: ImageEncoderType, hidden_dim: int):
        super().__init__()
        if swin_type == ImageEncoderType.SWIN_TRANSFORMER_TINY:
            self.encoder = _SwinTransformer((2, 2, 6, 2), 0.2)
        elif swin_type == ImageEncoderType.SWIN_TRANSFORMER_SMALL:
            self.encoder = _SwinTransformer((2, 2, 18, 2), 0.3)
        else:
            raise ValueError(f"Invalid Swin Transformer type: {swin_type}")
        self.encoder.head = nn.Linear(self.encoder.head.in_features, hidden_dim)


def image_encoder_factory(encoder_type: ImageEncoderType, hidden_dim: int, use_final_avgpool: bool, resolution: int):
    if encoder_type in (ImageEncoderType.RESNET18, ImageEncoderType.RESNET50):
        return ResNetImageEncoder(encoder_type, hidden_dim, use_final_avgpool, resolution)
    if encoder_type in (ImageEncoderType.SWIN_TRANSFORMER_TINY, ImageEncoderType.SWIN_TRANSFORMER_SMALL):
        return SwinTransformerImageEncoder(encoder_type, hidden_dim)
    raise ValueError(f"Invalid image encoder type: {encoder_type}")


def image_sequence_encoder_factory(encoder_type: SequenceEncoderType, image_encoder_type: ImageEncoderType, hidden_dim: int,
                                   num_layers: int, max_seq_len: int, use_final_avgpool: bool, resolution: int):
    image_encoder = image_encoder_factory(image_encoder_type, hidden_dim, use_final_avgpool, resolution)
    if encoder_type == SequenceEncoderType.TRANSFORMER:
        return TransformerImageSequenceEncoder(image_encoder, hidden_dim, num_layers, max_seq_len)
    if encoder_type == SequenceEncoderType.NONE:
        return image_encoder
    raise ValueError(f"Invalid sequence encoder type: {encoder_type}")
