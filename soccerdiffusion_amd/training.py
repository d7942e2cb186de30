"""Training side of the hot path: forward/backward of the denoiser and the context encoders
as autograd nodes whose every computation is a HIP kernel, the fused AdamW step on flat
buffers, and the data-parallel gradient all-reduce.

Reference semantics: one iteration of soccer_diffusion/ml/training/train.py:204-240
(normalise, t ~ randint, eps ~ randn, add_noise, forward, F.mse_loss, backward,
AdamW(lr).step, OneCycleLR.step).  The reference trains with torch's default dropout 0.1
(decoder.py:26-33 / encoder/base.py:29-40 never set it, train.py never calls .eval()):
``Dropout`` below reproduces it at torch's four kinds of sites with a counter-based Philox
mask that the backward kernels regenerate (nothing is stored); p = 0 is the parity path
(bit-level parity with torch's own mask stream is not defined: a different generator).

torch.autograd is used as the tape only (plumbing): each node below is one reference
block, its forward and backward are calls into libsoccerdiffusion_hip.so.
"""

from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
from torch.autograd import Function

from . import ops

Tensor = torch.Tensor


class _GradSink:
    """Where a block's parameter gradients go.  When the parameters' ``.grad`` buffers are
    preallocated (FusedAdamW aliases them to one flat buffer and zeroes it per step), the
    backward kernels ACCUMULATE straight into them (dW += dY^T X with fp32 atomics) and the
    autograd node returns None for those inputs: no zero-fill, no extra add kernel, no copy.
    Otherwise fresh tensors are returned to autograd as usual."""

    __slots__ = ("views",)

    def __init__(self, *params_and_slices):
        views = [] if params_and_slices else None
        for item in params_and_slices:
            p, sl = item if isinstance(item, tuple) else (item, None)
            g = p.grad
            if g is None or not g.is_cuda:
                views = None
                break
            views.append(g if sl is None else g[sl])
        self.views = views

    def target(self, i: int, like: Tensor):
        """(tensor to accumulate into, value to hand back to autograd)."""
        if self.views is not None:
            return self.views[i], None
        z = torch.zeros(like.shape, dtype=torch.float32, device=like.device)
        return z, z


# Transposed d x d weight blocks for the dX GEMMs, keyed by the block's address: FusedAdamW fills it with ONE gather over
# its flat parameter buffer after every step (the weights are final until the next one), instead of one strided copy
# kernel per block and backward node (41 launches, 3 % of the B=256 step).  An entry is valid while its optimizer is alive
# and the parameter's version counter is the one seen at the gather (load_state_dict, p.mul_(), ... bump it; a write
# through ``p.data`` does not - call ``optimizer.refresh_transposes()`` after one).
_WT_BLOCKS: dict = {}


def _wt_entry(W: Tensor, p: int, d: int):
    """(optimizer, float offset of block p's transpose in flat_wt) while the gather is current, else None."""
    ent = _WT_BLOCKS.get(W.data_ptr() + 4 * p * d * d)
    if ent is not None:
        owner, off, bd, pi = ent
        opt = owner()
        if opt is not None and bd == d and W.shape[1] == d and opt._wt_params[pi]._version == opt._wt_versions[pi]:
            return opt, off
    return None


def _transposed_block(W: Tensor, p: int, d: int) -> Tensor:
    ent = _wt_entry(W, p, d)
    if ent is not None and ent[0].flat_wt is not None:
        return ent[0].flat_wt[ent[1] : ent[1] + d * d].view(d, d)
    return W[p * d : (p + 1) * d].t().contiguous()


# The same registry serves the split planes of the row GEMMs: after every step FusedAdamW also splits each block and each
# transposed block into the fp16 hi | lo fragment planes the panel kernel multiplies with (ops.pack_weight_blocks, two
# launches), so the ~45 GEMMs of a step stream 1-KiB fragments instead of splitting fp32 weights in registers in every
# workgroup.  flat_wpk holds [planes of the blocks | planes of the transposed blocks], 2 d^2 halfs each, in flat_wt's order.
def _packed_weight(W: Tensor, p: int = 0, transposed: bool = False):
    """Address of the planes of block p of W (and of the blocks behind it), or None when W has no current planes."""
    d = W.shape[1]
    if W.stride() != (d, 1):
        return None
    ent = _wt_entry(W, p, d)
    if ent is None or ent[0].flat_wpk is None:
        return None
    opt, off = ent
    return opt.flat_wpk.data_ptr() + 2 * (2 * off + (2 * opt._wt_total if transposed else 0))


# Planes of whole weight matrices in the 16 x 16 x 32 fragment order of the trajectory kernels (csrc/sd_traj.h), for the layer-forward
# kernel sd_train_layer_fwd: a second, lazily built set beside flat_wpk.  A matrix registers with its optimizer on first use (in the
# eager warm-up steps of a graphed loop); from then on FusedAdamW.refresh_transposes repacks every registered matrix in one launch.
def _packed_weight_traj(W: Tensor, row0: int = 0, rows: Optional[int] = None):
    """Address of the trajectory-kernel planes of rows [row0, row0 + rows) of W (N x 256, row-major, owned by a FusedAdamW), else None."""
    d = W.shape[1]
    if d != 256 or W.stride() != (d, 1):
        return None
    ent = _wt_entry(W, 0, d)
    if ent is None:
        return None
    opt = ent[0]
    rows = W.shape[0] - row0 if rows is None else rows
    return opt.traj_planes(W, row0, rows)


def _linear(A: Tensor, W: Tensor, b, ln=None, res=None, drop=None) -> Tensor:
    """out = [res +] [dropout](LN?(A) W^T + b) on W's split planes when its optimizer keeps them, else on W itself."""
    wpk = _packed_weight(W)
    if wpk is not None:
        return ops.linear_packed(A, wpk, W.shape[0], b, ln=ln, res=res, drop=drop)
    if drop is not None:
        return ops.linear_dropout(A.contiguous(), W, b, res, drop)
    return ops.linear(A, W, b, ln=ln, res=res)


# ---- weight gradients ----------------------------------------------------------------------------------------------
# dW += dY^T X depends on dY and on a saved activation only.  (Round 2 could enqueue these GEMMs on a parallel branch of the captured
# graph; the branch overlapped - 1.0 of 5.2 ms of kernel time concurrent - but both sides slowed down by as much, 5.85 vs
# 5.83 ms per step, so the variant and its switch are gone.)
def _dw(dY: Tensor, X: Tensor, dW: Tensor, db: Optional[Tensor]) -> None:
    ops.gemm_tn(dY, X, dW, db)


def _dx_through_weight(dy2d: Tensor, W: Tensor) -> Tensor:
    """dX[R,d] = dY[R,N] @ W[N,d]: the forward panel kernel on W^T, one pass per d-wide
    column slice of dY (N = d, 2d or 3d), accumulated through the residual input."""
    N, d = W.shape
    out = None
    for p in range(N // d):
        wpk = _packed_weight(W, p, transposed=True)
        if wpk is not None:
            out = ops.linear_packed(dy2d[:, p * d : (p + 1) * d], wpk, d, None, res=out, out=out)
        else:
            out = ops.linear_strided(dy2d[:, p * d : (p + 1) * d], _transposed_block(W, p, d), res=out, out=out)
    return out


# ---- dropout sites (torch: nn.MultiheadAttention.dropout, TransformerDecoderLayer.dropout1/2/3 and .dropout) ----------
SITE_SA_PROBS, SITE_SA_OUT, SITE_CA_PROBS, SITE_CA_OUT, SITE_FFN_ACT, SITE_FFN_OUT = range(6)


class Dropout:
    """Dropout configuration of one transformer stack (the decoder or one context encoder).

    ``p`` defaults to torch's 0.1, as in the reference, and is live only in ``module.train()`` mode.  ``seed`` keys
    the Philox mask; every forward call of the stack draws a fresh call index, and a site is
    ``(call << 24) | (salt << 12) | (layer << 4) | kind``, so masks never repeat across calls, stacks, layers or
    sites.  ``for_call()`` returns the per-call object the autograd nodes use (``None`` when dropout is off)."""

    def __init__(self, p: float = 0.1, seed: Optional[int] = None, salt: int = 0):
        self.p, self.seed, self.salt, self.calls = float(p), seed, int(salt), 0

    def for_call(self, training: bool):
        if not training or self.p <= 0.0:
            return None
        if self.seed is None:
            self.seed = int(torch.initial_seed())
        self.calls += 1
        return _DropCall(self.p, self.seed, (self.calls << 24) | (self.salt << 12))


class _DropCall:
    __slots__ = ("p", "seed", "base")

    def __init__(self, p, seed, base):
        self.p, self.seed, self.base = p, seed, base

    def site(self, layer: int, kind: int) -> tuple:
        return (self.p, self.seed, self.base | (layer << 4) | kind)


class _LNLinear(Function):
    """y = act(LayerNorm(x) W^T + b) — LN1+QKV, LN2+Q, LN3+FFN1(+GELU [+ dropout])."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, W, b, gelu: bool, sink=None, drop=None, link=None):
        ctx.link = link
        x2 = x.reshape(-1, x.shape[-1])
        pre = _linear(x2, W, b, ln=(ln_w, ln_b))
        if gelu:
            y = ops.gelu_dropout_fwd(pre, drop) if drop is not None else ops.gelu_fwd(pre)
        else:
            y = pre
        ctx.save_for_backward(x2, ln_w, ln_b, W, pre if gelu else None)
        ctx.gelu = gelu
        ctx.drop = drop
        ctx.shape = x.shape
        ctx.sink = sink if sink is not None else _GradSink()
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, ln_w, ln_b, W, pre = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, W.shape[0])
        if ctx.gelu:
            dpre = ops.gelu_dropout_bwd(dy2, pre, ctx.drop) if ctx.drop is not None else ops.gelu_bwd(dy2, pre)
        else:
            dpre = dy2
        n, mean, rstd = ops.layernorm_fwd(x2, ln_w, ln_b)  # recomputed, not stored
        (dg, rg), (dbeta, rbeta) = ctx.sink.target(0, ln_w), ctx.sink.target(1, ln_w)
        (dW, rW), (db, rb) = ctx.sink.target(2, W), ctx.sink.target(3, W[:, 0])
        _dw(dpre, n, dW, db)
        dn = _dx_through_weight(dpre, W)
        # the residual branch's gradient of the same h (left by _LinearRes.backward, which ran first) is added inside the
        # LayerNorm backward kernel instead of by an autograd accumulation kernel
        dres = ctx.link.pop("dres", None) if ctx.link is not None else None
        dx = ops.layernorm_bwd_into(dn, x2, mean, rstd, ln_w, dg, dbeta, dres=dres)
        return dx.view(ctx.shape), rg, rbeta, rW, rb, None, None, None, None


class _LinearRes(Function):
    """y = res + dropout(a W^T + b) — attention out-projection and FFN2 with the residual add
    (torch: x + dropout1(sa_block(x)), x + dropout3(ff_block(x)))."""

    @staticmethod
    def forward(ctx, a, W, b, res, sink=None, drop=None, link=None):
        a2 = a.reshape(-1, a.shape[-1])
        ctx.save_for_backward(a2, W)
        ctx.shape = a.shape
        ctx.drop = drop
        ctx.link = link
        ctx.sink = sink if sink is not None else _GradSink()
        res2 = res.reshape(-1, W.shape[0]).contiguous()
        y = _linear(a2, W, b, res=res2, drop=drop)
        return y.view(*a.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        a2, W = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, W.shape[0])
        dym = ops.dropout(dy2, ctx.drop) if ctx.drop is not None else dy2   # gradient of the dropped branch; the residual gets dy
        (dW, rW), (db, rb) = ctx.sink.target(0, W), ctx.sink.target(1, W[:, 0])
        _dw(dym, a2, dW, db)
        da = _dx_through_weight(dym, W)
        if ctx.link is not None:   # h's other consumer is the _LNLinear that shares this link: it adds dy inside its kernel
            ctx.link["dres"] = dy2
            return da.view(ctx.shape), rW, rb, None, None, None, None
        return da.view(ctx.shape), rW, rb, dy, None, None, None


class _Linear(Function):
    """y = a W^T + b (no norm) — K/V projection of the un-normalised memory rows."""

    @staticmethod
    def forward(ctx, a, W, b, sink=None):
        a2 = a.reshape(-1, a.shape[-1]).contiguous()
        ctx.save_for_backward(a2, W)
        ctx.shape = a.shape
        ctx.sink = sink if sink is not None else _GradSink()
        return _linear(a2, W, b).view(*a.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        a2, W = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, W.shape[0])
        (dW, rW), (db, rb) = ctx.sink.target(0, W), ctx.sink.target(1, W[:, 0])
        _dw(dy2, a2, dW, db)
        da = _dx_through_weight(dy2, W) if ctx.needs_input_grad[0] else None
        return (da.view(ctx.shape) if da is not None else None), rW, rb, None


class _SelfAttention(Function):
    """softmax(q k^T / sqrt(hd)) v on a packed (B, T, 3d) q|k|v buffer."""

    @staticmethod
    def forward(ctx, qkv, heads: int, drop=None):
        d = qkv.shape[-1] // 3
        out, lse = ops.attention_lse(qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :], heads, drop)
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads = heads
        ctx.drop = drop
        return out

    @staticmethod
    def backward(ctx, dO):
        qkv, out, lse = ctx.saved_tensors
        d = qkv.shape[-1] // 3
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :], out, dO.contiguous(), lse,
                          dqkv[..., :d], dqkv[..., d : 2 * d], dqkv[..., 2 * d :], ctx.heads, ctx.drop)
        return dqkv, None, None


class _CrossAttention(Function):
    """Queries (B, T, d) against packed memory keys|values (B, M, 2d)."""

    @staticmethod
    def forward(ctx, q, kv, heads: int, drop=None):
        d = q.shape[-1]
        out, lse = ops.attention_lse(q, kv[..., :d], kv[..., d:], heads, drop)
        ctx.save_for_backward(q, kv, out, lse)
        ctx.heads = heads
        ctx.drop = drop
        return out

    @staticmethod
    def backward(ctx, dO):
        q, kv, out, lse = ctx.saved_tensors
        d = q.shape[-1]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ops.attention_bwd(q, kv[..., :d], kv[..., d:], out, dO.contiguous(), lse, dq, dkv[..., :d], dkv[..., d:], ctx.heads, ctx.drop)
        return dq, dkv, None, None


# abs-max words of tensors that cross from a fused stack to the embedding / fc_out nodes around it (the last layer's output, the
# gradient of the first layer's input).  The slot travels WITH the tensor object, as an attribute (weakref to the stack's table,
# address of the tensor's words, address of a free slot of the same table for the node's own small operand): an address-keyed
# registry could hand a reused allocation the stale maximum of a freed tensor.  The consumer reads the attribute
# from the very object it was handed (before any view) and also checks the shape and the tensor's version counter - an in-place
# update after registration (autograd's InputBuffer accumulating into a gradient it holds the last reference to, an in-place hook)
# would leave a stale, too-small maximum behind; anything else takes the block-floating GEMM.
def _amax_register(t: Tensor, table: Tensor, slot_addr: int, scratch_addr: int) -> None:
    import weakref
    t._sd_amax = (weakref.ref(table), slot_addr, scratch_addr, tuple(t.shape), t._version)


def _amax_lookup(t: Tensor):
    ent = getattr(t, "_sd_amax", None)
    if ent is None or ent[0]() is None or ent[3] != tuple(t.shape) or ent[4] != t._version:
        return None
    return ent[1], ent[2]


def _dw_skinny(dY: Tensor, X: Tensor, dW: Tensor, db, known: Tensor, small: Tensor, ent=None) -> None:
    """dW += dY^T X where one operand (``known``) carries registered abs-max words (``ent``: what _amax_lookup returned for the
    tensor ``known`` is a view of) and the other (``small``) is cheap to scan: the grouped fp16 GEMM with ragged tiles instead
    of the block-floating-point kernel (67 -> ~20 us for the J = 20 gradients)."""
    ok = ent is not None and dY.shape[-1] % 4 == 0 and X.shape[-1] % 4 == 0 and dY.data_ptr() % 16 == 0 and X.data_ptr() % 16 == 0
    if not ok:
        ops.gemm_tn(dY, X, dW, db)
        return
    ops.absmax(small, ent[1])
    ay, ax = (ent[0], ent[1]) if known is dY else (ent[1], ent[0])
    ops.gemm_tn_grouped([(dY, X, dW, db, ay, ax)])


def _grad_targets(*params):
    """(tensors to accumulate into, values to hand back to autograd): straight into the preallocated ``.grad`` buffers
    (FusedAdamW's flat buffer, zeroed per step: no zero-fill, no accumulation kernel) when every parameter has one."""
    if all(p.grad is not None and p.grad.is_cuda for p in params):
        return [p.grad for p in params], [None] * len(params)
    z = [torch.zeros_like(p) for p in params]
    return z, z


class _PatchEmbed(Function):
    """Conv1d(kernel = stride = p) (+ bias + positional table); p = 1 is nn.Linear(J -> d).
    The input is data (noisy trajectory / sensor history): no input gradient."""

    @staticmethod
    def forward(ctx, x, W, b, pe):
        ctx.save_for_backward(x, W, b)
        return ops.patch_embed(x.contiguous(), W, b, pe)

    @staticmethod
    def backward(ctx, dy):
        x, W, b = ctx.saved_tensors
        d = W.shape[0]
        if W.dim() == 2:
            patches = x.reshape(-1, x.shape[-1])
        else:  # (d, C, p): patch rows in (c, k) order, a layout copy
            _, C, p = W.shape
            B, S, _ = x.shape
            n = S // p
            patches = x[:, : n * p].reshape(B, n, p, C).permute(0, 1, 3, 2).reshape(B * n, C * p).contiguous()
        (dW, db), (rW, rb) = _grad_targets(W, b)
        ent = _amax_lookup(dy) if dy.is_contiguous() else None   # the very tensor the stack's backward produced
        dy2 = dy.contiguous().view(-1, d)
        _dw_skinny(dy2, patches, dW.view(d, patches.shape[1]), db, known=dy2, small=patches, ent=ent)
        dx = None
        if ctx.needs_input_grad[0]:
            # only the image tokens are a differentiable input (they come from the ResNet): kernel size 1, C = d
            if W.dim() == 3 and W.shape[2] == 1 and W.shape[1] == d:
                dx = _dx_through_weight(dy2, W.view(d, d)).view(x.shape)
            else:
                raise NotImplementedError("input gradient of a patch embedding is only implemented for kernel size 1 with C = hidden_dim")
        return dx, rW, rb, None


class _EmbedHead(Function):
    """The decoder stack's entry as ONE trajectory-owning launch (sd_train_head_fwd): h0 = Linear(J -> 256)(x) + positional rows, and
    - not differentiated here, as in _FusedLayer - n1 = LN1(h0), qkv = n1 Wqkv^T + b of layer 0.  The gradient of h0 goes where
    _PatchEmbed sends it."""

    @staticmethod
    def forward(ctx, x, W, b, pe, n1w, n1b, Wqkv, bqkv, amax_n1: int):
        ctx.save_for_backward(x, W, b)
        x = x.contiguous()
        w_emb = ops.pack_weight_traj(W.detach().contiguous())     # (256, J): a 4-us launch per step
        h0, n1, qkv = ops.train_head_fwd(x, w_emb.data_ptr(), b.detach(), pe, (n1w.detach(), n1b.detach()), _packed_weight_traj(Wqkv), bqkv.detach(),
                                         amax_n1)
        ctx.mark_non_differentiable(n1, qkv)
        return h0, n1, qkv

    @staticmethod
    def backward(ctx, dy, _dn, _dqkv):
        if ctx.needs_input_grad[0]:   # (_embed_head_ok sends a differentiable x to _PatchEmbed, which raises as well)
            raise NotImplementedError("the denoiser's embedding has no gradient with respect to the noisy trajectory")
        x, W, b = ctx.saved_tensors
        d = W.shape[0]
        patches = x.reshape(-1, x.shape[-1])
        (dW, db), (rW, rb) = _grad_targets(W, b)
        ent = _amax_lookup(dy) if dy.is_contiguous() else None
        dy2 = dy.contiguous().view(-1, d)
        _dw_skinny(dy2, patches, dW.view(d, patches.shape[1]), db, known=dy2, small=patches, ent=ent)
        return None, rW, rb, None, None, None, None, None, None


def _embed_head_ok(gen, x: Tensor, layers) -> bool:
    import os

    W = gen.embedding.weight
    J = x.shape[-1]
    return (os.environ.get("SD_TRAIN_TRAJ", "1") != "0" and x.is_cuda and not x.requires_grad and W.dim() == 2 and W.shape[0] == 256 and J % 4 == 0 and 4 <= J <= 32
            and x.shape[1] <= 100 and gen.num_heads == 4 and _packed_weight_traj(layers[0].self_attn.in_proj_weight) is not None)


class _FcOut(Function):
    """eps = h W^T + b with W (J, d)."""

    @staticmethod
    def forward(ctx, h, W, b):
        h2 = h.reshape(-1, h.shape[-1])
        ctx.save_for_backward(h2, W, b)
        ctx.amax = _amax_lookup(h) if h.is_contiguous() else None   # the stack's output object itself (h2 is then a view of it)
        ctx.shape = h.shape
        return ops.fc_out(h2, W, b).view(*h.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        h2, W, b = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, W.shape[0])
        (dW, db), (rW, rb) = _grad_targets(W, b)
        _dw_skinny(dy2, h2, dW, db, known=h2, small=dy2, ent=ctx.amax)
        dh = ops.small_k_matmul(dy2, W)
        return dh.view(ctx.shape), rW, rb


class _StepTokenFn(Function):
    @staticmethod
    def forward(ctx, steps, freq, token):
        ctx.half = token.shape[-1]
        ctx.save_for_backward(token)
        return ops.step_token(steps, freq, token)

    @staticmethod
    def backward(ctx, dtok):
        (token,) = ctx.saved_tensors
        (dtoken,), (rtoken,) = _grad_targets(token)
        ops.colsum(dtok.contiguous()[:, 0, ctx.half :], dtoken.view(-1))  # the learned half is expanded over the batch
        return None, None, rtoken


class _MSELoss(Function):
    """F.mse_loss(pred, target) with mean reduction (train.py:229)."""

    @staticmethod
    def forward(ctx, pred, target):
        loss, grad = ops.mse_loss(pred.contiguous(), target.contiguous())
        ctx.save_for_backward(grad)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return grad * gout, None  # gout is the scalar 1.0 of loss.backward()


def mse_loss(pred: Tensor, target: Tensor) -> Tensor:
    return _MSELoss.apply(pred, target)


def step_token_autograd(step_module, steps: Tensor) -> Tensor:
    steps = steps.contiguous()
    if steps.dtype not in (torch.int64, torch.float32):
        steps = steps.to(torch.float32 if steps.is_floating_point() else torch.int64)
    return _StepTokenFn.apply(steps, step_module._freq, step_module.token)


def _layer(lp, h, heads, memory=None, ffn_norm=None, dc=None, li: int = 0):
    """One pre-norm layer; ``dc`` (a _DropCall or None) turns on dropout at torch's sites of layer ``li``."""
    def site(kind):
        return dc.site(li, kind) if dc is not None else None

    # In each of the three sub-blocks h has exactly two consumers, the LayerNorm branch and the residual add, and the
    # residual's backward runs first: it hands its gradient to the LayerNorm backward through `link` (fused add).  Only
    # when h itself needs a gradient - otherwise autograd would never run the LayerNorm branch's backward for it.
    def link():
        return {} if h.requires_grad else None

    sa, n1 = lp.self_attn, lp.norm1
    lk = link()
    qkv = _LNLinear.apply(h, n1.weight, n1.bias, sa.in_proj_weight, sa.in_proj_bias, False,
                          _GradSink(n1.weight, n1.bias, sa.in_proj_weight, sa.in_proj_bias), None, lk)
    a = _SelfAttention.apply(qkv, heads, site(SITE_SA_PROBS))
    h = _LinearRes.apply(a, sa.out_proj.weight, sa.out_proj.bias, h, _GradSink(sa.out_proj.weight, sa.out_proj.bias), site(SITE_SA_OUT), lk)
    if memory is not None:
        d = h.shape[-1]
        ca, n2 = lp.multihead_attn, lp.norm2
        w, b = ca.in_proj_weight, ca.in_proj_bias
        lk = link()
        q = _LNLinear.apply(h, n2.weight, n2.bias, w[:d], b[:d], False,
                            _GradSink(n2.weight, n2.bias, (w, slice(0, d)), (b, slice(0, d))), None, lk)
        kv = _Linear.apply(memory, w[d:], b[d:], _GradSink((w, slice(d, 3 * d)), (b, slice(d, 3 * d))))  # memory is NOT layer-normed
        a = _CrossAttention.apply(q, kv, heads, site(SITE_CA_PROBS))
        h = _LinearRes.apply(a, ca.out_proj.weight, ca.out_proj.bias, h, _GradSink(ca.out_proj.weight, ca.out_proj.bias), site(SITE_CA_OUT), lk)
    lk = link()
    u = _LNLinear.apply(h, ffn_norm.weight, ffn_norm.bias, lp.linear1.weight, lp.linear1.bias, True,
                        _GradSink(ffn_norm.weight, ffn_norm.bias, lp.linear1.weight, lp.linear1.bias), site(SITE_FFN_ACT), lk)
    return _LinearRes.apply(u, lp.linear2.weight, lp.linear2.bias, h, _GradSink(lp.linear2.weight, lp.linear2.bias), site(SITE_FFN_OUT), lk)


# ---- fused layers: the row chains of a layer in one launch each (csrc/sd_train_chain.hip) ---------------------------
# Per decoder layer the forward is  attention, chain A (out-projection + residual, LN2 + Q), K/V of the memory, attention,
# chain B (out-projection + residual, LN3 + FFN1 + GELU + FFN2 + residual, the NEXT layer's LN1 + QKV); the backward is
# five chain launches, two attention backward kernels and seven weight-gradient GEMMs.  A layer therefore hands its
# successor (h, LN1(h), qkv); the successor's backward differentiates LN1 / in_proj (they are its parameters).
# Conditions (else the per-operation nodes above run): hidden_dim 64 / 128 / 256, dim_feedforward == hidden_dim, every
# weight owned by a FusedAdamW that keeps its split planes current.  SD_TRAIN_FUSED=0 turns the path off (A/B runs).
_DEC_PARAMS = ("norm1.weight", "norm1.bias", "self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
               "self_attn.out_proj.bias", "norm2.weight", "norm2.bias", "multihead_attn.in_proj_weight", "multihead_attn.in_proj_bias",
               "multihead_attn.out_proj.weight", "multihead_attn.out_proj.bias", "norm3.weight", "norm3.bias", "linear1.weight",
               "linear1.bias", "linear2.weight", "linear2.bias")
_ENC_PARAMS = ("norm1.weight", "norm1.bias", "self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
               "self_attn.out_proj.bias", "norm2.weight", "norm2.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias")


def _layer_params(lp, decoder: bool) -> list:
    out = []
    for name in (_DEC_PARAMS if decoder else _ENC_PARAMS):
        t = lp
        for part in name.split("."):
            t = getattr(t, part)
        out.append(t)
    return out


def _fused_ok(layers, d: int, decoder: bool) -> bool:
    if os.environ.get("SD_TRAIN_FUSED", "1") == "0" or d not in (64, 128, 256) or not layers:
        return False
    for lp in layers:
        P = _layer_params(lp, decoder)
        if any((not t.is_cuda) or t.dtype != torch.float32 for t in P):
            return False
        mats = [t for t in P if t.dim() == 2]
        if any(t.shape[1] != d or t.shape[0] % d != 0 for t in mats) or lp.linear1.weight.shape[0] != d:
            return False
        if any(_packed_weight(t) is None or _packed_weight(t, 0, transposed=True) is None for t in mats):
            return False
    return True


# abs-max words of a layer (row li of the stack's int32 table; bits of max |x|, atomically max-ed by the chains' row passes):
# the per-tensor scales of the grouped weight-gradient GEMM
AMAX_WORDS = 64   # SD_AMAX_WORDS: the producers spread their atomics over this many words per tensor
(_AX_N1, _AX_ASA, _AX_N2, _AX_ACA, _AX_NF, _AX_U, _AX_DY2, _AX_DPRE, _AX_DYC, _AX_DQ, _AX_DYS, _AX_DQKV, _AX_DKV, _AX_OUT, _AX_SCR,
 _AX_MEM, _AX_DX, _AX_SCR2) = range(18)   # _AX_OUT / _AX_DX: the layer's output / its input gradient (for fc_out / the embedding,
_AX_SLOTS = 20                             # with a scratch slot each for their small operand); _AX_MEM: the memory (row 0)


class _FusedCfg:
    __slots__ = ("heads", "dc", "li", "decoder", "next_ln", "next_w", "next_b", "amax")

    def __init__(self, heads, dc, li, decoder, nxt, amax):
        self.heads, self.dc, self.li, self.decoder, self.amax = heads, dc, li, decoder, amax
        if nxt is not None:
            self.next_ln = (nxt.norm1.weight.detach(), nxt.norm1.bias.detach())
            self.next_w, self.next_b = nxt.self_attn.in_proj_weight.detach(), nxt.self_attn.in_proj_bias.detach()
        else:
            self.next_ln = self.next_w = self.next_b = None

    def site(self, kind: int) -> int:
        return self.dc.site(self.li, kind)[2] if self.dc is not None else 0

    def ax(self, idx: int, layer_offset: int = 0) -> int:
        """Address of abs-max word ``idx`` of this layer (or of the next one)."""
        return self.amax.data_ptr() + 4 * AMAX_WORDS * ((self.li + layer_offset) * _AX_SLOTS + idx)

    @property
    def ax_mem(self) -> int:
        return self.amax.data_ptr() + 4 * AMAX_WORDS * _AX_MEM   # row 0

    def drop(self, kind: int):
        return self.dc.site(self.li, kind) if self.dc is not None else None

    @property
    def p(self) -> float:
        return self.dc.p if self.dc is not None else 0.0

    @property
    def seed(self) -> int:
        return self.dc.seed if self.dc is not None else 0


def _new(*shape, like: Tensor) -> Tensor:
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


# The memory side of a decoder layer - K/V projection of 11 rows per trajectory, its dX, the abs-max of dkv - is 44 panels per
# launch: 12 launches of ~17 us per step that leave five CUs in six idle.  They run on a second stream (a parallel branch of the
# captured graph) beside the 400-panel chains of the trajectory rows and are joined where their results are needed.
_MEM_SIDE: dict = {}


def _mem_side(device):
    s = _MEM_SIDE.get(device)
    if s is None:
        s = _MEM_SIDE[device] = torch.cuda.Stream(device=device)
    return s


TRAJ_LAYERS = [0]   # decoder layers whose forward ran as ONE trajectory-owning launch (tests assert which path ran)


def _traj_layer_weights(cfg, P, h: Tensor, memory: Tensor):
    """The plane addresses sd_train_layer_fwd needs, or None when the layer does not qualify (shape, SD_TRAIN_TRAJ=0, weights that no
    FusedAdamW keeps planes for): the per-op / row-chain launches run instead."""
    import os

    if os.environ.get("SD_TRAIN_TRAJ", "1") == "0" or not h.is_cuda or memory is None:
        return None
    B, T, d = h.shape
    if not ops.train_layer_fwd_ok(d, cfg.heads, T, memory.shape[1]):
        return None
    (n1w, n1b, Wqkv, bqkv, Wo, bo, n2w, n2b, Wc, bc, Woc, boc, nfw, nfb, W1, b1, W2, b2) = P
    for _ in range(2):   # a slice that registers on first use re-allocates the plane buffer: resolve the addresses again afterwards
        w = dict(w_o=_packed_weight_traj(Wo), w_q=_packed_weight_traj(Wc, 0, d), w_oc=_packed_weight_traj(Woc), w_1=_packed_weight_traj(W1),
                 w_2=_packed_weight_traj(W2), w_n=_packed_weight_traj(cfg.next_w) if cfg.next_w is not None else None)
    if any(v is None for k, v in w.items() if k != "w_n") or (cfg.next_w is not None and w["w_n"] is None):
        return None
    return w


class _FusedLayer(Function):
    """One pre-norm transformer layer (nn.TransformerDecoderLayer with memory, nn.TransformerEncoderLayer without) given
    (h, LN1(h), qkv = LN1(h) Wqkv^T + b); returns (h', LN1'(h'), qkv') for the next layer (empty tensors after the last)."""

    @staticmethod
    def forward(ctx, h, n1, qkv, memory, cfg: _FusedCfg, *P):
        B, T, d = h.shape
        R = B * T
        dec = cfg.decoder
        heads, p, seed = cfg.heads, cfg.p, cfg.seed
        if dec:
            (n1w, n1b, Wqkv, bqkv, Wo, bo, n2w, n2b, Wc, bc, Woc, boc, nfw, nfb, W1, b1, W2, b2) = P
        else:
            (n1w, n1b, Wqkv, bqkv, Wo, bo, nfw, nfb, W1, b1, W2, b2) = P
        h = h.contiguous()
        side = None
        if dec:   # the memory's K/V projection beside the self-attention and chain A
            M = memory.shape[1]
            mem2 = memory.reshape(B * M, d)
            kv = _new(B, M, 2 * d, like=h)
            side = _mem_side(h.device)
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    ops.linear_packed(mem2, _packed_weight(Wc, 1), 2 * d, bc[d:], out=kv.view(B * M, 2 * d))
            else:
                ops.linear_packed(mem2, _packed_weight(Wc, 1), 2 * d, bc[d:], out=kv.view(B * M, 2 * d))
        tw = _traj_layer_weights(cfg, P, h, memory) if dec else None
        if tw is not None:
            # ---- ONE launch for the whole layer forward: a workgroup per trajectory (csrc/sd_train_traj.hip) ----
            a_sa, h1, q, a_ca, h2, h3 = (_new(B, T, d, like=h) for _ in range(6))
            n2, nf, pre, u = (_new(R, d, like=h) for _ in range(4))
            lse_sa, lse_ca = _new(B, heads, T, like=h), _new(B, heads, T, like=h)
            if cfg.next_w is not None:
                nn1, qkv2 = _new(R, d, like=h), _new(B, T, 3 * d, like=h)
            else:
                nn1, qkv2 = _new(0, like=h), _new(0, like=h)
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            TRAJ_LAYERS[0] += 1
            ops.train_layer_fwd(
                B, T, M, heads,
                tensors=dict(h=h, qkv=qkv, a_sa=a_sa, lse_sa=lse_sa, h1=h1, n2=n2, q=q, kv=kv, a_ca=a_ca, lse_ca=lse_ca, h2=h2, nf=nf, pre=pre, u=u,
                             h3=h3, nn1=nn1 if cfg.next_w is not None else None, qkv2=qkv2 if cfg.next_w is not None else None,
                             b_o=bo, b_q=bc, b_oc=boc, b_1=b1, b_2=b2, b_n=cfg.next_b, n2_w=n2w, n2_b=n2b, n3_w=nfw, n3_b=nfb,
                             nn_w=cfg.next_ln[0] if cfg.next_w is not None else None, nn_b=cfg.next_ln[1] if cfg.next_w is not None else None),
                weights=tw, p=p, seed=seed,
                sites=(cfg.site(SITE_SA_PROBS), cfg.site(SITE_SA_OUT), cfg.site(SITE_CA_PROBS), cfg.site(SITE_CA_OUT), cfg.site(SITE_FFN_ACT),
                       cfg.site(SITE_FFN_OUT)),
                amax=(cfg.ax(_AX_ASA), cfg.ax(_AX_N2), cfg.ax(_AX_ACA), cfg.ax(_AX_NF), cfg.ax(_AX_U),
                      cfg.ax(_AX_N1, 1) if cfg.next_w is not None else None, cfg.ax(_AX_OUT) if cfg.next_w is None else None))
            if cfg.next_w is None:
                _amax_register(h3, cfg.amax, cfg.ax(_AX_OUT), cfg.ax(_AX_SCR))
            ctx.cfg = cfg
            ctx.n_c = 6
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(h, n1, qkv, memory, a_sa, lse_sa, h2, nf, pre, u, h1, n2, q, kv, a_ca, lse_ca, *P)
            ctx.mark_non_differentiable(nn1, qkv2)
            return h3, nn1, qkv2
        a_sa, lse_sa = ops.attention_lse(qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :], heads, cfg.drop(SITE_SA_PROBS))
        saved_c = ()
        if dec:
            h1, n2, q = _new(B, T, d, like=h), _new(R, d, like=h), _new(B, T, d, like=h)
            ops.train_fwd_chain(R, d, h, a=a_sa, wo=_packed_weight(Wo), bo=bo, h_out=h1, nln=(n2w, n2b), nn_out=n2,
                                wn=_packed_weight(Wc), bn=bc, y_out=q, n_next=1, p=p, seed=seed, sites=(cfg.site(SITE_SA_OUT), 0, 0),
                                amax=(cfg.ax(_AX_ASA), None, None, cfg.ax(_AX_N2)))
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            a_ca, lse_ca = ops.attention_lse(q, kv[..., :d], kv[..., d:], heads, cfg.drop(SITE_CA_PROBS))
            a_in, w_in, b_in, h_res, site_out, ax_a = a_ca, Woc, boc, h1, cfg.site(SITE_CA_OUT), _AX_ACA
            saved_c = (h1, n2, q, kv, a_ca, lse_ca)
        else:
            a_in, w_in, b_in, h_res, site_out, ax_a = a_sa, Wo, bo, h, cfg.site(SITE_SA_OUT), _AX_ASA
        h2, nf, pre, u, h3 = _new(B, T, d, like=h), _new(R, d, like=h), _new(R, d, like=h), _new(R, d, like=h), _new(B, T, d, like=h)
        if cfg.next_w is not None:
            nn1, qkv2 = _new(R, d, like=h), _new(B, T, 3 * d, like=h)
            nxt = dict(nln=cfg.next_ln, nn_out=nn1, wn=_packed_weight(cfg.next_w), bn=cfg.next_b, y_out=qkv2, n_next=3)
        else:
            nn1, qkv2 = _new(0, like=h), _new(0, like=h)
            nxt = {}
        ops.train_fwd_chain(R, d, h_res, a=a_in, wo=_packed_weight(w_in), bo=b_in, h_out=h2, ln=(nfw, nfb), n_out=nf,
                            w1=_packed_weight(W1), b1=b1, pre=pre, u=u, w2=_packed_weight(W2), b2=b2, h2_out=h3, p=p, seed=seed,
                            sites=(site_out, cfg.site(SITE_FFN_ACT), cfg.site(SITE_FFN_OUT)),
                            amax=(cfg.ax(ax_a), cfg.ax(_AX_NF), cfg.ax(_AX_U), cfg.ax(_AX_N1, 1), cfg.ax(_AX_OUT) if cfg.next_w is None else None),
                            **nxt)
        if cfg.next_w is None:   # the stack's output feeds fc_out (decoder): its weight gradient needs this abs-max
            _amax_register(h3, cfg.amax, cfg.ax(_AX_OUT), cfg.ax(_AX_SCR))
        ctx.cfg = cfg
        ctx.n_c = len(saved_c)
        ctx.set_materialize_grads(False)   # LN1'(h') and qkv' carry no gradient: autograd would hand zeros of their size to backward
        ctx.save_for_backward(h, n1, qkv, memory if dec else None, a_sa, lse_sa, h2, nf, pre, u, *saved_c, *P)
        ctx.mark_non_differentiable(nn1, qkv2)
        return h3, nn1, qkv2

    @staticmethod
    def backward(ctx, dh3, _dn, _dqkv):
        cfg = ctx.cfg
        dec, heads, p, seed = cfg.decoder, cfg.heads, cfg.p, cfg.seed
        sv = ctx.saved_tensors
        h, n1, qkv, memory, a_sa, lse_sa, h2, nf, pre, u = sv[:10]
        saved_c, P = sv[10 : 10 + ctx.n_c], sv[10 + ctx.n_c :]
        B, T, d = h.shape
        R = B * T
        if dec:
            (n1w, n1b, Wqkv, bqkv, Wo, bo, n2w, n2b, Wc, bc, Woc, boc, nfw, nfb, W1, b1, W2, b2) = P
            h1, n2, q, kv, a_ca, lse_ca = saved_c
        else:
            (n1w, n1b, Wqkv, bqkv, Wo, bo, nfw, nfb, W1, b1, W2, b2) = P
        # where the parameter gradients go: straight into the preallocated .grad buffers (FusedAdamW's flat buffer), else
        # fresh tensors handed back to autograd
        direct = all(t.grad is not None and t.grad.is_cuda for t in P)
        G = [t.grad if direct else torch.zeros_like(t) for t in P]
        g = dict(zip(_DEC_PARAMS if dec else _ENC_PARAMS, G))
        nf_name = "norm3" if dec else "norm2"
        wT = lambda W, blk=0: _packed_weight(W, blk, transposed=True)   # noqa: E731
        if dh3 is None:   # (gradients are not materialised: an unused layer output)
            dh3 = torch.zeros_like(h)
        dh3 = dh3.contiguous()
        dh3_2 = dh3.view(R, d)

        # feed-forward block: dy -> mask -> W2^T -> gelu' o mask -> W1^T -> LayerNorm backward (+ dy)
        dym, dpre, dh2 = (_new(R, d, like=h) if p > 0 else dh3_2), _new(R, d, like=h), _new(B, T, d, like=h)
        ops.train_bwd_chain(R, d, dh3_2, wT(W2), dh2, dym=dym if p > 0 else None, pre=pre, dpre=dpre, wt1=wT(W1), x=h2, ln_w=nfw,
                            dres=dh3, dg=g[nf_name + ".weight"], db=g[nf_name + ".bias"], p=p, seed=seed,
                            sites=(cfg.site(SITE_FFN_OUT), cfg.site(SITE_FFN_ACT)), amax=(cfg.ax(_AX_DY2), cfg.ax(_AX_DPRE)))
        # the weight gradients of the layer: one grouped launch at the end (hidden_dim a multiple of 128), else one each
        grouped = d % 128 == 0
        dws = [(dym, u, g["linear2.weight"], g["linear2.bias"], cfg.ax(_AX_DY2), cfg.ax(_AX_U)),
               (dpre, nf, g["linear1.weight"], g["linear1.bias"], cfg.ax(_AX_DPRE), cfg.ax(_AX_NF))]

        dmem = None
        if dec:
            # cross-attention block
            dym, da = (_new(R, d, like=h) if p > 0 else dh2.view(R, d)), _new(B, T, d, like=h)
            ops.train_bwd_chain(R, d, dh2.view(R, d), wT(Woc), da, dym=dym if p > 0 else None, p=p, seed=seed, sites=(cfg.site(SITE_CA_OUT), 0),
                                amax=(cfg.ax(_AX_DYC), None))
            dws.append((dym, a_ca.view(R, d), g["multihead_attn.out_proj.weight"], g["multihead_attn.out_proj.bias"], cfg.ax(_AX_DYC),
                        cfg.ax(_AX_ACA)))
            dq, dkv = torch.empty_like(q), torch.empty_like(kv)
            ops.attention_bwd(q, kv[..., :d], kv[..., d:], a_ca, da, lse_ca, dq, dkv[..., :d], dkv[..., d:], heads, cfg.drop(SITE_CA_PROBS))
            M = memory.shape[1]
            gWc, gbc = g["multihead_attn.in_proj_weight"], g["multihead_attn.in_proj_bias"]
            if ctx.needs_input_grad[3]:   # (the learned half of the step token sits in the memory)
                dmem = _new(*memory.shape, like=h)
            mside = _mem_side(h.device)
            if mside is not None:
                mside.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(mside if mside is not None else torch.cuda.current_stream()):
                if grouped:
                    ops.absmax(dkv.view(B * M, 2 * d), cfg.ax(_AX_DKV))
                    dws.append((dkv.view(B * M, 2 * d), memory.reshape(B * M, d), gWc[d:], gbc[d:], cfg.ax(_AX_DKV), cfg.ax_mem))
                else:
                    _dw(dkv.view(B * M, 2 * d), memory.reshape(B * M, d), gWc[d:], gbc[d:])
                if dmem is not None:   # both column passes in one launch
                    ops.train_bwd_chain(B * M, d, dkv.view(B * M, 2 * d), wT(Wc, 1), dmem.view(B * M, d), passes=2)
            dh1 = _new(B, T, d, like=h)
            ops.train_bwd_chain(R, d, dq.view(R, d), wT(Wc), dh1, x=h1, ln_w=n2w, dres=dh2, dg=g["norm2.weight"], db=g["norm2.bias"],
                                amax=(cfg.ax(_AX_DQ), None))
            dws.append((dq.view(R, d), n2, gWc[:d], gbc[:d], cfg.ax(_AX_DQ), cfg.ax(_AX_N2)))
            dres = dh1
        else:
            dres = dh2

        # self-attention block
        dym, da = (_new(R, d, like=h) if p > 0 else dres.view(R, d)), _new(B, T, d, like=h)
        ops.train_bwd_chain(R, d, dres.view(R, d), wT(Wo), da, dym=dym if p > 0 else None, p=p, seed=seed, sites=(cfg.site(SITE_SA_OUT), 0),
                            amax=(cfg.ax(_AX_DYS), None))
        dws.append((dym, a_sa.view(R, d), g["self_attn.out_proj.weight"], g["self_attn.out_proj.bias"], cfg.ax(_AX_DYS), cfg.ax(_AX_ASA)))
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :], a_sa, da, lse_sa, dqkv[..., :d], dqkv[..., d : 2 * d],
                          dqkv[..., 2 * d :], heads, cfg.drop(SITE_SA_PROBS))
        dh = _new(B, T, d, like=h)
        ops.train_bwd_chain(R, d, dqkv.view(R, 3 * d), wT(Wqkv), dh, passes=3, x=h, ln_w=n1w, dres=dres, dg=g["norm1.weight"],
                            db=g["norm1.bias"], amax=(cfg.ax(_AX_DQKV), None, cfg.ax(_AX_DX) if cfg.li == 0 else None))
        if cfg.li == 0:   # the gradient of the stack's input feeds the embedding's weight gradient
            _amax_register(dh, cfg.amax, cfg.ax(_AX_DX), cfg.ax(_AX_SCR2))
        dws.append((dqkv.view(R, 3 * d), n1, g["self_attn.in_proj_weight"], g["self_attn.in_proj_bias"], cfg.ax(_AX_DQKV), cfg.ax(_AX_N1)))
        if dec and _mem_side(h.device) is not None:
            torch.cuda.current_stream().wait_stream(_mem_side(h.device))   # dmem and the abs-max of dkv
        if grouped:
            ops.gemm_tn_grouped(dws)
        else:
            for dY, X, dW, db, _, _ in dws:
                _dw(dY, X, dW, db)
        return (dh, None, None, dmem, None) + tuple(None if direct else t for t in G)


FUSED_STACKS = [0]   # how many layer stacks went through the fused path (tests assert that it is the path that ran)


def _fused_stack(layers, h, heads: int, memory, dc, decoder: bool, hooks: bool, embed=None) -> Tensor:
    """``embed`` = (x, gen): the stack starts from the raw trajectory - embedding, LayerNorm 1 and the Q | K | V projection of layer 0 in
    one trajectory-owning launch (_EmbedHead) - instead of from an embedded ``h``."""
    FUSED_STACKS[0] += 1
    lp0 = layers[0]
    dev = embed[0].device if embed is not None else h.device
    d = lp0.norm1.weight.shape[0]
    amax = torch.zeros(len(layers) + 1, _AX_SLOTS, AMAX_WORDS, dtype=torch.int32, device=dev)
    if memory is not None and d % 128 == 0:
        ops.absmax(memory.reshape(-1, d), amax.data_ptr() + 4 * AMAX_WORDS * _AX_MEM)
    if embed is not None:
        x, gen = embed
        pe = gen.positional_encoding.pe[0, : x.shape[1]].contiguous()
        h, n1, qkv = _EmbedHead.apply(x, gen.embedding.weight, gen.embedding.bias, pe, lp0.norm1.weight, lp0.norm1.bias, lp0.self_attn.in_proj_weight,
                                      lp0.self_attn.in_proj_bias, amax.data_ptr() + 4 * AMAX_WORDS * _AX_N1)
        B, T, d = h.shape
    else:
        B, T, d = h.shape
        h = h.contiguous()
        with torch.no_grad():
            n1, qkv = _new(B * T, d, like=h), _new(B, T, 3 * d, like=h)
            ops.train_fwd_chain(B * T, d, h.detach(), nln=(lp0.norm1.weight, lp0.norm1.bias), nn_out=n1,
                                wn=_packed_weight(lp0.self_attn.in_proj_weight), bn=lp0.self_attn.in_proj_bias, y_out=qkv, n_next=3,
                                amax=(None, None, None, amax.data_ptr() + 4 * AMAX_WORDS * _AX_N1))
    for li, lp in enumerate(layers):
        if hooks:
            _layer_input_hook(h, li)
        cfg = _FusedCfg(heads, dc, li, decoder, layers[li + 1] if li + 1 < len(layers) else None, amax)
        h, n1, qkv = _FusedLayer.apply(h, n1, qkv, memory, cfg, *_layer_params(lp, decoder))
    return h


def denoiser_forward_autograd(gen, x: Tensor, memory: Tensor) -> Tensor:
    """Differentiable DiffusionActionGenerator.forward (reference ml/model/decoder.py:38-54)."""
    T = x.shape[1]
    pe = gen.positional_encoding.pe[0, :T].contiguous()
    memory = memory.contiguous()
    dc = gen.dropout.for_call(gen.training)
    layers = list(gen.transformer_decoder.layers)
    d_model = gen.embedding.weight.shape[0]
    if _fused_ok(layers, d_model, decoder=True) and _embed_head_ok(gen, x, layers):
        h = _fused_stack(layers, None, gen.num_heads, memory, dc, decoder=True, hooks=True, embed=(x, gen))
        return _FcOut.apply(h, gen.fc_out.weight, gen.fc_out.bias)
    h = _PatchEmbed.apply(x, gen.embedding.weight, gen.embedding.bias, pe)
    if _fused_ok(layers, h.shape[-1], decoder=True):
        h = _fused_stack(layers, h, gen.num_heads, memory, dc, decoder=True, hooks=True)
    else:
        for li, lp in enumerate(layers):
            _layer_input_hook(h, li)
            h = _layer(lp, h, gen.num_heads, memory=memory, ffn_norm=lp.norm3, dc=dc, li=li)
    return _FcOut.apply(h, gen.fc_out.weight, gen.fc_out.bias)


def encoder_forward_autograd(enc, x: Tensor) -> Tensor:
    """Differentiable BaseEncoder.forward (reference ml/model/encoder/base.py:41-53)."""
    n = x.shape[1] // enc.patch_size
    pe = enc.positional_encoding.pe[0, :n].contiguous()
    h = _PatchEmbed.apply(x, enc.embedding.weight, enc.embedding.bias, pe)
    dc = enc.dropout.for_call(enc.training)
    layers = list(enc.transformer_encoder.layers)
    if _fused_ok(layers, h.shape[-1], decoder=False):
        return _fused_stack(layers, h, enc.num_heads, None, dc, decoder=False, hooks=False)
    for li, lp in enumerate(layers):
        h = _layer(lp, h, enc.num_heads, memory=None, ffn_norm=lp.norm2, dc=dc, li=li)
    return h


# --------------------------------------------------------------------------------------
# optimizer: torch.optim.AdamW semantics on ONE flat fp32 buffer per quantity
# --------------------------------------------------------------------------------------
class FusedAdamW(torch.optim.Optimizer):
    """AdamW(lr) with torch's defaults (betas 0.9/0.999, eps 1e-8, weight_decay 1e-2) as the
    reference constructs it (train.py:162).  Parameters, gradients and both moments live in
    four flat buffers (parameters are re-pointed to views), so a step is one kernel launch
    and the data-parallel gradient exchange is one all-reduce.  ``state_dict()`` has torch
    AdamW's layout (``state[i] = {step, exp_avg, exp_avg_sq}``), so checkpoints interchange.
    Being a torch Optimizer, ``OneCycleLR`` drives ``lr`` and ``betas[0]`` as in the reference."""

    def __init__(self, params: Iterable[Tensor], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self._step = 0
        at = 0
        for p in params:
            k = p.numel()
            self.flat_param[at : at + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_param[at : at + k].view(p.shape)
            p.grad = self.flat_grad[at : at + k].view(p.shape)
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.flat_m[at : at + k].view(p.shape),
                             "exp_avg_sq": self.flat_v[at : at + k].view(p.shape)}
            at += k
        # gather index of every d x d block of the (N = k d, d) matrices, transposed (see _WT_BLOCKS)
        idx, self._wt_blocks, self._wt_params, at, out_at = [], [], [], 0, 0
        for p in params:
            k = p.numel()
            if p.dim() == 2 and p.shape[1] in (64, 128, 256, 512) and p.shape[0] % p.shape[1] == 0:
                N, d = p.shape
                idx.append(at + torch.arange(k, dtype=torch.int64).view(N // d, d, d).transpose(1, 2).reshape(-1))
                for blk in range(N // d):
                    self._wt_blocks.append((self.flat_param.data_ptr() + 4 * (at + blk * d * d), out_at + blk * d * d, d,
                                            len(self._wt_params)))
                self._wt_params.append(p)
                out_at += k
            at += k
        self._wt_total = out_at
        # split planes of the same blocks (see _packed_weight): per block width, (source offsets, first block) of the launch
        self.flat_wpk, self._pk_launches = None, []
        if idx and dev.type == "cuda" and os.environ.get("SD_TRAIN_PACKED", "1") != "0":
            self.flat_wpk = torch.empty(4 * out_at, dtype=torch.float16, device=dev)
            base = self.flat_param.data_ptr()
            for d in sorted({b[2] for b in self._wt_blocks}):
                run = [b for b in self._wt_blocks if b[2] == d]
                # blocks of one width are contiguous in flat_wt only if no other width interleaves: pack run by run
                runs, cur = [], [run[0]]
                for b in run[1:]:
                    if b[1] == cur[-1][1] + d * d:
                        cur.append(b)
                    else:
                        runs.append(cur); cur = [b]
                runs.append(cur)
                for r in runs:
                    fwd = torch.tensor([(b[0] - base) // 4 for b in r], dtype=torch.int64, device=dev)
                    wt = torch.tensor([b[1] for b in r], dtype=torch.int64, device=dev)
                    self._pk_launches.append((d, len(r), fwd, wt, 2 * r[0][1]))
        # without planes (no GPU, SD_TRAIN_PACKED=0): an fp32 copy of every W^T block from one gather per step
        self._wt_index = torch.cat(idx).to(dev) if idx and self.flat_wpk is None else None
        self.flat_wt = torch.empty(out_at, dtype=torch.float32, device=dev) if idx and self.flat_wpk is None else None
        self._wt_versions = []
        self.refresh_transposes()

    def refresh_transposes(self) -> None:
        """The per-step derived copies of the weights: the split planes of every d x d block and of its transpose (two
        launches per block width; the transposition happens inside the pack kernel), or - without planes - an fp32 gather
        of the transposed blocks (one kernel; see _WT_BLOCKS)."""
        ops.bump_weights_generation()   # every path that rewrites flat_param ends here (step, step_from_device_hyper, broadcast)
        if not self._wt_blocks or not self.flat_param.is_cuda:
            return
        import weakref
        if self.flat_wt is not None:
            torch.index_select(self.flat_param, 0, self._wt_index, out=self.flat_wt)
        for d, n, fwd, wt, half_off in self._pk_launches:
            ops.pack_weight_blocks(self.flat_param, fwd, n, d, self.flat_wpk[half_off:])
            ops.pack_weight_blocks(self.flat_param, fwd, n, d, self.flat_wpk[2 * self._wt_total + half_off:], transposed=True)
        self._wt_versions = [p._version for p in self._wt_params]
        self._refresh_traj()
        if not getattr(self, "_wt_registered", False):
            owner = weakref.ref(self)
            for key, off, d, pi in self._wt_blocks:
                _WT_BLOCKS[key] = (owner, off, d, pi)   # no tensor references: a dead optimizer's entries are inert
            self._wt_registered = True

    def traj_planes(self, W: Tensor, row0: int, rows: int):
        """Address of the trajectory-kernel planes (ops.pack_weight_traj layout) of rows [row0, row0 + rows) of parameter W, which lives
        in this optimizer's flat buffer; registers the slice on first use and repacks all registered slices."""
        src = (W.data_ptr() - self.flat_param.data_ptr()) // 4 + row0 * 256
        if src < 0 or src + rows * 256 > self.flat_param.numel() or rows % 16:
            return None
        reg = self.__dict__.setdefault("_traj_reg", {"index": {}, "src": [], "rows": [], "dst": [], "halfs": 0, "planes": None, "dev": None})
        at = reg["index"].get((src, rows))
        if at is None:
            at = reg["halfs"]
            reg["index"][(src, rows)] = at
            reg["src"].append(src); reg["rows"].append(rows); reg["dst"].append(at)
            reg["halfs"] += rows * 256 * 2
            reg["planes"] = None   # (re)allocated and repacked below
        if reg["planes"] is None:
            dev = self.flat_param.device
            reg["planes"] = torch.empty(reg["halfs"], dtype=torch.float16, device=dev)
            reg["dev"] = (torch.tensor(reg["src"], dtype=torch.int64, device=dev), torch.tensor(reg["rows"], dtype=torch.int32, device=dev),
                          torch.tensor(reg["dst"], dtype=torch.int64, device=dev))
            self._refresh_traj()
        return reg["planes"].data_ptr() + 2 * at

    def _refresh_traj(self) -> None:
        reg = self.__dict__.get("_traj_reg")
        if reg and reg["planes"] is not None:
            s, r, d = reg["dev"]
            ops.pack_weight_traj_multi(self.flat_param, s, r, d, max(reg["rows"]), reg["planes"])

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()
        at = 0
        for p in self.param_groups[0]["params"]:  # keep .grad aliased to the flat buffer
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * at:
                p.grad = self.flat_grad[at : at + k].view(p.shape)
            at += k

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self._step += 1
        ops.adamw_step(self.flat_param, self.flat_grad, self.flat_m, self.flat_v, g["lr"], g["betas"][0], g["betas"][1],
                       g["eps"], g["weight_decay"], self._step)
        self.refresh_transposes()

    @torch.no_grad()
    def step_from_device_hyper(self, hyper7: Tensor) -> None:
        """The update with its scalars read from device memory (``hyper_for_step``): what a captured graph replays.
        Does NOT advance ``_step`` - the caller that fills ``hyper7`` does."""
        ops.adamw_step_dev(self.flat_param, self.flat_grad, self.flat_m, self.flat_v, hyper7)
        self.refresh_transposes()

    def hyper_for_step(self, step: int, out) -> None:
        """The seven scalars of update number ``step`` at the CURRENT lr / betas of the param group -> ``out`` (7 floats)."""
        g = self.param_groups[0]
        ops.adamw_hyper(g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], step, out)

    def state_dict(self):
        for p in self.param_groups[0]["params"]:   # torch AdamW's per-parameter step counters, all equal here
            self.state[p]["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        views = {id(p): (self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in self.param_groups[0]["params"]}
        super().load_state_dict(state_dict)
        steps = []
        for p in self.param_groups[0]["params"]:
            st = self.state[p]
            m, v = views[id(p)]
            m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = m, v
            steps.append(int(float(st["step"])))
        self._step = max(steps) if steps else 0


def broadcast_parameters(optimizer: FusedAdamW, model=None, src: int = 0, group=None) -> None:
    """Data-parallel start: every rank takes rank ``src``'s replica - the optimizer's flat parameter buffer and both
    moment buffers (three broadcasts), plus every tensor of ``model`` the optimizer does not own (buffers such as
    ``mean`` / ``std`` and BatchNorm statistics, frozen parameters).  Without it replicas that were initialised from
    different RNG states would apply the averaged gradient to different weights and drift apart silently."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return
    for buf in (optimizer.flat_param, optimizer.flat_m, optimizer.flat_v):
        dist.broadcast(buf, src=src, group=group)
    step = torch.tensor([optimizer._step], dtype=torch.int64, device=optimizer.flat_param.device)
    dist.broadcast(step, src=src, group=group)
    optimizer._step = int(step.item())
    if model is not None:
        lo = optimizer.flat_param.data_ptr()
        hi = lo + 4 * optimizer.flat_param.numel()
        for t in list(model.parameters()) + list(model.buffers()):
            if t.numel() == 0 or lo <= t.data_ptr() < hi:
                continue
            if t.dtype == torch.int64 and t.dim() == 0:   # BatchNorm's num_batches_tracked
                tmp = t.detach().reshape(1).clone()
                dist.broadcast(tmp, src=src, group=group)
                t.data.copy_(tmp.reshape(()))
            else:
                dist.broadcast(t.detach(), src=src, group=group)
    optimizer.refresh_transposes()


def assert_replicas_equal(optimizer: FusedAdamW, group=None) -> None:
    """Data-parallel invariant: every rank holds bit-identical parameters (same start by broadcast, same averaged gradient,
    same deterministic update).  Two scalars per rank (sum and sum of squares of the flat parameter buffer, in fp64) are
    MIN- and MAX-reduced; a difference means the replicas have drifted apart - raise instead of training on silently."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return
    p = optimizer.flat_param.double()
    sig = torch.stack([p.sum(), (p * p).sum()])
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"data-parallel replicas diverged: parameter checksums differ across ranks ({lo.tolist()} .. {hi.tolist()})")


def allreduce_gradients(optimizer: FusedAdamW, world_size: int, group=None) -> None:
    """Data-parallel gradient exchange: ONE sum all-reduce of the flat fp32 gradient buffer
    over RCCL/xGMI (gloo on CPU tests), then the mean.  The reference is single-device."""
    import torch.distributed as dist

    if world_size <= 1:
        return
    dist.all_reduce(optimizer.flat_grad, op=dist.ReduceOp.SUM, group=group)
    optimizer.flat_grad.mul_(1.0 / world_size)


class BucketedAllReduce:
    """Gradient exchange overlapped with the backward: the flat gradient buffer is cut into
    one bucket per decoder layer (fc_out rides with the last layer, whose gradients are complete first) plus one for
    everything else (embedding, step token, context encoders).  ``ready(l)`` - called from a tensor hook when the
    gradient with respect to layer l's INPUT exists, i.e. when every gradient kernel of layer l has been enqueued -
    starts that bucket's asynchronous sum all-reduce (RCCL runs it on its own stream behind the work enqueued so far,
    while the backward of layers l-1 .. 0 keeps the compute stream busy); ``finish()`` reduces the rest, waits for all
    of them and applies the 1 / world mean.  The buckets partition the buffer, so the result equals
    ``allreduce_gradients`` exactly (same sum order per element: one all-reduce each)."""

    def __init__(self, optimizer: FusedAdamW, decoder, world_size: int, group=None):
        self.opt, self.world, self.group = optimizer, world_size, group
        base = optimizer.flat_param.data_ptr()
        layers = list(decoder.transformer_decoder.layers)
        self.ranges = []
        for i, lp in enumerate(layers):
            ps = list(lp.parameters()) + (list(decoder.fc_out.parameters()) if i == len(layers) - 1 else [])
            offs = sorted(((p.data_ptr() - base) // 4, p.numel()) for p in ps)
            lo, hi = offs[0][0], offs[-1][0] + offs[-1][1]
            if hi - lo != sum(n for _, n in offs) or lo < 0 or hi > optimizer.flat_param.numel():
                raise RuntimeError("layer parameters are not one contiguous range of the flat buffer")
            self.ranges.append((lo, hi))
        self.ranges.sort()
        for (a, b), (c, d) in zip(self.ranges, self.ranges[1:]):
            if b != c:
                raise RuntimeError("decoder layers are not adjacent in the flat buffer")
        self.handles, self.done = [], set()

    def ready(self, layer: int) -> None:
        import torch.distributed as dist

        if self.world <= 1 or layer in self.done:
            return
        self.done.add(layer)
        lo, hi = self.ranges[layer]
        self.handles.append(dist.all_reduce(self.opt.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        import torch.distributed as dist

        if self.world <= 1:
            return
        for l in range(len(self.ranges)):
            self.ready(l)          # a layer whose hook never fired (no gradient path) is reduced here
        n = self.opt.flat_grad.numel()
        lo, hi = self.ranges[0][0], self.ranges[-1][1]
        for a, b in ((0, lo), (hi, n)):
            if b > a:
                self.handles.append(dist.all_reduce(self.opt.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in self.handles:
            h.wait()
        self.handles, self.done = [], set()
        self.opt.flat_grad.mul_(1.0 / self.world)


_BUCKETS: Optional[BucketedAllReduce] = None   # the exchange in progress (set by train_step around loss.backward())


def _layer_input_hook(h: Tensor, layer: int) -> None:
    """Fires when the gradient of decoder layer ``layer``'s input exists: its bucket is complete."""
    if _BUCKETS is not None and h.requires_grad:
        h.register_hook(lambda g, l=layer: (_BUCKETS.ready(l) if _BUCKETS is not None else None, None)[1])


def train_step(model, optimizer: FusedAdamW, lr_scheduler, scheduler, joint_targets: Tensor, context=None,
               input_data=None, noise: Optional[Tensor] = None, timesteps: Optional[Tensor] = None,
               world_size: int = 1, generator: Optional[torch.Generator] = None, bucketed: bool = True) -> Tensor:
    """One iteration of the reference loop body (train.py:204-240) on already-normalised
    targets.  ``context`` given = the --decoder-pretraining path (train.py:221-224).
    Data parallel (``world_size`` > 1): the gradient all-reduce runs per decoder layer, overlapped with the rest of the
    backward (``BucketedAllReduce``); ``bucketed=False`` = one all-reduce after it."""
    global _BUCKETS
    B = joint_targets.shape[0]
    dev = joint_targets.device
    optimizer.zero_grad()
    if timesteps is None:
        timesteps = torch.randint(0, scheduler.config["num_train_timesteps"], (B,), device=dev, generator=generator).long()
    if noise is None:
        noise = torch.randn(joint_targets.shape, device=dev, generator=generator)
    noisy = scheduler.add_noise(joint_targets, noise, timesteps)
    buckets = None
    if world_size > 1 and bucketed:
        buckets = getattr(optimizer, "_buckets", None)
        if buckets is None or buckets.world != world_size:
            buckets = optimizer._buckets = BucketedAllReduce(optimizer, model.diffusion_action_generator, world_size)
    _BUCKETS = buckets            # the forward registers one hook per decoder layer while this is set
    try:
        if context is not None:
            pred = model.forward_with_context(context, noisy, timesteps)
        else:
            pred = model(input_data, noisy, timesteps)
        loss = mse_loss(pred, noise)
        loss.backward()
        if buckets is not None:
            buckets.finish()
        else:
            allreduce_gradients(optimizer, world_size)
    finally:
        _BUCKETS = None
    optimizer.step()
    if lr_scheduler is not None:
        lr_scheduler.step()
    return loss.detach()


class GraphedTrainStep:
    """``train_step`` captured into a hipGraph and replayed (decoder-pretraining path or full model, static shapes).

    The B = 256 step is ~120 kernel launches issued from Python through autograd; rocprofv3 shows the GPU idle for
    ~18 % of the step waiting for them.  A replay has no host work beyond one 32-byte upload.  What changes from step to
    step cannot sit in kernel arguments, which a graph freezes, so it comes from device memory instead:
      * AdamW's scalars (OneCycleLR's lr and cycled beta1, the bias corrections) - ``sd_adamw_step_dev``;
      * the per-step part of the dropout key - ``sd_set_dropout_epoch`` (process-wide; reset by ``close()``);
      * timesteps and noise come from ``generator`` (registered with the graph: torch advances its Philox offset per replay).
    The first ``eager_steps`` calls run the ordinary ``train_step`` (they are real training steps and warm every lazy
    initialisation up); the next call captures, and from then on every call replays.
    Data parallel (``world_size`` > 1): EVERY call is the eager ``train_step`` with its per-layer ``BucketedAllReduce`` - the one
    data-parallel step of this package (``cli train`` runs the same).  Overlapping the exchange with the backward needs the host
    to issue layer l's all-reduce while the GPU still runs the backward of layers l-1 .. 0; a single captured graph cannot
    signal the host or a side stream in mid-replay on this stack (torch refuses external event-record nodes on ROCm,
    tools/exp/graph_event.py), and since the fused row chains (~60 launches per step) the eager step is within 1 % of the
    replay (1 000-step soak: 3.67 vs 3.65 ms).  ``split_update=True`` keeps the round-2 form (graph up to the end of the
    backward, then ONE flat all-reduce and the update) for A/B runs and its parity test."""

    def __init__(self, model, optimizer: FusedAdamW, lr_scheduler, scheduler, world_size: int = 1,
                 generator: Optional[torch.Generator] = None, eager_steps: int = 2, split_update: Optional[bool] = None):
        self.model, self.opt, self.lr_sched, self.sched = model, optimizer, lr_scheduler, scheduler
        self.world, self.gen, self.eager_left = world_size, generator, eager_steps
        # data parallel: the eager bucketed step (see the class comment) unless split_update is forced
        self.split = False if split_update is None else bool(split_update)
        self.eager_dp = world_size > 1 and not self.split
        self.graph = None
        dev = optimizer.flat_param.device
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)       # 7 AdamW scalars + the dropout epoch word
        self._ring = [torch.zeros(8, dtype=torch.float32).pin_memory() for _ in range(8)]
        self._events = [None] * 8
        self._n = 0
        self._epoch = 0
        self._static = None

    # ---- inputs -------------------------------------------------------------------------
    def _stage(self, targets, context, input_data):
        if self._static is None:
            self._static = {"targets": targets.clone(),
                            "context": None if context is None else [c.clone() for c in context],
                            "input": None if input_data is None else {k: v.clone() for k, v in input_data.items()}}
            return
        st = self._static
        st["targets"].copy_(targets)
        if context is not None:
            for dst, src in zip(st["context"], context):
                dst.copy_(src)
        if input_data is not None:
            for k, v in input_data.items():
                st["input"][k].copy_(v)

    def _upload_hyper(self):
        slot = self._n % len(self._ring)
        if self._events[slot] is not None:
            self._events[slot].synchronize()     # the upload that last used this pinned buffer has run
        host = self._ring[slot]
        self.opt.hyper_for_step(self.opt._step + 1, host)
        self._epoch = (self._epoch + 1) & 0x7FFFFFFF
        host[7:8].view(torch.int32)[0] = self._epoch
        self.hyper.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[slot] = ev
        self._n += 1

    def _body(self):
        st = self._static
        B = st["targets"].shape[0]
        dev = st["targets"].device
        self.opt.zero_grad()
        t = torch.randint(0, self.sched.config["num_train_timesteps"], (B,), device=dev, generator=self.gen).long()
        noise = torch.randn(st["targets"].shape, device=dev, generator=self.gen)
        noisy = self.sched.add_noise(st["targets"], noise, t)
        if st["context"] is not None:
            pred = self.model.forward_with_context(st["context"], noisy, t)
        else:
            pred = self.model(st["input"], noisy, t)
        loss = mse_loss(pred, noise)
        loss.backward()
        if not self.split:
            self.opt.step_from_device_hyper(self.hyper[:7])
        return loss.detach()

    def __call__(self, joint_targets: Tensor, context=None, input_data=None) -> Tensor:
        if self.eager_left > 0 or self.eager_dp:
            self.eager_left = max(self.eager_left - 1, 0)
            return train_step(self.model, self.opt, self.lr_sched, self.sched, joint_targets, context=context,
                              input_data=input_data, world_size=self.world, generator=self.gen)
        self._stage(joint_targets, context, input_data)
        if self.graph is None:
            ops.set_dropout_epoch(self.hyper[7:8])
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            if self.gen is not None:
                self.graph.register_generator_state(self.gen)
            with torch.cuda.graph(self.graph):
                self._loss = self._body()
        self._upload_hyper()
        self.graph.replay()
        ops.bump_weights_generation()   # the replayed update rewrote the flat parameter buffer without any Python-side hook
        if self.split:
            allreduce_gradients(self.opt, self.world)
            self.opt.step_from_device_hyper(self.hyper[:7])
        self.opt._step += 1
        if self.lr_sched is not None:
            self.lr_sched.step()
        return self._loss

    def close(self) -> None:
        """Detaches the process-wide dropout epoch word (it lives in this object's ``hyper`` buffer)."""
        if self.graph is not None:
            torch.cuda.synchronize()
            ops.set_dropout_epoch(None)
            self.graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
