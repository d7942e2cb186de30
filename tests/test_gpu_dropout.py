"""Training dropout: the reference trains with torch's default p = 0.1 at the attention probabilities, after
each attention out-projection, after the GELU and after linear2 (decoder.py:26-33; train.py never calls .eval()).
Checked here: the Philox mask itself (keep rate, scaling, independence across sites / seeds, determinism), every fused
kernel against the mask applied by hand, and a full training step - loss and EVERY parameter gradient - against the
oracle's autograd applying the very masks the kernels regenerate."""

import math

import pytest
import torch

from conftest import rel_err
from oracle import ddim_ref
from oracle import denoiser_ref as ref

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from soccerdiffusion_amd import ops as o

    return o


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_mask_statistics(ops, p):
    rows, width = 4096, 250           # width not a multiple of 4: padded rows
    m = ops.dropout_mask(rows, width, (p, 1234, 77), "cuda")
    vals = torch.unique(m)
    assert set(vals.tolist()) <= {0.0, float(torch.tensor(1.0 / (1.0 - p), dtype=torch.float32))}
    keep = float((m != 0).double().mean())
    n = rows * width
    assert abs(keep - (1 - p)) < 5 * math.sqrt(p * (1 - p) / n)           # 5 sigma
    assert abs(float(m.double().mean()) - 1.0) < 5 * math.sqrt(p / (1 - p) / n)   # E[mask] = 1: unbiased scaling
    # rows and columns are not correlated: keep rate per row / per column within 6 sigma, adjacent-element agreement at chance
    row_keep = (m != 0).double().mean(dim=1)
    assert float((row_keep - (1 - p)).abs().max()) < 6 * math.sqrt(p * (1 - p) / width)
    col_keep = (m != 0).double().mean(dim=0)
    assert float((col_keep - (1 - p)).abs().max()) < 6 * math.sqrt(p * (1 - p) / rows)
    k = (m != 0)
    agree = float((k[:, 1:] == k[:, :-1]).double().mean())
    assert abs(agree - (p * p + (1 - p) ** 2)) < 0.005
    # deterministic in (seed, site); different in either
    assert torch.equal(m, ops.dropout_mask(rows, width, (p, 1234, 77), "cuda"))
    for other in ((p, 1235, 77), (p, 1234, 78), (p, 1234, 77 + (1 << 32))):
        m2 = ops.dropout_mask(rows, width, other, "cuda")
        same = float(((m2 != 0) == k).double().mean())
        assert abs(same - (p * p + (1 - p) ** 2)) < 0.005
    # p = 0: everything kept, scale 1
    assert torch.equal(ops.dropout_mask(8, 12, (0.0, 1, 1), "cuda"), torch.ones(8, 12, device="cuda"))


def test_elementwise_and_fused_kernels_apply_exactly_that_mask(ops):
    g = torch.Generator().manual_seed(0)
    drop = (0.1, 99, 5)
    R, d = 333, 256
    x = torch.randn(R, d, generator=g).cuda()
    m = ops.dropout_mask(R, d, drop, "cuda")
    assert torch.equal(ops.dropout(x, drop), x * m)
    # gelu + dropout, forward and backward
    got = ops.gelu_dropout_fwd(x, drop)
    assert rel_err(got, ref.gelu_erf(x.cpu()) * m.cpu()) < 1e-6
    assert torch.equal(got == 0, (m == 0) | (ref.gelu_erf(x.cpu()).cuda() * m == 0))
    dy = torch.randn(R, d, generator=g).cuda()
    xc = x.cpu().double().requires_grad_(True)
    (ref.gelu_erf(xc) * m.cpu().double() * dy.cpu().double()).sum().backward()
    assert rel_err(ops.gelu_dropout_bwd(dy, x, drop), xc.grad) < 1e-5
    # out = res + dropout(A W^T + b): the fused epilogue of the panel GEMM, rows not a multiple of 64, N = d and 3d
    for N in (d, 3 * d):
        W = (torch.randn(N, d, generator=g) / 16).cuda()
        b = torch.randn(N, generator=g).cuda()
        res = torch.randn(R, N, generator=g).cuda()
        mN = ops.dropout_mask(R, N, drop, "cuda")
        want = res.cpu().double() + (x.cpu().double() @ W.cpu().double().T + b.cpu().double()) * mN.cpu().double()
        got = ops.linear_dropout(x, W, b, res, drop)
        assert rel_err(got, want) < 1e-5
        plain = ops.linear(x, W, b)
        assert torch.equal(got[mN == 0], res[mN == 0])                    # dropped elements are exactly the residual
        assert rel_err(got[mN != 0], (res + plain * mN)[mN != 0]) < 1e-6


@pytest.mark.parametrize("B,T,S,d,heads,packed", [(3, 100, 100, 256, 4, True), (2, 100, 11, 256, 4, False), (2, 16, 16, 64, 4, True),
                                                  (2, 37, 37, 128, 4, True), (1, 130, 130, 256, 4, True)])
def test_attention_probability_dropout_forward_and_backward(ops, B, T, S, d, heads, packed):
    """O = (softmax(S) o m) V and its gradients: split-fp16 kernel (packed q|k|v, head dim 64, T <= 128), fp32 kernel
    (cross-attention with 11 keys; other head dims; T > 128), and attention_bwd - against fp64 autograd with the mask
    the kernels regenerate (rows = (b, h, q), width = S, padded to 4)."""
    g = torch.Generator().manual_seed(4)
    drop = (0.1, 7, 3)
    if packed:
        qkv = torch.randn(B, T, 3 * d, generator=g).cuda()
        q, k, v = qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :]
    else:
        q = torch.randn(B, T, d, generator=g).cuda()
        kv = torch.randn(B, S, 2 * d, generator=g).cuda()
        k, v = kv[..., :d], kv[..., d:]
    m = ops.dropout_mask(B * heads * T, S, drop, "cuda").view(B, heads, T, S)
    out, lse = ops.attention_lse(q, k, v, heads, drop)
    qd, kd, vd = (t.detach().cpu().double().requires_grad_(True) for t in (q, k, v))
    want = ref.attention(qd, kd, vd, heads, masks=lambda kind, shape: m.cpu().double(), kind=0)
    assert rel_err(out, want) < 2e-6
    # the normaliser is the un-dropped one: lse equals the p = 0 call's
    _, lse0 = ops.attention_lse(q, k, v, heads)
    assert rel_err(lse, lse0) < 1e-6
    dO = torch.randn(B, T, d, generator=g).cuda()
    (want * dO.cpu().double()).sum().backward()
    dq, dk, dv = torch.empty_like(q.contiguous()), torch.empty(B, S, d, device="cuda"), torch.empty(B, S, d, device="cuda")
    ops.attention_bwd(q, k, v, out, dO, lse, dq, dk, dv, heads, drop)
    assert rel_err(dq, qd.grad) < 1e-5 and rel_err(dk, kd.grad) < 1e-5 and rel_err(dv, vd.grad) < 1e-5


def _decoder_masks(ops, gen, call, B, T, M, d, heads):
    """(layer, kind, shape) -> the mask the HIP kernels regenerate for forward call number ``call`` of this decoder."""
    from soccerdiffusion_amd import training as tr

    dc = tr._DropCall(gen.dropout.p, gen.dropout.seed, (call << 24) | (gen.dropout.salt << 12))

    def masks(layer, kind, shape):
        drop = dc.site(layer, kind)
        if kind in (tr.SITE_SA_PROBS, tr.SITE_CA_PROBS):
            Bq, H, Tq, S = shape
            return ops.dropout_mask(Bq * H * Tq, S, drop, "cuda").view(shape).cpu()
        rows = shape[0] * shape[1]
        return ops.dropout_mask(rows, shape[2], drop, "cuda").view(shape).cpu()

    return masks


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("d,L,T,B", [(256, 2, 100, 3), (64, 2, 16, 2)])
def test_training_step_with_dropout_matches_oracle_with_the_same_masks(ops, d, L, T, B, fused):
    """One decoder-pretraining step in train() mode at p = 0.1: prediction, loss and every parameter gradient equal the
    oracle's autograd when it applies the same masks at torch's six sites per layer."""
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.synthetic import synthetic_state_dict
    from test_gpu_model import _build

    J, Mc = 20, 10
    sd = synthetic_state_dict(d, J, L, seed=5)
    m = _build(dict(d=d, J=J, L=L, T=T), full=False).cuda()
    m.load_state_dict(sd)
    m.train()
    m.set_dropout(0.1, seed=4242)
    if fused:   # split weight planes of a FusedAdamW switch the stack to the fused row chains
        m._opt = training.FusedAdamW(m.parameters(), lr=1e-3)
    stacks_before = training.FUSED_STACKS[0]
    g = torch.Generator().manual_seed(3)
    x0, eps = torch.randn(B, T, J, generator=g), torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, Mc, d, generator=g)
    t = torch.tensor([980, 500, 3][:B])
    x_t = ddim_ref.add_noise(x0, eps, t, ddim_ref.alphas_cumprod())
    gen = m.diffusion_action_generator
    calls_before = gen.dropout.calls
    pred = m.forward_with_context([ctx.cuda()], x_t.cuda(), t.cuda())
    assert gen.dropout.calls == calls_before + 1
    assert training.FUSED_STACKS[0] - stacks_before == (1 if fused else 0)
    loss = training.mse_loss(pred, eps.cuda())
    loss.backward()
    masks = _decoder_masks(ops, gen, gen.dropout.calls, B, T, Mc + 1, d, 4)
    want_pred, want_loss, want = ref.train_loss_and_grads(sd, x_t, t, eps, context=[ctx], dropout_masks=masks)
    assert rel_err(pred, want_pred) < TOL
    assert abs(float(loss) - float(want_loss)) / float(want_loss) < 1e-5
    named = dict(m.named_parameters())
    scale = max(float(v.norm()) for v in want.values())
    for k, gw in want.items():
        got = named[k].grad.detach().cpu()
        rel = float((got.double() - gw.double()).norm()) / max(float(gw.norm()), 1e-3 * scale)
        assert rel < TOL, (k, rel)
    # dropout really happened: the p = 0 prediction differs, and a second call draws fresh masks
    m.set_dropout(0.0)
    clean = m.forward_with_context([ctx.cuda()], x_t.cuda(), t.cuda())
    assert rel_err(pred, clean) > 1e-2
    m.set_dropout(0.1)
    again = m.forward_with_context([ctx.cuda()], x_t.cuda(), t.cuda())
    assert rel_err(again, pred) > 1e-2
    # eval() switches it off whatever p is, and routes back to the inference kernels
    m.eval()
    with torch.no_grad():
        ev = m.forward_with_context([ctx.cuda()], x_t.cuda(), t.cuda())
    assert rel_err(ev, clean) < 1e-5


def test_train_mode_without_grad_still_drops_and_encoders_drop_too(ops):
    """torch semantics: dropout follows module.training, not the grad mode - the reference's distillation teacher is
    never put into eval mode (distill.py:127-131), so its no_grad rollout and its context encoders run with dropout."""
    from soccerdiffusion_amd.synthetic import synthetic_state_dict
    from test_gpu_model import _build

    c = dict(d=64, J=20, L=2, T=16, ctx_len=20, patch=5, enc_layers=1)
    m = _build(c, full=True).cuda()
    m.set_dropout(0.1, seed=1)
    g = torch.Generator().manual_seed(0)
    inp = {"joint_command_history": torch.randn(2, 20, 20, generator=g).cuda(), "rotation": torch.randn(2, 20, 4, generator=g).cuda(),
           "joint_state": torch.randn(2, 20, 20, generator=g).cuda(), "game_state": torch.tensor([1, 3]).cuda()}
    x = torch.randn(2, 16, 20, generator=g).cuda()
    m.train()
    with torch.no_grad():
        e1 = m.encode_input_data(inp)
        e2 = m.encode_input_data(inp)
        s1 = m.sample(e1, x, 4, with_dropout=True)
        s2 = m.sample(e1, x, 4, with_dropout=True)
        native = m.sample(e1, x, 4)
    assert rel_err(e1[0], e2[0]) > 1e-3          # encoder dropout live, fresh masks per call
    assert rel_err(s1, s2) > 1e-3 and torch.isfinite(s1).all()
    m.eval()
    with torch.no_grad():
        e3, e4 = m.encode_input_data(inp), m.encode_input_data(inp)
        clean = m.sample(e1, x, 4, with_dropout=True)      # eval mode: the flag is inert
    assert torch.equal(e3[0], e4[0])
    assert torch.equal(clean, native)
