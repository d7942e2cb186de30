"""GPU parity of the training-side single ops against torch CPU autograd of the same math."""

import math

import pytest
import torch

from conftest import rel_err
from oracle import denoiser_ref as ref

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from soccerdiffusion_amd import ops as o

    return o


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("R,N,K", [(1000, 256, 256), (77, 768, 256), (300, 64, 64), (513, 256, 20), (200, 20, 256), (64, 128, 200), (5, 512, 512)])
def test_gemm_tn(ops, R, N, K):
    dY, X = _rand(R, N, seed=1), _rand(R, K, seed=2)
    dW = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda")
    ops.gemm_tn(dY.cuda(), X.cuda(), dW, db)
    assert rel_err(dW, dY.double().T @ X.double()) < TOL
    assert rel_err(db, dY.double().sum(0)) < TOL
    ops.gemm_tn(dY.cuda(), X.cuda(), dW, db)  # accumulates
    assert rel_err(dW, 2 * (dY.double().T @ X.double())) < TOL


def test_gemm_tn_staged_kernel_slices_and_dynamic_range(ops):
    """The LDS-staged kernel (N, K multiples of 128): operands that are column slices of packed (R, 3d) buffers, rows
    whose magnitudes span 1e-6 .. 1e+3 (block floating point per 32-row slab: gradients have no a-priori scale), a row
    count that is not a multiple of the slab, and agreement with the register-staged kernel's numerics class (fp64 ref)."""
    R, d = 1237, 256
    g = torch.Generator().manual_seed(7)
    big = torch.randn(R, 3 * d, generator=g)
    mag = 10.0 ** (torch.rand(R, 1, generator=g) * 9 - 6)
    big = (big * mag).cuda()
    X = (torch.randn(R, 2 * d, generator=g) * 10.0 ** (torch.rand(R, 1, generator=g) * 4 - 2)).cuda()
    for sl, xs in ((slice(0, d), slice(0, d)), (slice(d, 3 * d), slice(d, 2 * d)), (slice(0, 3 * d), slice(0, 2 * d))):
        dY, Xs = big[:, sl], X[:, xs]
        N, K = dY.shape[1], Xs.shape[1]
        dW = torch.zeros(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        ops.gemm_tn(dY, Xs, dW, db)
        assert rel_err(dW, dY.double().cpu().T @ Xs.double().cpu()) < 1e-6
        assert rel_err(db, dY.double().cpu().sum(0)) < 1e-5


def test_gemm_tn_column_slices(ops):
    R, d = 300, 64
    big = _rand(R, 3 * d, seed=1).cuda()
    X = _rand(R, d, seed=2).cuda()
    dW = torch.zeros(d, d, device="cuda")
    ops.gemm_tn(big[:, d : 2 * d], X, dW)
    assert rel_err(dW, big[:, d : 2 * d].double().cpu().T @ X.double().cpu()) < TOL


@pytest.mark.parametrize("d", [64, 128, 256, 512])
def test_layernorm_fwd_bwd(ops, d):
    R = 333
    x = _rand(R, d, seed=1).requires_grad_(True)
    g = (1 + _rand(d, seed=2, scale=0.1)).requires_grad_(True)
    b = _rand(d, seed=3, scale=0.1).requires_grad_(True)
    dy = _rand(R, d, seed=4)
    dres = _rand(R, d, seed=5)
    y = ref.layer_norm(x, g, b)
    y.backward(dy)
    yg, mean, rstd = ops.layernorm_fwd(x.detach().cuda(), g.detach().cuda(), b.detach().cuda())
    assert rel_err(yg, y) < 1e-6
    dx, dg, db = ops.layernorm_bwd(dy.cuda(), x.detach().cuda(), mean, rstd, g.detach().cuda(), dres=dres.cuda())
    assert rel_err(dx, x.grad + dres) < TOL
    assert rel_err(dg, g.grad) < TOL and rel_err(db, b.grad) < TOL


def test_gelu_fwd_bwd(ops):
    u = _rand(10000, seed=1, scale=2.0).requires_grad_(True)
    dy = _rand(10000, seed=2)
    y = ref.gelu_erf(u)
    y.backward(dy)
    assert rel_err(ops.gelu_fwd(u.detach().cuda()), y) < 1e-6
    assert rel_err(ops.gelu_bwd(dy.cuda(), u.detach().cuda()), u.grad) < 1e-5


@pytest.mark.parametrize("d,heads", [(64, 4), (128, 4), (256, 4), (512, 4)])
@pytest.mark.parametrize("Tq,S", [(16, 16), (100, 100), (10, 31), (100, 11), (130, 70), (7, 312)])
def test_attention_bwd(ops, d, heads, Tq, S):
    B = 2
    q = _rand(B, Tq, d, seed=1).requires_grad_(True)
    k = _rand(B, S, d, seed=2).requires_grad_(True)
    v = _rand(B, S, d, seed=3).requires_grad_(True)
    dO = _rand(B, Tq, d, seed=4)
    out = ref.attention(q, k, v, heads)
    out.backward(dO)
    qg, kg, vg = q.detach().cuda(), k.detach().cuda(), v.detach().cuda()
    og, lse = ops.attention_lse(qg, kg, vg, heads)
    assert rel_err(og, out) < TOL
    dq, dk, dv = torch.empty_like(qg), torch.full_like(kg, float("nan")), torch.full_like(vg, float("nan"))
    ops.attention_bwd(qg, kg, vg, og, dO.cuda(), lse, dq, dk, dv, heads)
    assert rel_err(dq, q.grad) < TOL
    assert rel_err(dk, k.grad) < TOL
    assert rel_err(dv, v.grad) < TOL


def test_attention_bwd_packed_qkv(ops):
    """q, k, v (and their gradients) as column slices of one packed (B, T, 3d) buffer."""
    B, T, d, heads = 3, 100, 256, 4
    qkv = _rand(B, T, 3 * d, seed=1)
    q, k, v = [t.clone().requires_grad_(True) for t in qkv.split(d, dim=-1)]
    dO = _rand(B, T, d, seed=2)
    ref.attention(q, k, v, heads).backward(dO)
    g = qkv.cuda()
    og, lse = ops.attention_lse(g[..., :d], g[..., d : 2 * d], g[..., 2 * d :], heads)
    dqkv = torch.empty_like(g)
    ops.attention_bwd(g[..., :d], g[..., d : 2 * d], g[..., 2 * d :], og, dO.cuda(), lse, dqkv[..., :d], dqkv[..., d : 2 * d],
                      dqkv[..., 2 * d :], heads)
    assert rel_err(dqkv, torch.cat([q.grad, k.grad, v.grad], -1)) < TOL


def test_linear_strided_accumulates_slices(ops):
    R, d = 200, 128
    dY = _rand(R, 3 * d, seed=1)
    W = _rand(3 * d, d, seed=2, scale=1 / math.sqrt(d))
    want = dY @ W  # dX = dY W
    g = dY.cuda()
    out = None
    for p in range(3):
        Wt = W[p * d : (p + 1) * d].t().contiguous().cuda()
        out = ops.linear_strided(g[:, p * d : (p + 1) * d], Wt, res=out, out=out)
    assert rel_err(out, want) < TOL


@pytest.mark.parametrize("d", [64, 128, 256, 512])
def test_linear_packed_matches_fp64_and_the_unpacked_kernel(ops, d):
    """Row GEMM on pre-split weight planes (what FusedAdamW keeps per step): LN, residual, residual + dropout, column-slice
    input, several blocks per call, block offsets out of order."""
    R, nb = 203, 3
    W = _rand(nb * d, d, seed=3, scale=1 / math.sqrt(d))
    bias = _rand(nb * d, seed=4, scale=0.1)
    A = _rand(R, 2 * d, seed=5)
    res = _rand(R, nb * d, seed=6)
    g, b = 1 + 0.1 * _rand(d, seed=7), 0.1 * _rand(d, seed=8)
    Wc, Ac = W.cuda(), A.cuda()
    # blocks stored in reverse order in the source buffer: the offsets array, not the layout, defines block b
    src = torch.cat([Wc[(nb - 1 - i) * d : (nb - i) * d].reshape(-1) for i in range(nb)])
    off = torch.tensor([(nb - 1 - i) * d * d for i in range(nb)], dtype=torch.int64, device="cuda")
    planes = torch.empty(2 * d * d * nb, dtype=torch.float16, device="cuda")
    ops.pack_weight_blocks(src, off, nb, d, planes)
    a = Ac[:, d:]   # row stride 2d
    a64 = A[:, d:].double()
    for ln, rs in ((None, None), ((g, b), None), (None, res)):
        want = (torch.nn.functional.layer_norm(a64, (d,), g.double(), b.double(), 1e-5) if ln else a64) @ W.double().t() + bias.double()
        if rs is not None:
            want = want + rs.double()
        got = ops.linear_packed(a, planes.data_ptr(), nb * d, bias.cuda(), ln=tuple(t.cuda() for t in ln) if ln else None,
                                res=rs.cuda() if rs is not None else None)
        assert rel_err(got, want.float()) < 1e-5
        plain = ops.linear(a.contiguous(), Wc, bias.cuda(), ln=tuple(t.cuda() for t in ln) if ln else None,
                           res=rs.cuda() if rs is not None else None)
        assert rel_err(got, plain) < 1e-5
    drop = (0.1, 1234, 77)
    got = ops.linear_packed(a, planes.data_ptr(), nb * d, bias.cuda(), res=res.cuda(), drop=drop)
    mask = ops.dropout_mask(R, nb * d, drop, "cuda").cpu().double()
    want = res.double() + mask * (a64 @ W.double().t() + bias.double())
    assert rel_err(got, want.float()) < 1e-5
    with pytest.raises(RuntimeError):
        ops.linear_packed(a, planes.data_ptr(), nb * d, bias.cuda(), ln=(g.cuda(), b.cuda()), res=res.cuda())


def test_optimizer_keeps_split_planes_current(ops):
    """FusedAdamW repacks the planes after every step; the autograd linear uses them and falls back when they are stale."""
    from soccerdiffusion_amd import training as tr

    d = 128
    W = torch.nn.Parameter(_rand(3 * d, d, seed=1, scale=0.1).cuda())
    bvec = torch.nn.Parameter(_rand(3 * d, seed=2, scale=0.1).cuda())
    opt = tr.FusedAdamW([W, bvec], lr=1e-2)
    assert opt.flat_wpk is not None and tr._packed_weight(W) is not None and tr._packed_weight(W[d:]) is not None
    x = _rand(70, d, seed=3).cuda()
    for _ in range(2):
        y = tr._linear(x, W, bvec)
        assert rel_err(y, x.double().cpu() @ W.detach().double().cpu().t() + bvec.detach().double().cpu()) < 1e-5
        dx = tr._dx_through_weight(y, W)
        assert rel_err(dx, y.double().cpu() @ W.detach().double().cpu()) < 1e-5
        W.grad.copy_(_rand(3 * d, d, seed=9).cuda())
        opt.step()
    with torch.no_grad():
        W.mul_(2.0)   # bumps the version: the planes are stale until the next refresh
    assert tr._packed_weight(W) is None
    assert rel_err(tr._linear(x, W, bvec), x.double().cpu() @ W.detach().double().cpu().t() + bvec.detach().double().cpu()) < 1e-5
    opt.refresh_transposes()
    assert tr._packed_weight(W) is not None


def _amax_word(t):
    """Device int32 word with the bits of max |t| (what the fused chains leave behind for the grouped GEMM)."""
    w = torch.zeros(64, dtype=torch.int32, device=t.device)    # SD_AMAX_WORDS words, the maximum counts
    w[int(t.numel()) % 64] = t.abs().max().reshape(1).view(torch.int32)[0]
    return w


def test_gemm_tn_grouped_matches_fp64(ops):
    """Several weight gradients in one launch with one scale per operand tensor: column-slice operands, ragged row counts,
    rows whose magnitudes differ by 10^4, accumulation onto existing contents, optional bias gradient."""
    g = torch.Generator().manual_seed(0)
    cases = [(1000, 256, 256, True), (77, 384, 128, False), (2816, 512, 256, True), (25, 128, 128, True),
             (1500, 20, 256, True), (1500, 256, 20, True), (333, 132, 68, False)]   # ragged tiles: the J = 20 gradients
    probs, wants, keep = [], [], []
    for i, (R, N, K, bias) in enumerate(cases):
        rowscale = torch.exp(torch.randn(R, 1, generator=g) * 2.3)            # e^(+-2.3 sigma): four decades between rows
        dY = (torch.randn(R, N + 8, generator=g) * rowscale * 1e-3).cuda()
        X = torch.randn(R, K + 4, generator=g).cuda()
        dW0 = torch.randn(N, K, generator=g).cuda()
        db0 = torch.randn(N, generator=g).cuda()
        dW, db = dW0.clone(), db0.clone()
        ay, ax = _amax_word(dY[:, :N]), _amax_word(X[:, :K])
        keep += [dY, X, ay, ax]
        probs.append((dY[:, :N], X[:, :K], dW, db if bias else None, ay.data_ptr(), ax.data_ptr()))
        wants.append((dW0.double().cpu() + dY[:, :N].double().cpu().t() @ X[:, :K].double().cpu(), db0.double().cpu() + dY[:, :N].double().cpu().sum(0), dW, db, bias, dW0, db0))
    ops.gemm_tn_grouped(probs)
    for wW, wb, dW, db, bias, dW0, db0 in wants:
        # error relative to the update, not to the random contents it is added to
        assert rel_err(dW - dW0, (wW - dW0.double().cpu()).float()) < 1e-5
        if bias:
            assert rel_err(db - db0, (wb - db0.double().cpu()).float()) < 1e-5
        else:
            assert torch.equal(db, db0)
    # more than 8 problems -> several launches; an abs-max word that is an upper bound (not the exact maximum) is fine
    R, N, K = 300, 128, 128
    dY, X = torch.randn(R, N, generator=g).cuda(), torch.randn(R, K, generator=g).cuda()
    ay, ax = _amax_word(dY * 3), _amax_word(X * 1.5)
    outs = [torch.zeros(N, K, device="cuda") for _ in range(11)]
    ops.gemm_tn_grouped([(dY, X, o, None, ay.data_ptr(), ax.data_ptr()) for o in outs])
    want = (dY.double().cpu().t() @ X.double().cpu()).float()
    for o in outs:
        assert rel_err(o, want) < 1e-5
    with pytest.raises(RuntimeError):
        ops.gemm_tn_grouped([(dY[:, :102], X, outs[0], None, ay.data_ptr(), ax.data_ptr())])   # N not a multiple of 4
    # sd_op_absmax: the words of an operand that no chain produced
    words = torch.zeros(64, dtype=torch.int32, device="cuda")
    big = torch.randn(700, 260, generator=g).cuda()
    ops.absmax(big[:, 4:24], words.data_ptr())
    assert float(words.view(torch.float32).max()) == float(big[:, 4:24].abs().max())


def test_small_k_matmul_and_colsum(ops):
    A, Bm = _rand(1000, 20, seed=1), _rand(20, 256, seed=2)
    assert rel_err(ops.small_k_matmul(A.cuda(), Bm.cuda()), A @ Bm) < 1e-5
    src = _rand(50, 11, 256, seed=3).cuda()
    out = torch.zeros(128, device="cuda")
    ops.colsum(src[:, -1, 128:], out)
    assert rel_err(out, src[:, -1, 128:].double().cpu().sum(0)) < 1e-5


def test_mse_loss(ops):
    a, b = _rand(256, 100, 20, seed=1), _rand(256, 100, 20, seed=2)
    loss, grad = ops.mse_loss(a.cuda(), b.cuda())
    want = torch.nn.functional.mse_loss(a.double(), b.double())
    assert abs(float(loss) - float(want)) / float(want) < 1e-6
    assert rel_err(grad, 2 * (a - b) / a.numel()) < 1e-6


def test_adamw_matches_torch(ops):
    n = 10007
    p0, g = _rand(n, seed=1), _rand(n, seed=2, scale=0.01)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-3)
    pg, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 6):
        p.grad = g * step
        opt.step()
        ops.adamw_step(pg, (g * step).cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step)
    assert rel_err(pg, p.detach()) < 1e-6
    st = opt.state[p]
    assert rel_err(m, st["exp_avg"]) < 1e-6 and rel_err(v, st["exp_avg_sq"]) < 1e-6
