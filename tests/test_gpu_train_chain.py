"""GPU parity of the fused training row chains (sd_train_fwd_chain / sd_train_bwd_chain) against fp64 torch on the CPU:
every stored tensor of the forward, every output of the backward (dx, dpre, dym, dgamma, dbeta), with and without dropout
(the Philox masks come from ops.dropout_mask, the same function the kernels evaluate), ragged last panels."""

import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from soccerdiffusion_amd import ops as o

    return o


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


class _Planes:
    """Split planes of W's d x d blocks and of their transposes, as FusedAdamW keeps them."""

    def __init__(self, ops, W):
        N, d = W.shape
        nb = N // d
        Wc = W.cuda().contiguous()
        Wt = torch.cat([Wc[b * d : (b + 1) * d].t().contiguous().reshape(-1) for b in range(nb)])
        off = torch.arange(nb, dtype=torch.int64, device="cuda") * d * d
        self.fwd = torch.empty(2 * d * d * nb, dtype=torch.float16, device="cuda")
        self.t = torch.empty(2 * d * d * nb, dtype=torch.float16, device="cuda")
        ops.pack_weight_blocks(Wc.reshape(-1), off, nb, d, self.fwd)
        ops.pack_weight_blocks(Wc.reshape(-1), off, nb, d, self.t, transposed=True)


def _ln(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def _gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def _mask(ops, R, d, p, seed, site):
    if p == 0:
        return torch.ones(R, d, dtype=torch.float64)
    return ops.dropout_mask(R, d, (p, seed, site), "cuda").cpu().double()


@pytest.mark.parametrize("d,R,p", [(256, 200, 0.0), (256, 130, 0.1), (128, 64, 0.1), (64, 77, 0.0), (64, 1, 0.25), (128, 1003, 0.1), (256, 4100, 0.0)])
def test_forward_chain_feed_forward_block(ops, d, R, p):
    s = 1 / math.sqrt(d)
    a, h = _rand(R, d, seed=1), _rand(R, d, seed=2)
    Wo, W1, W2, Wn = _rand(d, d, seed=3, scale=s), _rand(d, d, seed=4, scale=s), _rand(d, d, seed=5, scale=s), _rand(3 * d, d, seed=6, scale=s)
    bo, b1, b2, bn = (_rand(n, seed=7 + i, scale=0.1) for i, n in enumerate((d, d, d, 3 * d)))
    g3, be3, g1, be1 = 1 + 0.1 * _rand(d, seed=11), 0.1 * _rand(d, seed=12), 1 + 0.1 * _rand(d, seed=13), 0.1 * _rand(d, seed=14)
    seed, sites = 99, (1001, 1002, 1003)
    m_out, m_act, m_ffn = (_mask(ops, R, d, p, seed, st) for st in sites)
    D = lambda t: t.double()   # noqa: E731
    h1 = D(h) + m_out * (D(a) @ D(Wo).t() + D(bo))
    n = _ln(h1, D(g3), D(be3))
    pre = n @ D(W1).t() + D(b1)
    u = m_act * _gelu(pre)
    h2 = h1 + m_ffn * (u @ D(W2).t() + D(b2))
    nn = _ln(h2, D(g1), D(be1))
    y = nn @ D(Wn).t() + D(bn)

    pk = {k: _Planes(ops, w) for k, w in (("wo", Wo), ("w1", W1), ("w2", W2), ("wn", Wn))}
    new = lambda *sh: torch.full(sh, float("nan"), device="cuda")   # noqa: E731
    out = dict(h_out=new(R, d), n_out=new(R, d), pre=new(R, d), u=new(R, d), h2_out=new(R, d), nn_out=new(R, d), y_out=new(R, 3 * d))
    c = lambda t: t.cuda()   # noqa: E731
    ops.train_fwd_chain(R, d, c(h), a=c(a), wo=pk["wo"].fwd.data_ptr(), bo=c(bo), ln=(c(g3), c(be3)), w1=pk["w1"].fwd.data_ptr(), b1=c(b1),
                        w2=pk["w2"].fwd.data_ptr(), b2=c(b2), nln=(c(g1), c(be1)), wn=pk["wn"].fwd.data_ptr(), bn=c(bn), n_next=3,
                        p=p, seed=seed, sites=sites, **out)
    for name, want in (("h_out", h1), ("n_out", n), ("pre", pre), ("u", u), ("h2_out", h2), ("nn_out", nn), ("y_out", y)):
        assert rel_err(out[name], want.float()) < TOL, name
    # the abs-max words (the grouped weight-gradient GEMM's scales): exactly max |x| of a, LN3 output, u, LN1' output
    ax = torch.zeros(4, 64, dtype=torch.int32, device="cuda")    # SD_AMAX_WORDS words per tensor, the maximum counts
    ops.train_fwd_chain(R, d, c(h), a=c(a), wo=pk["wo"].fwd.data_ptr(), bo=c(bo), ln=(c(g3), c(be3)), w1=pk["w1"].fwd.data_ptr(), b1=c(b1),
                        w2=pk["w2"].fwd.data_ptr(), b2=c(b2), nln=(c(g1), c(be1)), wn=pk["wn"].fwd.data_ptr(), bn=c(bn), n_next=3,
                        p=p, seed=seed, sites=sites, amax=tuple(ax.data_ptr() + 256 * i for i in range(4)), **out)
    got = ax.view(torch.float32).max(dim=1).values.cpu()
    for i, t in enumerate((a, out["n_out"], out["u"], out["nn_out"])):
        m = float(t.abs().max())
        if R % 64 == 0 or i in (0, 2):     # rows past the end of a ragged panel are LayerNorm(0) = beta: an upper bound only
            assert got[i] == m, i
        else:
            assert m <= got[i] <= max(m, float(max(be3.abs().max(), be1.abs().max()))), i

    # the same chain without a next projection (last layer), and the out-projection + next projection form (chain A)
    out2 = dict(h_out=new(R, d), n_out=new(R, d), pre=new(R, d), u=new(R, d), h2_out=new(R, d))
    ops.train_fwd_chain(R, d, c(h), a=c(a), wo=pk["wo"].fwd.data_ptr(), bo=c(bo), ln=(c(g3), c(be3)), w1=pk["w1"].fwd.data_ptr(), b1=c(b1),
                        w2=pk["w2"].fwd.data_ptr(), b2=c(b2), p=p, seed=seed, sites=sites, **out2)
    assert rel_err(out2["h2_out"], h2.float()) < TOL
    outA = dict(h_out=new(R, d), nn_out=new(R, d), y_out=new(R, d))
    ops.train_fwd_chain(R, d, c(h), a=c(a), wo=pk["wo"].fwd.data_ptr(), bo=c(bo), nln=(c(g3), c(be3)), wn=pk["w1"].fwd.data_ptr(), bn=c(b1),
                        n_next=1, p=p, seed=seed, sites=sites, **outA)
    assert rel_err(outA["h_out"], h1.float()) < TOL
    assert rel_err(outA["nn_out"], n.float()) < TOL
    assert rel_err(outA["y_out"], pre.float()) < TOL
    # head form: LayerNorm + projection of h itself
    outH = dict(nn_out=new(R, d), y_out=new(R, 3 * d))
    ops.train_fwd_chain(R, d, c(h), nln=(c(g1), c(be1)), wn=pk["wn"].fwd.data_ptr(), bn=c(bn), n_next=3, **outH)
    nh = _ln(D(h), D(g1), D(be1))
    assert rel_err(outH["nn_out"], nh.float()) < TOL
    assert rel_err(outH["y_out"], (nh @ D(Wn).t() + D(bn)).float()) < TOL


@pytest.mark.parametrize("d,R,p", [(256, 200, 0.0), (256, 130, 0.1), (128, 65, 0.1), (64, 77, 0.0), (64, 999, 0.1), (256, 4100, 0.1)])
def test_backward_chains(ops, d, R, p):
    s = 1 / math.sqrt(d)
    seed, sites = 5, (2001, 2002)
    dy = _rand(R, d, seed=1, scale=1e-3)          # gradients are small: the per-row scale has to cope
    x, pre = _rand(R, d, seed=2), _rand(R, d, seed=3)
    W2, W1, W3 = _rand(d, d, seed=4, scale=s), _rand(d, d, seed=5, scale=s), _rand(3 * d, d, seed=6, scale=s)
    gam = 1 + 0.1 * _rand(d, seed=7)
    m_in, m_act = _mask(ops, R, d, p, seed, sites[0]), _mask(ops, R, d, p, seed, sites[1])
    D = lambda t: t.double()   # noqa: E731
    c = lambda t: t.cuda()     # noqa: E731
    new = lambda *sh: torch.full(sh, float("nan"), device="cuda")   # noqa: E731

    def ln_bwd(t, xx, g, dres):
        xr = xx.clone().requires_grad_(True)
        gr = g.clone().requires_grad_(True)
        br = torch.zeros_like(g).requires_grad_(True)
        _ln(xr, gr, br).backward(t)
        return xr.grad + dres, gr.grad, br.grad

    # feed-forward form
    dym = m_in * D(dy)
    pr = D(pre).clone().requires_grad_(True)
    _gelu(pr).backward(torch.ones_like(pr))
    dpre = (dym @ D(W2)) * pr.grad * m_act
    dx, dg, db = ln_bwd(dpre @ D(W1), D(x), D(gam), D(dy))
    p2, p1 = _Planes(ops, W2), _Planes(ops, W1)
    o = dict(dym=new(R, d) if p > 0 else None, dpre=new(R, d), dg=torch.ones(d, device="cuda"), db=torch.ones(d, device="cuda"))
    odx = new(R, d)
    ops.train_bwd_chain(R, d, c(dy), p2.t.data_ptr(), odx, pre=c(pre), wt1=p1.t.data_ptr(), x=c(x), ln_w=c(gam), dres=c(dy),
                        p=p, seed=seed, sites=sites, **o)
    if p > 0:
        assert rel_err(o["dym"], dym.float()) < TOL
    assert rel_err(o["dpre"], dpre.float()) < TOL
    assert rel_err(odx, dx.float()) < TOL
    ax = torch.zeros(2, 64, dtype=torch.int32, device="cuda")
    ops.train_bwd_chain(R, d, c(dy), p2.t.data_ptr(), odx, pre=c(pre), wt1=p1.t.data_ptr(), x=c(x), ln_w=c(gam), dres=c(dy),
                        p=p, seed=seed, sites=sites, amax=(ax.data_ptr(), ax.data_ptr() + 256),
                        **dict(o, dg=torch.zeros(d, device="cuda"), db=torch.zeros(d, device="cuda")))
    got = ax.view(torch.float32).max(dim=1).values.cpu()
    assert got[0] == float((o["dym"] if p > 0 else dy).abs().max()) and got[1] == float(o["dpre"].abs().max())
    assert rel_err(o["dg"] - 1, dg.float()) < 2e-5      # accumulated onto the buffer's contents (fp32 atomics)
    assert rel_err(o["db"] - 1, db.float()) < 2e-5

    # out-projection form: dx = mask(dy) W
    odx, odym = new(R, d), new(R, d)
    ops.train_bwd_chain(R, d, c(dy), p2.t.data_ptr(), odx, dym=odym if p > 0 else None, p=p, seed=seed, sites=sites)
    assert rel_err(odx, (dym @ D(W2)).float()) < TOL
    if p > 0:
        assert rel_err(odym, dym.float()) < TOL

    # LayerNorm + projection form, three passes over a (R, 3d) gradient with a row stride; never masked
    dY3 = _rand(R, 3 * d + 8, seed=9, scale=1e-2)
    p3 = _Planes(ops, W3)
    dx3, dg3, db3 = ln_bwd(D(dY3[:, : 3 * d]) @ D(W3), D(x), D(gam), D(dy))
    odx, odg, odb = new(R, d), torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    ops.train_bwd_chain(R, d, c(dY3)[:, : 3 * d], p3.t.data_ptr(), odx, passes=3, x=c(x), ln_w=c(gam), dres=c(dy), dg=odg, db=odb,
                        p=p, seed=seed, sites=sites)
    assert rel_err(odx, dx3.float()) < TOL
    assert rel_err(odg, dg3.float()) < 2e-5
    assert rel_err(odb, db3.float()) < 2e-5
    # one pass, no residual gradient
    dx1, _, _ = ln_bwd(D(dY3[:, :d]) @ D(W3[:d]), D(x), D(gam), torch.zeros(R, d, dtype=torch.float64))
    odx = new(R, d)
    ops.train_bwd_chain(R, d, c(dY3)[:, :d], p3.t.data_ptr(), odx, x=c(x), ln_w=c(gam), dg=odg, db=odb)
    assert rel_err(odx, dx1.float()) < TOL


def test_chain_argument_errors(ops):
    d, R = 64, 10
    h = torch.zeros(R, d, device="cuda")
    with pytest.raises(RuntimeError):
        ops.train_fwd_chain(R, d, h)                      # nothing to do
    with pytest.raises(RuntimeError):
        ops.train_fwd_chain(R, 96, h, nln=(h[0], h[0]), nn_out=h, wn=h.data_ptr(), bn=h[0], y_out=h, n_next=1)   # hidden_dim
    with pytest.raises(RuntimeError):
        ops.train_bwd_chain(R, d, h, h.data_ptr(), h, pre=h)   # GELU stage without dpre / wt1 / LayerNorm stage
