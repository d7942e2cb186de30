"""GPU parity of the single-op C-ABI entry points against the CPU oracle blocks."""

import math

import pytest
import torch

from conftest import rel_err
from oracle import denoiser_ref as ref

pytestmark = pytest.mark.gpu
TOL = 1e-4  # BASELINE.json: <= 1e-4 relative fp32 vs the CPU path


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from soccerdiffusion_amd import ops as o

    return o


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("d", [64, 128, 256, 512])
@pytest.mark.parametrize("R", [1, 63, 64, 200, 1000])
def test_linear_plain_and_ln_gelu_res(ops, d, R):
    A = _rand(R, d, seed=1)
    for N in (d, 3 * d):
        W = _rand(N, d, seed=2, scale=1 / math.sqrt(d))
        b = _rand(N, seed=3, scale=0.1)
        g = 1 + _rand(d, seed=4, scale=0.1)
        be = _rand(d, seed=5, scale=0.1)
        res = _rand(R, N, seed=6)
        dev = lambda t: t.cuda()
        want = A @ W.T + b
        assert rel_err(ops.linear(dev(A), dev(W), dev(b)), want) < TOL
        want_ln = ref.layer_norm(A, g, be) @ W.T + b
        assert rel_err(ops.linear(dev(A), dev(W), dev(b), ln=(dev(g), dev(be))), want_ln) < TOL
        assert rel_err(ops.linear(dev(A), dev(W), dev(b), ln=(dev(g), dev(be)), act="gelu"), ref.gelu_erf(want_ln)) < TOL
        r = dev(res)
        got = ops.linear(dev(A), dev(W), dev(b), res=r, out=r)  # in place on the residual, as the layers use it
        assert rel_err(got, res + want) < TOL


@pytest.mark.parametrize("d,heads", [(64, 4), (128, 4), (256, 4), (512, 4), (256, 8)])
@pytest.mark.parametrize("Tq,S", [(16, 16), (100, 100), (10, 11), (100, 11), (7, 312), (130, 130), (1, 1)])
def test_attention(ops, d, heads, Tq, S):
    B = 3
    q, k, v = _rand(B, Tq, d, seed=1), _rand(B, S, d, seed=2), _rand(B, S, d, seed=3)
    want = ref.attention(q, k, v, heads)
    assert rel_err(ops.attention(q.cuda(), k.cuda(), v.cuda(), heads), want) < TOL
    # one extra key/value row shared by the batch (the step token in the sampler)
    ke, ve = _rand(d, seed=4), _rand(d, seed=5)
    k2 = torch.cat([k, ke.expand(B, 1, d)], 1)
    v2 = torch.cat([v, ve.expand(B, 1, d)], 1)
    want = ref.attention(q, k2, v2, heads)
    assert rel_err(ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, extra=(ke.cuda(), ve.cuda())), want) < TOL


def test_attention_large_scores_are_stable(ops):
    """Scores far outside exp range in later key chunks exercise the online-softmax rescale."""
    B, Tq, S, d, heads = 2, 40, 300, 256, 4
    q, k, v = _rand(B, Tq, d, seed=1, scale=4.0), _rand(B, S, d, seed=2, scale=4.0), _rand(B, S, d, seed=3)
    k[:, 200] = q[:, 5] * 3  # spike: the running max jumps in the second chunk
    want = ref.attention(q.double(), k.double(), v.double(), heads)
    assert rel_err(ops.attention(q.cuda(), k.cuda(), v.cuda(), heads), want) < TOL


@pytest.mark.parametrize("d", [64, 256, 512])
@pytest.mark.parametrize("C,p,S", [(20, 1, 16), (20, 1, 100), (22, 1, 10), (20, 10, 100), (4, 5, 100), (5, 1, 100), (22, 10, 100)])
def test_patch_embed(ops, d, C, p, S):
    B = 3
    x = _rand(B, S, C, seed=1)
    w = _rand(d, C, p, seed=2, scale=1 / math.sqrt(C * p))
    b = _rand(d, seed=3, scale=0.1)
    n = S // p
    pe = ref.positional_table(d, n)
    patches = x[:, : n * p].reshape(B, n, p, C).permute(0, 1, 3, 2).reshape(B, n, C * p)
    want = patches @ w.reshape(d, C * p).T + b + pe
    wd = w.cuda() if p > 1 else w.reshape(d, C).cuda()
    assert rel_err(ops.patch_embed(x.cuda(), wd, b.cuda(), pe.cuda()), want) < TOL


@pytest.mark.parametrize("d,J,R", [(64, 20, 32), (256, 20, 1000), (256, 22, 65), (512, 22, 100), (128, 1, 5), (256, 33, 130), (64, 64, 7)])
def test_fc_out_and_fused_ddim(ops, d, J, R):
    from oracle import ddim_ref

    h = _rand(R, d, seed=1)
    W = _rand(J, d, seed=2, scale=1 / math.sqrt(d))
    b = _rand(J, seed=3, scale=0.1)
    x = _rand(R, J, seed=4)
    want = h @ W.T + b
    assert rel_err(ops.fc_out(h.cuda(), W.cuda(), b.cuda()), want) < TOL
    acp = ddim_ref.alphas_cumprod()
    coef = ops.ddim_coefficients([980], acp, 50)[0]
    xg = x.cuda()
    eps = ops.fc_out(h.cuda(), W.cuda(), b.cuda(), x_io=xg, coef4=coef)
    assert rel_err(eps, want) < TOL
    assert rel_err(xg, ddim_ref.step(want, 980, x, 50, acp)) < TOL


def test_step_token_bitwise_table_and_values(ops):
    d = 256
    token = _rand(1, d // 2, seed=1)
    freq = ops.step_frequencies(d)
    for steps in (torch.tensor([0, 1, 20, 500, 980, 999]), torch.tensor([0.0, 0.5, 512.0])):
        want = ref.step_token(steps, token, d)
        got = ops.step_token(steps.cuda(), freq.cuda(), token.cuda())
        assert got.shape == want.shape
        assert float((got.cpu() - want).abs().max()) < 2e-6  # sin/cos of up to 999 rad in fp32


def test_game_state_embed(ops):
    table = _rand(4, 128, seed=1)
    idx = torch.tensor([0, 3, 2, 1, 3])
    assert torch.equal(ops.game_state_embed(idx.cuda(), table.cuda()).cpu(), table[idx].unsqueeze(1))


def test_ddim_add_noise_and_step(ops):
    from oracle import ddim_ref

    acp = ddim_ref.alphas_cumprod()
    x0, eps = _rand(5, 100, 20, seed=1), _rand(5, 100, 20, seed=2)
    t = torch.tensor([0, 20, 500, 980, 999])
    assert rel_err(ops.ddim_add_noise(x0.cuda(), eps.cuda(), t.cuda(), acp.cuda()), ddim_ref.add_noise(x0, eps, t, acp)) < 1e-6
    for tt in (980, 500, 0):
        coef = ops.ddim_coefficients([tt], acp, 50)[0]
        assert rel_err(ops.ddim_step(eps.cuda(), x0.cuda(), coef), ddim_ref.step(eps, tt, x0, 50, acp)) < 1e-6


def test_bad_arguments_raise(ops):
    A = torch.zeros(4, 96, device="cuda")
    W = torch.zeros(96, 96, device="cuda")
    with pytest.raises(RuntimeError, match="hidden_dim"):
        ops.linear(A, W)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.linear(torch.zeros(4, 64), torch.zeros(64, 64))
