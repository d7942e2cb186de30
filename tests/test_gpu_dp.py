"""Data parallelism on one GPU (VERDICT r1 #1c): a training step over two "virtual ranks" - two half batches whose
flat gradients are summed and halved, exactly what training.allreduce_gradients does over RCCL - must equal the
full-batch step, on the HIP path (the world-2 gloo tests on CPU cover the collectives themselves)."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(seed=0, layers=2):
    import bench
    from soccerdiffusion_amd import cli

    torch.manual_seed(seed)
    m = cli.build_model(dict(bench.C2_PARAMS, num_decoder_layers=layers)).cuda().train()
    if hasattr(m, "set_dropout"):
        m.set_dropout(0.0)
    return m


def test_two_virtual_ranks_equal_one_full_batch_step():
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.scheduler import DDIMScheduler

    B, T, J, d = 16, 100, 20, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    x0 = torch.randn(B, T, J, device="cuda", generator=g)
    ctx = torch.randn(B, 10, d, device="cuda", generator=g)
    noise = torch.randn(B, T, J, device="cuda", generator=g)
    t = torch.randint(0, 1000, (B,), device="cuda", generator=g)
    ns = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)

    full = _model()
    opt_f = training.FusedAdamW(full.parameters(), lr=1e-3)
    loss_f = training.train_step(full, opt_f, None, ns, x0, context=[ctx], noise=noise, timesteps=t)
    grad_f = opt_f.flat_grad.clone()

    dp = _model()
    opt_d = training.FusedAdamW(dp.parameters(), lr=1e-3)
    assert torch.equal(opt_d.flat_param, torch.cat([p.detach().reshape(-1) for p in _model().parameters()]))
    halves, losses = [], []
    for r in range(2):   # what rank r computes before the all-reduce
        sl = slice(r * B // 2, (r + 1) * B // 2)
        opt_d.zero_grad()
        noisy = ns.add_noise(x0[sl], noise[sl], t[sl])
        loss = training.mse_loss(dp.forward_with_context([ctx[sl]], noisy, t[sl]), noise[sl])
        loss.backward()
        halves.append(opt_d.flat_grad.clone())
        losses.append(float(loss))
    opt_d.flat_grad.copy_((halves[0] + halves[1]) * 0.5)   # all-reduce(SUM) then 1 / world
    opt_d.step()

    assert abs(0.5 * (losses[0] + losses[1]) - float(loss_f)) < 1e-5 * abs(float(loss_f))
    rel = float((opt_d.flat_grad - grad_f).norm() / grad_f.norm())
    assert rel < 2e-5, rel
    relp = float((opt_d.flat_param - opt_f.flat_param).norm() / opt_f.flat_param.norm())
    assert relp < 1e-6, relp


def test_broadcast_then_step_world1_is_identity():
    """broadcast_parameters is a no-op without a process group (cli train at WORLD_SIZE = 1)."""
    from soccerdiffusion_amd import training

    m = _model(layers=1)
    opt = training.FusedAdamW(m.parameters(), lr=1e-3)
    before = opt.flat_param.clone()
    training.broadcast_parameters(opt, m)
    assert torch.equal(before, opt.flat_param)
