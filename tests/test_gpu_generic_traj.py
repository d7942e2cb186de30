"""The generic trajectory step kernels (csrc/sd_trajg.hip): hidden_dim 128 / 256 / 512 with any number of memory rows - the reference's
own shipped shapes (ml/training/config/default.yaml: hidden_dim 128, 100 + 100 + 100 + 10 + 1 context rows at encoder_patch_size 1;
larger_model.yaml: hidden_dim 512, 8 layers) and every hidden_dim-256 shape beyond sd_traj.h's 64 memory rows.  x after EVERY DDIM step
and every noise prediction against the fp32 CPU oracle (reference blocks: ml/model/decoder.py:26-54 under the loop of
ml/inference/plot.py:122-131), through sd_ddim_sample_eps and through the loop form (sd_sampler_prepare / sd_sampler_eps)."""

import pytest
import torch

from conftest import rel_err
from oracle import ddim_ref
from oracle import denoiser_ref as ref

pytestmark = pytest.mark.gpu
TOL = 1e-4

SHAPES = [
    # d, T, Mc, J, L, B
    (128, 10, 311, 20, 4, 3),    # default.yaml
    (512, 10, 311, 20, 8, 2),    # larger_model.yaml / larger_model_distill.yaml
    (128, 10, 311, 22, 4, 2),    # ... with the real database's 22 joints
    (512, 10, 311, 22, 2, 2),
    (256, 10, 311, 22, 2, 2),    # hidden_dim 256 beyond 64 memory rows
    (256, 100, 100, 20, 2, 2),
    (256, 64, 64, 20, 2, 2),
    (256, 100, 311, 20, 1, 1),
    (128, 100, 10, 20, 2, 3),    # BASELINE's horizon at hidden_dim 128: 7 token tiles, ragged last tile
    (128, 97, 0, 22, 1, 2),      # no context rows (decoder pretraining): the step token alone
    (128, 33, 40, 7, 2, 2),
    (128, 64, 5, 20, 2, 2),
    (128, 81, 17, 4, 2, 2),
    (512, 48, 10, 20, 2, 2),     # the largest horizon at hidden_dim 512
    (512, 17, 70, 22, 2, 2),
    (512, 1, 3, 4, 1, 2),        # one token
    (512, 33, 0, 20, 1, 2),
    (128, 10, 32, 20, 1, 2),     # pair edges of the streamed memory: 31 / 32 / 33 context rows, one row
    (128, 16, 33, 20, 1, 2),
    (128, 16, 31, 20, 1, 2),
    (128, 10, 1, 20, 1, 2),
    (128, 1, 1, 1, 1, 1),
]


@pytest.mark.parametrize("d,T,Mc,J,L,B", SHAPES)
def test_generic_step_kernel_every_step(d, T, Mc, J, L, B):
    from soccerdiffusion_amd import _lib, ops

    n_steps = 4
    assert _lib.load().sd_sampler_mode(d, 4, T, Mc, J) == 3
    sd = ref.synthetic_state_dict(d, J, L, seed=17 + T + d)
    g = torch.Generator().manual_seed(T * 7 + Mc + d)
    x_T = torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, Mc, d, generator=g) if Mc else None
    acp = ddim_ref.alphas_cumprod()
    ts = ddim_ref.timesteps(n_steps).tolist()
    want_eps = []

    def oracle(x, t):
        e = ref.forward_with_context(sd, [ctx] if Mc else [], x, torch.full((B,), t, dtype=torch.int64))
        want_eps.append(e)
        return e

    want = ddim_ref.sample(oracle, x_T, n_steps, acp)
    packed = ops.pack_denoiser(sd, "cuda", max_len=T)
    toks = ops.step_token(torch.tensor(ts).cuda(), ops.step_frequencies(d).cuda(), sd["step_encoding.token"].cuda()).reshape(n_steps, d)
    coef = ops.ddim_coefficients(ts, acp, n_steps)
    cg = ctx.cuda() if Mc else None
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    xm, trm, epm = ops.ddim_sample(packed, cg, toks, coef, x_T.cuda(), trace=True, eps_trace=True, max_mode=3, status=status)
    errs = [rel_err(trm[i], want[i]) for i in range(n_steps)]
    eerr = [rel_err(epm[i], want_eps[i]) for i in range(n_steps)]
    assert all(e < TOL for e in errs), errs
    assert all(e < TOL for e in eerr), eerr
    assert int(status.item()) == 0 and torch.isfinite(xm).all()
    # the older kernels (row panels / unfused chains) agree at fp32 rounding level
    x2 = ops.ddim_sample(packed, cg, toks, coef, x_T.cuda(), max_mode=2)
    assert rel_err(xm, x2.cpu()) < 2e-5


@pytest.mark.parametrize("name,d,L,Mc", [("default", 128, 4, 311), ("larger_model", 512, 8, 311), ("decoder_only", 256, 4, 0), ("sim_scratch", 256, 6, 50)])
@pytest.mark.parametrize("J", [20, 22])
def test_every_shipped_yaml_takes_a_trajectory_kernel(name, d, L, Mc, J):
    """sd_sampler_mode >= 3 for all five YAMLs (larger_model_distill has larger_model's shapes) at 20 and at 22 joints."""
    from soccerdiffusion_amd import _lib

    assert _lib.load().sd_sampler_mode(d, 4, 10, Mc, J) >= 3


def test_loop_form_on_the_generic_kernels():
    """forward_with_context in the reference's loop (plot.py:122-131) at default.yaml's decoder shape: one generic step launch per call, the
    311 context rows projected and packed once."""
    from test_gpu_loop_form import _loop_cache, _model

    from soccerdiffusion_amd.scheduler import DDIMScheduler

    d, J, L, T, B, Mc, n = 128, 22, 2, 10, 2, 311, 5
    m, sd = _model(d, J, L, T)
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn(B, T, J, generator=g)
    ctxs = [torch.randn(B, 300, d, generator=g), torch.randn(B, 11, d, generator=g)]
    want_eps = []

    def oracle(x, t):
        e = ref.forward_with_context(sd, ctxs, x, torch.full((B,), t, dtype=torch.int64))
        want_eps.append(e)
        return e

    want = ddim_ref.sample(oracle, x_T, n, ddim_ref.alphas_cumprod())
    sched = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
    sched.set_timesteps(n)
    cg = [c.cuda() for c in ctxs]
    traj = x_T.cuda()
    with torch.no_grad():
        for i, t in enumerate(sched.timesteps):
            eps = m.forward_with_context(cg, traj, torch.full((B,), int(t), device="cuda"))
            assert rel_err(eps, want_eps[i]) < TOL, i
            traj = sched.step(eps, t, traj).prev_sample
    assert rel_err(traj, want[-1]) < TOL
    ls = next(iter(_loop_cache(m).values()))
    assert ls.supported and ls.prepares == 1
    with torch.no_grad():   # per-sample steps on the generic kernels
        steps = torch.tensor([980, 17])
        assert rel_err(m.forward_with_context(cg, x_T.cuda(), steps.cuda()), ref.forward_with_context(sd, ctxs, x_T, steps)) < TOL


def test_generic_rollout_replays_from_a_hipgraph_and_serves_the_robot_batch():
    """default.yaml's decoder shape at the robot's B = 1: model.sample(use_graph=True) (the rollout captured into a hipGraph, generic kernels
    inside) equals the eager call bit for bit and the oracle at 1e-4; a second context goes through the same graph."""
    from test_gpu_loop_form import _model

    d, J, L, T, B, Mc, n = 128, 20, 4, 10, 1, 311, 30
    m, sd = _model(d, J, L, T)
    g = torch.Generator().manual_seed(8)
    for rep in range(2):
        x_T = torch.randn(B, T, J, generator=g)
        ctx = torch.randn(B, Mc, d, generator=g)
        want = ddim_ref.sample(lambda x, t: ref.forward_with_context(sd, [ctx], x, torch.full((B,), t, dtype=torch.int64)), x_T, n,
                               ddim_ref.alphas_cumprod())[-1]
        eager = m.sample([ctx.cuda()], x_T.cuda(), n)
        graphed = m.sample([ctx.cuda()], x_T.cuda(), n, use_graph=True)
        assert torch.equal(eager, graphed), rep
        assert rel_err(eager, want) < TOL
