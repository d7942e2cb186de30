"""The reference's OWN loop form on the trajectory kernels: ``for t in scheduler.timesteps: eps = model.forward_with_context(ctx, x,
t); x = scheduler.step(eps, t, x).prev_sample`` (soccer_diffusion/ml/inference/plot.py:122-131, ml/training/distill.py:179-189,
ml/inference/ros.py:301-310) reaches ``traj_step_kernel`` through ``sd_sampler_prepare`` / ``sd_sampler_eps`` (ops.LoopSampler): every
noise prediction and the final sample against the CPU oracle, the context folded ONCE per loop, per-sample steps, the (1,) step of
plot.py, an empty context, joint counts that are not multiples of four (the real database's 22), cache invalidation."""

import pytest
import torch

from conftest import rel_err
from oracle import ddim_ref
from oracle import denoiser_ref as ref

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(d, J, L, T, seed=5):
    from soccerdiffusion_amd.ml.model import End2EndDiffusionTransformer
    from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, SequenceEncoderType
    from soccerdiffusion_amd.ml.model.encoder.imu import IMUEncoder
    from soccerdiffusion_amd.synthetic import synthetic_state_dict

    m = End2EndDiffusionTransformer(
        num_joints=J, hidden_dim=d, use_action_history=False, num_action_history_encoder_layers=1, max_action_context_length=20,
        encoder_patch_size=5, use_imu=False, imu_orientation_embedding_method=IMUEncoder.OrientationEmbeddingMethod.QUATERNION,
        num_imu_encoder_layers=1, imu_context_length=20, use_joint_states=False, joint_state_encoder_layers=1,
        joint_state_context_length=20, use_images=False, image_encoder_type=ImageEncoderType.RESNET18,
        image_sequence_encoder_type=SequenceEncoderType.TRANSFORMER, num_image_sequence_encoder_layers=1, image_context_length=0,
        image_use_final_avgpool=True, image_resolution=480, use_gamestate=False, num_decoder_layers=L, trajectory_prediction_length=T)
    sd = synthetic_state_dict(d, J, L, seed=seed)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


def _loop_cache(m):
    from soccerdiffusion_amd.ml.model import model as mm

    return mm._model_cache(m, "loop")


@pytest.mark.parametrize("T,Mc,J,L,B,n", [(100, 10, 20, 4, 3, 10), (10, 0, 20, 2, 2, 6), (10, 50, 22, 2, 2, 5), (100, 10, 22, 2, 2, 4),
                                          (33, 7, 21, 2, 3, 4), (16, 10, 3, 1, 2, 3)])
def test_scheduler_loop_runs_the_step_kernel(T, Mc, J, L, B, n):
    from soccerdiffusion_amd import _lib
    from soccerdiffusion_amd.scheduler import DDIMScheduler

    d = 256
    assert _lib.load().sd_sampler_mode(d, 4, T, Mc, J) == 3
    m, sd = _model(d, J, L, T)
    g = torch.Generator().manual_seed(T + Mc)
    x_T = torch.randn(B, T, J, generator=g)
    ctxs = [torch.randn(B, Mc - Mc // 2, d, generator=g), torch.randn(B, Mc // 2, d, generator=g)] if Mc > 1 else []
    if Mc == 1:
        ctxs = [torch.randn(B, 1, d, generator=g)]
    acp = ddim_ref.alphas_cumprod()
    want_eps = []

    def oracle(x, t):
        e = ref.forward_with_context(sd, ctxs, x, torch.full((B,), t, dtype=torch.int64))
        want_eps.append(e)
        return e

    want = ddim_ref.sample(oracle, x_T, n, acp)
    sched = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
    sched.config["num_train_timesteps"] = 1000
    sched.set_timesteps(n)
    ctx_gpu = [c.cuda() for c in ctxs]
    traj = x_T.cuda()
    x_in = traj.clone()
    with torch.no_grad():
        for i, t in enumerate(sched.timesteps):
            eps = m.forward_with_context(ctx_gpu, traj, torch.full((B,), int(t), device="cuda"))
            assert rel_err(eps, want_eps[i]) < TOL, i
            traj = sched.step(eps, t, traj).prev_sample
            assert rel_err(traj, want[i]) < TOL, i
    assert torch.equal(x_in, x_T.cuda())   # inputs are never written (distill.py reuses noisy_trajectory)
    cache = _loop_cache(m)
    assert len(cache) == 1
    ls = next(iter(cache.values()))
    assert ls.supported and ls.prepares == 1   # weights split and context folded once for the whole loop
    native = m.sample(ctx_gpu, x_T.cuda(), n)
    assert rel_err(native, traj) < 5e-6


def test_per_sample_steps_single_step_and_float_steps():
    """train.py-style per-sample timesteps (validation), plot.py's ``torch.tensor([t])`` for a whole batch, and the distilled student's
    float zeros (distill.py:193-195)."""
    d, J, L, T, B, Mc = 256, 20, 2, 10, 3, 10
    m, sd = _model(d, J, L, T)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, Mc, d, generator=g)
    cg = [ctx.cuda()]
    with torch.no_grad():
        steps = torch.tensor([980, 500, 0])
        assert rel_err(m.forward_with_context(cg, x.cuda(), steps.cuda()), ref.forward_with_context(sd, [ctx], x, steps)) < TOL
        one = torch.tensor([640])
        assert rel_err(m.forward_with_context(cg, x.cuda(), one.cuda()), ref.forward_with_context(sd, [ctx], x, one.expand(B))) < TOL
        fz = torch.zeros(B)
        assert rel_err(m.forward_with_context(cg, x.cuda(), fz.cuda()), ref.forward_with_context(sd, [ctx], x, fz)) < TOL
        ff = torch.tensor([3.5, 250.25, 999.0])
        assert rel_err(m.forward_with_context(cg, x.cuda(), ff.cuda()), ref.forward_with_context(sd, [ctx], x, ff)) < TOL
    assert all(ls.supported for ls in _loop_cache(m).values())


def test_loop_cache_follows_weights_and_context():
    """A changed context tensor (new object, or the same object written in place) and changed weights (in place through torch, or
    through FusedAdamW's native update, which moves no version counter) must be seen."""
    from soccerdiffusion_amd.training import FusedAdamW

    d, J, L, T, B, Mc = 256, 20, 2, 16, 2, 5
    m, sd = _model(d, J, L, T)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, Mc, d, generator=g)
    step = torch.full((B,), 300)
    cg = ctx.cuda()

    def check():
        cur = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        with torch.no_grad():
            got = m.forward_with_context([cg], x.cuda(), step.cuda())
        assert rel_err(got, ref.forward_with_context(cur, [cg.cpu()], x, step)) < TOL

    check()
    ls = next(iter(_loop_cache(m).values()))
    assert ls.prepares == 1
    check()
    assert ls.prepares == 1
    cg.mul_(0.5)            # same tensor object, new contents
    check()
    assert ls.prepares == 2
    cg = (ctx * 2).cuda()   # new object
    check()
    assert ls.prepares == 3
    with torch.no_grad():
        m.diffusion_action_generator.transformer_decoder.layers[0].linear1.weight.mul_(1.5)
    check()
    assert ls.prepares == 4
    # FusedAdamW: parameters become views of a flat buffer updated by a native kernel
    for p in m.parameters():
        p.requires_grad_(True)
    opt = FusedAdamW(m.parameters(), lr=1e-2)
    for p in m.parameters():
        p.grad.normal_()
    opt.step()
    for p in m.parameters():
        p.requires_grad_(False)
    check()


def test_unsupported_shapes_fall_back():
    """hidden_dim 64 (BASELINE configs[0]) is not a trajectory-kernel shape: the call lands on sd_denoiser_forward as before."""
    d, J, L, T, B, Mc = 64, 20, 2, 16, 2, 10
    m, sd = _model(d, J, L, T)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, Mc, d, generator=g)
    step = torch.tensor([900, 100])
    with torch.no_grad():
        got = m.forward_with_context([ctx.cuda()], x.cuda(), step.cuda())
    assert rel_err(got, ref.forward_with_context(sd, [ctx], x, step)) < TOL
    assert not any(ls.supported for ls in _loop_cache(m).values())
