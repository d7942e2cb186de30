"""GPU parity of the training step: loss and EVERY parameter gradient against the values the
reference itself produced (tests/golden, dropout p=0), the fused AdamW + OneCycleLR
trajectory against torch's own optimizers, and checkpoint interchange."""

import io

import pytest
import torch

from conftest import rel_err
from oracle import ddim_ref
from oracle import denoiser_ref as ref
from test_gpu_model import _build


def _build0(c, full):
    """The parity path: dropout p = 0 (the golden gradients and the oracle's autograd have none)."""
    return _build(c, full).set_dropout(0.0)

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _check_grads(model, golden_grads):
    named = dict(model.named_parameters())
    assert set(golden_grads) == {k for k, p in named.items() if p.grad is not None}
    scale = max(float(g.norm()) for g in golden_grads.values())
    worst = 0.0
    for k, g in golden_grads.items():
        got = named[k].grad.detach().cpu()
        err = float((got.double() - g.double()).norm())
        # relative to the gradient's own norm, with a floor for gradients that are ~0 by
        # construction (key bias of softmax attention)
        rel = err / max(float(g.norm()), 1e-3 * scale)
        worst = max(worst, rel)
        assert rel < TOL, (k, rel)
    return worst


def _arm_fused(m, fused):
    """fused: give the model a FusedAdamW - its split weight planes switch the layer stacks to the fused row chains
    (training._fused_stack); without one the per-operation autograd nodes run.  Returns the check to call afterwards."""
    from soccerdiffusion_amd import training

    if not fused:
        return lambda n: None
    m._opt = training.FusedAdamW(m.parameters(), lr=1e-3)
    before = training.FUSED_STACKS[0]

    def check(n):
        assert training.FUSED_STACKS[0] - before == n, "the fused chains did not run"

    return check


@pytest.mark.parametrize("fused", [False, True])
def test_decoder_pretraining_step_gradients_golden(g1, fused):
    from soccerdiffusion_amd import training

    m = _build0(g1["config"], full=False).cuda()
    m.load_state_dict(g1["state_dict"])
    m.train()
    ran = _arm_fused(m, fused)
    tr = g1["train"]
    pred = m.forward_with_context([g1["ctx"].cuda()], g1["x"].cuda(), g1["steps_int"].cuda())
    loss = training.mse_loss(pred, tr["noise"].cuda())
    assert rel_err(pred, tr["pred"]) < TOL
    assert abs(float(loss) - float(tr["loss"])) / float(tr["loss"]) < 1e-5
    loss.backward()
    _check_grads(m, tr["grads"])
    ran(1)


@pytest.mark.parametrize("fused", [False, True])
def test_full_model_step_gradients_golden(g2, fused):
    from soccerdiffusion_amd import training

    m = _build0(g2["config"], full=True).cuda()
    m.load_state_dict(g2["state_dict"])
    m.train()
    ran = _arm_fused(m, fused)
    tr = g2["train"]
    inp = {k: v.cuda() for k, v in g2["input_data"].items()}
    pred = m(inp, g2["x"].cuda(), g2["steps"].cuda())
    loss = training.mse_loss(pred, tr["noise"].cuda())
    assert rel_err(pred, tr["pred"]) < TOL
    loss.backward()
    _check_grads(m, tr["grads"])
    ran(4)   # the decoder and the three sequence encoders (joint commands, IMU, joint states)


@pytest.mark.parametrize("fused", [False, True])
def test_c2_shape_gradients_vs_oracle(fused):
    """BASELINE config 2 shape (d=256, L=4, T=100, M=11), small batch, vs CPU autograd of the oracle."""
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.synthetic import synthetic_state_dict

    c = dict(d=256, J=20, L=4, T=100)
    sd = synthetic_state_dict(256, 20, 4, seed=5)
    m = _build0(c, full=False).cuda()
    m.load_state_dict(sd)
    ran = _arm_fused(m, fused)
    g = torch.Generator().manual_seed(3)
    B = 3
    x0, eps = torch.randn(B, 100, 20, generator=g), torch.randn(B, 100, 20, generator=g)
    ctx = torch.randn(B, 10, 256, generator=g)
    t = torch.tensor([980, 500, 3])
    acp = ddim_ref.alphas_cumprod()
    x_t = ddim_ref.add_noise(x0, eps, t, acp)
    _, want_loss, want = ref.train_loss_and_grads(sd, x_t, t, eps, context=[ctx])
    pred = m.forward_with_context([ctx.cuda()], x_t.cuda(), t.cuda())
    loss = training.mse_loss(pred, eps.cuda())
    loss.backward()
    assert abs(float(loss) - float(want_loss)) / float(want_loss) < 1e-5
    _check_grads(m, want)
    ran(1)


def test_fused_adamw_onecycle_matches_torch(g1):
    """5 optimisation steps: GPU (HIP fwd/bwd + fused AdamW) vs CPU (oracle autograd + torch AdamW), same OneCycleLR."""
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.scheduler import DDIMScheduler

    m = _build0(g1["config"], full=False).cuda()
    m.load_state_dict(g1["state_dict"])
    m.train()
    opt = training.FusedAdamW(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=20)
    noise_sched = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)

    cpu = {k: v.clone() for k, v in g1["state_dict"].items()}
    cpu_params = {k: torch.nn.Parameter(v.clone()) for k, v in cpu.items() if v.is_floating_point() and k not in ("mean", "std")}
    copt = torch.optim.AdamW(list(cpu_params.values()), lr=1e-3)
    csched = torch.optim.lr_scheduler.OneCycleLR(copt, max_lr=1e-3, total_steps=20)
    acp = ddim_ref.alphas_cumprod()
    g = torch.Generator().manual_seed(0)
    for it in range(5):
        x0, eps = torch.randn(2, 16, 20, generator=g), torch.randn(2, 16, 20, generator=g)
        t = torch.randint(0, 1000, (2,), generator=g)
        loss = training.train_step(m, opt, sched, noise_sched, x0.cuda(), context=[g1["ctx"].cuda()], noise=eps.cuda(),
                                   timesteps=t.cuda())
        sd_now = {**cpu, **{k: p.detach() for k, p in cpu_params.items()}}
        _, closs, grads = ref.train_loss_and_grads(sd_now, ddim_ref.add_noise(x0, eps, t, acp), t, eps, context=[g1["ctx"]])
        copt.zero_grad()
        for k, p in cpu_params.items():
            p.grad = grads.get(k, torch.zeros_like(p))
        copt.step()
        csched.step()
        assert abs(float(loss) - float(closs)) / float(closs) < 1e-4, it
        assert abs(opt.param_groups[0]["lr"] - copt.param_groups[0]["lr"]) < 1e-12
        assert opt.param_groups[0]["betas"][0] == copt.param_groups[0]["betas"][0]  # cycle_momentum
    have = dict(m.named_parameters())
    d = g1["config"]["d"]
    for k, p in cpu_params.items():
        a, b = have[k].detach().cpu(), p.detach()
        if k.endswith("in_proj_bias"):
            # the key bias has an exactly-zero true gradient (softmax is shift invariant per query), so
            # Adam's m/sqrt(v) turns pure rounding noise into O(lr) steps there, on any implementation
            a, b = torch.cat([a[:d], a[2 * d :]]), torch.cat([b[:d], b[2 * d :]])
        assert rel_err(a, b) < 1e-4, k


def test_checkpoint_dict_interchange(g1, tmp_path):
    """The checkpoint dictionary of train.py:243-250 round-trips, and torch's own AdamW /
    OneCycleLR accept the optimizer and lr-scheduler entries."""
    from soccerdiffusion_amd import training

    m = _build0(g1["config"], full=False).cuda()
    m.load_state_dict(g1["state_dict"])
    m.train()
    opt = training.FusedAdamW(m.parameters(), lr=1e-4)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=10)
    from soccerdiffusion_amd.scheduler import DDIMScheduler

    ns = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
    for _ in range(2):
        training.train_step(m, opt, sched, ns, g1["x"].cuda(), context=[g1["ctx"].cuda()])
    ckpt = {"model_state_dict": m.state_dict(), "optimizer_state_dict": opt.state_dict(),
            "lr_scheduler_state_dict": sched.state_dict(), "hyperparams": {"hidden_dim": 64, "num_joints": 20},
            "current_epoch": 0}
    path = tmp_path / "ckpt.pth"
    torch.save(ckpt, path)
    back = torch.load(path, weights_only=True)  # the reference always loads with weights_only=True
    assert set(back) == {"model_state_dict", "optimizer_state_dict", "lr_scheduler_state_dict", "hyperparams", "current_epoch"}
    # torch's AdamW on a same-shaped model accepts the optimizer state
    m2 = _build0(g1["config"], full=False)
    m2.load_state_dict(back["model_state_dict"])
    topt = torch.optim.AdamW(m2.parameters(), lr=1e-4)
    topt.load_state_dict(back["optimizer_state_dict"])
    first = next(iter(topt.state.values()))
    assert set(first) >= {"step", "exp_avg", "exp_avg_sq"} and float(first["step"]) == 2.0
    # and our optimizer resumes from it
    m3 = _build0(g1["config"], full=False).cuda()
    m3.load_state_dict(back["model_state_dict"])
    opt3 = training.FusedAdamW(m3.parameters(), lr=1e-4)
    opt3.load_state_dict(back["optimizer_state_dict"])
    assert opt3._step == 2
    assert rel_err(opt3.flat_m, opt.flat_m) < 1e-7 and rel_err(opt3.flat_v, opt.flat_v) < 1e-7
    l1 = training.train_step(m, opt, None, ns, g1["x"].cuda(), context=[g1["ctx"].cuda()], noise=g1["train"]["noise"].cuda(),
                             timesteps=g1["steps_int"].cuda())
    l3 = training.train_step(m3, opt3, None, ns, g1["x"].cuda(), context=[g1["ctx"].cuda()], noise=g1["train"]["noise"].cuda(),
                             timesteps=g1["steps_int"].cuda())
    assert abs(float(l1) - float(l3)) < 1e-6
    assert rel_err(opt3.flat_param, opt.flat_param) < 1e-6


@pytest.mark.parametrize("packed", [True, False])
def test_transposed_weight_blocks_follow_the_optimizer(g1, packed, monkeypatch):
    """The dX GEMMs read per-step copies of W^T that FusedAdamW refreshes after every update - split fp16 planes (packed) or,
    with SD_TRAIN_PACKED=0, an fp32 gather: they must equal the current weights after every step, and a torch in-place write
    to a parameter must retire them until the next step."""
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.scheduler import DDIMScheduler

    if not packed:
        monkeypatch.setenv("SD_TRAIN_PACKED", "0")
    m = _build0(g1["config"], full=False).cuda()
    m.load_state_dict(g1["state_dict"])
    m.train()
    opt = training.FusedAdamW(m.parameters(), lr=1e-3)
    assert (opt.flat_wpk is not None) == packed and (opt.flat_wt is None) == packed
    noise_sched = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
    mats = [p for p in m.parameters() if p.dim() == 2 and p.shape[1] % 64 == 0 and p.shape[0] % p.shape[1] == 0]
    assert mats
    probe = torch.randn(5, 64, generator=torch.Generator().manual_seed(7)).cuda()

    def check_all(expect_cached, only=None):
        for W in (mats if only is None else only):
            d = W.shape[1]
            for blk in range(W.shape[0] // d):
                want = W.detach()[blk * d : (blk + 1) * d].t()
                if packed:
                    wpk = training._packed_weight(W, blk, transposed=True)
                    assert (wpk is not None) == expect_cached
                    if wpk is not None:   # probe (W^T)^T = probe W_blk through the planes of the transposed block
                        from soccerdiffusion_amd import ops
                        got = ops.linear_packed(probe, wpk, d, None)
                        assert rel_err(got, probe.double().cpu() @ want.double().cpu().t()) < 1e-5
                else:
                    wt = training._transposed_block(W, blk, d)
                    assert torch.equal(wt, want)
                    cached = wt.data_ptr() >= opt.flat_wt.data_ptr() and wt.data_ptr() < opt.flat_wt.data_ptr() + 4 * opt.flat_wt.numel()
                    assert cached == expect_cached

    check_all(True)
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        x0, eps = torch.randn(2, 16, 20, generator=g), torch.randn(2, 16, 20, generator=g)
        t = torch.randint(0, 1000, (2,), generator=g)
        training.train_step(m, opt, None, noise_sched, x0.cuda(), context=[g1["ctx"].cuda()], noise=eps.cuda(), timesteps=t.cuda())
        check_all(True)
    with torch.no_grad():
        mats[0].mul_(1.5)          # outside the optimizer: the gathered blocks are stale now
    check_all(False, only=mats[:1])
    check_all(True, only=mats[1:])
    opt.step()
    check_all(True)


@pytest.mark.parametrize("fused", [False, True])
def test_distillation_step_matches_oracle(fused):
    """One iteration of the reference's distillation loop (ml/training/distill.py:165-204): the teacher's n-step DDIM rollout
    from pure noise under no_grad is the target, the student predicts it in ONE forward with the step token of t = 0 from the
    same noise and context, MSE, backward.  Target, loss and every student gradient against the oracle's rollout + autograd."""
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.synthetic import synthetic_state_dict
    from test_gpu_model import _build

    d, J, L, T, B, Mc, n_teacher = 64, 20, 2, 16, 3, 10, 10
    sd_t, sd_s = synthetic_state_dict(d, J, L, seed=21), synthetic_state_dict(d, J, L, seed=22)
    teacher = _build(dict(d=d, J=J, L=L, T=T), full=False).cuda()
    student = _build(dict(d=d, J=J, L=L, T=T), full=False).cuda()
    teacher.load_state_dict(sd_t)
    student.load_state_dict(sd_s)
    teacher.eval()
    student.train()
    student.set_dropout(0.0)
    ran = _arm_fused(student, fused)
    g = torch.Generator().manual_seed(5)
    noisy = torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, Mc, d, generator=g)
    with torch.no_grad():
        target = teacher.sample([ctx.cuda()], noisy.cuda(), n_teacher)
    pred = student.forward_with_context([ctx.cuda()], noisy.cuda(), torch.zeros(B, device="cuda"))
    loss = training.mse_loss(pred, target)
    loss.backward()
    ran(1)
    # oracle: the same loop on the CPU
    acp = ddim_ref.alphas_cumprod()
    want_target = ddim_ref.sample(lambda x, t: ref.forward_with_context(sd_t, [ctx], x, torch.full((B,), t, dtype=torch.int64)), noisy,
                                  n_teacher, acp)[-1]
    assert rel_err(target, want_target) < TOL
    _, want_loss, want = ref.train_loss_and_grads(sd_s, noisy, torch.zeros(B), want_target, context=[ctx])
    assert abs(float(loss) - float(want_loss)) / float(want_loss) < 1e-4
    _check_grads(student, want)


@pytest.mark.parametrize("T,Mc,B,L,p", [(100, 10, 3, 4, 0.0), (100, 10, 5, 2, 0.1), (10, 0, 4, 2, 0.1), (37, 15, 2, 2, 0.1), (64, 3, 2, 3, 0.0), (1, 5, 2, 1, 0.1)])
def test_trajectory_layer_forward_equals_the_row_chain_path(monkeypatch, T, Mc, B, L, p):
    """sd_train_layer_fwd (csrc/sd_train_traj.hip: one launch per decoder layer, a workgroup per trajectory) against the four launches it
    replaces (attention cores + row chains A / B; SD_TRAIN_TRAJ=0): the same prediction, loss and - through the UNCHANGED backward, which
    reads the tensors and regenerates the dropout masks the forward left behind - the same gradient of every parameter, at fp32 rounding
    level, with and without dropout, on full, ragged and single-token horizons and 1 .. 16 memory rows."""
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.synthetic import synthetic_state_dict

    d, J = 256, 20
    sd = synthetic_state_dict(d, J, L, seed=8)
    g = torch.Generator().manual_seed(T + Mc)
    x_t, eps = torch.randn(B, T, J, generator=g).cuda(), torch.randn(B, T, J, generator=g).cuda()
    ctx = [torch.randn(B, Mc, d, generator=g).cuda()] if Mc else []
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    out = {}
    for traj in ("0", "1"):
        monkeypatch.setenv("SD_TRAIN_TRAJ", traj)
        m = _build0(dict(d=d, J=J, L=L, T=T), full=False).cuda()
        m.load_state_dict(sd)
        m.train()
        m.set_dropout(p, seed=99)
        m._opt = training.FusedAdamW(m.parameters(), lr=1e-3)
        before = training.TRAJ_LAYERS[0]
        pred = m.forward_with_context(ctx, x_t, t)
        assert training.TRAJ_LAYERS[0] - before == (L if traj == "1" else 0)
        loss = training.mse_loss(pred, eps)
        loss.backward()
        out[traj] = (pred.detach().clone(), float(loss), {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None})
    assert rel_err(out["1"][0], out["0"][0]) < 2e-6
    assert abs(out["1"][1] - out["0"][1]) / out["0"][1] < 1e-6
    scale = max(float(v.norm()) for v in out["0"][2].values())
    for k, gw in out["0"][2].items():
        rel = float((out["1"][2][k].double() - gw.double()).norm()) / max(float(gw.norm()), 1e-3 * scale)
        assert rel < 2e-5, (k, rel)
