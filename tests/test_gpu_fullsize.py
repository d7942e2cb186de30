"""BASELINE configs[2] at its full size (B = 4096, 50 steps): parity of individual trajectories with the oracle, batch
invariance, the hipGraph-captured rollout, and the range guard of the split-fp16 kernels."""

import warnings

import pytest
import torch

from conftest import rel_err
from oracle import ddim_ref
from oracle import denoiser_ref as ref

pytestmark = pytest.mark.gpu
D, L, T, J, MC, N = 256, 4, 100, 20, 10, 50


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from soccerdiffusion_amd import ops as o

    return o


def _setup(ops, sd, n_steps, d=D):
    acp = ddim_ref.alphas_cumprod()
    ts = ddim_ref.timesteps(n_steps).tolist()
    toks = ops.step_token(torch.tensor(ts).cuda(), ops.step_frequencies(d).cuda(), sd["step_encoding.token"].cuda()).reshape(n_steps, d)
    return acp, ts, toks, ops.ddim_coefficients(ts, acp, n_steps)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("mode", [4, 3, 2])
def test_b4096_trajectories_match_oracle_and_small_batch(ops, mode):
    """The bench shape itself: 6 400 panels through the XCD relabelling, the 7-GB workspace carve and the fold gate.
    Trajectory i of the 4096-batch must equal (a) the fp32 CPU oracle run on that trajectory alone - after EVERY
    reported step, 1e-4 of north_star - and (b) the same trajectory sampled in a batch of 10 (different panel
    alignment, different abs-max scales of the folded blocks: fp32 rounding level, not bitwise).
    Picks: first / last panel, trajectories whose panels straddle a neighbour (every one but 0 mod 16), the last
    trajectory of XCD 0's block and the first of XCD 1's (panels 800 x), one inside every other XCD's block."""
    from soccerdiffusion_amd import _lib

    B = 4096
    assert _lib.load().sd_sampler_mode(D, 4, T, MC, J) == 3   # modes 3 / 4 = one workgroup per trajectory (4096 workgroups); 2 = 6 400 panels
    sd = ref.synthetic_state_dict(D, J, L, seed=7)
    acp, ts, toks, coef = _setup(ops, sd, N)
    x_T = torch.randn(B, T, J, generator=torch.Generator().manual_seed(1234))
    ctx = torch.randn(B, MC, D, generator=torch.Generator().manual_seed(1235))
    packed = ops.pack_denoiser(sd, "cuda", max_len=T)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    got = ops.ddim_sample(packed, ctx.cuda(), toks, coef, x_T.cuda(), status=status, max_mode=mode)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert torch.isfinite(got).all()
    picks = [0, 1, 511, 512, 1100, 1700, 2300, 2900, 3500, 4095]
    xs, cs = x_T[picks].contiguous(), ctx[picks].contiguous()
    n = len(picks)
    want = ddim_ref.sample(lambda x, t: ref.forward_with_context(sd, [cs], x, torch.full((n,), t, dtype=torch.int64)), xs, N, acp)[-1]
    errs = [rel_err(got[b], want[i]) for i, b in enumerate(picks)]
    assert all(e < 1e-4 for e in errs), errs
    small = ops.ddim_sample(packed, cs.cuda(), toks, coef, xs.cuda(), max_mode=mode, status=status)
    inv = [rel_err(got[b], small[i]) for i, b in enumerate(picks)]
    assert all(e < 1e-5 for e in inv), inv   # the same kernels on the same trajectory: batch-size independent
    # nothing else in the batch is degenerate: per-trajectory norms are in a sane band
    norms = got.flatten(1).norm(dim=1)
    assert float(norms.min()) > 0.0 and float(norms.max()) < 1e4


def test_wide_memory_rollout_at_batch_size(ops):
    """17 .. 64 memory rows at a batch that fills the chip four times (B = 1024, horizon 100, 51 memory rows = the reference's
    sim_scratch.yaml context, 4 key tiles): the folded blocks of (trajectory, key tile) pairs are addressed per workgroup - picked
    trajectories against the fp32 CPU oracle on that trajectory alone and against the same trajectories sampled in a batch of 6."""
    B, Mc, n_steps, L_ = 1024, 50, 10, 2
    sd = ref.synthetic_state_dict(D, J, L_, seed=17)
    acp, ts, toks, coef = _setup(ops, sd, n_steps)
    x_T = torch.randn(B, T, J, generator=torch.Generator().manual_seed(21))
    ctx = torch.randn(B, Mc, D, generator=torch.Generator().manual_seed(22))
    packed = ops.pack_denoiser(sd, "cuda", max_len=T)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    got = ops.ddim_sample(packed, ctx.cuda(), toks, coef, x_T.cuda(), status=status)
    assert int(status.item()) == 0 and torch.isfinite(got).all()
    picks = [0, 1, 255, 256, 700, 1023]
    xs, cs = x_T[picks].contiguous(), ctx[picks].contiguous()
    n = len(picks)
    want = ddim_ref.sample(lambda x, t: ref.forward_with_context(sd, [cs], x, torch.full((n,), t, dtype=torch.int64)), xs, n_steps, acp)[-1]
    errs = [rel_err(got[b], want[i]) for i, b in enumerate(picks)]
    assert all(e < 1e-5 for e in errs), errs
    small = ops.ddim_sample(packed, cs.cuda(), toks, coef, xs.cuda())
    assert all(rel_err(got[b], small[i]) < 1e-5 for i, b in enumerate(picks))


@pytest.mark.parametrize("d,L_,T_,Mc,B,n_steps", [(256, 2, 100, 10, 37, 8), (256, 2, 10, 31, 3, 6), (64, 2, 16, 10, 2, 10)])
def test_graphed_sampler_replay_is_bit_identical_to_eager(ops, d, L_, T_, Mc, B, n_steps):
    """ops.GraphedSampler (hipGraph-captured rollout, BASELINE configs[2]): replays equal the eager call bit for bit,
    for two different inputs through the SAME captured graph, in every sampler mode (2, 0 with fp16 chains, 0 fp32)."""
    sd = ref.synthetic_state_dict(d, J, L_, seed=11)
    _, _, toks, coef = _setup(ops, sd, n_steps, d)
    packed = ops.pack_denoiser(sd, "cuda", max_len=T_)
    gs = ops.GraphedSampler(packed, B, T_, Mc, toks, coef)
    for seed in (1, 2):
        g = torch.Generator().manual_seed(seed)
        x_T = torch.randn(B, T_, J, generator=g).cuda()
        ctx = torch.randn(B, Mc, d, generator=g).cuda()
        eager = ops.ddim_sample(packed, ctx, toks, coef, x_T)
        replay = gs(ctx, x_T)
        torch.cuda.synchronize()
        assert torch.equal(eager, replay)
        assert int(gs.status.item()) == 0
    # and the result is the oracle's
    acp = ddim_ref.alphas_cumprod()
    want = ddim_ref.sample(lambda x, t: ref.forward_with_context(sd, [ctx.cpu()], x, torch.full((B,), t, dtype=torch.int64)),
                           x_T.cpu(), n_steps, acp)[-1]
    assert rel_err(replay, want) < 1e-4


def test_model_sample_graph_path_matches_eager(ops):
    """End2EndDiffusionTransformer.sample(use_graph=True) == sample() (the module-level route to the captured rollout)."""
    from soccerdiffusion_amd import cli

    import bench

    p = dict(bench.C2_PARAMS, num_decoder_layers=2)
    torch.manual_seed(3)
    m = cli.build_model(p).cuda().eval()
    x = torch.randn(5, T, J, device="cuda")
    ctx = [torch.randn(5, MC, D, device="cuda")]
    a = m.sample(ctx, x, 6)
    b = m.sample(ctx, x, 6, use_graph=True)
    c = m.sample(ctx, x, 6, use_graph=True)
    assert torch.equal(a, b) and torch.equal(b, c)


def test_range_guard_flags_overflow_and_falls_back_to_fp32(ops):
    """the split-fp16 kernels use a fixed activation scale of 8, so a LayerNorm weight in the thousands
    drives fp16 hi parts to infinity.  That must never come back as silent garbage: sd_ddim_sample_ex sets the status
    word, and ops.ddim_sample_guarded (what model.sample calls) repeats the rollout on the fp32-MFMA kernels, whose
    result holds the usual 1e-4 against the oracle."""
    d, L_, B, n_steps = 256, 2, 3, 4
    sd = ref.synthetic_state_dict(d, J, L_, seed=13)
    key = "diffusion_action_generator.transformer_decoder.layers.0.norm3.weight"
    sd[key] = sd[key] * 4000.0
    acp, ts, toks, coef = _setup(ops, sd, n_steps, d)
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, MC, d, generator=g)
    packed = ops.pack_denoiser(sd, "cuda", max_len=T)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    raw = ops.ddim_sample(packed, ctx.cuda(), toks, coef, x_T.cuda(), status=status)
    torch.cuda.synchronize()
    assert int(status.item()) == ops.STATUS_NONFINITE
    assert not torch.isfinite(raw).all()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = ops.ddim_sample_guarded(packed, ctx.cuda(), toks, coef, x_T.cuda())
    assert any(issubclass(x.category, RuntimeWarning) for x in w)
    assert torch.isfinite(got).all()
    want = ddim_ref.sample(lambda x, t: ref.forward_with_context(sd, [ctx], x, torch.full((B,), t, dtype=torch.int64)), x_T, n_steps, acp)[-1]
    assert rel_err(got, want) < 1e-4
    # the capped call alone is clean, and an in-range model never trips the guard
    ok = ops.ddim_sample(packed, ctx.cuda(), toks, coef, x_T.cuda(), status=status, max_mode=1)
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and torch.equal(ok, got)
    # non-finite INPUT is reported as such, not retried forever
    bad = x_T.clone()
    bad[0, 0, 0] = float("nan")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(FloatingPointError):
            ops.ddim_sample_guarded(packed, ctx.cuda(), toks, coef, bad.cuda())


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_c2_training_step_at_full_batch_fused_equals_per_operation_path(ops, monkeypatch, p):
    """BASELINE configs[1] at its size (B = 256: 400 panels on 256 CUs, ~500 workgroups in the grouped weight-gradient launch,
    the XCD relabelling, abs-max words spread over 64 slots): every parameter gradient of the fused row chains equals the
    per-operation autograd nodes' (the path the reference's golden gradients pin at small sizes) - same weights, same batch,
    same dropout masks - and a few trajectories equal the oracle's forward."""
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.synthetic import synthetic_state_dict
    from test_gpu_model import _build

    B, Mc = 256, 10
    sd = synthetic_state_dict(D, J, L, seed=5)
    m = _build(dict(d=D, J=J, L=L, T=T), full=False).cuda()
    m.load_state_dict(sd)
    m.train()
    m.set_dropout(p, seed=99)
    opt = training.FusedAdamW(m.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(11)
    x_t, eps = torch.randn(B, T, J, generator=g).cuda(), torch.randn(B, T, J, generator=g).cuda()
    ctx = torch.randn(B, Mc, D, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    gen = m.diffusion_action_generator
    grads, preds = {}, {}
    for mode in ("fused", "nodes"):
        monkeypatch.setenv("SD_TRAIN_FUSED", "1" if mode == "fused" else "0")
        gen.dropout.calls = 7          # the same call index: the same masks
        opt.zero_grad()
        before = training.FUSED_STACKS[0]
        pred = m.forward_with_context([ctx], x_t, t)
        assert training.FUSED_STACKS[0] - before == (1 if mode == "fused" else 0)
        training.mse_loss(pred, eps).backward()
        torch.cuda.synchronize()
        grads[mode] = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}
        preds[mode] = pred.detach()
    assert rel_err(preds["fused"], preds["nodes"]) < 2e-6
    scale = max(float(v.norm()) for v in grads["nodes"].values())
    for k, gn in grads["nodes"].items():
        rel = float((grads["fused"][k].double() - gn.double()).norm()) / max(float(gn.norm()), 1e-3 * scale)
        assert rel < 2e-5, (k, rel)
    if p == 0.0:
        picks = [0, 100, 255]
        want = ref.forward_with_context(sd, [ctx[picks].cpu()], x_t[picks].cpu(), t[picks].cpu())
        assert rel_err(preds["fused"][picks], want) < 1e-4
