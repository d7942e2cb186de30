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


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_graphed_train_step_matches_eager(p):
    """training.GraphedTrainStep (the step replayed from a hipGraph: AdamW scalars and the dropout epoch from device
    memory, timesteps / noise from the registered generator): at p = 0 the parameters after 7 steps equal the eager loop's
    (same generator seed, same OneCycleLR); at p = 0.1 replays draw fresh masks (losses differ from step to step, the
    run stays finite and trains)."""
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.scheduler import DDIMScheduler

    B, T, J, d = 8, 100, 20, 256
    g0 = torch.Generator(device="cuda").manual_seed(5)
    x0 = torch.randn(B, T, J, device="cuda", generator=g0)
    ctx = [torch.randn(B, 10, d, device="cuda", generator=g0)]
    ns = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)

    def run(graphed, steps=7, split=None):
        m = _model(seed=1)
        m.set_dropout(p, seed=99)
        opt = training.FusedAdamW(m.parameters(), lr=1e-3)
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=20)
        gen = torch.Generator(device="cuda").manual_seed(11)
        losses = []
        if graphed:
            gs = training.GraphedTrainStep(m, opt, sch, ns, generator=gen, eager_steps=2, split_update=split)
            for _ in range(steps):
                losses.append(float(gs(x0, context=ctx)))
            gs.close()
        else:
            for _ in range(steps):
                losses.append(float(training.train_step(m, opt, sch, ns, x0, context=ctx, generator=gen)))
        return opt.flat_param.clone(), losses, opt._step, sch.get_last_lr()[0], opt.state_dict()

    pg, lg, sg, lrg, sdg = run(True)
    assert sg == 7 and all(torch.isfinite(torch.tensor(lg)))
    assert float(sdg["state"][0]["step"]) == 7.0
    if p == 0.0:
        pe, le, se, lre, _ = run(False)
        assert se == sg and abs(lre - lrg) < 1e-12
        assert max(abs(a - b) for a, b in zip(lg, le)) < 1e-5 * max(le), (lg, le)
        rel = float((pg - pe).norm() / pe.norm())
        assert rel < 1e-5, rel   # fp32 atomics of the weight-gradient GEMMs: summation order differs run to run
        # the data-parallel form (graph ends after the backward, all-reduce and update issued eagerly after each replay)
        ps, ls, ss, _, _ = run(True, split=True)
        assert ss == sg and float((ps - pe).norm() / pe.norm()) < 1e-5
    else:
        assert len({round(v, 6) for v in lg}) == len(lg)      # every replay drew new noise / masks
        assert lg[-1] < lg[0] * 1.5


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["sample", "train"])
def test_bench_two_ranks_rehearsal_on_one_gpu(mode):
    """The N-rank code path of bench.py end to end - the parent starts 2 ranks itself, rendezvous on 127.0.0.1, rank-0
    broadcast, barrier + MAX-over-ranks timing, the gradient all-reduce behind the graphed step, one JSON line from rank 0 -
    rehearsed on ONE GPU (SD_BENCH_SHARE_GPU=1: both ranks on cuda:0, gloo instead of RCCL; an 8-GPU node is the driver's to
    run).  Checks what a scaling record needs: n_gpus = 2, whole-job value = 2 x batch x steps / time, weak scaling."""
    import json
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SD_BENCH_SHARE_GPU"] = "1"
    args = ["--gpus", "2", "--mode", mode, "--steps", "2", "--warmup", "1", "--batch", "64" if mode == "sample" else "16", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    per_gpu = 64 if mode == "sample" else 16
    assert abs(d["value"] - 2 * per_gpu * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.01
    assert d["cpu_baseline"] is None
    # the line verifies its own world: size, backend, one entry per rank with the device it ran on (here both ranks on one GPU)
    w = d["world"]
    assert w["size"] == 2 and w["backend"] == "gloo" and w["shared_gpu_rehearsal"] is True
    assert [r["rank"] for r in w["ranks"]] == [0, 1] and len({r["pid"] for r in w["ranks"]}) == 2
    assert all("MI3" in r["name"] or "AMD" in r["name"] for r in w["ranks"]) and w["distinct_devices"] == 1
    if mode == "train":
        assert d["train"]["allreduce_bytes"] > 10_000_000 and d["train"]["allreduce_ms"] > 0
        t = d["train"]
        assert abs(t["allreduce_busbw_gbs"] - t["allreduce_bytes"] * 2 * (2 - 1) / 2 / (t["allreduce_ms"] * 1e-3) / 1e9) / t["allreduce_busbw_gbs"] < 0.02
        assert "all-reduce" in d["config"]["parallelism"]
        # the step the N-rank benchmark times is the one cli train runs: eager, per-layer buckets overlapped with the backward;
        # the exchange is reported alone and as what it still costs on the critical path
        assert "overlapped" in d["train"]["allreduce_form"] and isinstance(d["train"]["allreduce_exposed_ms"], float)
        assert "no multi-GPU hardware run" in d["train"]["note"]


def test_data_parallel_step_is_the_eager_bucketed_one():
    """At world_size > 1 GraphedTrainStep does not capture: every call is training.train_step with the per-layer bucketed all-reduce
    (a captured graph cannot signal a side stream in mid-replay on ROCm torch), so cli train and bench.py --mode train share ONE
    data-parallel step; split_update=True keeps the older graph + flat all-reduce form selectable."""
    from soccerdiffusion_amd import training
    from soccerdiffusion_amd.scheduler import DDIMScheduler

    m = _model(layers=1)
    opt = training.FusedAdamW(m.parameters(), lr=1e-3)
    ns = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
    dp = training.GraphedTrainStep(m, opt, None, ns, world_size=2)
    assert dp.eager_dp and not dp.split and dp.graph is None
    assert not training.GraphedTrainStep(m, opt, None, ns, world_size=2, split_update=True).eager_dp
    assert not training.GraphedTrainStep(m, opt, None, ns, world_size=1).eager_dp
