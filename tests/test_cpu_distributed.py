"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: flat gradient buffer,
one all-reduce, identical averaged gradients on every rank; and bench.py's rank sharding."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_cpu_abi_and_host import REPO  # noqa: F401  (sys.path side effect via conftest)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tiny_model():
    from soccerdiffusion_amd.ml.model import End2EndDiffusionTransformer
    from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, SequenceEncoderType
    from soccerdiffusion_amd.ml.model.encoder.imu import IMUEncoder

    torch.manual_seed(0)
    return End2EndDiffusionTransformer(
        num_joints=20, hidden_dim=64, use_action_history=False, num_action_history_encoder_layers=1,
        max_action_context_length=20, encoder_patch_size=5, use_imu=False,
        imu_orientation_embedding_method=IMUEncoder.OrientationEmbeddingMethod.QUATERNION, num_imu_encoder_layers=1,
        imu_context_length=20, use_joint_states=False, joint_state_encoder_layers=1, joint_state_context_length=20,
        use_images=False, image_encoder_type=ImageEncoderType.RESNET18, image_sequence_encoder_type=SequenceEncoderType.TRANSFORMER,
        num_image_sequence_encoder_layers=1, image_context_length=0, image_use_final_avgpool=True, image_resolution=480,
        use_gamestate=False, num_decoder_layers=2, trajectory_prediction_length=16)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from soccerdiffusion_amd import training

        model = _tiny_model()
        opt = training.FusedAdamW(model.parameters(), lr=1e-3)
        n = opt.flat_param.numel()
        assert n == sum(p.numel() for p in model.parameters())
        # parameters and gradients alias the flat buffers
        first = next(model.parameters())
        assert first.data_ptr() == opt.flat_param.data_ptr() and first.grad.data_ptr() == opt.flat_grad.data_ptr()
        # every rank holds a different local gradient; after the exchange all hold the mean
        g = torch.Generator().manual_seed(100 + rank)
        opt.flat_grad.copy_(torch.randn(n, generator=g))
        local = opt.flat_grad.clone()
        training.allreduce_gradients(opt, world)
        gathered = [torch.empty(n) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        ok = torch.allclose(opt.flat_grad, want, atol=1e-7)
        same = [torch.empty(n) for _ in range(world)]
        dist.all_gather(same, opt.flat_grad)
        ok = ok and all(torch.equal(same[0], s) for s in same)
        # the per-parameter .grad views see the averaged values
        ok = ok and torch.equal(first.grad.reshape(-1), opt.flat_grad[: first.numel()])
        # zero_grad keeps the aliasing
        opt.zero_grad()
        ok = ok and float(opt.flat_grad.abs().sum()) == 0.0 and first.grad.data_ptr() == opt.flat_grad.data_ptr()
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_gradient_allreduce_world2():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


def test_allreduce_is_noop_for_world1():
    from soccerdiffusion_amd import training

    model = _tiny_model()
    opt = training.FusedAdamW(model.parameters(), lr=1e-3)
    opt.flat_grad.fill_(3.0)
    training.allreduce_gradients(opt, 1)  # no process group needed
    assert float(opt.flat_grad[0]) == 3.0


def test_bench_flop_model_matches_survey():
    import bench

    f = bench.flops_per_traj_step()
    assert f["total"] == 478_478_336  # 478.48 MFLOP / trajectory / step
    assert abs(bench.N_DDIM * f["total"] / 1e9 - 23.92) < 0.01


def test_bench_layer_kernel_algorithmic_flops():
    """the dominant kernel's 4 launches per DDIM step own L*16Td^2 + L*4TMd + 4TJd = 425.98 MFLOP per
    trajectory (minus the first step's head, 1/50 of 6Td^2 + 2TJd, which decoder_head_kernel runs)."""
    import bench

    T, D, L, M, J = bench.T, bench.D, bench.L, bench.M, bench.J
    full = L * 16 * T * D * D + L * 4 * T * M * D + 4 * T * J * D
    assert abs(full - 425.98e6) < 0.01e6
    merged = bench.layer_kernel_algorithmic_flops_per_traj_step(True)
    assert abs(merged - (full - (6 * T * D * D + 2 * T * J * D) / bench.N_DDIM)) < 1.0
    assert bench.layer_kernel_algorithmic_flops_per_traj_step(False) == full - 6 * T * D * D - 2 * T * J * D


def _bcast_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from soccerdiffusion_amd import training

        model = _tiny_model_seeded(1000 + rank)   # ranks start from DIFFERENT replicas (the round-1 cli bug)
        model.mean.fill_(float(rank))            # a buffer the optimizer does not own
        opt = training.FusedAdamW(model.parameters(), lr=1e-3)
        opt.flat_m.fill_(float(rank))
        opt._step = 7 * rank
        before = opt.flat_param.clone()
        training.broadcast_parameters(opt, model)
        gathered = [torch.empty_like(opt.flat_param) for _ in range(world)]
        dist.all_gather(gathered, opt.flat_param)
        ok = all(torch.equal(gathered[0], g) for g in gathered)
        ok = ok and (rank == 0) == bool(torch.equal(before, opt.flat_param))   # rank 1 really changed
        ok = ok and float(model.mean[0]) == 0.0 and float(opt.flat_m[0]) == 0.0 and opt._step == 0
        # parameters still alias the flat buffer
        first = next(model.parameters())
        ok = ok and first.data_ptr() == opt.flat_param.data_ptr()
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _tiny_model_seeded(seed):
    m = _tiny_model()            # seeds with 0 ...
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():        # ... so perturb every parameter with a rank-specific stream
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.01)
    return m


@pytest.mark.timeout(120)
def test_broadcast_parameters_world2():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_bcast_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


@pytest.mark.parametrize("n_total,bs,world", [(4095, 1024, 2), (4097, 1024, 2), (2049, 1024, 2), (100, 32, 8), (7, 4, 2), (5, 8, 4)])
def test_shard_plan_is_rank_independent(n_total, bs, world):
    """ranks must agree on the number of optimizer steps (= all-reduces) per epoch and on OneCycleLR's
    total_steps whatever n_total % world is."""
    from soccerdiffusion_amd.cli import shard_plan

    plans = [shard_plan(n_total, bs, r, world) for r in range(world)]
    steps = {p[1] for p in plans}
    sizes = {len(p[0]) for p in plans}
    assert len(steps) == 1 and len(sizes) == 1
    n_steps, per_rank = steps.pop(), sizes.pop()
    assert per_rank == n_total // world and n_steps >= 1
    # every step of every rank has at least one sample, and ranks own disjoint samples
    assert (n_steps - 1) * bs < per_rank
    seen = torch.cat([p[0] for p in plans])
    assert len(set(seen.tolist())) == len(seen)


def test_shard_plan_single_rank_keeps_short_last_batch():
    from soccerdiffusion_amd.cli import shard_plan

    shard, steps = shard_plan(1000, 256, 0, 1)
    assert len(shard) == 1000 and steps == 4       # ceil: the reference's DataLoader keeps the short last batch


def _run_bench(*argv, env=None):
    import subprocess
    import sys

    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


@pytest.mark.timeout(300)
def test_bench_refuses_a_silent_single_gpu_run():
    """`bench.py --gpus N` must start N ranks or fail: never print an n_gpus=1 line for --gpus 8.
    This container has no GPU, so the launcher path must exit non-zero before touching one."""
    r = _run_bench("--gpus", "8")
    assert r.returncode != 0 and "n_gpus" not in r.stdout
    assert "--gpus 8" in r.stderr
    # a world size that disagrees with --gpus is refused as well
    r = _run_bench("--gpus", "1", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    r = _run_bench("--gpus", "4", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "n_gpus" not in r.stdout


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from soccerdiffusion_amd import training

        model = _tiny_model()
        opt = training.FusedAdamW(model.parameters(), lr=1e-3)
        n = opt.flat_param.numel()
        br = training.BucketedAllReduce(opt, model.diffusion_action_generator, world)
        # the buckets are the decoder layers, adjacent, inside the buffer, and leave a head (step token, embedding) before them
        ok = len(br.ranges) == 2 and br.ranges[0][1] == br.ranges[1][0] and br.ranges[0][0] > 0 and br.ranges[-1][1] == n
        g = torch.Generator().manual_seed(200 + rank)
        local = torch.randn(n, generator=g)
        gathered = [torch.empty(n) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        for order in ((1, 0), (0,), ()):            # backward order; a hook that never fires; none at all
            opt.flat_grad.copy_(local)
            for l in order:
                br.ready(l)
            br.ready(order[0]) if order else None   # a second call for the same layer is a no-op
            br.finish()
            ok = ok and torch.allclose(opt.flat_grad, want, atol=1e-6)
        # and it equals the single all-reduce bit for bit (one sum per element either way)
        opt.flat_grad.copy_(local)
        training.allreduce_gradients(opt, world)
        single = opt.flat_grad.clone()
        opt.flat_grad.copy_(local)
        br.ready(1); br.ready(0); br.finish()
        ok = ok and torch.equal(single, opt.flat_grad)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_allreduce_world2():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_bucket_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}
