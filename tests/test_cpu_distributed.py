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
