"""End-to-end through the `cli train` / `cli sample` entry points on the GPU (BASELINE config 1 shape)."""

import math
import os
import subprocess
import sys

import pytest
import torch
import yaml

from conftest import REPO, rel_err

pytestmark = pytest.mark.gpu

CFG = dict(hidden_dim=64, action_context_length=20, trajectory_prediction_length=16, epochs=2, batch_size=32, lr=1e-3,
           train_denoising_timesteps=1000, image_context_length=0, imu_context_length=20, num_imu_encoder_layers=1,
           joint_state_context_length=20, num_normalization_samples=100, num_joints=20, use_action_history=True,
           num_action_history_encoder_layers=1, use_imu=True, imu_orientation_embedding_method="quaternion",
           use_joint_states=True, joint_state_encoder_layers=1, use_images=False, image_sequence_encoder_type="transformer",
           image_encoder_type="resnet18", num_image_sequence_encoder_layers=1, num_decoder_layers=2,
           distill_teacher_inference_steps=30, use_gamestate=True, encoder_patch_size=5)


def _run(*argv, **extra_env):
    env = dict(os.environ, PYTHONPATH=REPO, **extra_env)
    return subprocess.run([sys.executable, "-m", "soccerdiffusion_amd.cli", *argv], cwd=REPO, env=env, capture_output=True, text=True)


def test_train_then_sample_full_model(tmp_path):
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(yaml.safe_dump(CFG))
    ckpt = tmp_path / "model.pth"
    r = _run("train", "-c", str(cfg), "-o", str(ckpt), "--synthetic", "256")
    assert r.returncode == 0, r.stderr[-2000:]
    losses = [float(l.split("Loss:")[1].split(",")[0]) for l in r.stdout.splitlines() if "Loss:" in l]
    assert len(losses) >= 2 and all(math.isfinite(x) for x in losses)
    back = torch.load(ckpt, weights_only=True)
    assert set(back) == {"model_state_dict", "optimizer_state_dict", "lr_scheduler_state_dict", "hyperparams", "current_epoch"}
    assert back["hyperparams"] == CFG and back["current_epoch"] == 1
    assert "mean" in back["model_state_dict"] and not any(k.endswith(".pe") for k in back["model_state_dict"])
    # resume from the checkpoint (-p), hyper-parameters come from it
    r = _run("train", "-p", str(ckpt), "-o", str(tmp_path / "resumed.pth"), "--synthetic", "256")
    assert r.returncode == 0, r.stderr[-2000:]
    out = tmp_path / "samples.pt"
    r = _run("sample", str(ckpt), "--steps", "10", "--num_samples", "6", "-o", str(out), "--synthetic", "64")
    assert r.returncode == 0, r.stderr[-2000:]
    s = torch.load(out, weights_only=True)
    assert s["trajectories"].shape == (6, 16, 20) and torch.isfinite(s["trajectories"]).all()


def test_decoder_pretraining_loss_goes_down(tmp_path):
    cfg = dict(CFG, use_action_history=False, use_imu=False, use_joint_states=False, use_gamestate=False, epochs=12, lr=3e-3)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    r = _run("train", "-c", str(path), "-o", str(tmp_path / "dec.pth"), "--decoder-pretraining", "--synthetic", "640")
    assert r.returncode == 0, r.stderr[-2000:]
    losses = [float(l.split("Loss:")[1].split(",")[0]) for l in r.stdout.splitlines() if "Loss:" in l]
    assert losses[-1] < 0.7 * losses[0], losses
    r = _run("sample", str(tmp_path / "dec.pth"), "--steps", "10", "--num_samples", "4", "-o", str(tmp_path / "s.pt"))
    assert r.returncode == 0, r.stderr[-2000:]


def test_normalize_roundtrip():
    from soccerdiffusion_amd import ops

    x = torch.randn(5, 16, 20)
    mean, std = torch.randn(20), 0.5 + torch.rand(20)
    n = ops.normalize(x.cuda(), mean.cuda(), std.cuda())
    assert rel_err(n, (x - mean) / std) < 1e-6
    assert rel_err(ops.normalize(n, mean.cuda(), std.cuda(), inverse=True), x) < 1e-6


def test_distill_then_single_step_sample(tmp_path):
    """distill.py's flow: teacher checkpoint -> student flagged `distilled_decoder`, trained to hit the teacher's
    30-step sample in one forward; `cli sample` then takes the single-step branch (plot.py:118-121)."""
    cfg = dict(CFG, epochs=3, lr=2e-3, distill_teacher_inference_steps=10)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    teacher = tmp_path / "teacher.pth"
    r = _run("train", "-c", str(path), "-o", str(teacher), "--synthetic", "256")
    assert r.returncode == 0, r.stderr[-2000:]
    student = tmp_path / "student.pth"
    r = _run("distill", str(path), str(teacher), "-o", str(student), "--synthetic", "256")
    assert r.returncode == 0, r.stderr[-2000:]
    losses = [float(l.split("Loss:")[1].split(",")[0]) for l in r.stdout.splitlines() if "Loss:" in l]
    assert len(losses) == 3 and losses[-1] < losses[0], losses
    t, s = torch.load(teacher, weights_only=True), torch.load(student, weights_only=True)
    assert s["hyperparams"]["distilled_decoder"] is True and "distilled_decoder" not in t["hyperparams"]
    # encoders are untouched by distillation, the decoder is not
    k_enc, k_dec = "imu_encoder.embedding.weight", "diffusion_action_generator.fc_out.weight"
    assert torch.equal(t["model_state_dict"][k_enc], s["model_state_dict"][k_enc])
    assert not torch.equal(t["model_state_dict"][k_dec], s["model_state_dict"][k_dec])
    r = _run("sample", str(student), "--num_samples", "4", "-o", str(tmp_path / "s.pt"), "--synthetic", "32")
    assert r.returncode == 0, r.stderr[-2000:]
    assert torch.isfinite(torch.load(tmp_path / "s.pt", weights_only=True)["trajectories"]).all()


def test_train_and_sample_from_sqlite_database(tmp_path):
    """`--db`: the reference's SQLite schema through the pre-extracting feed, batches gathered in HBM."""
    from test_cpu_dataset import _make_db

    db = tmp_path / "db.sqlite3"
    _make_db(str(db), lengths=(400, 150)).close()
    cfg = dict(CFG, num_joints=22, epochs=2, batch_size=64)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    ckpt = tmp_path / "m.pth"
    r = _run("train", "-c", str(path), "-o", str(ckpt), "--db", str(db))
    assert r.returncode == 0, r.stderr[-2000:]
    losses = [float(l.split("Loss:")[1].split(",")[0]) for l in r.stdout.splitlines() if "Loss:" in l]
    assert losses and all(math.isfinite(x) for x in losses)
    sd = torch.load(ckpt, weights_only=True)["model_state_dict"]
    assert sd["mean"].shape == (22,) and float(sd["mean"].mean()) > 2.0  # joint angles live around pi
    r = _run("sample", str(ckpt), "--steps", "10", "--num_samples", "5", "-o", str(tmp_path / "s.pt"), "--db", str(db))
    assert r.returncode == 0, r.stderr[-2000:]
    out = torch.load(tmp_path / "s.pt", weights_only=True)["trajectories"]
    assert out.shape == (5, 16, 22) and torch.isfinite(out).all()


def test_train_with_image_context_from_sqlite_database(tmp_path):
    """BASELINE config 5 in miniature: frames from the Image table -> ResNet-18 token per frame -> 8-head HIP
    sequence encoder -> memory of the denoiser; one epoch of `cli train --db` (image_resolution 120, hidden 128)."""
    from test_cpu_dataset import _make_db

    db = tmp_path / "db.sqlite3"
    _make_db(str(db), lengths=(120, 60)).close()
    cfg = dict(CFG, hidden_dim=128, num_joints=22, epochs=1, batch_size=16, use_images=True, image_context_length=2,
               image_resolution=120, image_use_final_avgpool=True, use_imu=False, use_joint_states=False, num_decoder_layers=1)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    ckpt = tmp_path / "m.pth"
    r = _run("train", "-c", str(path), "-o", str(ckpt), "--db", str(db))
    assert r.returncode == 0, r.stderr[-2000:]
    losses = [float(l.split("Loss:")[1].split(",")[0]) for l in r.stdout.splitlines() if "Loss:" in l]
    assert losses and all(math.isfinite(x) for x in losses)
    sd = torch.load(ckpt, weights_only=True)["model_state_dict"]
    assert "image_sequence_encoder.image_encoder.encoder.conv1.weight" in sd


@pytest.mark.timeout(900)
def test_two_rank_train_and_distill_rehearsal_on_one_gpu(tmp_path):
    """`cli train` / `cli distill` under torchrun with 2 ranks (replicas used to start from different weights and
    could disagree on the number of steps): rank-0 broadcast, equal shards with a common step count (257 samples, batch 32),
    the per-layer all-reduce hooked into the backward, dropout with per-rank masks - and the invariant
    `training.assert_replicas_equal` checked at the end of every epoch (it raises when the parameter checksums differ).
    Rehearsed on ONE GPU: SD_BENCH_SHARE_GPU=1 puts both ranks on cuda:0 and uses gloo instead of RCCL."""
    import socket

    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(yaml.safe_dump(dict(CFG, epochs=2)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=REPO, SD_BENCH_SHARE_GPU="1")

    def torchrun(*argv):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", "soccerdiffusion_amd.cli", *argv]
        return subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=420)

    ckpt = tmp_path / "dp.pth"
    r = torchrun("train", "-c", str(cfg), "-o", str(ckpt), "--synthetic", "257")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "diverged" not in r.stderr
    back = torch.load(ckpt, weights_only=True)
    assert back["current_epoch"] == 1
    # 257 samples / 2 ranks = 128 each -> 4 steps per epoch on BOTH ranks: OneCycleLR was built for 8 steps
    assert back["lr_scheduler_state_dict"]["total_steps"] == 8
    assert all(torch.isfinite(v).all() for v in back["model_state_dict"].values() if v.is_floating_point())
    r = torchrun("distill", str(cfg), str(ckpt), "-o", str(tmp_path / "student.pth"), "--synthetic", "130")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert torch.load(tmp_path / "student.pth", weights_only=True)["hyperparams"]["distilled_decoder"] is True


@pytest.mark.timeout(900)
def test_train_image_conditioned_non_square_frames(tmp_path):
    """BASELINE configs[4] in miniature through `cli train`: NON-square frames (96 x 128; the benchmark's are
    480 x 640) with image_use_final_avgpool True - the reference's no-avgpool head assumes square frames
    (ml/model/encoder/image.py:69-83) - ResNet-18 per frame -> token -> 8-head HIP sequence encoder -> denoiser d = 256, B = 4."""
    cfg = dict(CFG, hidden_dim=256, epochs=2, batch_size=4, use_images=True, image_context_length=3, image_resolution=96,
               image_use_final_avgpool=True, use_imu=False, use_joint_states=False, use_action_history=False, num_decoder_layers=2,
               trajectory_prediction_length=100)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    ckpt = tmp_path / "m.pth"
    r = _run("train", "-c", str(path), "-o", str(ckpt), "--synthetic", "12", "--image-size", "96x128")
    assert r.returncode == 0, r.stderr[-2000:]
    losses = [float(l.split("Loss:")[1].split(",")[0]) for l in r.stdout.splitlines() if "Loss:" in l]
    assert len(losses) >= 2 and all(math.isfinite(x) for x in losses)   # one line per epoch (every 20th iteration)
    sd = torch.load(ckpt, weights_only=True)["model_state_dict"]
    assert sd["image_sequence_encoder.image_encoder.encoder.conv1.weight"].shape == (64, 3, 7, 7)


@pytest.mark.timeout(1500)
def test_configs4_per_gpu_share_at_full_frame_size(tmp_path):
    """BASELINE configs[4]'s per-GPU share AT SIZE (VERDICT r3 #1c): B = 128 over 8 GPUs = 16 trajectories x 10 frames of 480 x 640 RGB,
    ResNet-18 per frame (torch.nn / MIOpen: parity unpinned vs torchvision) -> token -> 8-head HIP sequence encoder -> denoiser d = 256,
    L = 4, T = 100, J = 20.  (a) through `cli train` (reference loop ml/training/train.py:204-240): finite, decreasing-or-equal loss
    lines and a checkpoint that holds the backbone; (b) the hand-written half - sequence encoder + step token + denoiser - against the
    oracle GIVEN the backbone's tokens (reference wiring ml/model/model.py:159-179, ml/model/encoder/image.py:38-52, 107-128), 1e-4."""
    from oracle import denoiser_ref as ref

    import bench
    from soccerdiffusion_amd import cli

    # (a)
    cfg = dict(CFG, hidden_dim=256, epochs=2, batch_size=16, use_images=True, image_context_length=10, image_resolution=480,
               image_use_final_avgpool=True, use_imu=False, use_joint_states=False, use_action_history=False, use_gamestate=False,
               num_decoder_layers=4, trajectory_prediction_length=100, num_image_sequence_encoder_layers=2, lr=1e-4)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    ckpt = tmp_path / "c5.pth"
    r = _run("train", "-c", str(path), "-o", str(ckpt), "--synthetic", "16", "--image-size", "480x640", MIOPEN_FIND_MODE="FAST")
    assert r.returncode == 0, r.stderr[-2000:]
    losses = [float(l.split("Loss:")[1].split(",")[0]) for l in r.stdout.splitlines() if "Loss:" in l]
    assert len(losses) >= 2 and all(math.isfinite(x) and 0.0 < x < 10.0 for x in losses), losses
    back = torch.load(ckpt, weights_only=True)
    sd = back["model_state_dict"]
    assert sd["image_sequence_encoder.image_encoder.encoder.conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["image_sequence_encoder.image_encoder.encoder.fc.weight"].shape[0] == 256
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
    # (b) the trained weights, eval mode, fresh frames
    # (no MIOPEN_FIND_MODE in THIS process: it would change the solvers the later backbone tests get)
    dev = torch.device("cuda:0")
    params = dict(bench.C2_PARAMS, **{k: cfg[k] for k in ("use_images", "image_context_length", "image_resolution", "image_use_final_avgpool",
                                                          "num_image_sequence_encoder_layers")})
    model = cli.build_model(dict(back["hyperparams"])).to(dev).eval()
    model.load_state_dict(sd)
    assert params["image_context_length"] == 10
    B, F, T_, J_ = 16, 10, 100, 20
    g = torch.Generator().manual_seed(11)
    frames = torch.rand(B, F, 3, 480, 640, generator=g).to(dev)
    x = torch.randn(B, T_, J_, generator=g)
    step = torch.randint(0, 1000, (B,), generator=g)
    with torch.no_grad():
        tokens = model.image_sequence_encoder.image_encoder(frames)          # the backbone (library convolutions)
        got = model({"image_data": frames}, x.to(dev), step.to(dev))
        ctx = model.encode_input_data({"image_data": frames})
    assert tokens.shape == (B, F, 256) and [tuple(c.shape) for c in ctx] == [(B, F, 256)]
    cpu_sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    seq_sd = {k[len("image_sequence_encoder.transformer_encoder."):]: v for k, v in cpu_sd.items()
              if k.startswith("image_sequence_encoder.transformer_encoder.")}
    # BaseEncoder(input_dim = d, patch 1, 8 heads) on the tokens: k = 1 embedding, PE, 2 x (self-attention + FFN) (image.py:107-128)
    assert "transformer_encoder.layers.1.norm1.weight" in seq_sd and not any(".layers.2." in k for k in seq_sd)
    h = ref.encoder_forward(seq_sd, tokens.cpu(), "", heads=8)
    assert rel_err(ctx[0], h) < 1e-4
    want = ref.forward_with_context(cpu_sd, [h], x, step)
    assert rel_err(got, want) < 1e-4
