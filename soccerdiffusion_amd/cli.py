"""`train` and `sample` entry points with the flags, YAML schema and checkpoint dictionary of
the reference's scripts (soccer_diffusion/ml/training/train.py:26-253,
soccer_diffusion/ml/inference/plot.py:21-135).

    python -m soccerdiffusion_amd.cli train -c cfg.yaml [-p ckpt] [-o out] [--decoder-pretraining] [--pretrained-decoder p]
    python -m soccerdiffusion_amd.cli sample ckpt [--steps 30] [--num_samples 10]
    python -m soccerdiffusion_amd.cli distill cfg.yaml teacher_ckpt [-o out]     (ml/training/distill.py:25-224)

Differences, all additive: data comes from the reference's SQLite database (`--db file`,
read once into HBM by soccerdiffusion_amd/dataset.py, image frames included), from a tensor
file (`--data file.pt`: dict with `joint_command` (N,T,J) and the optional context keys of
the reference's `Result` dataclass) or from a synthetic sine-wave generator (`--synthetic N`); wandb and matplotlib are not used; under torchrun
(WORLD_SIZE > 1) training is data parallel with one RCCL all-reduce of the flat gradient
per step and `sample` shards the rollouts over the ranks.
"""

from __future__ import annotations

import argparse
import logging
import math
import os
import sys
from typing import Optional

import torch
import yaml

logger = logging.getLogger("soccerdiffusion_amd")

CONTEXT_KEYS = ("joint_command_history", "rotation", "joint_state", "image_data", "game_state")


def build_model(params: dict):
    """End2EndDiffusionTransformer(**hyperparams) exactly as train.py:113-139 / plot.py:38-64 do."""
    from .ml.model import End2EndDiffusionTransformer
    from .ml.model.encoder.image import ImageEncoderType, SequenceEncoderType
    from .ml.model.encoder.imu import IMUEncoder

    return End2EndDiffusionTransformer(
        num_joints=params["num_joints"],
        hidden_dim=params["hidden_dim"],
        use_action_history=params["use_action_history"],
        num_action_history_encoder_layers=params["num_action_history_encoder_layers"],
        max_action_context_length=params["action_context_length"],
        use_imu=params["use_imu"],
        imu_orientation_embedding_method=IMUEncoder.OrientationEmbeddingMethod(params["imu_orientation_embedding_method"]),
        num_imu_encoder_layers=params["num_imu_encoder_layers"],
        imu_context_length=params["imu_context_length"],
        use_joint_states=params["use_joint_states"],
        joint_state_encoder_layers=params["joint_state_encoder_layers"],
        joint_state_context_length=params["joint_state_context_length"],
        use_images=params["use_images"],
        image_sequence_encoder_type=SequenceEncoderType(params["image_sequence_encoder_type"]),
        image_encoder_type=ImageEncoderType(params["image_encoder_type"]),
        num_image_sequence_encoder_layers=params["num_image_sequence_encoder_layers"],
        image_context_length=params["image_context_length"],
        image_use_final_avgpool=params.get("image_use_final_avgpool", True),
        image_resolution=params.get("image_resolution", 480),
        num_decoder_layers=params["num_decoder_layers"],
        trajectory_prediction_length=params["trajectory_prediction_length"],
        use_gamestate=params["use_gamestate"],
        encoder_patch_size=params["encoder_patch_size"],
    )


def synthetic_dataset(n: int, params: dict, seed: int = 0, image_size: Optional[tuple] = None) -> dict:
    """Sine-wave joints around pi (the reference stores angles in [0, 2pi)), random unit
    quaternions, random game states — shaped like the reference's `Result` fields."""
    g = torch.Generator().manual_seed(seed)
    J, T = params["num_joints"], params["trajectory_prediction_length"]
    Ha, Hi, Hj = params["action_context_length"], params["imu_context_length"], params["joint_state_context_length"]
    total = Ha + T
    phase = torch.rand(n, 1, J, generator=g) * 2 * math.pi
    freq = 0.5 + torch.rand(n, 1, J, generator=g)
    t = torch.arange(total).view(1, total, 1) / 50.0
    wave = math.pi + torch.sin(2 * math.pi * freq * t + phase)
    quat = torch.randn(n, Hi, 4, generator=g)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    feat = 5 if params["imu_orientation_embedding_method"] == "five_dim" else 4
    rot = quat if feat == 4 else torch.cat([quat[..., :3], torch.sin(quat[..., 3:]), torch.cos(quat[..., 3:])], -1)
    extra = {}
    if params.get("use_images"):  # (n, F, 3, H, W) noise frames; the reference's dataset delivers square R x R ones
        R = params.get("image_resolution", 480)
        H, W = image_size if image_size is not None else (R, R)
        extra["image_data"] = torch.rand(n, params["image_context_length"], 3, H, W, generator=g)
    return {
        **extra,
        "joint_command": wave[:, Ha:].contiguous(),
        "joint_command_history": wave[:, :Ha].contiguous(),
        "joint_state": (wave[:, Ha - Hj : Ha] + 0.01 * torch.randn(n, Hj, J, generator=g)).contiguous(),
        "rotation": rot.contiguous(),
        "game_state": torch.randint(0, 4, (n,), generator=g),
    }


class DataSource:
    """Uniform access for the loops: number of samples, device batches by index, and the
    joint commands of a few samples for the normaliser fit."""

    def __init__(self, tensors: Optional[dict] = None, dataset=None, device=None):
        self.dataset = dataset.to(device) if dataset is not None else None
        self.tensors = {k: v.to(device) for k, v in tensors.items()} if tensors is not None else None
        self.n = len(dataset) if dataset is not None else tensors["joint_command"].shape[0]

    def batch(self, idx: torch.Tensor) -> dict:
        if self.dataset is not None:
            return self.dataset.batch(idx)
        dev = self.tensors["joint_command"].device
        return {k: v[idx.to(dev)] for k, v in self.tensors.items()}


def open_database(path: str, params: dict):
    """The reference's SQLite database (dataset/models.py schema) through the pre-extracting feed."""
    from .dataset import SoccerDiffusionDataset

    return SoccerDiffusionDataset(
        db_path=path, num_joints=params["num_joints"], num_frames_video=params["image_context_length"],
        num_samples_joint_trajectory_future=params["trajectory_prediction_length"],
        num_samples_joint_trajectory=params["action_context_length"], num_samples_imu=params["imu_context_length"],
        num_samples_joint_states=params["joint_state_context_length"],
        imu_representation=params["imu_orientation_embedding_method"], use_action_history=params["use_action_history"],
        use_imu=params["use_imu"], use_joint_states=params["use_joint_states"], use_images=params["use_images"],
        use_game_state=params["use_gamestate"], image_resolution=params.get("image_resolution", 480))


def data_source(args, params: dict, device) -> DataSource:
    chosen = [n for n in ("db", "data", "synthetic") if getattr(args, n, None) is not None]
    if len(chosen) != 1:
        raise SystemExit("give exactly one data source: --db FILE (the reference's SQLite database), --data FILE.pt "
                         f"or --synthetic N (got: {', '.join('--' + c for c in chosen) or 'none'})")
    if getattr(args, "db", None):
        return DataSource(dataset=open_database(args.db, params), device=device)
    return DataSource(tensors=load_data(args, params), device=device)


def load_data(args, params: dict) -> dict:
    if args.data:
        data = torch.load(args.data, map_location="cpu", weights_only=True)
        if "joint_command" not in data:
            raise SystemExit("--data file must hold a dict with a 'joint_command' (N, T, J) tensor")
        return data
    size = None
    if getattr(args, "image_size", None):
        try:
            h, w = (int(v) for v in args.image_size.lower().split("x"))
        except ValueError:
            raise SystemExit("--image-size expects HxW, e.g. 480x640") from None
        if h != w and not params.get("image_use_final_avgpool", True):
            raise SystemExit("non-square frames need image_use_final_avgpool: True (the no-avgpool head assumes a square map, "
                             "reference encoder/image.py:69-83)")
        size = (h, w)
    return synthetic_dataset(args.synthetic, params, seed=args.seed, image_size=size)


def shard_plan(n_total: int, batch_size: int, rank: int, world: int):
    """Sample indices of this rank and the number of optimizer steps per epoch, the SAME on every rank.

    world == 1: all samples, ceil(n / bs) steps, the last batch may be short (the reference's DataLoader default,
    train.py:199-201).  world > 1: every rank owns exactly n_total // world samples (every world-th one; the
    n_total % world trailing samples are dropped) and runs floor(per_rank / bs) full batches - at least one, possibly
    short, when per_rank < bs - so that every rank issues the same number of all-reduces and OneCycleLR gets the same
    total_steps everywhere."""
    if world <= 1:
        return torch.arange(n_total), max(1, math.ceil(n_total / batch_size))
    per_rank = n_total // world
    if per_rank == 0:
        raise SystemExit(f"{n_total} samples cannot be sharded over {world} ranks")
    shard = torch.arange(rank, per_rank * world, world)
    return shard, max(1, per_rank // batch_size)


def _dist_env():
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("SD_BENCH_SHARE_GPU") == "1":   # rehearsal of the multi-rank path on a 1-GPU box (tests): all ranks on cuda:0
        local = 0
    return rank, world, local


def _init_dist(device) -> None:
    import torch.distributed as dist

    if os.environ.get("SD_BENCH_SHARE_GPU") == "1":
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=device)   # RCCL over xGMI


def cmd_train(args) -> int:
    assert args.config is not None or args.checkpoint is not None, "Either a config file or a checkpoint must be provided"
    from . import ops, training
    from .scheduler import DDIMScheduler

    rank, world, local = _dist_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        _init_dist(device)

    checkpoint = None
    params: dict = {}
    if args.checkpoint is not None:
        checkpoint = torch.load(args.checkpoint, map_location="cpu", weights_only=True)
        params = dict(checkpoint["hyperparams"])
    if args.config is not None:
        with open(args.config) as f:
            config_params = yaml.safe_load(f)
        if checkpoint is not None:  # train.py:57-70: differences are warned, then the YAML REPLACES the checkpoint's dict
            logger.warning("Both a configuration file and a checkpoint are provided. "
                           "The configuration file will be used for the hyperparameters.")
            for key, value in config_params.items():
                if key not in params:
                    logger.warning("Key '%s' is not present in the checkpoint", key)
                elif value != params[key]:
                    logger.warning("Key '%s' has a different value in the checkpoint: %r != %r", key, params[key], value)
        params = dict(config_params)
    if params["train_denoising_timesteps"] != 1000:
        raise SystemExit("train_denoising_timesteps must be 1000 (the scheduler table length; see scheduler.py)")

    gen = torch.Generator().manual_seed(args.seed + rank)
    source = data_source(args, params, device)
    n_total = source.n
    norm_idx = torch.randint(0, n_total, (params["num_normalization_samples"],), generator=torch.Generator().manual_seed(args.seed))
    from .dataset import fit_normalizer as fit_rows

    mean, std = fit_rows(source.batch(norm_idx)["joint_command"].cpu())  # train.py:108-110

    torch.manual_seed(args.seed)  # the same initial replica on every rank ...
    model = build_model(params).to(device)
    model.mean.copy_(mean)
    model.std.copy_(std)
    if checkpoint is not None:
        model.load_state_dict(checkpoint["model_state_dict"])
    if args.pretrained_decoder is not None:
        pre = torch.load(args.pretrained_decoder, map_location="cpu", weights_only=True)
        model.load_state_dict(pre["model_state_dict"], strict=False)
    model.train()
    model.set_dropout(args.dropout, seed=args.seed + 7919 * rank)   # every rank its own mask stream

    optimizer = training.FusedAdamW(model.parameters(), lr=params["lr"])
    if checkpoint is not None and "optimizer_state_dict" in checkpoint:
        optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
    if world > 1:  # ... and rank 0's parameters, moments and buffers are what every rank starts from, whatever happened above
        training.broadcast_parameters(optimizer, model)
    bs = params["batch_size"]
    shard, steps_per_epoch = shard_plan(n_total, bs, rank, world)
    lr_scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=params["lr"], total_steps=params["epochs"] * steps_per_epoch)
    scheduler = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
    scheduler.config["num_train_timesteps"] = params["train_denoising_timesteps"]

    dev_gen = torch.Generator(device=device).manual_seed(args.seed + 1000 * rank)
    for epoch in range(params["epochs"]):
        order = shard[torch.randperm(len(shard), generator=gen)]
        for i in range(steps_per_epoch):
            idx = order[i * bs : (i + 1) * bs]
            batch = source.batch(idx)
            targets = ops.normalize(batch["joint_command"].contiguous(), model.mean, model.std)
            if args.decoder_pretraining:
                ctx = [torch.randn((len(idx), 10, params["hidden_dim"]), device=device, generator=dev_gen)]
                loss = training.train_step(model, optimizer, lr_scheduler, scheduler, targets, context=ctx,
                                           world_size=world, generator=dev_gen)
            else:
                inp = {k: batch[k].contiguous() for k in CONTEXT_KEYS if k in batch}
                loss = training.train_step(model, optimizer, lr_scheduler, scheduler, targets, input_data=inp,
                                           world_size=world, generator=dev_gen)
            if i % 20 == 0 and rank == 0:
                print(f"Epoch {epoch}, it {i}, Loss: {float(loss):.05f}, LR: {lr_scheduler.get_last_lr()[0]:0.7f}", flush=True)
        training.assert_replicas_equal(optimizer)   # one pair of tiny all-reduces per epoch
        if rank == 0:
            torch.save({"model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                        "lr_scheduler_state_dict": lr_scheduler.state_dict(), "hyperparams": params,
                        "current_epoch": epoch}, args.output)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return 0


def cmd_distill(args) -> int:
    """Single-step distillation (reference ml/training/distill.py:155-221): per batch the teacher
    runs its `distill_teacher_inference_steps`-step DDIM rollout from pure noise under no_grad
    (one native call), the student predicts the sample in ONE forward at t = 0 re-using the
    teacher's context tokens, and is trained with MSE to the teacher's sample."""
    from . import training

    rank, world, local = _dist_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        _init_dist(device)
    checkpoint = torch.load(args.checkpoint, map_location="cpu", weights_only=True)
    teacher_params = checkpoint["hyperparams"]
    with open(args.config) as f:
        params = yaml.safe_load(f)
    for key, value in params.items():
        if key not in teacher_params:
            logger.warning("parameter %s in the config is not in the checkpoint's hyperparameters", key)
        elif value != teacher_params[key]:
            logger.warning("parameter %s differs from the teacher checkpoint: %r != %r", key, teacher_params[key], value)
    params["distilled_decoder"] = True  # flags the student (distill.py:62)

    teacher = build_model(params).to(device)
    teacher.load_state_dict(checkpoint["model_state_dict"])
    # distill.py:127-131 never calls teacher_model.eval(): the reference's teacher rollout (and its context encoders) run
    # with dropout 0.1 live.  --teacher-dropout reproduces that (a Python loop over the training kernels); the default
    # is the clean teacher on the native sampler.
    teacher.train(args.teacher_dropout)
    teacher.set_dropout(args.dropout if args.teacher_dropout else 0.0, seed=args.seed + 104729 + 7919 * rank)
    student = build_model(params).to(device)
    student.load_state_dict(checkpoint["model_state_dict"])
    student.train()
    student.set_dropout(args.dropout, seed=args.seed + 7919 * rank)
    # the student's context encoders never see a gradient (the context comes from the teacher under
    # no_grad), so torch's AdamW leaves them untouched; the flat optimizer therefore only owns the rest
    trainable = [p for n, p in student.named_parameters() if n.startswith(("diffusion_action_generator.", "step_encoding."))]
    optimizer = training.FusedAdamW(trainable, lr=params["lr"])
    if world > 1:
        training.broadcast_parameters(optimizer, student)

    gen = torch.Generator().manual_seed(args.seed + rank)
    source = data_source(args, params, device)
    n_total = source.n
    bs = params["batch_size"]
    shard, steps_per_epoch = shard_plan(n_total, bs, rank, world)
    lr_scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=params["lr"], total_steps=params["epochs"] * steps_per_epoch)
    dev_gen = torch.Generator(device=device).manual_seed(args.seed + 1000 * rank)
    n_teacher = params["distill_teacher_inference_steps"]
    for epoch in range(params["epochs"]):
        order = shard[torch.randperm(len(shard), generator=gen)]
        mean_loss = 0.0
        for i in range(steps_per_epoch):
            idx = order[i * bs : (i + 1) * bs]
            batch = source.batch(idx)
            noisy = torch.randn(batch["joint_command"].shape, device=device, generator=dev_gen)
            optimizer.zero_grad()
            with torch.no_grad():
                if params_need_context(params):
                    embedded = teacher.encode_input_data({k: batch[k].contiguous() for k in CONTEXT_KEYS if k in batch})
                else:
                    embedded = [torch.randn(len(idx), 10, params["hidden_dim"], device=device, generator=dev_gen)]
                target = teacher.sample(embedded, noisy, n_teacher, with_dropout=args.teacher_dropout)
            pred = student.forward_with_context(embedded, noisy, torch.zeros(len(idx), device=device))
            loss = training.mse_loss(pred, target)
            loss.backward()
            training.allreduce_gradients(optimizer, world)
            optimizer.step()
            lr_scheduler.step()
            mean_loss += float(loss)
            if i % 20 == 0 and rank == 0:
                print(f"Epoch {epoch}, it {i}, Loss: {mean_loss / (i + 1):.05f}, LR: {lr_scheduler.get_last_lr()[0]:0.7f}", flush=True)
        training.assert_replicas_equal(optimizer)
        if rank == 0:
            torch.save({"model_state_dict": student.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                        "lr_scheduler_state_dict": lr_scheduler.state_dict(), "hyperparams": params,
                        "current_epoch": epoch}, args.output)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return 0


def cmd_sample(args) -> int:
    from . import ops

    rank, world, local = _dist_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    checkpoint = torch.load(args.checkpoint, map_location="cpu", weights_only=True)
    params = checkpoint["hyperparams"]
    model = build_model(params).to(device)
    model.load_state_dict(checkpoint["model_state_dict"])
    model.eval()
    n = args.num_samples
    mine = torch.arange(rank, n, world)  # embarrassingly parallel over ranks, no collective
    source = data_source(args, params, device) if (args.data or getattr(args, "db", None) or params_need_context(params)) else None
    gen = torch.Generator(device=device).manual_seed(args.seed + rank)
    B = len(mine)
    if B == 0:
        return 0
    x_T = torch.randn(B, params["trajectory_prediction_length"], params["num_joints"], device=device, generator=gen)
    with torch.no_grad():
        if params_need_context(params):
            batch = source.batch(mine % source.n)
            inp = {k: batch[k].contiguous() for k in CONTEXT_KEYS if k in batch}
            context = model.encode_input_data(inp)
        else:
            context = [torch.randn(B, 10, params["hidden_dim"], device=device, generator=gen)]
        if params.get("distilled_decoder", False):  # single forward at t = 0 (plot.py:118-121)
            traj = model.forward_with_context(context, x_T, torch.zeros(B, device=device))
        else:
            traj = model.sample(context, x_T, args.steps)
        traj = ops.normalize(traj.contiguous(), model.mean, model.std, inverse=True)
    out = args.output if world == 1 else f"{args.output}.rank{rank}"
    torch.save({"trajectories": traj.cpu(), "noise": x_T.cpu(), "steps": args.steps, "indices": mine}, out)
    if rank == 0:
        print(f"sampled {n} trajectories of shape {tuple(traj.shape[1:])} with {args.steps} DDIM steps -> {args.output}")
    return 0


def params_need_context(params: dict) -> bool:
    return any(params.get(k, False) for k in ("use_action_history", "use_imu", "use_joint_states", "use_images", "use_gamestate"))


def main(argv: Optional[list] = None) -> int:
    logging.basicConfig(level=os.environ.get("LOGLEVEL", "INFO"))
    ap = argparse.ArgumentParser(prog="cli", description="SoccerDiffusion denoiser on MI355X: train / sample")
    sub = ap.add_subparsers(dest="command", required=True)
    tr = sub.add_parser("train", help="train the model (flags of the reference's train.py)")
    tr.add_argument("--config", "-c", type=str, default=None, help="Path to the configuration file")
    tr.add_argument("--checkpoint", "-p", type=str, default=None, help="Path to the checkpoint to load")
    tr.add_argument("--output", "-o", type=str, default="trajectory_transformer_model.pth", help="Path to save the model")
    tr.add_argument("--decoder-pretraining", action="store_true", help="Train the decoder only, on random context")
    tr.add_argument("--pretrained-decoder", type=str, default=None, help="Checkpoint whose decoder weights are loaded (strict=False)")
    tr.add_argument("--dropout", type=float, default=0.1, help="dropout probability of the transformer layers in training "
                    "(the reference never sets it: torch's default 0.1, decoder.py:26-33); 0 = the parity path")
    sa = sub.add_parser("sample", help="sample trajectories from a checkpoint (flags of the reference's plot.py)")
    sa.add_argument("checkpoint", type=str, help="Path to the checkpoint to load")
    sa.add_argument("--steps", type=int, default=30, help="Number of denoising steps")
    sa.add_argument("--num_samples", type=int, default=10, help="Number of samples to generate")
    sa.add_argument("--output", "-o", type=str, default="samples.pt", help="Where to save the sampled trajectories")
    di = sub.add_parser("distill", help="distil the multi-step model into a single-step model (flags of the reference's distill.py)")
    di.add_argument("config", type=str, help="Path to the training configuration file")
    di.add_argument("checkpoint", type=str, help="Path to the checkpoint to load for the teacher model")
    di.add_argument("--output", "-o", type=str, default="distilled_trajectory_transformer_model.pth", help="Path to save the distilled model")
    di.add_argument("--dropout", type=float, default=0.1, help="dropout probability of the student (and of the teacher with --teacher-dropout)")
    di.add_argument("--teacher-dropout", action="store_true", help="leave the teacher in train mode during its rollout, as the "
                    "reference's distill.py does (it never calls teacher_model.eval())")
    for p in (tr, sa, di):
        p.add_argument("--data", type=str, default=None, help="tensor file with joint_command (+ context keys)")
        p.add_argument("--db", type=str, default=None, help="SQLite database in the reference's schema (SOCCER_DIFFUSION_DB_PATH of the reference)")
        p.add_argument("--synthetic", type=int, default=None, metavar="N", help="train / sample on N synthetic sine-wave samples")
        p.add_argument("--image-size", type=str, default=None, metavar="HxW", help="--synthetic only: frame size when it is not the "
                       "square image_resolution of the config (BASELINE's image-conditioned case uses 480x640)")
        p.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("soccerdiffusion_amd needs an MI355X (no CPU fallback)")
    return {"train": cmd_train, "sample": cmd_sample, "distill": cmd_distill}[args.command](args)


if __name__ == "__main__":
    sys.exit(main())
