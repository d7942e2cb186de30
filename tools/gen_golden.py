#!/usr/bin/env python3
"""Generate tests/golden/*.pt by running the REFERENCE's own model modules.

Runs only in the build container (needs /root/reference); never on the GPU box.  The
reference package cannot be imported normally (its ``__init__`` wants package metadata
and a writable log dir; torchvision is absent), so parent packages are registered as
stub modules whose ``__path__`` points into the reference checkout.
Outputs are DATA ONLY: seeds, inputs, weights and the tensors the reference produced.

    python tools/gen_golden.py            # rewrites tests/golden/
"""

from __future__ import annotations

import importlib
import os
import sys
import types
import warnings

import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference/soccer_diffusion"
OUT = os.path.join(REPO, "tests", "golden")


def _stub(name, path=None, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    if path:
        m.__path__ = [path]
    sys.modules[name] = m


def import_reference():
    _stub("soccer_diffusion", REF, DB_PATH=None, LOGGING_PATH="/tmp/x.log", SESSION_ID="oracle")
    for n, p in [
        ("soccer_diffusion.ml", "/ml"),
        ("soccer_diffusion.ml.model", "/ml/model"),
        ("soccer_diffusion.ml.model.encoder", "/ml/model/encoder"),
        ("soccer_diffusion.dataset", "/dataset"),
    ]:
        _stub(n, REF + p)
    _stub("torchvision")
    _stub("torchvision.models", resnet18=None, resnet50=None, swin_s=None, swin_t=None)
    _stub("torchvision.models.resnet", ResNet18_Weights=None, ResNet50_Weights=None)
    model = importlib.import_module("soccer_diffusion.ml.model.model")
    imu = importlib.import_module("soccer_diffusion.ml.model.encoder.imu")
    image = importlib.import_module("soccer_diffusion.ml.model.encoder.image")
    return model, imu, image


def zero_dropout(model: nn.Module):
    """Training-mode parity is only defined at p=0."""
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0


def build(model_mod, imu_mod, image_mod, *, d, J, L, T, full, patch=5, ctx_len=20, enc_layers=1):
    return model_mod.End2EndDiffusionTransformer(
        num_joints=J,
        hidden_dim=d,
        use_action_history=full,
        num_action_history_encoder_layers=enc_layers,
        max_action_context_length=ctx_len,
        encoder_patch_size=patch,
        use_imu=full,
        imu_orientation_embedding_method=imu_mod.IMUEncoder.OrientationEmbeddingMethod.QUATERNION,
        num_imu_encoder_layers=enc_layers,
        imu_context_length=ctx_len,
        use_joint_states=full,
        joint_state_encoder_layers=enc_layers,
        joint_state_context_length=ctx_len,
        use_images=False,
        image_encoder_type=image_mod.ImageEncoderType.RESNET18,
        image_sequence_encoder_type=image_mod.SequenceEncoderType.TRANSFORMER,
        num_image_sequence_encoder_layers=1,
        image_context_length=0,
        image_use_final_avgpool=True,
        image_resolution=480,
        use_gamestate=full,
        num_decoder_layers=L,
        trajectory_prediction_length=T,
    )


def randomise(model: nn.Module, seed: int):
    """Non-trivial biases / LN affine so every term of the math is exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.1)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1 + (torch.rand(p.shape, generator=g) * 2 - 1) * 0.1)
        model.mean.copy_(torch.randn(model.mean.shape, generator=g))
        model.std.copy_(0.5 + torch.rand(model.std.shape, generator=g))


def gen_dataset_golden(out_dir: str = OUT) -> str:
    """G4: the reference's OWN data feed (soccer_diffusion/dataset/pytorch.py:40-414) run on a small SQLite file whose tables are
    created from the reference's own SQLAlchemy schema (dataset/models.py).  The fixture holds the database file's bytes (data), the
    constructor arguments and what the reference class returned: items at the window edges (start-up padding, last samples of a
    recording, the first of the next), a collated batch, `len`, `sample_boundaries`, the joint order and `Normalizer.fit`.
    Image path excluded (cv2 / torchvision absent: `use_images=False`, stubs are import-only); IMU in quaternion mode
    (`quats_to_5d` needs transforms3d, absent)."""
    import math
    import sqlite3
    import tempfile

    import numpy as np

    if "soccer_diffusion" not in sys.modules:
        import_reference()
    # import-only stand-ins for packages the image lacks; none of them is called with use_images=False / quaternion IMU
    _stub("cv2", resize=None, INTER_AREA=3)
    _stub("torchvision.transforms", v2=types.SimpleNamespace())
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    _stub("transforms3d")
    _stub("transforms3d.quaternions", quat2axangle=None)
    _stub("soccer_diffusion.utils", REF + "/utils")
    import logging

    sys.modules["soccer_diffusion.dataset"].logger = logging.getLogger("dataset")   # what dataset/__init__.py:27 would define
    models = importlib.import_module("soccer_diffusion.dataset.models")
    pyt = importlib.import_module("soccer_diffusion.dataset.pytorch")
    imu = importlib.import_module("soccer_diffusion.ml.model.encoder.imu")
    from sqlalchemy import create_engine

    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "g4.sqlite3")
    engine = create_engine(f"sqlite:///{path}")
    models.Base.metadata.create_all(engine)   # the reference's schema, every table
    engine.dispose()
    names = models.JointStates.get_ordered_joint_names()
    con = sqlite3.connect(path)
    cur = con.cursor()
    rng = np.random.default_rng(4)
    lengths = (140, 61)
    for rid, n in enumerate(lengths, start=1):
        cur.execute("INSERT INTO Recording (_id, allow_public, original_file, team_name, robot_type, simulated, img_width, img_height, "
                    "img_width_scaling, img_height_scaling, location) VALUES (?, 0, ?, 'team', 'wolfgang', 0, 480, 480, 1.0, 1.0, 'lab')",
                    (rid, f"rec{rid}.mcap"))
        for i in rng.permutation(n):   # inserted out of order: only ORDER BY stamp gives the sequence
            stamp = float(i) / 100.0
            for table, off in (("JointCommands", 0.0), ("JointStates", 0.37)):
                vals = [float((math.pi + math.sin(0.07 * i + 0.9 * j + off + rid)) % (2 * math.pi)) for j in range(len(names))]
                cols = ", ".join(f'"{c}"' for c in names)
                cur.execute(f"INSERT INTO {table} (stamp, recording_id, {cols}) VALUES (?, ?, {', '.join('?' * len(names))})", (stamp, rid, *vals))
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            cur.execute("INSERT INTO Rotation (stamp, recording_id, x, y, z, w) VALUES (?, ?, ?, ?, ?, ?)", (stamp, rid, *[float(v) for v in q]))
    for stamp, state in ((0.25, "POSITIONING"), (0.60, "PLAYING"), (1.10, "STOPPED")):   # recording 2 has no game state: UNKNOWN
        cur.execute("INSERT INTO GameState (stamp, recording_id, state) VALUES (?, 1, ?)", (stamp, state))
    con.commit()
    con.close()
    kw = dict(num_samples_imu=25, num_samples_joint_states=20, num_samples_joint_trajectory=30, num_samples_joint_trajectory_future=10,
              sampling_rate=100, trajectory_stride=1, num_joints=len(names), use_images=False, use_imu=True, use_joint_states=True,
              use_action_history=True, use_game_state=True)
    con = sqlite3.connect(path)
    ds = pyt.SoccerDiffusionDataset(db_connection=con, imu_representation=imu.IMUEncoder.OrientationEmbeddingMethod.QUATERNION, **kw)
    idxs = [0, 1, 9, 19, 20, 24, 25, 29, 30, 31, 60, 110, 129, 130, 131, 150, 180]
    items = {}
    for i in idxs:
        r = ds[i]
        items[i] = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in pyt.asdict(r).items() if v is not None}
    batch = pyt.SoccerDiffusionDataset.collate_fn([ds[i] for i in idxs])
    norm_idx = [3, 77, 140, 5, 129, 42]
    normalizer = pyt.Normalizer.fit(torch.cat([ds[i].joint_command for i in norm_idx], dim=0))   # train.py:106-110
    # a strided view of the same file (trajectory_stride 3), two items
    ds3 = pyt.SoccerDiffusionDataset(db_connection=con, imu_representation=imu.IMUEncoder.OrientationEmbeddingMethod.QUATERNION,
                                     **dict(kw, trajectory_stride=3, use_game_state=False))
    items3 = {i: {k: v.clone() for k, v in pyt.asdict(ds3[i]).items() if v is not None} for i in (2, 44, 50)}
    con.close()
    out = os.path.join(out_dir, "g4_dataset.pt")
    torch.save(
        {
            "sqlite_bytes": torch.frombuffer(bytearray(open(path, "rb").read()), dtype=torch.uint8).clone(),
            "kwargs": kw, "imu_representation": "quaternion",
            "joint_names": [str(n) for n in ds.joint_names], "len": len(ds), "sample_boundaries": [list(b) for b in ds.sample_boundaries],
            "idxs": idxs, "items": items,
            "batch": {k: v.clone() for k, v in pyt.asdict(batch).items() if v is not None},
            "norm_idx": norm_idx, "norm_mean": normalizer.mean.clone(), "norm_std": normalizer.std.clone(),
            "stride3_len": len(ds3), "stride3_boundaries": [list(b) for b in ds3.sample_boundaries], "stride3_items": items3,
            "robot_state_values": [str(v) for v in models.RobotState.values()],
        },
        out,
    )
    return out


def main():
    warnings.filterwarnings("ignore")
    os.makedirs(OUT, exist_ok=True)
    model_mod, imu_mod, image_mod = import_reference()
    from oracle.denoiser_ref import synthetic_state_dict

    # ---- G1: tiny decoder-only (BASELINE config 1 shape): weights = reference init -------
    torch.manual_seed(0)
    d, J, L, T, B, M = 64, 20, 2, 16, 2, 10
    m = build(model_mod, imu_mod, image_mod, d=d, J=J, L=L, T=T, full=False)
    randomise(m, 11)
    m.eval()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, M, d, generator=g)
    steps_int = torch.tensor([900, 37], dtype=torch.int64)
    steps_float = torch.tensor([0.0, 512.0])
    with torch.no_grad():
        tok_int = m.step_encoding(steps_int)
        tok_float = m.step_encoding(steps_float)
        eps_int = m.forward_with_context([ctx], x, steps_int)
        eps_float = m.forward_with_context([ctx], x, steps_float)
        mem = torch.cat([ctx, tok_int], dim=1)
        dec = m.diffusion_action_generator(x, mem)
        pe = m.diffusion_action_generator.positional_encoding.pe[0].clone()
        # short horizon through a longer PE table (misc.py:65 slices)
        eps_short = m.forward_with_context([ctx], x[:, :7], steps_int)
    # training-mode gradients at dropout p=0
    m.train()
    zero_dropout(m)
    noise = torch.randn(B, T, J, generator=g)
    m.zero_grad()
    pred = m.forward_with_context([ctx], x, steps_int)
    loss = torch.nn.functional.mse_loss(pred, noise)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    torch.save(
        {
            "config": dict(d=d, J=J, L=L, T=T, B=B, M=M),
            "state_dict": {k: v.clone() for k, v in m.state_dict().items()},
            "x": x, "ctx": ctx, "steps_int": steps_int, "steps_float": steps_float,
            "step_token_int": tok_int, "step_token_float": tok_float,
            "eps_int": eps_int, "eps_float": eps_float, "decoder_out": dec, "pe": pe,
            "eps_short": eps_short,
            "train": {"noise": noise, "pred": pred.detach(), "loss": loss.detach(), "grads": grads},
        },
        os.path.join(OUT, "g1_tiny_decoder.pt"),
    )
    print("g1", float(eps_int.abs().mean()), float(loss))

    # ---- G2: tiny full model (3 context encoders + game state) ------------------------
    torch.manual_seed(1)
    d, J, L, T, B = 64, 20, 2, 16, 3
    m = build(model_mod, imu_mod, image_mod, d=d, J=J, L=L, T=T, full=True, patch=5, ctx_len=20, enc_layers=2)
    randomise(m, 12)
    m.eval()
    g = torch.Generator().manual_seed(4321)
    inp = {
        "joint_command_history": torch.randn(B, 20, J, generator=g),
        "rotation": torch.randn(B, 20, 4, generator=g),
        "joint_state": torch.randn(B, 20, J, generator=g),
        "game_state": torch.tensor([0, 3, 2], dtype=torch.int64),
    }
    x = torch.randn(B, T, J, generator=g)
    steps = torch.tensor([980, 500, 0], dtype=torch.int64)
    with torch.no_grad():
        enc = m.encode_input_data(inp)
        eps = m(inp, x, steps)
    m.train()
    zero_dropout(m)
    noise = torch.randn(B, T, J, generator=g)
    m.zero_grad()
    pred = m(inp, x, steps)
    loss = torch.nn.functional.mse_loss(pred, noise)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    torch.save(
        {
            "config": dict(d=d, J=J, L=L, T=T, B=B, patch=5, ctx_len=20, enc_layers=2),
            "state_dict": {k: v.clone() for k, v in m.state_dict().items()},
            "input_data": inp, "x": x, "steps": steps, "encoded": enc, "eps": eps,
            "train": {"noise": noise, "pred": pred.detach(), "loss": loss.detach(), "grads": grads},
        },
        os.path.join(OUT, "g2_tiny_full.pt"),
    )
    print("g2", float(eps.abs().mean()), [tuple(e.shape) for e in enc])

    # ---- G3: BASELINE config 2/3 shape (d=256, L=4, T=100, M=11): weights by seed -------
    d, J, L, T, B, M = 256, 20, 4, 100, 2, 10
    sd = synthetic_state_dict(d, J, L, seed=7)
    m = build(model_mod, imu_mod, image_mod, d=d, J=J, L=L, T=T, full=False)
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, M, d, generator=g)
    steps = torch.tensor([980, 20], dtype=torch.int64)
    with torch.no_grad():
        eps = m.forward_with_context([ctx], x, steps)
    checksum = torch.stack([v.double().sum() for v in sd.values()]).sum()
    torch.save(
        {
            "config": dict(d=d, J=J, L=L, T=T, B=B, M=M, weight_seed=7, input_seed=1234),
            "weight_checksum": checksum, "x": x, "ctx": ctx, "steps": steps, "eps": eps,
        },
        os.path.join(OUT, "g3_c2_decoder.pt"),
    )
    print("g3", float(eps.abs().mean()), float(checksum))
    print("g4", gen_dataset_golden())


if __name__ == "__main__":
    if "--dataset-only" in sys.argv:
        warnings.filterwarnings("ignore")
        print(gen_dataset_golden())
        sys.exit(0)
    main()
