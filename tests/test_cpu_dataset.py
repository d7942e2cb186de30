"""The pre-extracting data feed against the reference's per-item SQL semantics
(soccer_diffusion/dataset/pytorch.py:128-384), restated here with the same queries
(LIMIT/OFFSET windows ordered by stamp, front padding, last game state <= stamp) on a
SQLite file that uses the reference's table and column names (dataset/models.py)."""

import math
import sqlite3

import numpy as np
import pytest
import torch

from soccerdiffusion_amd.dataset import JOINT_NAMES_22, ROBOT_STATES, SoccerDiffusionDataset, fit_normalizer, quats_to_5d


def _make_db(path, lengths=(180, 75)):
    con = sqlite3.connect(path)
    cur = con.cursor()
    jcols = ", ".join(f'"{n}" FLOAT' for n in JOINT_NAMES_22)
    cur.execute("CREATE TABLE Recording (_id INTEGER PRIMARY KEY, team_name TEXT, start_time TEXT, location TEXT, original_file TEXT)")
    for t in ("JointCommands", "JointStates"):
        cur.execute(f"CREATE TABLE {t} (_id INTEGER PRIMARY KEY AUTOINCREMENT, stamp FLOAT, recording_id INTEGER, {jcols})")
    cur.execute("CREATE TABLE Rotation (_id INTEGER PRIMARY KEY AUTOINCREMENT, stamp FLOAT, recording_id INTEGER, x FLOAT, y FLOAT, z FLOAT, w FLOAT)")
    cur.execute("CREATE TABLE GameState (_id INTEGER PRIMARY KEY AUTOINCREMENT, stamp FLOAT, recording_id INTEGER, state TEXT)")
    cur.execute("CREATE TABLE Image (_id INTEGER PRIMARY KEY AUTOINCREMENT, stamp FLOAT, recording_id INTEGER, data BLOB)")
    rng = np.random.default_rng(0)
    for rid, n_img in ((1, 9), (2, 2)):   # ~4 fps, inserted out of order; recording 2 has fewer frames than the context asks for
        for j in rng.permutation(n_img):
            frame = rng.integers(0, 256, size=(480, 480, 3), dtype=np.uint8)
            cur.execute("INSERT INTO Image (stamp, recording_id, data) VALUES (?, ?, ?)", (0.3 + 0.27 * j, rid, frame.tobytes()))
    for rid, n in enumerate(lengths, start=1):
        cur.execute("INSERT INTO Recording VALUES (?, 'team', '2024', 'lab', 'f.mcap')", (rid,))
        order = rng.permutation(n)  # rows are inserted out of order: only ORDER BY stamp gives the sequence
        for i in order:
            stamp = i / 50.0
            for t, off in (("JointCommands", 0.0), ("JointStates", 0.5)):
                vals = [(math.pi + math.sin(0.1 * i + j + off)) % (2 * math.pi) for j in range(22)]
                cur.execute(f"INSERT INTO {t} (stamp, recording_id, {', '.join(chr(34) + n + chr(34) for n in JOINT_NAMES_22)}) VALUES (?, ?, {', '.join('?' * 22)})",
                            (stamp, rid, *vals))
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            cur.execute("INSERT INTO Rotation (stamp, recording_id, x, y, z, w) VALUES (?, ?, ?, ?, ?, ?)", (stamp, rid, *q))
        for stamp, state in ((0.4, "POSITIONING"), (1.0, "PLAYING"), (2.5, "STOPPED")):
            if rid == 1:
                cur.execute("INSERT INTO GameState (stamp, recording_id, state) VALUES (?, ?, ?)", (stamp, rid, state))
    con.commit()
    return con


def _reference_item(con, names, idx, boundaries, T, H, Hs, Hi, stride=1, rate=50):
    """The reference's __getitem__ restated query by query."""
    for start, end, rid in boundaries:
        if start <= idx < end:
            break
    i = (idx - start) * stride
    cols = ", ".join(f'"{n}"' for n in names)

    def q(table, c, off, num):
        rows = con.execute(f"SELECT {c} FROM {table} WHERE recording_id = {rid} ORDER BY stamp ASC LIMIT {num} OFFSET {off}").fetchall()
        return np.asarray(rows, dtype=np.float32).reshape(len(rows), len(c.split(",")))

    def hist(table, c, n, pad=None):
        s = max(0, i - n)
        rows = q(table, c, s, i - s)
        if rows.shape[0] < n:
            fill = np.zeros((n - rows.shape[0], rows.shape[1] if rows.size else len(c.split(","))), np.float32)
            if pad is not None:
                fill[:] = pad
            rows = np.concatenate([fill, rows.reshape(-1, fill.shape[1])], 0)
        return rows

    gs = con.execute("SELECT state FROM GameState WHERE recording_id = ? AND stamp <= ? ORDER BY stamp DESC LIMIT 1", (rid, i / rate)).fetchone()
    return dict(joint_command=q("JointCommands", cols, i, T), joint_command_history=hist("JointCommands", cols, H),
                joint_state=hist("JointStates", cols, Hs), rotation=hist("Rotation", "x, y, z, w", Hi, pad=[0, 0, 0, 1]),
                game_state=ROBOT_STATES.index(gs[0]) if gs else ROBOT_STATES.index("UNKNOWN"))


@pytest.fixture(scope="module")
def db(tmp_path_factory):
    return _make_db(str(tmp_path_factory.mktemp("db") / "db.sqlite3"))


def test_items_and_batches_match_reference_queries(db):
    T, H, Hs, Hi = 10, 30, 20, 25
    ds = SoccerDiffusionDataset(db, num_samples_imu=Hi, num_samples_joint_states=Hs, num_samples_joint_trajectory=H,
                                num_samples_joint_trajectory_future=T, sampling_rate=50, num_joints=22, use_images=False)
    assert ds.joint_names == JOINT_NAMES_22
    assert len(ds) == (180 - T) + (75 - T) and ds.sample_boundaries == [(0, 170, 1), (170, 235, 2)]
    idxs = [0, 1, 5, 19, 20, 29, 30, 100, 169, 170, 171, 200, 234]  # start-up padding, both recordings, last samples
    for idx in idxs:
        want = _reference_item(db, JOINT_NAMES_22, idx, ds.sample_boundaries, T, H, Hs, Hi)
        got = ds[idx]
        for k in ("joint_command", "joint_command_history", "joint_state", "rotation"):
            assert torch.equal(getattr(got, k), torch.tensor(want[k])), (idx, k)
        assert int(got.game_state) == want["game_state"], idx
    b = ds.batch(torch.tensor(idxs))
    c = SoccerDiffusionDataset.collate_fn([ds[i] for i in idxs])
    for k in ("joint_command", "joint_command_history", "joint_state", "rotation", "game_state"):
        assert torch.equal(b[k], getattr(c, k)), k
    assert b["joint_command"].shape == (len(idxs), T, 22) and b["game_state"].dtype == torch.int64
    assert torch.equal(ds.tensors()["joint_command"][7], ds[7].joint_command)


def test_stride_twenty_joints_and_five_dim(db):
    ds = SoccerDiffusionDataset(db, num_samples_imu=8, imu_representation="five_dim", num_samples_joint_states=8,
                                num_samples_joint_trajectory=8, num_samples_joint_trajectory_future=4, sampling_rate=50,
                                trajectory_stride=3, num_joints=20, use_images=False, use_game_state=False)
    assert len(ds.joint_names) == 20 and "LElbowYaw" not in ds.joint_names
    assert len(ds) == int((180 - 4) / 3) + int((75 - 4) / 3)
    item = ds[2]  # joint-command index 6: two rows of identity-quaternion padding in front
    assert item.joint_command.shape == (4, 20) and item.rotation.shape == (8, 5) and item.game_state is None
    ident = torch.tensor(quats_to_5d(np.array([[0.0, 0.0, 0.0, 1.0]]))[0]).float()
    assert torch.equal(item.rotation[0], ident) and torch.equal(item.rotation[1], ident)
    assert torch.allclose(ident, torch.tensor([1.0, 0.0, 0.0, 0.0, 1.0]))
    want = _reference_item(db, ds.joint_names, 2, ds.sample_boundaries, 4, 8, 8, 8, stride=3)
    assert torch.equal(item.joint_command, torch.tensor(want["joint_command"]))
    assert torch.allclose(item.rotation[2:], torch.tensor(quats_to_5d(want["rotation"][2:])).float())


def test_quats_to_5d_axis_angle():
    ang = 1.3
    axis = np.array([0.0, 0.6, 0.8])
    q = np.concatenate([axis * math.sin(ang / 2), [math.cos(ang / 2)]])[None]
    out = quats_to_5d(q)[0]
    assert np.allclose(out[:3], axis) and np.allclose(out[3:], [math.sin(ang), math.cos(ang)])


def test_fit_normalizer_is_unbiased_std():
    x = torch.randn(7, 5, 3)
    mean, std = fit_normalizer(x)
    rows = x.reshape(-1, 3)
    assert torch.allclose(mean, rows.mean(0)) and torch.allclose(std, rows.std(0, unbiased=True))


def _reference_images(con, rid, stamp, F, fps, R):
    """query_image_data (dataset/pytorch.py:173-229) restated: the SQL as it is, the frames' preprocessing in numpy
    (cv2 and torchvision are absent: INTER_AREA for an integer factor is the rounded block mean)."""
    ctx = (F + 1) / fps
    rows = con.execute("SELECT stamp, data FROM Image WHERE recording_id = ? AND stamp BETWEEN ? - ? AND ? ORDER BY stamp ASC",
                       (rid, stamp, ctx, stamp)).fetchall()
    rows = rows[-F:] if len(rows) > F else rows
    frames, stamps = [], []
    for st, data in rows:
        img = np.frombuffer(data, dtype=np.uint8).reshape(480, 480, 3).astype(np.float64)
        k = 480 // R
        if k > 1:
            img = np.rint(img.reshape(R, k, R, k, 3).mean(axis=(1, 3)))
        img = (img / 255.0 - np.array((0.485, 0.456, 0.406))) / np.array((0.229, 0.224, 0.225))
        frames.append(img.transpose(2, 0, 1))
        stamps.append(st)
    pad = F - len(frames)
    frames = [np.zeros((3, R, R))] * pad + frames
    stamps = [stamp - ctx] * pad + stamps
    return np.asarray(stamps, np.float32), np.stack(frames).astype(np.float32)


@pytest.mark.parametrize("R", [480, 120])
def test_images_match_reference_query(db, R):
    F, fps = 3, 2
    ds = SoccerDiffusionDataset(db, num_samples_imu=5, num_samples_joint_states=5, num_samples_joint_trajectory=5,
                                num_samples_joint_trajectory_future=4, sampling_rate=50, num_joints=22, use_images=True,
                                num_frames_video=F, max_fps_video=fps, image_resolution=R)
    picks = [0, 20, 60, 100, 175, 176 + 3, 176 + 60]
    batch = ds.batch(torch.tensor(picks))
    assert batch["image_data"].shape == (len(picks), F, 3, R, R) and batch["image_data"].dtype == torch.float32
    for n, idx in enumerate(picks):
        for start, end, rid in ds.sample_boundaries:
            if start <= idx < end:
                break
        stamp = (idx - start) / 50
        want_st, want = _reference_images(db, rid, stamp, F, fps, R)
        item = ds[idx]
        for got_st, got in ((item.image_stamps, item.image_data), (batch["image_stamps"][n], batch["image_data"][n])):
            assert np.allclose(got_st.numpy(), want_st, atol=1e-6)
            assert np.abs(got.numpy() - want).max() < 1e-5
    # frames before any image / beyond the context are zero padding
    assert float(batch["image_data"][0].abs().max()) == 0.0


def test_image_resolution_must_divide_480(db):
    with pytest.raises(NotImplementedError):
        SoccerDiffusionDataset(db, use_images=True, image_resolution=224)
