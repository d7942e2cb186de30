"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares,
the ctypes signatures cover the header, host-side tables equal the oracle's, the module
mirror has the reference's state_dict keys, and the product path refuses to run on CPU."""

import os
import re
import subprocess

import pytest
import torch

from conftest import REPO
from oracle import ddim_ref
from oracle import denoiser_ref as ref

HEADER = os.path.join(REPO, "include", "soccerdiffusion_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from soccerdiffusion_amd import build

    build.build()
    from soccerdiffusion_amd import _lib

    return _lib


def test_library_exports_every_declared_symbol(lib):
    declared = _declared_symbols()
    assert len(declared) >= 15
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sd_[a-z0-9_]+)", out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert exported == set(declared), "exported symbol missing from the header: %s" % sorted(exported - set(declared))


def test_ctypes_signatures_cover_header(lib):
    assert sorted(lib.SIGNATURES) == _declared_symbols()
    h = lib.load()
    assert h.sd_abi_version() == 1
    assert h.sd_workspace_floats(2, 16, 11, 64, 2, 10) > 0
    assert h.sd_last_error() == b"ok"


def test_struct_layout_matches_header(lib):
    import ctypes as C

    assert C.sizeof(lib.LayerWeights) == 18 * 8
    assert C.sizeof(lib.DenoiserWeights) == 4 * 4 + 5 * 8 + 2 * 4 + 8
    assert C.sizeof(lib.EncoderWeights) == 6 * 4 + 4 * 8
    hdr = open(HEADER).read()
    block = hdr[hdr.index("typedef struct sd_layer_weights") : hdr.index("} sd_layer_weights;")]
    fields = re.findall(r"\*\s*([a-z0-9_]+)\s*[;,]", block)
    assert tuple(fields) == lib.LayerWeights.FIELDS


def test_argument_errors_without_gpu(lib):
    h = lib.load()
    assert h.sd_ddim_step(None, None, None, 1.0, 0.0, 1.0, 0.0, 10, None) == -1
    assert b"sd_ddim_step" in h.sd_last_error()
    assert h.sd_op_linear(None, None, None, None, None, None, None, 4, 64, 64, 0, None) == -1
    with pytest.raises(RuntimeError, match="invalid argument -1"):
        lib.check(-1, "x")


def test_host_tables_equal_oracle():
    from soccerdiffusion_amd import ops

    assert torch.equal(ops.alphas_cumprod(), ddim_ref.alphas_cumprod())
    for n in (10, 30, 50, 1000):
        assert ops.ddim_timesteps(n) == ddim_ref.timesteps(n).tolist()
    for d, T in ((64, 16), (256, 100)):
        assert torch.equal(ops.positional_table(d, T), ref.positional_table(d, T))
    acp = ops.alphas_cumprod()
    ts = ops.ddim_timesteps(50)
    coef = ops.ddim_coefficients(ts, acp, 50)
    for i, t in enumerate(ts):
        a_t, a_p = ddim_ref.step_coefficients(t, 50, acp)
        want = [a_t.sqrt(), (1 - a_t).sqrt(), a_p.sqrt(), (1 - a_p).sqrt()]
        assert [float(c) for c in coef[i]] == [float(w) for w in want]
    assert float(coef[-1][2]) == 1.0 and float(coef[-1][3]) == 0.0  # final_alpha_cumprod = 1
    # step frequencies: the values StepToken multiplies t with (misc.py:32)
    tok = ref.step_token(torch.tensor([1.0]), torch.zeros(1, 32), 64)
    assert torch.allclose(tok[0, 0, :16], torch.sin(ops.step_frequencies(64)))


def test_module_state_dict_keys_match_reference_checkpoints(g1, g2):
    from soccerdiffusion_amd.ml.model import End2EndDiffusionTransformer
    from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, SequenceEncoderType
    from soccerdiffusion_amd.ml.model.encoder.imu import IMUEncoder

    def build(c, full):
        return End2EndDiffusionTransformer(
            num_joints=c["J"], hidden_dim=c["d"], use_action_history=full, num_action_history_encoder_layers=c.get("enc_layers", 1),
            max_action_context_length=20, encoder_patch_size=5, use_imu=full,
            imu_orientation_embedding_method=IMUEncoder.OrientationEmbeddingMethod("quaternion"), num_imu_encoder_layers=c.get("enc_layers", 1),
            imu_context_length=20, use_joint_states=full, joint_state_encoder_layers=c.get("enc_layers", 1), joint_state_context_length=20,
            use_images=False, image_encoder_type=ImageEncoderType("resnet18"), image_sequence_encoder_type=SequenceEncoderType("transformer"),
            num_image_sequence_encoder_layers=1, image_context_length=0, image_use_final_avgpool=True, image_resolution=480,
            use_gamestate=full, num_decoder_layers=c["L"], trajectory_prediction_length=c["T"])

    for g, full in ((g1, False), (g2, True)):
        m = build(g["config"], full)
        want = g["state_dict"]
        have = m.state_dict()
        assert set(have) == set(want)
        for k in want:
            assert have[k].shape == want[k].shape, k
        m.load_state_dict(want)  # strict
        assert not any(k.endswith(".pe") or "_freq" in k for k in have)  # tables are non-persistent


def test_product_path_has_no_cpu_fallback(g1):
    from soccerdiffusion_amd import ops

    sd = g1["state_dict"]
    packed = ops.pack_denoiser(sd, "cpu", max_len=16)  # descriptors can be built anywhere ...
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.denoiser_forward(packed, g1["x"], torch.cat([g1["ctx"], g1["step_token_int"]], 1))  # ... compute cannot


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(REPO, "soccerdiffusion_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(root, f)


def test_scheduler_rejects_unsupported_configs():
    from soccerdiffusion_amd.scheduler import DDIMScheduler

    s = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
    assert s.config["num_train_timesteps"] == 1000
    s.set_timesteps(30)
    assert s.timesteps.tolist() == ddim_ref.timesteps(30).tolist()
    with pytest.raises(NotImplementedError):
        DDIMScheduler(beta_schedule="linear")
    s.config["num_train_timesteps"] = 500  # the reference's post-construction override (train.py:186)
    with pytest.raises(ValueError):
        s.set_timesteps(10)


def test_image_encoder_state_dict_keys_follow_torchvision():
    """Reference checkpoints hold the backbone under torchvision's names (image.py:61-73)."""
    from soccerdiffusion_amd.ml.model.encoder.image import (ImageEncoderType, ResNetImageEncoder, SequenceEncoderType,
                                                            image_sequence_encoder_factory)

    enc = image_sequence_encoder_factory(SequenceEncoderType.TRANSFORMER, ImageEncoderType.RESNET18, 64, 1, 10, False, 480)
    keys = set(enc.state_dict())
    for k in ("image_encoder.encoder.conv1.weight", "image_encoder.encoder.bn1.running_var",
              "image_encoder.encoder.layer1.1.conv2.weight", "image_encoder.encoder.layer3.0.downsample.0.weight",
              "image_encoder.encoder.layer4.1.bn2.num_batches_tracked", "image_encoder.encoder.avgpool.weight",
              "image_encoder.encoder.fc.bias", "transformer_encoder.embedding.weight",
              "transformer_encoder.transformer_encoder.layers.0.self_attn.in_proj_weight"):
        assert k in keys, k
    assert not any("layer1.0.downsample" in k for k in keys)
    assert ResNetImageEncoder.calculate_output_size(480) == 15 and ResNetImageEncoder.calculate_output_size(224) == 7
    assert enc.image_encoder.encoder.fc.in_features == 15 * 15 * 32
    backbone = sum(p.numel() for n, p in enc.image_encoder.encoder.named_parameters() if not n.startswith(("fc", "avgpool")))
    assert backbone == 11_176_512  # torchvision resnet18 without its classifier
    r50 = ResNetImageEncoder(ImageEncoderType.RESNET50, 64, True, 224)
    assert sum(p.numel() for p in r50.parameters()) == 23_508_032 + 2048 * 64 + 64


def test_swin_image_encoders_follow_torchvision():
    """swin_t / swin_s restated (image.py:84-98): torchvision's parameter counts and state_dict keys, head -> hidden_dim."""
    import torch

    from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, image_encoder_factory

    t = image_encoder_factory(ImageEncoderType.SWIN_TRANSFORMER_TINY, 64, True, 224).eval()
    assert sum(p.numel() for p in t.parameters()) == 28_288_354 - (768 * 1000 + 1000) + 768 * 64 + 64
    keys = set(t.state_dict())
    for k in ("encoder.features.0.0.weight", "encoder.features.0.2.bias", "encoder.features.1.1.attn.relative_position_bias_table",
              "encoder.features.1.1.attn.relative_position_index", "encoder.features.3.0.attn.qkv.weight", "encoder.features.5.5.mlp.3.bias",
              "encoder.features.6.reduction.weight", "encoder.features.7.1.norm2.weight", "encoder.norm.weight", "encoder.head.bias"):
        assert k in keys, k
    with torch.no_grad():
        y = t(torch.randn(1, 2, 3, 96, 96))   # 96 / 4 = 24 = windows of 7 with padding; later stages smaller than a window
    assert y.shape == (1, 2, 64) and torch.isfinite(y).all()
    s = image_encoder_factory(ImageEncoderType.SWIN_TRANSFORMER_SMALL, 64, True, 224)
    assert sum(p.numel() for p in s.parameters()) == 49_606_258 - (768 * 1000 + 1000) + 768 * 64 + 64


def test_grouped_weight_gradient_partition_terminates_and_covers():
    """sd_gemm_tn_grouped's search for slabs-per-workgroup must end for ANY group - also one with more output
    tiles than a chip-full of workgroups (it used to spin forever there).  Host arithmetic only, no launch."""
    import ctypes as C

    from soccerdiffusion_amd import _lib

    lib = _lib.load()

    def plan(probs):
        n = len(probs)
        R = (C.c_long * n)(*[p[0] for p in probs])
        N = (C.c_long * n)(*[p[1] for p in probs])
        K = (C.c_long * n)(*[p[2] for p in probs])
        tot = C.c_long(0)
        per = lib.sd_gemm_tn_grouped_plan(R, N, K, n, C.byref(tot))
        return per, tot.value

    # the training step's layer launch: six d x d gradients over 25 600 rows + two ragged J = 20 ones -> one chip-full
    per, tot = plan([(25600, 256, 256)] * 6 + [(25600, 20, 256), (25600, 256, 20)])
    assert per >= 8 and 0 < tot <= 512
    # more tiles than workgroups fit: eight 1152 x 1152 gradients (81 tiles each) - terminates, runs as several rounds
    per, tot = plan([(4096, 1152, 1152)] * 8)
    assert per >= 128 and tot == 8 * 81          # every workgroup takes a whole tile's 128 slabs
    per, tot = plan([(64, 4096, 4096)])          # one large matrix, two slabs: 1024 tiles
    assert per >= 2 and tot == 1024
    # coverage: chunks x per-workgroup slabs cover every slab of every problem
    for probs in ([(1000, 256, 256)], [(37, 128, 20), (100000, 256, 768)], [(25600, 256, 256)] * 16):
        per, tot = plan(probs)
        want = 0
        for R_, N_, K_ in probs:
            slabs = (R_ + 31) // 32
            cs = min(per, slabs)
            cs += cs & 1
            want += ((N_ + 127) // 128) * ((K_ + 127) // 128) * ((slabs + cs - 1) // cs)
        assert tot == want


def test_workspace_grows_with_the_key_tiles_of_the_memory():
    """sd_workspace_floats: the folded cross-attention blocks of the sampler are laid out per (trajectory, key tile of 16 memory rows)
    - one tile up to 16 rows (context + the step token), then one more per 16 (traj_step_wide_kernel, up to 64 rows)."""
    from soccerdiffusion_amd import _lib

    lib = _lib.load()
    B, T, d, L, n = 8, 10, 256, 4, 30
    w = [lib.sd_workspace_floats(B, T, M, d, L, n) for M in (0, 15, 16, 31, 32, 63)]
    assert w[0] < w[1] < w[2] < w[3] < w[4] < w[5]
    per_tile = 2 * L * B * 64 * 2 * d          # fp32 blocks + their fp16 hi | lo planes (2 halfs per float), per key tile
    rows = lambda M: 2 * L * B * (M + 1) * 2 * d   # kv + kvtmp
    assert w[2] - w[1] >= per_tile + rows(16) - rows(15)          # 16 -> 17 rows: a second tile
    # within one tile the K/V rows grow - and the generic trajectory kernels' K / V^T planes (csrc/sd_trajg.hip), one pair of 32 rows at a time
    pair = lambda M: 2 * L * B * ((M + 31) // 32) * 32 * d
    assert abs((w[1] - w[0]) - (rows(15) - rows(0)) - (pair(15) - pair(0))) <= 64 * 12
    assert lib.sd_sampler_mode(256, 4, 10, 50, 20) == 3 and lib.sd_sampler_mode(256, 4, 100, 63, 20) == 3   # 51 / 64 memory rows
    # 65 rows and more, hidden_dim 128 / 512, 22 joints: the generic trajectory kernels - every shipped YAML's shape is mode 3
    assert lib.sd_sampler_mode(256, 4, 10, 64, 20) == 3 and lib.sd_sampler_mode(128, 4, 10, 311, 22) == 3 and lib.sd_sampler_mode(512, 4, 10, 311, 20) == 3
    assert lib.sd_sampler_mode(512, 4, 100, 10, 20) != 3 and lib.sd_sampler_mode(64, 4, 16, 10, 20) != 3     # a 100-token panel at 512 does not fit the LDS
