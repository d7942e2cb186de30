#!/usr/bin/env python3
"""Small-batch rollout latency (the robot's shape: B = 1, horizon 10, 30 DDIM steps, d = 256, L = 4; reference loop
soccer_diffusion/ml/inference/ros.py:301-310, budget 0.2 s per rollout) per kernel selection of sd_ddim_sample_ex: the trajectory
kernel family (modes 3 / 4: any T <= 100 with <= 16 memory rows) against the row-panel / unfused-chain kernels (max_mode 2), for the
memory sizes of the shipped configs: 50 context rows (sim_scratch.yaml: 20 + 20 + 10) and 31 (three 100-sample modalities at patch 10) - the
wide instantiation, 2 .. 4 key tiles -, 10 rows, none (decoder_only.yaml)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccerdiffusion_amd import _lib, cli, ops

params = dict(hidden_dim=256, action_context_length=100, trajectory_prediction_length=10, epochs=1, batch_size=1, lr=1e-4,
              train_denoising_timesteps=1000, image_context_length=10, imu_context_length=100, num_imu_encoder_layers=2,
              joint_state_context_length=100, num_normalization_samples=1000, num_joints=20, use_action_history=False,
              num_action_history_encoder_layers=2, use_imu=False, imu_orientation_embedding_method="quaternion",
              use_joint_states=False, joint_state_encoder_layers=2, use_images=False, image_sequence_encoder_type="transformer",
              image_encoder_type="resnet18", num_image_sequence_encoder_layers=1, num_decoder_layers=4,
              distill_teacher_inference_steps=30, use_gamestate=False, encoder_patch_size=10)
torch.manual_seed(0)
m = cli.build_model(params).cuda().eval()
T, n_steps = 10, 30
ts = ops.ddim_timesteps(n_steps)
coef = ops.ddim_coefficients(ts, ops.alphas_cumprod(), n_steps)
toks = m.step_encoding.table(ts, torch.device("cuda"))
packed = m.diffusion_action_generator.packed()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


for B in (1, 16, 256):
    for Mc in (50, 31, 10, 0):
        x = torch.randn(B, T, 20, device="cuda")
        ctx = torch.randn(B, Mc, 256, device="cuda") if Mc else None
        rec = {"B": B, "T": T, "memory_rows": Mc + 1, "steps": n_steps, "sd_sampler_mode": _lib.load().sd_sampler_mode(256, 4, T, Mc, 20)}
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        for mode in (2, 3, 4):
            rec[f"eager_max_mode_{mode}_ms"] = timed(lambda: ops.ddim_sample(packed, ctx, toks, coef, x, status=status, max_mode=mode))
            gs = ops.GraphedSampler(packed, B, T, Mc, toks, coef, max_mode=mode)
            rec[f"hipgraph_max_mode_{mode}_ms"] = timed(lambda: gs.replay_into(ctx, x))
            del gs
        print(json.dumps(rec), flush=True)

# ---- the robot's image-conditioned rollout (reference default.yaml: 10 frames of 224 x 224, ResNet-18 without the final avgpool, one
# sequence-encoder layer; B = 1, T = 10, 30 steps): encode_input_data + sample, hand-written backbone kernels vs the library route
del m, packed
img = dict(params, use_images=True, image_resolution=224, image_use_final_avgpool=False)
torch.manual_seed(0)
m = cli.build_model(img).cuda().eval()
for B in (1, 16):
    data = {"image_data": torch.rand(B, 10, 3, 224, 224, device="cuda")}
    x = torch.randn(B, T, 20, device="cuda")
    rec = {"B": B, "T": T, "steps": n_steps, "frames": 10, "frame_size": [224, 224], "case": "image-conditioned rollout (encode + sample)"}
    with torch.no_grad():
        for route in ("hip", "torch"):
            if route == "torch":
                os.environ["SD_CONV"] = "torch"
            rec[f"encode_{route}_backbone_ms"] = timed(lambda: m.encode_input_data(data))
            rec[f"encode_and_sample_{route}_backbone_ms"] = timed(lambda: m.sample(m.encode_input_data(data), x, n_steps))
            os.environ.pop("SD_CONV", None)
    print(json.dumps(rec), flush=True)
