#!/usr/bin/env python3
"""Small-batch rollout latency (the robot's shape: B=1, horizon 10, 30 DDIM steps, d=256, L=4):
eager launches vs the hipGraph-captured rollout.  The reference's budget is 0.2 s per rollout."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccerdiffusion_amd import cli

params = dict(hidden_dim=256, action_context_length=100, trajectory_prediction_length=10, epochs=1, batch_size=1, lr=1e-4,
              train_denoising_timesteps=1000, image_context_length=10, imu_context_length=100, num_imu_encoder_layers=2,
              joint_state_context_length=100, num_normalization_samples=1000, num_joints=20, use_action_history=False,
              num_action_history_encoder_layers=2, use_imu=False, imu_orientation_embedding_method="quaternion",
              use_joint_states=False, joint_state_encoder_layers=2, use_images=False, image_sequence_encoder_type="transformer",
              image_encoder_type="resnet18", num_image_sequence_encoder_layers=1, num_decoder_layers=4,
              distill_teacher_inference_steps=30, use_gamestate=False, encoder_patch_size=10)
torch.manual_seed(0)
m = cli.build_model(params).cuda().eval()
for B in (1, 16):
    x = torch.randn(B, 10, 20, device="cuda")
    ctx = [torch.randn(B, 31, 256, device="cuda")]
    out = {}
    for name, kw in (("eager", {}), ("hipgraph", {"use_graph": True})):
        for _ in range(3):
            m.sample(ctx, x, 30, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            m.sample(ctx, x, 30, **kw)
        torch.cuda.synchronize()
        out[name + "_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
    print(json.dumps({"B": B, "T": 10, "M": 32, "steps": 30, **out}))
