#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU: image-conditioned training step - ResNet-18 per frame + 8-head sequence encoder +
denoiser (d=256, L=4, horizon 100), B trajectories x image_context_length frames of 480x640 RGB, image_use_final_avgpool
True (the no-avgpool head assumes square frames: reference encoder/image.py:69-83).  Reports trajectories/s, frames/s
and the share of the step spent in the backbone (its forward + backward timed alone on the same frames).
The backbone trains on this package's kernels (conv_training.py; SD_CONV=torch: torch.nn / MIOpen, for the A/B); it restates torchvision's
ResNet-18 (architecture parity unpinned vs torchvision)."""
import argparse, json, os, sys, threading, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # the exhaustive convolution search at 1 280 x 3 x 480 x 640 takes many minutes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soccerdiffusion_amd import cli, training
from soccerdiffusion_amd.scheduler import DDIMScheduler


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--size", type=str, default="480x640")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--channels-last", action="store_true", help="backbone and frames in torch.channels_last (NHWC): A/B of the MIOpen layout")
    args = ap.parse_args()
    H, W = (int(v) for v in args.size.split("x"))
    t_start = time.time()
    phase = ["init"]

    def heartbeat():   # a long first step (MIOpen kernel selection) must not look like a hang
        while phase[0] != "done":
            print(f"[bench_c5] {time.time() - t_start:6.0f} s  phase: {phase[0]}", file=sys.stderr, flush=True)
            time.sleep(45)

    threading.Thread(target=heartbeat, daemon=True).start()
    dev = torch.device("cuda", 0)
    params = dict(bench.C2_PARAMS, use_images=True, image_context_length=args.frames, image_use_final_avgpool=True,
                  image_resolution=H, batch_size=args.batch)
    torch.manual_seed(0)
    model = cli.build_model(params).to(dev).train()
    model.set_dropout(0.1, seed=1)
    opt = training.FusedAdamW(model.parameters(), lr=1e-4)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=args.steps + 4)
    ns = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
    g = torch.Generator(device=dev).manual_seed(2)
    B, F = args.batch, args.frames
    frames = torch.rand(B, F, 3, H, W, device=dev, generator=g)
    if args.channels_last:   # the per-frame encoder sees (B * F, 3, H, W): NHWC storage for it and for its weights
        model.image_sequence_encoder.image_encoder.to(memory_format=torch.channels_last)
        frames = frames.view(B * F, 3, H, W).contiguous(memory_format=torch.channels_last).view(B, F, 3, H, W)
    x0 = torch.randn(B, bench.T, bench.J, device=dev, generator=g)
    inp = {"image_data": frames}

    def step():
        return training.train_step(model, opt, sch, ns, x0, input_data=inp, generator=g)

    torch.cuda.reset_peak_memory_stats()
    phase[0] = "warm-up step (MIOpen kernel selection)"
    step()                                            # warm-up (MIOpen kernel selection)
    torch.cuda.synchronize()
    phase[0] = "timed steps"
    per_step = []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        per_step.append(time.perf_counter() - t0)
    dt = min(per_step)     # steady state: the allocator still grows its pools in the first steps at this footprint
    stats = torch.cuda.memory_stats()
    peak = torch.cuda.max_memory_allocated() / 2**30
    # the backbone alone: forward + backward of the per-frame encoder on the same frames
    enc = model.image_sequence_encoder.image_encoder
    def backbone():
        for p in enc.parameters():
            p.grad = None
        enc(frames).sum().backward()
    phase[0] = "backbone alone"
    backbone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        backbone()
    torch.cuda.synchronize()
    db = (time.perf_counter() - t0) / args.steps
    phase[0] = "done"
    print(json.dumps({"workload": f"BASELINE configs[4] on 1 GPU: ResNet-18 + denoiser training step, B={B} x {F} frames of {H}x{W}, "
                                  f"d=256 L=4 T=100 J=20, avgpool head, dropout 0.1, backbone layout {'channels_last' if args.channels_last else 'NCHW'}",
                      "ms_per_step": round(dt * 1e3, 1), "trajectories_per_s": round(B / dt, 1), "frames_per_s": round(B * F / dt, 1),
                      "backbone_fwd_bwd_ms": round(db * 1e3, 1), "backbone_share": round(db / dt, 3),
                      "per_step_ms": [round(v * 1e3, 1) for v in per_step],
                      "allocator": {"alloc_retries": stats.get("num_alloc_retries", 0), "device_mallocs": stats.get("num_device_alloc", 0),
                                    "device_frees": stats.get("num_device_free", 0)},
                      "peak_hbm_gib": round(peak, 1), "loss": float(loss)}))


if __name__ == "__main__":
    main()
