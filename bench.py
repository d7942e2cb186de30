#!/usr/bin/env python3
"""Headline benchmark: denoised joint-trajectories/s, 50-step DDIM, H=100, J=20.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One "step" = one complete 50-step DDIM rollout of a batch of B trajectories through
``sd_ddim_sample`` (BASELINE.json config 3: d=256, L=4, 4 heads, T=100, J=20, M=11 memory
tokens, B=4096 per GPU), inputs resident in HBM.  N>1: one process per GPU (launched by
torch.distributed.run), every rank samples its own B trajectories — the path shards over
independent trajectories with no data-path collective (weak scaling); the only
collectives are the timing barrier and the MAX over ranks of the elapsed time.

Prints ONE JSON line (rank 0) with the contract keys plus ``roofline`` and ``cpu_baseline``.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

D, L, HEADS, T, J, MC, N_DDIM = 256, 4, 4, 100, 20, 10, 50
M = MC + 1
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2516.8  # MI355X_MICROARCH.md: BF16/F16 MFMA dense (~2.5 PF = 16 x the fp32 matrix rate)


def flops_per_traj_step():
    """SURVEY.md §8(d) algorithmic FLOPs per trajectory per denoiser step, by kernel class."""
    gemm = L * 16 * T * D * D                       # QKV, out, q_c, out_c, W1, W2 (row GEMMs)
    kv = L * 4 * M * D * D                          # memory K/V projection
    attn = L * (4 * T * T * D + 4 * T * M * D)      # self + cross attention cores
    io = 4 * T * J * D                              # embedding + fc_out
    return {"gemm": gemm, "kv": kv, "attn": attn, "io": io, "total": gemm + kv + attn + io}


def executed_flops_per_traj():
    """FLOPs this implementation executes per trajectory: the sampler caches the memory K/V over the rollout and
    folds the cross-attention Q and out projections into them, so per step and layer the 16 T d^2 of
    row GEMMs become 12 T d^2 + 4 T d (heads*M); the fold itself is 8 Mc d^2 per layer, once."""
    layer_chain = (L * 12 - 6) * T * D * D + L * 4 * T * D * HEADS * M + 2 * T * D * J   # decoder_layer_kernel, L launches
    head = 2 * T * J * D + 6 * T * D * D                                                 # decoder_head_kernel
    attn = L * 4 * T * T * D                                                             # self-attention cores
    once = L * 8 * MC * D * D
    return {"layer_chain": layer_chain, "head": head, "step": layer_chain + head + attn, "once": once,
            "rollout": N_DDIM * (layer_chain + head + attn) + once}


def cpu_baseline(sd, seconds_budget=25.0):
    """The CPU oracle (a stock-PyTorch restatement of the reference path, validated against
    the reference's own modules) on the host cores of this box: a bounded sample of the
    same workload."""
    from oracle import ddim_ref
    from oracle import denoiser_ref as ref

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    Bc = 32
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(Bc, T, J, generator=g)
    ctx = torch.randn(Bc, MC, D, generator=torch.Generator().manual_seed(1235))
    acp = ddim_ref.alphas_cumprod()
    ts = ddim_ref.timesteps(N_DDIM).tolist()

    def one(x, t):
        with torch.no_grad():
            eps = ref.forward_with_context(sd, [ctx], x, torch.full((Bc,), t, dtype=torch.int64))
        return ddim_ref.step(eps, t, x, N_DDIM, acp)

    x = one(x, ts[0])  # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    done = 0
    for t in ts:
        x = one(x, t)
        done += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return {
        "value": Bc * (done / N_DDIM) / dt,
        "unit": "trajectories/s",
        "cores": threads,
        "kind": "port",
        "sample": f"B={Bc} trajectories x {done} of {N_DDIM} DDIM steps (d={D}, L={L}, T={T}, J={J}, M={M}), "
                  f"oracle/denoiser_ref.py on {threads} torch threads, fp32",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU per rollout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from soccerdiffusion_amd.synthetic import synthetic_state_dict
    from soccerdiffusion_amd import _lib, ops

    lib = _lib.load()
    B = args.batch
    sd = synthetic_state_dict(D, J, L, seed=7)
    packed = ops.pack_denoiser(sd, dev, heads=HEADS, max_len=T)
    ts = ops.ddim_timesteps(N_DDIM)
    acp = ops.alphas_cumprod()
    coef = ops.ddim_coefficients(ts, acp, N_DDIM)
    toks = ops.step_token(torch.tensor(ts, device=dev), ops.step_frequencies(D).to(dev),
                          sd["step_encoding.token"].to(dev)).reshape(N_DDIM, D).contiguous()
    x_T = torch.randn(B, T, J, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    ctx = torch.randn(B, MC, D, generator=torch.Generator().manual_seed(1235 + rank)).to(dev)
    x = torch.empty_like(x_T)

    def rollout():
        x.copy_(x_T)
        ops.ddim_sample(packed, ctx, toks, coef, x, inplace=True)

    for _ in range(args.warmup):
        rollout()
    timing = not args.no_kernel_timing
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    if timing:
        lib.sd_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rollout()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lib.sd_profile_enable(0)
    if dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(x).all(), "sampler produced non-finite values"

    roofline = None
    if timing:
        n = len(_lib.KERNEL_CLASSES)
        ms = (C.c_double * n)()
        cnt = (C.c_long * n)()
        _lib.check(lib.sd_profile_collect(ms, cnt, n), "sd_profile_collect")
        f = flops_per_traj_step()
        names = _lib.KERNEL_CLASSES
        dl = names.index("decoder_layer_kernel")
        # Dominant kernel: decoder_layer_kernel, one launch per (DDIM step, layer).  Per trajectory it does the row
        # GEMMs of SURVEY 8(d) except layer 0's LN1+QKV, which decoder_head_kernel does (10Td^2 per layer + the next
        # layer's 6Td^2 QKV for all but the last), the cross-attention cores 4TMd and, in the last layer, fc_out
        # 2TdJ.  Summed over the L launches of a step:
        # In the sampler the Q and out projections of the cross-attention are folded into the cached memory
        # (executed_flops_per_traj): `achieved` counts the FLOPs the kernel really executes, not the reference's.
        ex = executed_flops_per_traj()
        per_traj_step_dl = ex["layer_chain"]
        mode = lib.sd_sampler_mode(D, HEADS, T, MC, J)
        # mode 2 runs the head of steps 1.. inside the previous step's last layer launch: those FLOPs and
        # bytes belong to this kernel class
        merged = mode == 2 and os.environ.get("SD_MERGE_HEAD", "1") != "0"
        if merged:
            per_traj_step_dl += ex["head"] * (N_DDIM - 1) / N_DDIM
        tail_units = 2 + (4 * (N_DDIM - 1) / N_DDIM if merged else 0)   # a, h in (+ h, q|k|v out), in units of B T d floats
        dl_flops = args.steps * B * N_DDIM * per_traj_step_dl
        dl_s = ms[dl] / 1e3
        achieved = dl_flops / dl_s / 1e12
        total_flops = args.steps * B * N_DDIM * f["total"]
        # mode 2: every product of the layer chain is 3 v_mfma_f32_32x32x16_f16 on split (hi + lo) operands:
        # the matrix pipe executes 3x the algorithmic FLOPs, priced against the fp16 MFMA peak; fc_out (2TdJ) stays fp32
        peak, kernel_name, mfma_factor = PEAK_F32_MFMA_TFLOPS, "decoder_layer_kernel<256>", 1.0
        if mode == 2:
            peak, kernel_name, mfma_factor = PEAK_F16_MFMA_TFLOPS, "decoder_layer_f16_kernel<256>", 3.0
        alg_tflops = achieved
        achieved = achieved * mfma_factor
        pmc = None
        pmc_file = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                pmc = json.load(fh).get("decoder_layer_kernel_bytes_per_launch")
        sq = None
        sq_file = os.path.join(REPO, "profiles", "pmc_sq.json")
        if os.path.exists(sq_file) and mode == 2:
            with open(sq_file) as fh:
                sq = json.load(fh)
        roofline = {
            "bound": "mfma",
            "kernel": kernel_name,
            "achieved": round(achieved, 2),
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            "sampler_mode": {0: "fp32 MFMA", 1: "fp32 MFMA, folded cross-attention",
                             2: "fp16x3 split-operand MFMA (fp32 accumulate), folded cross-attention"}[mode],
            "algorithmic_tflops": round(alg_tflops, 2),   # executed FLOPs counted once; the fp32 MFMA peak is 157.3
            # the same kernel against the HBM roof: per launch it must read a and h, write h and the next q|k|v (the last
            # layer writes x only) and read each trajectory's folded cross-attention blocks once
            "hbm": (lambda by: {"algorithmic_bytes_per_launch_avg": by, "achieved": round(by / (dl_s / max(int(cnt[dl]), 1)) / 1e9, 1),
                                "peak": 8000.0, "unit": "GB/s", "frac": round(by / (dl_s / max(int(cnt[dl]), 1)) / 8e12, 4)})(
                B * T * D * 4 * ((L - 1) * 6 + tail_units) / L + B * 64 * 2 * D * 4 * (1 if mode else 0)),
            "executed_mfma_flops_per_algorithmic_flop": mfma_factor,
            "traffic": pmc,
            # the hardware's own count from the committed SQ counter pass (profiles/pmc_sq.json): MFMA-pipe busy cycles over
            # SIMD-cycles at the clock the kernel really ran at (frac above is priced at the nominal 2.4 GHz)
            "mfma_busy_pmc": round(sq["decoder_layer_kernel_mfma_busy"], 4) if sq else None,
            "effective_clock_ghz_pmc": round(sq["decoder_layer_kernel_effective_clock_ghz"], 3) if sq else None,
            "next_step_head_merged_into_last_layer": bool(merged),
            "launches": int(cnt[dl]),
            "avg_launch_ms": round(ms[dl] / max(int(cnt[dl]), 1), 5),
            "flops_per_launch_avg": dl_flops / max(int(cnt[dl]), 1),
            "kernel_time_share": {k: round(ms[i] / 1e3 / elapsed, 4) for i, k in enumerate(names)},
            "whole_path": {   # executed = what the GPU did; reference_algorithm = SURVEY 8(d) F_step x 50 over the same time
                "achieved": round(mfma_factor * args.steps * B * ex["rollout"] / elapsed / 1e12, 2),
                "frac": round(mfma_factor * args.steps * B * ex["rollout"] / elapsed / 1e12 / peak, 4),
                "algorithmic_tflops": round(args.steps * B * ex["rollout"] / elapsed / 1e12, 2),
                "flops_per_trajectory": ex["rollout"],
                "reference_algorithm_tflops": round(total_flops / elapsed / 1e12, 2),
                "reference_flops_per_trajectory": N_DDIM * f["total"],
            },
        }

    if rank == 0:
        value = world * B * args.steps / elapsed
        line = {
            "metric": "denoised joint-trajectories/s (50-step DDIM, H=100, J=20)",
            "value": round(value, 2),
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if lib.sd_sampler_mode(D, HEADS, T, MC, J) != 2 else
                     "f32 (operands split into fp16 hi+lo, 3 fp16 MFMAs per product, fp32 accumulate: 22-bit operands; "
                     "50-step rollout error vs the fp64 oracle 3.6e-7, the fp32 CPU oracle's own 3.6e-7)",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[2]: 50-step DDIM sampling, B=%d parallel rollouts per GPU, "
                            "transformer denoiser d=256 L=4 heads=4, horizon T=100, J=20, memory M=11 "
                            "(10 context tokens + step token); one step = one full rollout" % B,
                "batch_per_gpu": B, "ddim_steps": N_DDIM, "horizon": T, "joints": J, "hidden_dim": D,
                "decoder_layers": L, "memory_tokens": M, "parallelism": f"dp{world} (independent rollouts, no collective)",
            },
            "roofline": roofline,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(sd),
        }
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
