#!/usr/bin/env python3
"""Headline benchmark: denoised joint-trajectories/s, 50-step DDIM, H=100, J=20.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode sample|train] [--batch B]

``--mode sample`` (default, BASELINE.json configs[2]): one "step" = one complete 50-step DDIM rollout of a batch of
B = 4096 trajectories per GPU through ``sd_ddim_sample`` (d=256, L=4, 4 heads, T=100, J=20, M=11 memory tokens),
inputs resident in HBM.  Ranks sample independent trajectories: no data-path collective (weak scaling).

``--mode train`` (BASELINE.json configs[1] / [3]): one "step" = one training iteration of the decoder-pretraining
path (add_noise, forward, MSE, backward, RCCL all-reduce of the flat gradient, AdamW, OneCycleLR) at B = 256
trajectories per GPU through ``training.train_step`` (weak scaling, one collective per step).

N > 1: one process per GPU.  Under ``torch.distributed.run`` (WORLD_SIZE set) this process is one rank; started
plainly as ``python bench.py --gpus N`` the parent launches N fresh ranks itself BEFORE it touches the GPU and exits
with their exit code.  A world size that disagrees with ``--gpus`` is an error, never a silent 1-GPU run.

Prints ONE JSON line (rank 0) with the contract keys plus ``roofline`` and ``cpu_baseline``; at N = 1 in sample mode
the line also carries the hipGraph replay of the same rollout, north_star's B = 256 sampling shape and the C2
training step (each timed after the headline region).
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # only the torch.nn comparison leg of `image_backbone_train` ever reaches MIOpen

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

D, L, HEADS, T, J, MC, N_DDIM = 256, 4, 4, 100, 20, 10, 50
M = MC + 1
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2516.8  # MI355X_MICROARCH.md: BF16/F16 MFMA dense (~2.5 PF = 16 x the fp32 matrix rate)
PEAK_HBM_GBS = 8000.0
FUSED_BYTES_PER_TRAJ_STEP = 2 * T * J * 4 + M * D * 4   # x in + eps out + memory = 27 264 B
TRAIN_B = 256


def flops_per_traj_step():
    """SURVEY.md §8(d) algorithmic FLOPs per trajectory per denoiser step, by kernel class."""
    gemm = L * 16 * T * D * D                       # QKV, out, q_c, out_c, W1, W2 (row GEMMs)
    kv = L * 4 * M * D * D                          # memory K/V projection
    attn = L * (4 * T * T * D + 4 * T * M * D)      # self + cross attention cores
    io = 4 * T * J * D                              # embedding + fc_out
    return {"gemm": gemm, "kv": kv, "attn": attn, "io": io, "total": gemm + kv + attn + io}


def executed_flops_per_traj():
    """FLOPs this implementation executes per trajectory: the sampler caches the memory K/V over the rollout and
    folds the cross-attention Q and out projections into them (NOTEBOOK.md 5.5), so per step and layer the 16 T d^2 of
    row GEMMs become 12 T d^2 + 4 T d (heads*M); the fold itself is 8 Mc d^2 per layer, once."""
    layer_chain = (L * 12 - 6) * T * D * D + L * 4 * T * D * HEADS * M + 2 * T * D * J   # decoder_layer_kernel, L launches
    head = 2 * T * J * D + 6 * T * D * D                                                 # decoder_head_kernel
    attn = L * 4 * T * T * D                                                             # self-attention cores
    once = L * 8 * MC * D * D
    return {"layer_chain": layer_chain, "head": head, "step": layer_chain + head + attn, "once": once,
            "rollout": N_DDIM * (layer_chain + head + attn) + once}


def traj_step_flops_per_traj_step(mode: int = 4):
    """Sampler modes 3 / 4 (csrc/sd_traj.h): one launch per DDIM step owns everything of SURVEY 8(d)'s F_step except the memory
    K/V projection (once per rollout).  Executed: 16x16x32 fp16 MFMAs (16 384 FLOP each), three per product - in mode 4 two at the
    Q | K | V projection (NOTEBOOK.md 5.11 / 5.12) - on 7 token tiles of 16 (T = 100 padded to 112), the cross-attention in its folded form."""
    f = flops_per_traj_step()
    qkv = 4 * 84 * 8                                                            # products of 16x16x32 tiles: Q | K | V projection, per layer
    rest = 4 * (49 * 2 + 28 * 4 + 112 * 2) + 28 * 8 + 112 * 3 + 2 * 112 * 8    # scores, PV, out-projection; folded cross-attention; W1, W2
    mfma = L * ((2 if mode == 4 else 3) * qkv + 3 * rest) + 3 * (16 * 7 + 14 * 8)   # mode 4: Q | K | V reads one activation plane; + embedding + fc_out
    return {"algorithmic": f["total"] - f["kv"], "executed": mfma * 16384.0, "mfma_instructions": mfma}


def layer_kernel_algorithmic_flops_per_traj_step(merged: bool):
    """SURVEY §8(d) FLOPs owned by the L launches of the layer kernel in one DDIM step, per trajectory: all row GEMMs
    of the reference algorithm (16 T d^2 per layer, the two cross-attention projections the fold removes INCLUDED -
    algorithmic, not executed), the cross-attention cores 4 T M d, fc_out 2 T d J and - when the next step's head runs
    inside the last layer's launch - its embedding 2 T J d; layer 0's LN1+QKV (6 T d^2) of the FIRST step belongs to
    decoder_head_kernel."""
    f = L * 16 * T * D * D + L * 4 * T * M * D + 2 * T * D * J
    if merged:
        f += 2 * T * J * D - (6 * T * D * D + 2 * T * J * D) / N_DDIM
    else:
        f -= 6 * T * D * D
    return f


def physical_cores() -> int:
    """Distinct (package, core) pairs among the CPUs this process may run on."""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id") as fh:
                core = fh.read().strip()
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id") as fh:
                pkg = fh.read().strip()
            seen.add((pkg, core))
        except OSError:
            seen.add(("?", str(c)))
    return max(1, len(seen))


def cpu_quota() -> float:
    """CPUs this process may actually use: the cgroup quota when there is one (a box's affinity mask can list every
    host CPU while the container is throttled to a fraction of them), else the affinity count."""
    n = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            with open(path) as fh:
                quota, period = fh.read().split()[:2]
            if quota != "max":
                n = min(n, float(quota) / float(period))
        except (OSError, ValueError):
            pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            q = float(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            per = float(fh.read())
        if q > 0:
            n = min(n, q / per)
    except (OSError, ValueError):
        pass
    return max(1.0, n)


def cpu_baseline(sd, seconds_budget=25.0):
    """the CPU oracle (a stock-PyTorch restatement of the reference path, validated against the
    reference's own modules) on the host cores of this box, fp32, no_grad, B = 256: the thread count is the fastest of
    a short sweep (one denoiser step each: more threads than the container's CPU share only thrash), then one more
    warm-up step and as many timed steps of the 50-step rollout as the budget allows (a bounded sample of the same
    workload)."""
    import torch
    from oracle import ddim_ref
    from oracle import denoiser_ref as ref

    phys, quota = physical_cores(), cpu_quota()
    Bc = 256
    x = torch.randn(Bc, T, J, generator=torch.Generator().manual_seed(1234))
    ctx = torch.randn(Bc, MC, D, generator=torch.Generator().manual_seed(1235))
    acp = ddim_ref.alphas_cumprod()
    ts = ddim_ref.timesteps(N_DDIM).tolist()

    def one(x, t):
        with torch.no_grad():
            eps = ref.forward_with_context(sd, [ctx], x, torch.full((Bc,), t, dtype=torch.int64))
        return ddim_ref.step(eps, t, x, N_DDIM, acp)

    cands = sorted({c for c in (8, 16, 32, 64, int(quota), phys) if 1 <= c <= phys})
    sweep, t_sweep = {}, time.perf_counter()
    for c in cands:
        torch.set_num_threads(c)
        one(x, ts[0])                      # thread-pool spin-up at this width
        t0 = time.perf_counter()
        one(x, ts[0])
        sweep[c] = time.perf_counter() - t0
        if time.perf_counter() - t_sweep > 20.0:
            break
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    x = one(x, ts[0])  # warm-up at the chosen width
    t0 = time.perf_counter()
    done = 0
    for t in ts:
        x = one(x, t)
        done += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return {
        "value": round(Bc * (done / N_DDIM) / dt, 3),
        "unit": "trajectories/s",
        "cores": threads,
        "kind": "port",
        "seconds_per_denoiser_step": round(dt / done, 4),
        "thread_sweep_seconds_per_step": {str(k): round(v, 3) for k, v in sweep.items()},
        "sample": f"B={Bc} trajectories x {done} of {N_DDIM} DDIM steps after warm-up (d={D}, L={L}, T={T}, J={J}, M={M}), "
                  f"oracle/denoiser_ref.py + oracle/ddim_ref.py, fp32, no_grad, {threads} torch threads (fastest of the sweep; "
                  f"this box: {phys} physical cores in the affinity mask, cgroup CPU quota {quota:g})",
    }


def cpu_baseline_train(sd, seconds_budget=15.0):
    """The same contract for --mode train: the oracle's training step (forward, MSE, torch CPU autograd backward; no optimizer:
    AdamW on 2.6 M parameters is noise next to it) at p = 0 at SURVEY 8(d)'s batch, B = 256 trajectories per step - a bounded
    number of steps (as many as the budget allows after one warm-up step, at least one), at the thread count the box's CPU share
    allows."""
    import torch
    from oracle import ddim_ref
    from oracle import denoiser_ref as ref

    quota = cpu_quota()
    threads = max(1, min(int(quota), physical_cores()))
    torch.set_num_threads(threads)
    Bc = TRAIN_B
    g = torch.Generator().manual_seed(4321)
    x0, eps = torch.randn(Bc, T, J, generator=g), torch.randn(Bc, T, J, generator=g)
    ctx = torch.randn(Bc, MC, D, generator=g)
    t = torch.randint(0, 1000, (Bc,), generator=g)
    x_t = ddim_ref.add_noise(x0, eps, t, ddim_ref.alphas_cumprod())
    ref.train_loss_and_grads(sd, x_t, t, eps, context=[ctx])   # warm-up
    t0, done = time.perf_counter(), 0
    while True:
        ref.train_loss_and_grads(sd, x_t, t, eps, context=[ctx])
        done += 1
        if time.perf_counter() - t0 > seconds_budget or done >= 400:
            break
    dt = time.perf_counter() - t0
    return {"value": round(Bc * done / dt, 3), "unit": "trajectories/s", "cores": threads, "kind": "port",
            "seconds_per_step": round(dt / done, 4),
            "sample": f"{done} training steps of B={Bc} trajectories after warm-up (forward + MSE + backward of the d={D}, L={L}, T={T}, "
                      f"J={J}, M={M} denoiser at p = 0, oracle/denoiser_ref.py under torch CPU autograd, fp32, {threads} torch threads; "
                      f"cgroup CPU quota {quota:g})"}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (torch.distributed.run, rendezvous on
    127.0.0.1) before this process makes any GPU call, and hand their exit code back."""
    import torch

    have = torch.cuda.device_count()   # counting devices does not initialise the GPU
    if os.environ.get("SD_BENCH_SHARE_GPU") == "1":   # rehearsal on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL
        have = max(have, n)
    if have < n:
        sys.stderr.write(f"bench.py: --gpus {n} but this node exposes {have} GPU(s)\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd)


# ------------------------------------------------------------------------------------------------------------------
# training leg (BASELINE configs[1]: C2; configs[3]: C2 x N with the RCCL gradient all-reduce)
# ------------------------------------------------------------------------------------------------------------------
C2_PARAMS = dict(hidden_dim=D, action_context_length=100, trajectory_prediction_length=T, epochs=1, batch_size=TRAIN_B,
                 lr=1e-4, train_denoising_timesteps=1000, image_context_length=10, imu_context_length=100,
                 num_imu_encoder_layers=2, joint_state_context_length=100, num_normalization_samples=1000, num_joints=J,
                 use_action_history=False, num_action_history_encoder_layers=2, use_imu=False,
                 imu_orientation_embedding_method="quaternion", use_joint_states=False, joint_state_encoder_layers=2,
                 use_images=False, image_sequence_encoder_type="transformer", image_encoder_type="resnet18",
                 num_image_sequence_encoder_layers=1, num_decoder_layers=L, distill_teacher_inference_steps=30,
                 use_gamestate=False, encoder_patch_size=10)


class TrainLeg:
    """C2: decoder d=256 L=4, B trajectories per GPU of horizon 100 x 20 joints, memory = 10 random context tokens + the
    step token (the reference's --decoder-pretraining path, train.py:221-224), dropout as `dropout` (the reference
    trains with torch's default 0.1)."""

    def __init__(self, dev, rank, world, batch, total_steps, dropout=0.1, graph=True):
        import torch
        from soccerdiffusion_amd import cli, training
        from soccerdiffusion_amd.scheduler import DDIMScheduler

        self.torch, self.training, self.world, self.B, self.dev = torch, training, world, batch, dev
        torch.manual_seed(0)   # every rank builds the same replica ...
        self.model = cli.build_model(C2_PARAMS).to(dev).train()
        if hasattr(self.model, "set_dropout"):
            self.model.set_dropout(dropout, seed=1234 + 7919 * rank)   # every rank its own Philox key, as cli.cmd_train
        self.dropout = dropout if hasattr(self.model, "set_dropout") else 0.0
        self.opt = training.FusedAdamW(self.model.parameters(), lr=1e-4)
        if world > 1:   # ... and rank 0's parameters are broadcast anyway (cli.cmd_train does the same)
            training.broadcast_parameters(self.opt, self.model)
        self.lr = torch.optim.lr_scheduler.OneCycleLR(self.opt, max_lr=1e-4, total_steps=total_steps + 64)   # + the graph's eager calls and exposed_allreduce_ms' 2 x 10 steps
        self.ns = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
        self.g = torch.Generator(device=dev).manual_seed(1 + rank)
        self.x0 = torch.randn(batch, T, J, device=dev, generator=self.g)
        self.ctx = [torch.randn(batch, MC, D, device=dev, generator=self.g)]
        self.graphed = None
        if graph:   # the step replayed from a hipGraph (training.GraphedTrainStep); the first two calls run eagerly
            self.graphed = training.GraphedTrainStep(self.model, self.opt, self.lr, self.ns, world_size=world, generator=self.g)

    def step(self):
        if self.graphed is not None:
            return self.graphed(self.x0, context=self.ctx)
        return self.training.train_step(self.model, self.opt, self.lr, self.ns, self.x0, context=self.ctx,
                                        world_size=self.world, generator=self.g)

    def close(self):
        if self.graphed is not None:
            self.graphed.close()

    def exposed_allreduce_ms(self, steps=10):
        """Step time with the exchange minus step time without it (the same eager step with world_size = 1 semantics on this
        rank's shard): what the overlapped, per-layer all-reduce still costs on the critical path."""
        torch = self.torch
        if self.world <= 1:
            return 0.0

        def timed(world):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.training.train_step(self.model, self.opt, self.lr, self.ns, self.x0, context=self.ctx, world_size=world, generator=self.g)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3

        with_x = timed(self.world)
        without = timed(1)   # replicas diverge from here on: call this last
        return with_x - without

    def allreduce_ms(self, iters=10):
        """The gradient exchange alone (flat fp32 buffer, sum + mean) timed with events on the current stream."""
        torch = self.torch
        if self.world <= 1:
            return 0.0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.training.allreduce_gradients(self.opt, self.world)
        torch.cuda.synchronize()
        a.record()
        for _ in range(iters):
            self.training.allreduce_gradients(self.opt, self.world)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters


def time_train(leg: TrainLeg, steps: int, warmup: int, dist):
    torch = leg.torch
    for _ in range(warmup):
        leg.step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = leg.step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device=leg.dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, float(loss)


def train_chain_roofline(leg: "TrainLeg", step_ms: float, n=4):
    """Roofline of the training step's dominant kernel class - the layer chains: since round 4 the trajectory-owning forward launches
    (sd_train_head_fwd, sd_train_layer_fwd: embedding, both attention cores and every row GEMM of a layer's forward) and the backward
    row chains (sd_train_bwd_chain; 57 % of the step in profiles/r04_train_kernel_stats_and_traffic.txt): n EAGER steps with a
    HIP-event pair around every launch (the graphed step cannot be instrumented), their summed time per step against the
    algorithmic FLOPs they own - the row GEMMs of forward and dX, 2 x 16 T d^2, plus the forward's attention cores, 4 T^2 d + 4 T M d,
    per layer and trajectory (the weight gradients are the grouped GEMM's, the attention backward its own kernel's)."""
    import ctypes as C

    from soccerdiffusion_amd import _lib

    lib, torch = _lib.load(), leg.torch
    eager = lambda: leg.training.train_step(leg.model, leg.opt, leg.lr, leg.ns, leg.x0, context=leg.ctx, world_size=leg.world,   # noqa: E731
                                            generator=leg.g)
    eager()
    torch.cuda.synchronize()
    lib.sd_profile_enable(1)
    for _ in range(n):
        eager()
    torch.cuda.synchronize()
    lib.sd_profile_enable(0)
    k = len(_lib.KERNEL_CLASSES)
    ms, cnt = (C.c_double * k)(), (C.c_long * k)()
    _lib.check(lib.sd_profile_collect(ms, cnt, k), "sd_profile_collect")
    ch = _lib.KERNEL_CLASSES.index("decoder_layer_kernel")
    chain_ms = ms[ch] / n
    flops = leg.B * L * (32 * T * D * D + 4 * T * T * D + 4 * T * M * D)
    ach = flops / max(chain_ms, 1e-9) / 1e9
    return {"bound": "mfma", "kernel": "layer chains (train_head_fwd_kernel / train_layer_fwd_kernel, csrc/sd_train_traj.hip; "
                                       "train_bwd_chain_kernel, csrc/sd_train_chain.hip)",
            "achieved": round(ach, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4),
            "vs_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 3),
            "definition": "algorithmic FLOPs of the forward (16 T d^2 + 4 T^2 d + 4 T M d per layer and trajectory) and of the dX row GEMMs "
                          "(16 T d^2) / summed HIP-event time of the chain-class launches in one eager step",
            "chain_ms_per_step": round(chain_ms, 4), "chain_launches_per_step": int(cnt[ch] // n),
            "share_of_graphed_step": round(chain_ms / max(step_ms, 1e-9), 4),
            "class_ms_per_eager_step": {nm: round(ms[i] / n, 4) for i, nm in enumerate(_lib.KERNEL_CLASSES) if ms[i] > 0},
            **train_traffic_record(chain_ms)}


def train_traffic_record(chain_ms):
    """HBM bytes the chain launches of ONE step move, from the committed FETCH_SIZE / WRITE_SIZE passes (profiles/pmc_traffic_train.json:
    another box and run than this line) - and what they are algorithmically: the row tensors the chains read and write."""
    p = os.path.join(REPO, "profiles", "pmc_traffic_train.json")
    if not os.path.exists(p):
        return {"traffic": None}
    with open(p) as fh:
        t = json.load(fh)
    tr = t.get("chain_bytes_per_step")
    return {"traffic": tr, "traffic_unit": "HBM bytes per step over the chain launches (2 x FETCH_SIZE + WRITE_SIZE)",
            "hbm_gbs_over_chain_time": round(tr / max(chain_ms, 1e-9) / 1e6, 1) if tr else None,
            "traffic_source": t.get("source"), "traffic_round": t.get("round")}


def train_record(elapsed, steps, world, B, loss, allreduce_ms, dropout, n_params):
    f = flops_per_traj_step()["total"]
    tf = 3 * f * B * world * steps / elapsed / 1e12
    return {
        "value": round(world * B * steps / elapsed, 1), "unit": "trajectories/s", "ms_per_step": round(elapsed / steps * 1e3, 3),
        "batch_per_gpu": B, "n_gpus": world,
        "algorithmic_tflops": round(tf, 2),   # 3 x F_step per trajectory: forward + backward
        "algorithmic_tflops_per_gpu": round(tf / world, 2),
        "frac_of_f16_mfma_peak": round(tf / world / PEAK_F16_MFMA_TFLOPS, 4),
        "frac_of_f32_mfma_peak": round(tf / world / PEAK_F32_MFMA_TFLOPS, 4),
        "allreduce_ms": round(allreduce_ms, 4), "allreduce_bytes": 4 * n_params,
        "dropout_p": dropout, "final_loss": round(loss, 6),
    }


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 3 rollouts / 30 training steps)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 1 rollout / 5 training steps)")
    ap.add_argument("--mode", choices=("sample", "train"), default="sample")
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU per step (default 4096 sample / 256 train)")
    ap.add_argument("--dropout", type=float, default=0.1, help="train mode: dropout probability (reference: 0.1)")
    ap.add_argument("--sampler-mode", type=int, choices=(0, 1, 2, 3, 4), default=3,
                    help="sample mode: cap of sd_ddim_sample_ex's kernel selection.  3 (default) = what End2EndDiffusionTransformer.sample and "
                         "an automatic sd_ddim_sample call run: the trajectory kernel with three fp16 products at every site (valid for any "
                         "weights); 4 = the opt-in two-product Q|K|V site (max_mode=4 / SD_SAMPLER_MODE=4), its SD_STATUS_SHARP_LOGITS guard "
                         "read after the timed region")
    ap.add_argument("--no-graph", action="store_true", help="train mode: issue the step's launches eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="sample mode: skip the hipGraph / B=256 / training sub-records")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))   # nothing below has run: this process never touched the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}` "
                         f"or plainly as `python bench.py --gpus {args.gpus}`")
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    share = os.environ.get("SD_BENCH_SHARE_GPU") == "1"   # rehearsal of the N-rank code path on one GPU (tests): gloo, all ranks on cuda:0
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        if dist.get_world_size() != args.gpus or dist.get_rank() != rank:
            raise SystemExit(f"bench.py: process group of {dist.get_world_size()} ranks (this one {dist.get_rank()}) for --gpus {args.gpus}")
    try:
        if args.mode == "train":
            run_train(args, rank, world, dev, dist)
        else:
            run_sample(args, rank, world, dev, dist)
    finally:
        if dist:
            dist.barrier()
            dist.destroy_process_group()


def rank_inventory(dev, rank, world, dist, share):
    """Self-verification of an N > 1 run (`world` in the JSON line): every rank's device - name, PCI address, uuid - gathered on all
    ranks, so the line itself shows N distinct GPUs (or, in the one-GPU rehearsal, that the ranks shared one), plus the backend."""
    import torch

    p = torch.cuda.get_device_properties(dev)
    pci = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    mine = {"rank": rank, "device_index": dev.index, "name": torch.cuda.get_device_name(dev), "pci": pci, "uuid": str(getattr(p, "uuid", "")),
            "pid": os.getpid()}
    ranks = [mine]
    if dist:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    return {"size": world, "backend": (dist.get_backend() if dist else None), "ranks": ranks,
            "distinct_devices": len({(r["pci"], r["uuid"]) for r in ranks}), "shared_gpu_rehearsal": bool(share)}


def run_train(args, rank, world, dev, dist):
    steps = args.steps if args.steps is not None else 30
    warmup = args.warmup if args.warmup is not None else 5
    B = args.batch or TRAIN_B
    leg = TrainLeg(dev, rank, world, B, steps + warmup, dropout=args.dropout, graph=not args.no_graph)
    elapsed, loss = time_train(leg, steps, max(warmup, 3 if leg.graphed is not None else 0), dist)   # capture happens on call 3
    ar = leg.allreduce_ms()
    ar_exposed = leg.exposed_allreduce_ms() if world > 1 else 0.0
    chain = None
    if world == 1 and not args.no_kernel_timing:
        try:
            chain = train_chain_roofline(leg, elapsed / steps * 1e3)
        except Exception as e:  # noqa: BLE001 - a sub-record must not take the line down
            chain = {"error": repr(e)[:300]}
    leg.close()
    inventory = rank_inventory(dev, rank, world, dist, os.environ.get("SD_BENCH_SHARE_GPU") == "1")   # (a collective: every rank calls it)
    if rank != 0:
        return
    rec = train_record(elapsed, steps, world, B, loss, ar, leg.dropout, leg.opt.flat_param.numel())
    rec["allreduce_exposed_ms"] = round(ar_exposed, 4)
    if world > 1 and ar > 0:   # bus bandwidth of the flat-gradient exchange as RCCL's tests define it: bytes x 2 (N - 1) / N / time
        rec["allreduce_busbw_gbs"] = round(rec["allreduce_bytes"] * 2 * (world - 1) / world / (ar * 1e-3) / 1e9, 2)
    if world > 1:
        rec["allreduce_form"] = ("per-layer buckets started from gradient hooks and overlapped with the rest of the backward (training.BucketedAllReduce; "
                                 "the step is issued eagerly at N > 1, the same step cli train runs); allreduce_ms = the flat exchange alone, "
                                 "allreduce_exposed_ms = step time with minus without the exchange")
        rec["note"] = "N > 1: correct by construction and rehearsed with two ranks on one GPU (gloo); no multi-GPU hardware run exists"
    line = {
        "metric": "training trajectories/s (fwd + bwd + AdamW per trajectory; denoiser d=256 L=4, H=100, J=20)",
        "value": rec["value"], "unit": "trajectories/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (operands split into fp16 hi+lo, 3 fp16 MFMAs per product, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[{1 if world == 1 else 3}]: one training step (add_noise, forward, MSE, backward, "
                               f"{'RCCL all-reduce of the flat fp32 gradient, ' if world > 1 else ''}AdamW, OneCycleLR) of the transformer "
                               f"denoiser d=256 L=4 heads=4, B={B} trajectories per GPU, horizon T=100, J=20, memory M=11 "
                               f"(decoder-pretraining path, reference train.py:204-240), dropout p={rec['dropout_p']}, "
                               f"{'the step replayed from a hipGraph' if leg.graphed is not None else 'launches issued eagerly'}",
                   "batch_per_gpu": B, "global_batch": B * world, "horizon": T, "joints": J, "hidden_dim": D, "decoder_layers": L,
                   "memory_tokens": M, "parallelism": f"dp{world}" + (" (one flat-gradient all-reduce per step)" if world > 1 else "")},
        "roofline": chain if chain and "error" not in chain else
                    {"bound": "mfma", "kernel": "whole training step", "achieved": rec["algorithmic_tflops_per_gpu"], "peak": PEAK_F16_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": rec["frac_of_f16_mfma_peak"], "traffic": None,
                     "definition": "3 x F_step (SURVEY 8(d)) x trajectories / wall time, per GPU"},
        "whole_step": {"achieved": rec["algorithmic_tflops_per_gpu"], "unit": "TFLOP/s", "frac_of_f16_mfma_peak": rec["frac_of_f16_mfma_peak"],
                       "definition": "3 x F_step (SURVEY 8(d)) x trajectories / wall time, per GPU"},
        "train": rec,
        "world": inventory,
        "cpu_baseline": None,
    }
    if not args.no_cpu_baseline and world == 1:
        from soccerdiffusion_amd.synthetic import synthetic_state_dict

        line["cpu_baseline"] = cpu_baseline_train(synthetic_state_dict(D, J, L, seed=7))
    print(json.dumps(line), flush=True)


def run_sample(args, rank, world, dev, dist):
    import numpy as np  # noqa: F401
    import torch

    from soccerdiffusion_amd import _lib, ops
    from soccerdiffusion_amd.synthetic import synthetic_state_dict

    steps = args.steps if args.steps is not None else 3
    warmup = args.warmup if args.warmup is not None else 1
    lib = _lib.load()
    B = args.batch or 4096
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
    guard = torch.zeros(1, dtype=torch.int32, device=dev)   # range-guard word of sd_ddim_sample_ex, read after the timed region

    # the kernels this call runs: sd_sampler_mode (3 where the trajectory kernel applies) capped by --sampler-mode (default 3 = the
    # product default, ops.default_sampler_cap()); 4 is opt-in
    auto_mode = lib.sd_sampler_mode(D, HEADS, T, MC, J)
    mode = args.sampler_mode if (auto_mode == 3 and args.sampler_mode >= 3) else min(auto_mode, args.sampler_mode)

    def rollout(m=mode):
        x.copy_(x_T)
        ops.ddim_sample(packed, ctx, toks, coef, x, inplace=True, status=guard, max_mode=m)

    for _ in range(warmup):
        rollout()
    timing = not args.no_kernel_timing
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    if timing:
        lib.sd_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
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
    assert int(guard.item()) == 0, "sampler range guard tripped (non-finite sample, or mode 4 outside its validated logit range)"

    roofline = None
    if timing:
        n = len(_lib.KERNEL_CLASSES)
        ms = (C.c_double * n)()
        cnt = (C.c_long * n)()
        _lib.check(lib.sd_profile_collect(ms, cnt, n), "sd_profile_collect")
        roofline = sample_roofline(ms, cnt, steps, B, elapsed, mode)

    inventory = rank_inventory(dev, rank, world, dist, os.environ.get("SD_BENCH_SHARE_GPU") == "1")   # (a collective: every rank calls it)
    if rank != 0:
        return
    extras = {}
    if world == 1 and not args.no_extras:
        extras = sample_extras(ops, packed, toks, coef, x_T, ctx, x, sd, dev, mode, guard)
    value = world * B * steps / elapsed
    line = {
        "metric": "denoised joint-trajectories/s (50-step DDIM, H=100, J=20)",
        "value": round(value, 2),
        "unit": "trajectories/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE_BY_MODE[mode],
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[2]: 50-step DDIM sampling, B=%d parallel rollouts per GPU, "
                        "transformer denoiser d=256 L=4 heads=4, horizon T=100, J=20, memory M=11 "
                        "(10 context tokens + step token); one step = one full rollout; the timed region issues the rollout's "
                        "launches eagerly with a HIP-event pair around each (the roofline leg); the hipGraph replay of the same "
                        "rollout is timed right after it (`hipgraph`).  Kernel selection: sampler mode %d (--sampler-mode; "
                        "`other_traj_mode` holds the same rollout in the other trajectory-kernel mode)" % (B, mode),
            "batch_per_gpu": B, "ddim_steps": N_DDIM, "horizon": T, "joints": J, "hidden_dim": D,
            "decoder_layers": L, "memory_tokens": M, "parallelism": f"dp{world} (independent rollouts, no collective)",
        },
        "roofline": roofline,
        "world": inventory,
        **extras,
        "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(sd),
    }
    print(json.dumps(line), flush=True)


DTYPE_BY_MODE = {
    0: "f32",
    1: "f32",
    2: "f32 (operands split into fp16 hi+lo, 3 fp16 MFMAs per product at every site, fp32 accumulate: 22-bit operands; "
       "50-step rollout error vs the fp64 oracle 3.6e-7, the fp32 CPU oracle's own 3.6e-7)",
    3: "f32 (operands split into fp16 hi+lo, 3 fp16 MFMAs per product at every site, fp32 accumulate: 22-bit operands; single "
       "noise prediction vs the fp64 oracle 8e-7 on these weights and 2e-6 on weights stressed to |logit| ~ 60 "
       "(tests/test_gpu_denoiser.py::test_mode3_noise_prediction_single_step); valid for any weights)",
    4: "f32 (operands split into fp16 hi+lo, fp32 accumulate; 3 fp16 MFMAs per product EXCEPT the self-attention's Q|K|V projection, "
       "which reads ONE fp16 plane of LayerNorm 1's output: 2 MFMAs per product, 11-bit activation operand at that site.  Error of a "
       "noise prediction vs the fp64 oracle ~ 1e-5 x max|attention logit|: 1.7e-5 on these weights (max |logit| 1.6; 50-step rollout "
       "5.1e-6), 1e-4 at |logit| ~ 9.  OPT-IN (max_mode=4 / SD_SAMPLER_MODE=4): the kernel sets SD_STATUS_SHARP_LOGITS beyond |logit| 5 "
       "- checked == 0 after the timed region - and ops.ddim_sample_guarded then repeats on mode 3, the default)",
}


def sample_roofline(ms, cnt, steps, B, elapsed, mode):
    from soccerdiffusion_amd import _lib

    names = _lib.KERNEL_CLASSES
    if mode >= 3:
        return traj_roofline(ms, cnt, steps, B, elapsed, mode)
    dl = names.index("decoder_layer_kernel")
    at = names.index("attention_kernel")
    f = flops_per_traj_step()
    ex = executed_flops_per_traj()
    # mode 2 runs the head of steps 1.. inside the previous step's last layer launch (NOTEBOOK.md 5.5)
    merged = mode == 2 and os.environ.get("SD_MERGE_HEAD", "1") != "0"
    launches = max(int(cnt[dl]), 1)
    dl_s = ms[dl] / 1e3
    avg_s = dl_s / launches
    # --- ALGORITHMIC FLOPs of the dominant kernel / its time ---------------------------------------
    alg_per_traj_step = layer_kernel_algorithmic_flops_per_traj_step(merged)
    alg_flops = steps * B * N_DDIM * alg_per_traj_step
    achieved = alg_flops / dl_s / 1e12
    # --- what the matrix pipe executed: folded projections removed, every product as 3 fp16 MFMAs in mode 2 ----
    exe_per_traj_step = ex["layer_chain"] + (ex["head"] * (N_DDIM - 1) / N_DDIM if merged else 0)
    mfma_factor = 3.0 if mode == 2 else 1.0
    executed = mfma_factor * steps * B * N_DDIM * exe_per_traj_step / dl_s / 1e12
    peak = PEAK_F16_MFMA_TFLOPS if mode == 2 else PEAK_F32_MFMA_TFLOPS
    kernel_name = "decoder_layer_f16_kernel<256>" if mode == 2 else "decoder_layer_kernel<256>"
    # --- HBM: this design's own per-launch bytes (a + h read, h + next q|k|v written, folded blocks read once) ----
    tail_units = 2 + (4 * (N_DDIM - 1) / N_DDIM if merged else 0)
    design_bytes = B * T * D * 4 * ((L - 1) * 6 + tail_units) / L + B * 64 * 2 * D * 4 * (1 if mode else 0)
    prof = profiles_record()
    traffic = prof.get("decoder_layer_kernel_bytes_per_launch") if prof else None
    att_traffic = prof.get("attention_kernel_bytes_per_launch") if prof else None
    fused_alg_bytes_per_step = B * FUSED_BYTES_PER_TRAJ_STEP + 10.59e6   # fused step + fp32 weights once per step
    hbm = {
        "algorithmic_bytes_per_launch_this_design": design_bytes,
        "achieved": round(design_bytes / avg_s / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": round(design_bytes / avg_s / 1e9 / PEAK_HBM_GBS, 4),
        "fused_algorithmic_bytes_per_ddim_step": fused_alg_bytes_per_step,
    }
    if traffic:
        hbm["traffic_over_this_design"] = round(traffic / design_bytes, 3)
        if att_traffic:
            hbm["traffic_over_algorithmic"] = round(L * (traffic + att_traffic) / fused_alg_bytes_per_step, 1)
            hbm["traffic_bytes_per_trajectory_step"] = round(L * (traffic + att_traffic) / 4096, 1)
    return {
        "bound": "mfma",
        "kernel": kernel_name,
        "achieved": round(achieved, 2),          # SURVEY 8(d) algorithmic FLOPs of this kernel class / its time
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4),
        "traffic": traffic,                      # HBM bytes per launch from the committed PMC passes (see from_profiles)
        "definition": "algorithmic FLOPs (SURVEY 8(d): 16Td^2 + 4TMd per layer + fc_out/embedding, folded projections "
                      "counted) of the L launches per DDIM step / summed HIP-event durations of those launches",
        "algorithmic_flops_per_launch_avg": alg_flops / launches,
        "launches": launches,
        "avg_launch_ms": round(avg_s * 1e3, 5),
        "sampler_mode": {0: "fp32 MFMA", 1: "fp32 MFMA, folded cross-attention",
                         2: "fp16x3 split-operand MFMA (fp32 accumulate), folded cross-attention"}[mode],
        # pipe occupancy, NOT useful work: executed products x 3 MFMAs each against the fp16 peak
        "mfma_pipe_occupancy": round(executed / peak, 4),
        "executed_mfma_tflops": round(executed, 2),
        "executed_mfma_flops_per_algorithmic_flop": round(mfma_factor * exe_per_traj_step / alg_per_traj_step, 3),
        "vs_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 3),   # what the split-fp16 design buys over an exact-fp32 MFMA
        "next_step_head_merged_into_last_layer": bool(merged),
        "hbm": hbm,
        "kernel_time_share": {k: round(ms[i] / 1e3 / elapsed, 4) for i, k in enumerate(names)},
        "attention_kernel": {"launches": int(cnt[at]), "avg_launch_ms": round(ms[at] / max(int(cnt[at]), 1), 5),
                             "algorithmic_tflops": round(steps * B * N_DDIM * L * 4 * T * T * D / max(ms[at] / 1e3, 1e-9) / 1e12, 2)},
        "whole_path": {   # SURVEY 8(d) F_step x 50 over the wall time of the timed region
            "achieved": round(steps * B * N_DDIM * f["total"] / elapsed / 1e12, 2),
            "frac": round(steps * B * N_DDIM * f["total"] / elapsed / 1e12 / peak, 4),
            "flops_per_trajectory": N_DDIM * f["total"],
            "executed_flops_per_trajectory": ex["rollout"],
        },
        "from_profiles": prof,
    }


def traj_roofline(ms, cnt, steps, B, elapsed, mode=4):
    """Roofline record of sampler modes 3 / 4: ONE kernel (traj_step_kernel) per DDIM step carries a trajectory through the
    whole denoiser step, so the dominant kernel IS the path."""
    from soccerdiffusion_amd import _lib

    names = _lib.KERNEL_CLASSES
    k = names.index("traj_step_kernel")
    f = flops_per_traj_step()
    tf = traj_step_flops_per_traj_step(mode)
    launches = max(int(cnt[k]), 1)
    k_s = ms[k] / 1e3
    avg_s = k_s / launches
    alg_flops = steps * B * N_DDIM * tf["algorithmic"]
    achieved = alg_flops / k_s / 1e12
    executed = steps * B * N_DDIM * tf["executed"] / k_s / 1e12
    peak = PEAK_F16_MFMA_TFLOPS
    # this design's own bytes per launch: x in + out, the folded cross-attention planes and score biases of every layer
    # (per trajectory), the split weights once
    design_bytes = B * (2 * T * J * 4 + L * (2 * 4 * 16 * D * 2 * 2 + 64 * 4)) + L * 6 * D * D * 4 + 2 * 32 * D * 4
    fused_alg_bytes_per_step = B * FUSED_BYTES_PER_TRAJ_STEP + 10.59e6
    prof = profiles_record(mode)
    traffic = prof.get("traj_step_kernel_bytes_per_launch") if prof else None
    hbm = {"algorithmic_bytes_per_launch_this_design": design_bytes, "achieved": round(design_bytes / avg_s / 1e9, 1), "peak": PEAK_HBM_GBS,
           "unit": "GB/s", "frac": round(design_bytes / avg_s / 1e9 / PEAK_HBM_GBS, 4),
           "fused_algorithmic_bytes_per_ddim_step": fused_alg_bytes_per_step,
           "design_bytes_per_trajectory_step": round(design_bytes / B, 1)}
    if traffic:
        hbm["traffic_over_this_design"] = round(traffic / design_bytes, 3)
        hbm["traffic_over_algorithmic"] = round(traffic / fused_alg_bytes_per_step, 1)
        hbm["traffic_bytes_per_trajectory_step"] = round(traffic / B, 1)
    return {
        "bound": "mfma",
        "kernel": "traj_step_kernel<7, %s>" % ("true" if mode == 3 else "false"),
        "achieved": round(achieved, 2),
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4),
        "traffic": traffic,
        "definition": "algorithmic FLOPs (SURVEY 8(d) F_step minus the memory K/V projection, which runs once per rollout) of the one "
                      "launch per DDIM step / summed HIP-event durations of those launches",
        "algorithmic_flops_per_launch_avg": alg_flops / launches,
        "launches": launches,
        "avg_launch_ms": round(avg_s * 1e3, 5),
        "sampler_mode": "%d: trajectory-owning step kernel (one workgroup per trajectory, self-attention inside; fp16x3 split-operand "
                        "MFMA, fp32 accumulate, folded cross-attention; %s)"
                        % (mode, "three products at every site" if mode == 3 else "two products at the Q|K|V projection, guarded"),
        "mfma_pipe_occupancy": round(executed / peak, 4),
        "executed_mfma_tflops": round(executed, 2),
        "executed_mfma_flops_per_algorithmic_flop": round(tf["executed"] / tf["algorithmic"], 3),
        "vs_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 3),
        "hbm": hbm,
        "kernel_time_share": {n: round(ms[i] / 1e3 / elapsed, 4) for i, n in enumerate(names)},
        "whole_path": {
            "achieved": round(steps * B * N_DDIM * f["total"] / elapsed / 1e12, 2),
            "frac": round(steps * B * N_DDIM * f["total"] / elapsed / 1e12 / peak, 4),
            "flops_per_trajectory": N_DDIM * f["total"],
        },
        "from_profiles": prof,
    }


def profiles_record(mode=None):
    """Counter-derived figures read from the committed profile summaries (profiles/*.json): measured by rocprofv3 on
    ANOTHER box and run than this line, so they are evidence with provenance, not live measurements.  Sampler mode 3
    (traj_step_kernel<7, true>) has passes of its own (pmc_*_mode3.json); the plain files are mode 4's (and the older modes')."""
    out = {}
    sfx = "_mode3" if mode == 3 else ""
    p = os.path.join(REPO, "profiles", f"pmc_traffic{sfx}.json")
    if os.path.exists(p):
        with open(p) as fh:
            t = json.load(fh)
        out["decoder_layer_kernel_bytes_per_launch"] = t.get("decoder_layer_kernel_bytes_per_launch")
        out["traj_step_kernel_bytes_per_launch"] = t.get("traj_step_kernel_bytes_per_launch")
        out["attention_kernel_bytes_per_launch"] = t.get("attention_kernel_bytes_per_launch")
        if out["attention_kernel_bytes_per_launch"] is None:
            for k, v in t.get("per_kernel", {}).items():
                if k.startswith("attention_f16"):
                    out["attention_kernel_bytes_per_launch"] = v.get("hbm_bytes_per_launch")
        out["traffic_source"] = t.get("source")
        out["traffic_round"] = t.get("round", "r01")
        out["traffic_kernel"] = t.get("dominant_kernel")
    p = os.path.join(REPO, "profiles", f"pmc_sq{sfx}.json")
    if os.path.exists(p):
        with open(p) as fh:
            s = json.load(fh)
        if "traj_step_kernel_mfma_busy" in s:
            out["mfma_busy_pmc"] = round(s["traj_step_kernel_mfma_busy"], 4)
            out["effective_clock_ghz_pmc"] = round(s["traj_step_kernel_effective_clock_ghz"], 3)
        else:
            out["mfma_busy_pmc"] = round(s["decoder_layer_kernel_mfma_busy"], 4)
            out["effective_clock_ghz_pmc"] = round(s["decoder_layer_kernel_effective_clock_ghz"], 3)
        out["sq_source"] = s.get("source")
        out["sq_round"] = s.get("round", "r01")
    if out:
        out["note"] = "from committed rocprofv3 passes (profiles/), a different box and run than this line"
    return out or None


def loop_form_record(sd, x_T, ctx, dev, x_native=None):
    """50 x (End2EndDiffusionTransformer.forward_with_context + DDIMScheduler.step) as a Python loop, at B = 256 and at the headline batch."""
    import torch

    from soccerdiffusion_amd import cli
    from soccerdiffusion_amd.scheduler import DDIMScheduler

    model = cli.build_model(C2_PARAMS).to(dev).eval()
    model.load_state_dict(sd)
    sched = DDIMScheduler(beta_schedule="squaredcos_cap_v2", clip_sample=False)
    sched.config["num_train_timesteps"] = 1000
    sched.set_timesteps(N_DDIM)
    rec = {"workload": "the reference's loop form (plot.py:122-131): 50 x [model.forward_with_context(ctx, x, t) -> one traj_step_kernel launch "
                       "through sd_sampler_eps + the step token's K/V and fold; scheduler.step -> sd_ddim_step], eager, sampler mode 3; context "
                       "folded once per loop (ops.LoopSampler)"}
    for Bs in sorted({256, x_T.shape[0]}):
        xs, cs = x_T[:Bs].contiguous(), [ctx[:Bs].contiguous()]

        def loop():
            traj = xs
            with torch.no_grad():
                for t in sched.timesteps:
                    eps = model.forward_with_context(cs, traj, torch.full((Bs,), int(t), device=dev))
                    traj = sched.step(eps, t, traj).prev_sample
            return traj

        y = loop()
        torch.cuda.synchronize()
        n = 3 if Bs > 1024 else 10
        t0 = time.perf_counter()
        for _ in range(n):
            y = loop()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        r = {"ms_per_rollout": round(dt * 1e3, 3), "value": round(Bs / dt, 2), "unit": "trajectories/s", "rollouts_timed": n}
        if x_native is not None and Bs == x_native.shape[0]:
            r["max_rel_diff_to_headline"] = float(((y - x_native).flatten(1).norm(dim=1) / x_native.flatten(1).norm(dim=1)).max())
        rec[f"b{Bs}"] = r
    return rec


def sample_extras(ops, packed, toks, coef, x_T, ctx, x, sd, dev, mode=4, guard=None):
    """After the headline region, N = 1: (a) the same rollout replayed from a hipGraph (BASELINE configs[2] names a
    hipGraph-captured step), checked bit for bit against the eager result; (b) north_star's B = 256 sampling shape;
    (c) the C2 training step."""
    import torch

    out = {}
    B = x_T.shape[0]
    # (0) the same rollout in the OTHER trajectory-kernel mode (3: three products everywhere / 4: two at the Q|K|V projection)
    if mode >= 3:
        try:
            from soccerdiffusion_amd import _lib

            lib = _lib.load()
            other = 7 - mode
            y = torch.empty_like(x)
            st = torch.zeros(1, dtype=torch.int32, device=dev)
            for i in range(3):
                if i == 1:
                    torch.cuda.synchronize()
                    lib.sd_profile_enable(1)   # a HIP-event pair around every launch of the two timed rollouts: this mode's own roofline
                    t0 = time.perf_counter()
                y.copy_(x_T)
                ops.ddim_sample(packed, ctx, toks, coef, y, inplace=True, status=st, max_mode=other)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            lib.sd_profile_enable(0)
            nk = len(_lib.KERNEL_CLASSES)
            kms, kcnt = (C.c_double * nk)(), (C.c_long * nk)()
            _lib.check(lib.sd_profile_collect(kms, kcnt, nk), "sd_profile_collect")
            out["other_traj_mode"] = {"sampler_mode": other, "ms_per_rollout": round(dt * 1e3, 3), "value": round(B / dt, 2),
                                      "unit": "trajectories/s", "status_word": int(st.item()),
                                      "max_rel_diff_to_headline": float(((y - x).flatten(1).norm(dim=1) / x.flatten(1).norm(dim=1)).max()),
                                      "dtype": DTYPE_BY_MODE[other], "roofline": traj_roofline(kms, kcnt, 2, B, 2 * dt, other)}
        except Exception as e:  # noqa: BLE001
            out["other_traj_mode"] = {"error": repr(e)[:300]}
    # (a) hipGraph replay at the headline batch
    try:
        gs = ops.GraphedSampler(packed, B, T, MC, toks, coef, max_mode=mode)
        y = gs(ctx, x_T)
        torch.cuda.synchronize()
        same = bool(torch.equal(y, x))
        n = 3
        t0 = time.perf_counter()
        for _ in range(n):
            gs.replay_into(ctx, x_T)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out["hipgraph"] = {"ms_per_rollout": round(dt * 1e3, 3), "value": round(B / dt, 2), "unit": "trajectories/s",
                           "replays_timed": n, "bit_identical_to_eager": same}
        del gs
    except Exception as e:  # noqa: BLE001 - a sub-record must not take the headline down
        out["hipgraph"] = {"error": repr(e)[:300]}
    # (b) B = 256 (north_star's sampling shape), hipGraph replay and eager
    try:
        Bs = 256
        xs, cs = x_T[:Bs].contiguous(), ctx[:Bs].contiguous()
        gs = ops.GraphedSampler(packed, Bs, T, MC, toks, coef, max_mode=mode)
        gs(cs, xs)
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            gs.replay_into(cs, xs)
        torch.cuda.synchronize()
        dg = (time.perf_counter() - t0) / n
        xe = torch.empty_like(xs)
        t0 = time.perf_counter()
        for _ in range(n):
            xe.copy_(xs)
            ops.ddim_sample(packed, cs, toks, coef, xe, inplace=True, status=guard, max_mode=mode)
        torch.cuda.synchronize()
        de = (time.perf_counter() - t0) / n
        out["b256"] = {"workload": "north_star shape: B=256, 50 DDIM steps, T=100, J=20 (one rollout = one step)",
                       "hipgraph_ms_per_rollout": round(dg * 1e3, 3), "eager_ms_per_rollout": round(de * 1e3, 3),
                       "value": round(Bs / min(dg, de), 2), "unit": "trajectories/s",
                       "algorithmic_tflops": round(Bs * N_DDIM * flops_per_traj_step()["total"] / min(dg, de) / 1e12, 2)}
        del gs
    except Exception as e:  # noqa: BLE001
        out["b256"] = {"error": repr(e)[:300]}
    # (b2) the reference's OWN loop form - for t in scheduler.timesteps: eps = model.forward_with_context(ctx, x, t); x = scheduler.step(eps,
    # t, x).prev_sample (plot.py:122-131, distill.py:179-189) - through the boundary class: every call reaches traj_step_kernel through
    # ops.LoopSampler (weights split and context folded once per loop), the DDIM update is its own launch
    try:
        out["loop_form"] = loop_form_record(sd, x_T, ctx, dev, x if B == 4096 else None)
    except Exception as e:  # noqa: BLE001
        out["loop_form"] = {"error": repr(e)[:300]}
    # (c) C2 training step on this GPU
    try:
        leg = TrainLeg(dev, 0, 1, TRAIN_B, 40)
        elapsed, loss = time_train(leg, 30, 5, None)
        leg.close()
        out["train"] = train_record(elapsed, 30, 1, TRAIN_B, loss, 0.0, leg.dropout, leg.opt.flat_param.numel())
        out["train"]["hipgraph"] = leg.graphed is not None
        try:
            leg2 = TrainLeg(dev, 0, 1, TRAIN_B, 40, graph=False)
            out["train"]["roofline"] = train_chain_roofline(leg2, out["train"]["ms_per_step"])
            del leg2
        except Exception as e:  # noqa: BLE001
            out["train"]["roofline"] = {"error": repr(e)[:300]}
        out["train"]["cpu_baseline"] = cpu_baseline_train(sd, seconds_budget=10.0)
        out["train"]["workload"] = "BASELINE.json configs[1]: C2 training step, B=256, d=256 L=4 T=100 J=20 M=11 (bench.py --mode train)"
    except Exception as e:  # noqa: BLE001
        out["train"] = {"error": repr(e)[:300]}
    # (d) SURVEY 8 row f2: the image backbone's inference forward on this repository's convolution kernels (csrc/sd_conv.hip) at
    # BASELINE configs[4]'s per-GPU share - 16 trajectories x 10 frames of 480 x 640 (the MIOpen route beside it: tools/bench_conv.py)
    try:
        from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, image_encoder_factory

        torch.manual_seed(0)
        enc = image_encoder_factory(ImageEncoderType.RESNET18, 256, True, 480).to(dev).eval()
        frames = torch.rand(16, 10, 3, 480, 640, device=dev)
        with torch.no_grad():
            enc(frames)
            torch.cuda.synchronize()
            n = 5
            t0 = time.perf_counter()
            for _ in range(n):
                enc(frames)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        flops = 2 * 1.814e9 * 480 * 640 / (224 * 224)   # ResNet-18: 1.814 GMAC per 224 x 224 frame, convolutions only
        out["image_backbone"] = {"workload": "BASELINE.json configs[4] per-GPU share, inference: ResNet-18 on 16 x 10 frames of 480 x 640 (all 20 convolutions hand-written: sd_stem_conv_bn_relu_pool, sd_conv3x3_bn_act, sd_conv_s2_bn_act)",
                                 "ms_per_forward": round(dt * 1e3, 3), "value": round(160 / dt, 1), "unit": "frames/s",
                                 "algorithmic_tflops": round(160 * flops / dt / 1e12, 1),
                                 "dtype": "f32 (operands split into fp16 hi+lo, 3 fp16 MFMAs per product, fp32 accumulate; 2e-6 vs torch's CPU operators)"}
        del enc, frames
    except Exception as e:  # noqa: BLE001
        out["image_backbone"] = {"error": repr(e)[:300]}
    # (e) ... and its TRAINING step (forward + backward of the backbone in train() mode: batch-statistics BatchNorm, every parameter's
    # gradient) on this repository's kernels (csrc/sd_conv_train.hip, conv_training.py) beside the torch.nn / MIOpen route on the same frames
    try:
        from soccerdiffusion_amd.ml.model.encoder.image import ImageEncoderType, image_encoder_factory

        torch.manual_seed(0)
        enc = image_encoder_factory(ImageEncoderType.RESNET18, 256, True, 480).to(dev).train()
        frames = torch.rand(16, 10, 3, 480, 640, device=dev)

        def fwd_bwd():
            for p in enc.parameters():
                p.grad = None
            enc(frames).sum().backward()

        rec = {}
        for route in ("hip", "torch"):
            if route == "torch":
                os.environ["SD_CONV"] = "torch"
            try:
                for _ in range(3):   # (three warm-up passes: MIOpen picks its backward kernels and the allocator settles over the first two)
                    fwd_bwd()
                torch.cuda.synchronize()
                n = 3
                t0 = time.perf_counter()
                for _ in range(n):
                    fwd_bwd()
                torch.cuda.synchronize()
                rec[route] = (time.perf_counter() - t0) / n
            finally:
                os.environ.pop("SD_CONV", None)
        flops = 3 * 2 * 1.814e9 * 480 * 640 / (224 * 224)   # forward + data gradient + weight gradient of every convolution
        out["image_backbone_train"] = {
            "workload": "BASELINE.json configs[4] per-GPU share, training: ResNet-18 forward + backward on 16 x 10 frames of 480 x 640, train() mode "
                        "(reference: train.py:226-240 trains the backbone with every step); every convolution (forward, data gradient, weight "
                        "gradient) and BatchNorm (batch statistics, backward) hand-written, the stem's BatchNorm + ReLU + max-pool fused (sd_bn_relu_pool_*), no MIOpen kernel",
            "ms_per_step": round(rec["hip"] * 1e3, 3), "value": round(160 / rec["hip"], 1), "unit": "frames/s",
            "algorithmic_tflops": round(160 * flops / rec["hip"] / 1e12, 1),
            "torch_nn_miopen_ms_per_step": round(rec["torch"] * 1e3, 3), "speedup_over_miopen_route": round(rec["torch"] / rec["hip"], 2),
            "dtype": "f32 (operands split into fp16 hi+lo, 3 fp16 MFMAs per product, fp32 accumulate; gradients 1e-5 vs torch CPU fp64 per kernel)"}
        del enc, frames
    except Exception as e:  # noqa: BLE001
        out["image_backbone_train"] = {"error": repr(e)[:300]}
    return out


if __name__ == "__main__":
    main()
