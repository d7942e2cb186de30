#!/usr/bin/env python3
"""profiles/pmc_traffic_train.json from the FETCH_SIZE / WRITE_SIZE passes of the training step (tools/profile_train_traffic.sh):
HBM bytes per launch of every kernel of the "row chain" class - the trajectory-owning forward launches (train_head_fwd_kernel,
train_layer_fwd_kernel) and the backward row chains (train_bwd_chain_kernel) - and their sum per step, which bench.py --mode train
prints beside the class's HIP-event time.   usage: python tools/summarize_train_traffic.py gpurun_out/prof_<tag> [round]"""
import json
import os
import sys

src = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
t = json.load(open(os.path.join(src, "pmc_traffic.json")))
per = t["per_kernel"]
chain = {k: v for k, v in per.items() if "train_layer_fwd_kernel" in k or "train_head_fwd_kernel" in k or "train_bwd_chain_kernel" in k or "train_fwd_chain_kernel" in k}
steps = min(v["dispatches"] for k, v in chain.items() if "train_layer_fwd_kernel" in k) // 4 if any("train_layer_fwd_kernel" in k for k in chain) else 7
launches = {k: v["dispatches"] / steps for k, v in chain.items()}
total = sum(v["hbm_bytes_per_launch"] * launches[k] for k, v in chain.items())
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over bench.py --mode train (tools/profile_train_traffic.sh); FETCH_SIZE doubled per "
                 "the gfx950 note of MI355X_MICROARCH.md; KiB units",
       "round": rnd, "workload": "bench.py --mode train (B=256, p=0.1)", "steps_profiled": steps,
       "chain_launches_per_step": launches, "chain_bytes_per_step": total,
       "note": "kernels outside the top 12 of both passes are not in the table (the head launch and the memory-side chain: < 2 % of the bytes)",
       "per_kernel_bytes_per_launch": {k: v["hbm_bytes_per_launch"] for k, v in per.items()}}
json.dump(out, open(os.path.join("profiles", "pmc_traffic_train.json"), "w"), indent=1)
print(json.dumps({"chain_bytes_per_step": total, "launches": launches}, indent=1))
