#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel-trace stats + FETCH_SIZE/WRITE_SIZE passes)
into a small text/JSON report that is committed under profiles/."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


report = {}
f = find("trace", "*kernel_stats.csv")
if f:
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    stats = []
    for r in rows:
        nm = short(r["Name"])
        calls = int(r["Calls"])
        tot = float(r["TotalDurationNs"]) / 1e6
        avg = float(r["AverageNs"]) / 1e3
        pct = float(r["Percentage"])
        stats.append(dict(kernel=nm, calls=calls, total_ms=tot, avg_us=avg, pct=pct))
        if pct >= 0.05:
            print(f"{nm:110s} {calls:7d} {tot:10.2f} {avg:10.1f} {pct:6.2f}")
    report["kernel_stats"] = stats

for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(tag, "*counter_collection.csv")
    if not f:
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += float(r["Counter_Value"])
        agg[k][1] += 1
    print(f"\n== {counter} per kernel (KiB units as reported; sum over dispatches, dispatches)")
    rep = {}
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"{k:110s} {v:16.0f} {n:7d}  per-dispatch {v / max(n, 1):14.1f}")
        rep[k] = dict(total=v, dispatches=n)
    report[counter] = rep

json.dump(report, open(os.path.join(out, "summary.json"), "w"), indent=1)

# HBM traffic per launch for bench.py's roofline.traffic: FETCH_SIZE and WRITE_SIZE are reported in KiB; FETCH_SIZE is
# doubled per the gfx950 note of MI355X_MICROARCH.md (HBM / rocprofv3 section); the passes are separate runs.
if "FETCH_SIZE" in report and "WRITE_SIZE" in report:
    per = {}
    for k, fv in report["FETCH_SIZE"].items():
        wv = report["WRITE_SIZE"].get(k)
        if not wv or not fv["dispatches"]:
            continue
        per[k] = {"hbm_bytes_per_launch": (2 * fv["total"] / fv["dispatches"] + wv["total"] / wv["dispatches"]) * 1024,
                  "dispatches": fv["dispatches"]}
    layer = [k for k in per if k.startswith("decoder_layer")]
    traj = [k for k in per if "traj_step_kernel" in k]
    dom = traj[0] if traj else (max(layer, key=lambda k: per[k]["dispatches"]) if layer else None)
    traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/profile_gpu.sh); FETCH_SIZE doubled per the "
                         "gfx950 note in MI355X_MICROARCH.md; KiB units",
               "workload": "bench.py --steps 1 (B=4096)", "dominant_kernel": dom,
               # bench.py's roofline averages over every launch of the layer class: so does this figure
               "decoder_layer_kernel_bytes_per_launch":
                   (sum(per[k]["hbm_bytes_per_launch"] * per[k]["dispatches"] for k in layer) / sum(per[k]["dispatches"] for k in layer))
                   if layer else None,
               # sampler mode 3: one launch per DDIM step carries the whole denoiser step
               "traj_step_kernel_bytes_per_launch": per[traj[0]]["hbm_bytes_per_launch"] if traj else None,
               "round": os.environ.get("SD_PROFILE_ROUND", "r04"),
               "per_kernel": per}
    json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("\n== HBM bytes per launch (2 x FETCH + WRITE):")
    for k, v in per.items():
        print(f"{k:110s} {v['hbm_bytes_per_launch'] / 1e9:8.3f} GB")
