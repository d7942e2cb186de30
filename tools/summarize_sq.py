#!/usr/bin/env python3
"""Per-kernel averages of the SQ counter pass of tools/profile_sq.sh (rocprofv3 --pmc, one row per dispatch and counter)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else int(os.environ.get("SD_SQ_TOP", "6"))   # kernels listed (by total duration)
files = glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)
if not files:
    sys.exit("no counter_collection.csv under " + out)
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
dur = defaultdict(float)
for r in csv.DictReader(open(files[0])):
    k = r["Kernel_Name"].split("(")[0][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in cnt[k]:
        cnt[k].add(r["Dispatch_Id"])
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
names = sorted({c for v in agg.values() for c in v})
summary = {}
print("per-dispatch averages (counter units as rocprofv3 reports them: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles, "
      "SQ_VALU_MFMA_BUSY_CYCLES in cycles, MI355X_MICROARCH.md)")
for k in sorted(agg, key=lambda k: -dur[k])[:TOP]:
    n = len(cnt[k])
    v = {c: agg[k][c] / n for c in names}
    print(f"\n{k}   dispatches {n}   avg duration under the counter pass {dur[k] / n / 1e3:.1f} us")
    for c in names:
        print(f"   {c:28s} {v[c]:16.0f}")
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    if gui and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # busy cycles (32 per v_mfma_f32_32x32x16_f16) are summed over the chip's 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE is
        # summed over the 8 XCDs (gui / duration = 8 x the shader clock), so the kernel's length in shader cycles is gui / 8
        cyc = gui / 8
        summary[k.replace("void ", "")] = {"mfma_busy_frac_of_simd_cycles": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                                           "effective_shader_clock_ghz": cyc / (dur[k] / n), "dispatches": n,
                                           "mfma_32cycle_instructions_per_dispatch": v["SQ_VALU_MFMA_BUSY_CYCLES"] / 32}
        print(f"   MFMA pipe busy: {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.1f} % of SIMD-cycles "
              f"(effective shader clock {cyc / (dur[k] / n):.2f} GHz; {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 32 / 1e6:.1f} M "
              f"32-cycle MFMAs per dispatch)")
    if v.get("SQ_WAVE_CYCLES"):
        w = v["SQ_WAVE_CYCLES"]
        print("   of a wave's cycles: parked (waitcnt / barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %%"
              % (100 * v.get("SQ_WAIT_ANY", 0) / w, 100 * v.get("SQ_WAIT_INST_ANY", 0) / w, 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / w))
    if v.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   LDS bank-conflict cycles / LDS active cycles: {100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE']:.1f} %")

import json
traj = [k for k in summary if "traj_step_kernel" in k]
if traj:
    k = traj[0]
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE (tools/profile_sq.sh); busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
               "round": os.environ.get("SD_PROFILE_ROUND", "r04"),
               "traj_step_kernel_mfma_busy": summary[k]["mfma_busy_frac_of_simd_cycles"],
               "traj_step_kernel_effective_clock_ghz": summary[k]["effective_shader_clock_ghz"],
               "per_kernel": summary}, open(os.path.join(out, "pmc_sq.json"), "w"), indent=1)
layer = [k for k in summary if k.startswith("decoder_layer")]
if layer and not traj:
    tot = sum(summary[k]["dispatches"] for k in layer)
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE (tools/profile_sq.sh); busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
               "decoder_layer_kernel_mfma_busy": sum(summary[k]["mfma_busy_frac_of_simd_cycles"] * summary[k]["dispatches"] for k in layer) / tot,
               "decoder_layer_kernel_effective_clock_ghz": sum(summary[k]["effective_shader_clock_ghz"] * summary[k]["dispatches"] for k in layer) / tot,
               "per_kernel": summary}, open(os.path.join(out, "pmc_sq.json"), "w"), indent=1)
