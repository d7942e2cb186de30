"""Per-kernel register / scratch / occupancy table from hipcc's resource-usage remarks.
usage: python tools/kernel_resources.py [substring]   (compiles sd_kernels.hip and sd_train.hip for gfx950)"""
import re
import subprocess
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else ""
for src in ("soccerdiffusion_amd/csrc/sd_kernels.hip", "soccerdiffusion_amd/csrc/sd_train.hip", "soccerdiffusion_amd/csrc/sd_train_chain.hip"):
    out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Iinclude", "-c", src, "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for name, r in rows.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.split("(")[0].replace("void ", "")
        if pat in dem:
            print(f"{dem:58s} vgpr {r.get('VGPRs', -1):4d} agpr {r.get('AGPRs', -1):4d} scratch {r.get('ScratchSize', -1):5d} "
                  f"vspill {r.get('VGPRs Spill', -1):4d} occ {r.get('Occupancy', -1)} lds {r.get('LDS Size', -1)}")
