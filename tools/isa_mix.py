"""Static instruction mix of selected kernels (loops in the fused kernels are fully unrolled, so static ~ dynamic).
usage: python tools/isa_mix.py <asm.s> <substring> [...]   (asm from: hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only)"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
pats = sys.argv[2:]
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for n, (i, name) in enumerate(starts):
    if not any(p in name for p in pats):
        continue
    end = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    c = collections.Counter()
    for line in lines[i + 1:end]:
        line = line.strip()
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"([a-z_0-9]+)\s", line + " ")
        if not m or line.startswith((".", ";")):
            continue
        op = m.group(1)
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            c["valu:" + op] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    print(name, {k: v for k, v in c.items() if ":" not in k})
    print("   top valu:", sorted(((v, k[5:]) for k, v in c.items() if k.startswith("valu:")), reverse=True)[:22])
