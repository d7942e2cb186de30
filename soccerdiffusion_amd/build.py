"""Builds the HIP shared library in-tree for gfx950 (hipcc cross-compiles without a GPU).

    python -m soccerdiffusion_amd.build [--force]

Output: soccerdiffusion_amd/lib/libsoccerdiffusion_hip.so (git-ignored; travels with gpurun).
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
SRC = [os.path.join(PKG, "csrc", "sd_kernels.hip"), os.path.join(PKG, "csrc", "sd_train.hip")]
HDR = [os.path.join(REPO, "include", "soccerdiffusion_hip.h"), os.path.join(PKG, "csrc", "sd_common.h"),
       os.path.join(PKG, "csrc", "sd_f16x3.h")]
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libsoccerdiffusion_hip.so")
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SRC + HDR)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wno-unused-value",
           "-I", os.path.join(REPO, "include"), *SRC, "-o", LIB + ".tmp"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building " + LIB)
    os.replace(LIB + ".tmp", LIB)
    if verbose:
        sys.stderr.write(res.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
