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
SRC = [os.path.join(PKG, "csrc", "sd_kernels.hip"), os.path.join(PKG, "csrc", "sd_train.hip"),
       os.path.join(PKG, "csrc", "sd_train_chain.hip"), os.path.join(PKG, "csrc", "sd_conv.hip"),
       os.path.join(PKG, "csrc", "sd_train_traj.hip"), os.path.join(PKG, "csrc", "sd_trajg.hip"),
       os.path.join(PKG, "csrc", "sd_conv_train.hip")]
HDR = [os.path.join(REPO, "include", "soccerdiffusion_hip.h"), os.path.join(PKG, "csrc", "sd_common.h"),
       os.path.join(PKG, "csrc", "sd_panel.h"), os.path.join(PKG, "csrc", "sd_f16x3.h"), os.path.join(PKG, "csrc", "sd_traj.h"),
       os.path.join(PKG, "csrc", "sd_trajg.h")]
# The sampler's translation unit is compiled WITHOUT packed fp32 vector instructions (v_pk_fma/mul/add_f32): they do not overlap with
# MFMAs - neither a wave's own nor its SIMD partner's - while plain fp32 instructions do (tools/exp/coissue3.hip; NOTEBOOK.md 5.11), and
# the trajectory kernel lives on that overlap: + 1.5 % sampler throughput.  The training units lose 0.6 % with the same flag: packed.
EXTRA_FLAGS = {"sd_kernels.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
               "sd_trajg.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]}
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libsoccerdiffusion_hip.so")
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def _obj(src: str) -> str:
    return os.path.join(LIB_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")


def _deps(src: str) -> list:
    """sd_f16x3.h and sd_traj.h are included by sd_kernels.hip only, sd_panel.h not by sd_train.hip."""
    hdr = [h for h in HDR if not (h.endswith("sd_f16x3.h") and not src.endswith("sd_kernels.hip"))
           and not (h.endswith("sd_trajg.h") and not (src.endswith("sd_kernels.hip") or src.endswith("sd_trajg.hip")))
           and not (h.endswith("sd_traj.h") and not (src.endswith("sd_kernels.hip") or src.endswith("sd_train_traj.hip")))
           and not (h.endswith("sd_panel.h") and (src.endswith("sd_train.hip") or src.endswith("sd_conv.hip") or src.endswith("sd_conv_train.hip")
                                                 or src.endswith("sd_trajg.hip")))]
    return [src] + hdr


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SRC + HDR + [os.path.abspath(__file__)])


def _run(cmd: list, what: str, verbose: bool) -> None:
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed " + what)
    if verbose:
        sys.stderr.write(res.stderr)


def packed_fp32_in_traj_kernels(obj: str = None) -> dict:
    """{kernel symbol: number of v_pk_{fma,mul,add}_f32 instructions} for every trajectory step kernel (traj_step_kernel / traj_step_wide_kernel /
    the generic family) in the sampler's object file.  EXTRA_FLAGS removes the packed fp32 operations from that translation unit's target
    features; the flag goes through -Xclang and a toolchain update could drop it silently, so build() checks its effect on the code."""
    import re
    import tempfile

    obj = obj or _obj(SRC[0])
    llvm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib", "llvm", "bin")
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--type=o", f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--input={fat}",
                        f"--output={co}", "--unbundle"], check=True, capture_output=True)
        asm = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
    counts, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
        if m:
            cur = m.group(1) if "traj_step" in m.group(1) else None
            if cur:
                counts[cur] = 0
        elif cur and re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
            counts[cur] += 1
    return counts


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per translation unit (recompiled only when it or a header it includes changed; the two compile in
    parallel), then one link.  The units share host functions only - no relocatable device code is needed."""
    if not force and not is_stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    flags = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wno-unused-value", "-I", os.path.join(REPO, "include")]
    if verbose:
        flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    jobs = []
    for src in SRC:
        obj = _obj(src)
        if force or not os.path.exists(obj) or any(os.path.getmtime(f) > os.path.getmtime(obj) for f in _deps(src) + [os.path.abspath(__file__)]):
            jobs.append((src, subprocess.Popen([hipcc(), *flags, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj + ".tmp"], stdout=subprocess.PIPE,
                                               stderr=subprocess.PIPE, text=True)))
    for src, proc in jobs:
        out, err = proc.communicate()
        if proc.returncode != 0:
            sys.stderr.write(out + err)
            raise RuntimeError("hipcc failed compiling " + src)
        os.replace(_obj(src) + ".tmp", _obj(src))
        if verbose:
            sys.stderr.write(err)
        if os.path.basename(src) == "sd_kernels.hip":
            counts = packed_fp32_in_traj_kernels(_obj(src))
            bad = {k: v for k, v in counts.items() if v}
            if not counts or bad:
                os.remove(_obj(src))
                raise RuntimeError(f"packed fp32 instructions in the trajectory step kernels (or none of them found): {bad or counts}; "
                                   "EXTRA_FLAGS['sd_kernels.hip'] no longer takes effect")
    _run([hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *[_obj(s) for s in SRC], "-o", LIB + ".tmp"], "linking " + LIB, verbose)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
