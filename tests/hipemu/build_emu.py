"""TEST INFRASTRUCTURE: compile the unmodified HIP sources for the host CPU against the
fiber-based HIP emulator (tests/hipemu/hip/hip_runtime.h) -> tests/hipemu/libfdtd_emu.so."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "tidy3d_amd", "csrc")
LIB = os.path.join(HERE, "libfdtd_emu.so")
DEPS = [os.path.join(CSRC, "fdtd_capi.hip"), os.path.join(CSRC, "fdtd_kernels.hpp"), os.path.join(CSRC, "fdtd_kernels2.hpp"), os.path.join(CSRC, "fdtd_fused2.hpp"), os.path.join(CSRC, "fdtd_fused2.hip"),
        os.path.join(CSRC, "fdtd_fused2c.hip"), os.path.join(CSRC, "fdtd_shell2.hip"), os.path.join(CSRC, "fdtd_shell2.hpp"), os.path.join(CSRC, "fdtd_shell2_host.hpp"), os.path.join(CSRC, "fdtd_strip.hpp"), os.path.join(CSRC, "fdtd_aniso.hpp"),
        os.path.join(ROOT, "include", "fdtd_hip.h"), os.path.join(HERE, "hip_emu.cpp"),
        os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "rccl", "rccl.h"),
        os.path.abspath(__file__)]


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and all(
            os.path.getmtime(d) <= os.path.getmtime(LIB) for d in DEPS):
        return LIB
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = shutil.which("clang++") or shutil.which("g++")
    cmd = [cxx, "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-ffp-contract=off", "-mfma", "-I" + HERE,
           "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-pass-failed",
           "-x", "c++", os.path.join(CSRC, "fdtd_capi.hip"), os.path.join(CSRC, "fdtd_fused2.hip"), os.path.join(CSRC, "fdtd_fused2c.hip"),
           os.path.join(CSRC, "fdtd_shell2.hip"), os.path.join(HERE, "hip_emu.cpp"),
           "-o", LIB]
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
