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
        os.path.join(CSRC, "fdtd_fused2c.hip"), os.path.join(CSRC, "fdtd_fused2d.hip"), os.path.join(CSRC, "fdtd_fused2w.hip"), os.path.join(CSRC, "fdtd_fused2s.hip"), os.path.join(CSRC, "fdtd_shell2.hip"), os.path.join(CSRC, "fdtd_shell2.hpp"), os.path.join(CSRC, "fdtd_shell2_host.hpp"), os.path.join(CSRC, "fdtd_strip.hpp"), os.path.join(CSRC, "fdtd_aniso.hpp"),
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
    # one object per translation unit, compiled side by side (the emulated library is rebuilt on every kernel edit: 4 min in one
    # compiler invocation, ~1.5 min this way), into a private directory so that two builds at once do not race
    import tempfile
    flags = [cxx, "-O2", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mfma", "-I" + HERE,
             "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-pass-failed"]
    srcs = [os.path.join(CSRC, "fdtd_capi.hip"), os.path.join(CSRC, "fdtd_fused2.hip"), os.path.join(CSRC, "fdtd_fused2c.hip"),
            os.path.join(CSRC, "fdtd_fused2d.hip"), os.path.join(CSRC, "fdtd_fused2w.hip"), os.path.join(CSRC, "fdtd_fused2s.hip"), os.path.join(CSRC, "fdtd_shell2.hip"),
            os.path.join(HERE, "hip_emu.cpp")]
    tmp = tempfile.mkdtemp(prefix="fdtd_emu_")
    try:
        procs, objs = [], []
        for src in srcs:
            obj = os.path.join(tmp, os.path.basename(src) + ".o")
            cmd = [*flags, "-x", "c++", "-c", src, "-o", obj]
            print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(obj)
        for cmd, pr in procs:
            if pr.wait() != 0:
                for _, other in procs:
                    if other.poll() is None:
                        other.kill()
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        out = os.path.join(tmp, "libfdtd_emu.so")
        cmd = [cxx, "-shared", "-fPIC", "-Wl,-Bsymbolic", *objs, "-o", out]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        shutil.move(out, LIB)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
