"""Build libfdtd_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfdtd_hip.so")
SOURCES = [os.path.join(CSRC, "fdtd_capi.hip"), os.path.join(CSRC, "fdtd_fused2.hip"), os.path.join(CSRC, "fdtd_fused2c.hip"), os.path.join(CSRC, "fdtd_fused2d.hip"), os.path.join(CSRC, "fdtd_fused2w.hip"), os.path.join(CSRC, "fdtd_fused2s.hip"), os.path.join(CSRC, "fdtd_shell2.hip")]
# per-source flags: the two-steps-per-sweep kernels are built with the SLP vectorizer off (fdtd_fused2.hpp)
SOURCE_FLAGS = {"fdtd_fused2.hip": ["-fno-slp-vectorize"], "fdtd_fused2c.hip": ["-fno-slp-vectorize"], "fdtd_fused2d.hip": ["-fno-slp-vectorize"], "fdtd_fused2w.hip": ["-fno-slp-vectorize"], "fdtd_fused2s.hip": ["-fno-slp-vectorize"],
                "fdtd_shell2.hip": ["-fno-slp-vectorize"]}
DEPS = SOURCES + [os.path.join(CSRC, "fdtd_kernels.hpp"), os.path.join(CSRC, "fdtd_kernels2.hpp"),
                  os.path.join(CSRC, "fdtd_fused2.hpp"), os.path.join(CSRC, "fdtd_shell2.hpp"), os.path.join(CSRC, "fdtd_shell2_host.hpp"), os.path.join(CSRC, "fdtd_strip.hpp"), os.path.join(CSRC, "fdtd_aniso.hpp"),
                  os.path.join(HERE, "..", "include", "fdtd_hip.h"), os.path.abspath(__file__)]


HOST_LIB = os.path.join(HERE, "libfdtd_host.so")
HOST_SOURCE = os.path.join(CSRC, "host_raster.cpp")
HOST_DEPS = [HOST_SOURCE, os.path.join(HERE, "..", "include", "fdtd_host.h")]


def build_host(force: bool = False, verbose: bool = True) -> str:
    """libfdtd_host.so (include/fdtd_host.h): the rasteriser's native host passes — plain C++ on threads, no HIP runtime.
    -ffp-contract=off: the sample coordinates and inside tests are the NumPy statements' IEEE operations, one for one."""
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in HOST_DEPS):
        return HOST_LIB
    cxx = shutil.which("g++") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tmp = HOST_LIB + ".%d.tmp" % os.getpid()
    cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wall", HOST_SOURCE, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, HOST_LIB)
    return HOST_LIB


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    build_host(force, verbose)
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # -ffp-contract=off: every path (vector body, tile-edge scalar code, chunk prologue, two-pass and
    # fused kernels) then performs the same IEEE operations in the same order, so results do not
    # depend on the launch geometry and the variants agree bit for bit (the kernels are HBM-bound;
    # the few extra VALU instructions are free).
    # -mllvm -disable-lsr: loop strength reduction turns the sweep's  uniform base + lane offset  addresses
    # into per-lane 64-bit induction pointers (a VGPR pair per array for the whole z-march): 110 -> 100
    # VGPRs for the plain sweep, 167 + spills -> 153 for the one that carries materials and CPML.
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
              "-mllvm", "-disable-lsr", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value", "-I/opt/rocm/include",
              *os.environ.get("FDTD_EXTRA_HIPCC_FLAGS", "").split()]     # e.g. -DFDTD_PLACEMENT_PROBE for scripts/probe_layout.py
    # objects go to a private directory (two builds at once — ranks, CI jobs — must not race on fixed object paths), and a
    # failed compile takes its siblings down before it is reported
    tmp = tempfile.mkdtemp(prefix="fdtd_build_")
    objs, procs = [], []
    try:
        for src in SOURCES:                    # the translation units compile side by side
            obj = os.path.join(tmp, os.path.basename(src) + ".o")
            cmd = [*common, *SOURCE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(obj)
        failed = None
        for cmd, pr in procs:
            rc = pr.wait()
            if rc != 0 and failed is None:
                failed = subprocess.CalledProcessError(rc, cmd)
                for _, other in procs:
                    if other.poll() is None:
                        other.kill()
        if failed is not None:
            raise failed
        out = os.path.join(tmp, "libfdtd_hip.so")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out, "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        shutil.move(out, LIB)
    finally:
        for _, pr in procs:
            if pr.poll() is None:
                pr.kill()
                pr.wait()
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
