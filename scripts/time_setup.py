#!/usr/bin/env python
"""Set-up time of BASELINE configs 3 and 5 (discretize: grid, rasteriser, sub-pixel pass, sources, monitors), phase by phase, with
the native host passes (tidy3d_amd/libfdtd_host.so) and with the NumPy passes they replace ($TIDY3D_AMD_NO_HOST_LIB=1), and a check
that both produce the same material indices and table.  python scripts/time_setup.py [3] [5]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import tidy3d_amd.schema as td  # noqa: E402
import tidy3d_amd.discretize as D  # noqa: E402
import tidy3d_amd.sources as S  # noqa: E402
from tidy3d_amd.constants import C_0  # noqa: E402


def config5():
    from cases import gold_johnson_christy
    dl = 0.005
    nxy, nz = 1024, 256 - 24
    pitch = 64 * dl
    au = gold_johnson_christy()
    f0 = 5e14
    pulse = td.GaussianPulse(freq0=f0, fwidth=1e14)
    L, Lz = nxy * dl, nz * dl
    discs = [td.Structure(geometry=td.Cylinder(center=(-L / 2 + (i + 0.5) * pitch, -L / 2 + (j + 0.5) * pitch, 0.0), radius=0.08, length=0.04, axis=2), medium=au)
             for i in range(16) for j in range(16)]
    slab = td.Structure(geometry=td.Box(center=(0, 0, -Lz / 4 - 0.02), size=(td.inf, td.inf, Lz / 2)), medium=td.Medium(permittivity=2.1))
    plane = (td.inf, td.inf, 0)
    return td.Simulation(size=(L, L, Lz), grid_spec=td.GridSpec.uniform(dl=dl), run_time=6e-14, structures=[slab] + discs,
                         sources=[td.PlaneWave(center=(0, 0, Lz / 2 - 0.1), size=plane, source_time=pulse, direction="-")],
                         monitors=[td.FluxMonitor(center=(0, 0, Lz / 2 - 0.05), size=plane, freqs=[f0], name="R"),
                                   td.FluxMonitor(center=(0, 0, -Lz / 2 + 0.1), size=plane, freqs=[f0], name="T"),
                                   td.FieldMonitor(center=(0, 0, 0.03), size=(td.inf, 0, 0), freqs=[f0], name="line", fields=["Ex"], colocate=False)],
                         boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml()), shutoff=1e-4)


def config3():
    lam = 1.55
    f0 = C_0 / lam
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 10)
    plane = (td.inf, td.inf, 0)
    return td.Simulation(
        size=(4.0, 2.0, 8.0), grid_spec=td.GridSpec.uniform(dl=0.01), run_time=2.6e-13, medium=td.Medium(permittivity=1.44 ** 2),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.45, 0.22, td.inf)), medium=td.Medium(permittivity=3.48 ** 2))],
        sources=[td.ModeSource(center=(0, 0, -3.5), size=plane, source_time=pulse, direction="+", mode_spec=td.ModeSpec(num_modes=1), mode_index=0)],
        monitors=[td.FluxMonitor(center=(0, 0, 3.0), size=plane, freqs=[f0], name="fwd"), td.FluxMonitor(center=(0, 0, -3.8), size=plane, freqs=[f0], name="bwd"),
                  td.ModeMonitor(center=(0, 0, 3.0), size=plane, freqs=[f0], mode_spec=td.ModeSpec(num_modes=1), name="mm"),
                  td.FieldMonitor(center=(0, 0, 1.0), size=(0, 0, 0), freqs=[f0], name="p1", fields=["Ex"]),
                  td.FieldMonitor(center=(0, 0, 2.0), size=(0, 0, 0), freqs=[f0], name="p2", fields=["Ex"])],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=12)), shutoff=1e-5)


tt = {}


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        tt[name] = tt.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(mod, name, g)


for n_ in ("rasterize", "_subpixel_average", "_filled", "plan_monitors", "apply_absorbers", "_truncate_spent_sources"):
    if hasattr(D, n_):
        wrap(D, n_)
wrap(S, "build_sources")
import tidy3d_amd.modesource as MSRC  # noqa: E402
wrap(MSRC, "mode_profile")                # (the eigen-solves of mode sources / monitors: not part of the rasteriser's budget)

which = [a for a in sys.argv[1:] if a in ("3", "5")] or ["3", "5"]
for name in which:
    sim = {"3": config3, "5": config5}[name]()
    sig = {}
    for native in (1, 0, 1):
        os.environ["TIDY3D_AMD_NO_HOST_LIB"] = "0" if native else "1"
        tt.clear()
        t0 = time.perf_counter()
        d = D.discretize(sim)
        dt = time.perf_counter() - t0
        h = hashlib.sha1(d.spec.mat_idx.tobytes()).hexdigest()[:16] + "/%d" % len(d.spec.media)
        sig[native] = h
        print(json.dumps({"config": int(name), "native_host_passes": bool(native), "cores": os.cpu_count(), "setup_s": round(dt, 3),
                          "phases_s": {k: round(v, 3) for k, v in tt.items()}, "shape": list(d.spec.shape), "mat_idx": h}), flush=True)
        if native and "--engine" in sys.argv:
            # the engine's own set-up (material words, ADE cell lists, uploads) on the device — cProfile's top lines
            import cProfile
            import pstats
            import torch  # noqa: F401
            from tidy3d_amd.engine import HipEngine
            pr = cProfile.Profile()
            t0 = time.perf_counter()
            pr.enable()
            e = HipEngine(d.spec)
            pr.disable()
            t1 = time.perf_counter()
            e.close()
            st = pstats.Stats(pr)
            top = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:12]
            print(json.dumps({"config": int(name), "engine_init_s": round(t1 - t0, 3),
                              "top_tottime": [[f"{os.path.basename(k[0])}:{k[1]}:{k[2]}", round(v[2], 3)] for k, v in top]}), flush=True)
        del d
    assert sig[0] == sig[1], sig
