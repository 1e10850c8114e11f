#!/usr/bin/env python
"""Step pairs whose sweeps advance the dispersive cells (FDTD_OPT_DISP, round 6) against single steps of the same engine build,
bit for bit, on bench.py's V3 / V4 problems (Lorentz sphere + CPML [+ flux box]) and on the same sphere inside PEC walls and
absorber layers; prints one JSON line per case with the timings of both.
    python scripts/check_disp_gpu.py [n ...]        (default 320 512)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402


def spec_for(n, steps, kind):
    import tidy3d_amd.schema as td
    from tidy3d_amd.discretize import discretize
    if kind in ("v3", "v4"):
        return bench.build_spec(n, steps, kind)
    dl = 0.05
    lor = td.Lorentz(eps_inf=2.0, coeffs=[(2.0, 4e14, 2e13)])
    size = (n * dl - 1e-6 * dl,) * 3
    bspec = td.BoundarySpec.all_sides(td.PECBoundary())
    if kind == "absorber":
        size = ((n - 80) * dl - 1e-6 * dl,) * 3
        bspec = td.BoundarySpec.all_sides(td.Absorber(num_layers=40))
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                        structures=[td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=100 * dl * n / 512), medium=lor)],
                        sources=[td.PointDipole(center=(0, 0, 0), source_time=td.GaussianPulse(freq0=2e14, fwidth=2e13), polarization="Ez")],
                        monitors=[td.FieldTimeMonitor(center=(0.3, 0.2, 0.1), size=(0, 0, 0), name="p", interval=1, colocate=False)] if kind == "pec_probe" else [],
                        boundary_spec=bspec, shutoff=0)
    d = discretize(sim, n_steps=steps)
    d.spec.decay_every = 0
    return d.spec


def run(spec, n, steps, twostep, disp=-1):
    with HipEngine(spec, variant=L.VARIANT_FUSED) as e:
        e.set_option(L.OPT_PLACEMENT_TRIES, 0)
        if twostep == 0:
            e.set_option(L.OPT_TWOSTEP, 0)
        if disp >= 0:
            e.set_option(L.OPT_DISP, disp)
        for c in range(6):
            arr = np.empty((n, n, n), dtype=np.float32)
            for k in range(n):
                arr[k] = np.random.default_rng(c * 100003 + k).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
            e.set_field(c, arr)
        st = e.run(7)          # odd: pairs + a single step, then a run that starts on another parity
        t0 = time.perf_counter()
        st = e.run(steps)
        dt = time.perf_counter() - t0
        f = [e.get_field(c) for c in range(6)]
        return f, e.results(), dt, int(st.fused2_pairs), int(st.disp_pairs), int(st.fused2_off_reason)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [320, 512]
    ok = True
    for n in sizes:
        for kind in ("v3", "v4", "pec", "pec_probe", "absorber"):
            steps = 40
            spec = spec_for(n, steps + 16, kind)
            ref = run(spec, n, steps, 0)
            got = run(spec, n, steps, -1)
            holes = run(spec, n, steps, -1, 0)
            same = all(np.array_equal(a, b) for a, b in zip(ref[0], got[0]))
            same_h = all(np.array_equal(a, b) for a, b in zip(ref[0], holes[0]))
            mons = all(np.array_equal(ref[1][k], got[1][k]) for k in ref[1])
            ok = ok and same and mons and got[4] > 0
            print(json.dumps({"n": n, "kind": kind, "bit_identical": bool(same), "monitors_identical": bool(mons),
                              "round5_form_bit_identical": bool(same_h), "pairs": got[3], "disp_pairs": got[4], "off_reason": got[5],
                              "ms_per_step_single": ref[2] / steps * 1e3, "ms_per_step_pairs": got[2] / steps * 1e3,
                              "ms_per_step_round5_form": holes[2] / steps * 1e3, "round5_pairs": holes[3],
                              "gcells_pairs": n ** 3 * steps / got[2] / 1e9, "gcells_single": n ** 3 * steps / ref[2] / 1e9,
                              "gcells_round5_form": n ** 3 * steps / holes[2] / 1e9}), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
