#!/usr/bin/env python
"""Step pairs with a point FieldTimeMonitor recording every step, 512^3 dielectric / Lorentz sphere in PEC walls (round 6:
the pair is SLOWER than without the probe — what costs?).  python scripts/probe_mon_pairs.py [n] [kind] [mon] [twostep]"""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tidy3d_amd.schema as td
from tidy3d_amd.discretize import discretize
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kinds = [sys.argv[2]] if len(sys.argv) > 2 else ["vac", "diel", "lor"]
mons = [int(sys.argv[3])] if len(sys.argv) > 3 else [0, 1, 2]
twos = [int(sys.argv[4])] if len(sys.argv) > 4 else [-1, 0]
dl = 0.05
for kind in kinds:
    med = td.Medium(permittivity=4.0) if kind == "diel" else td.Lorentz(eps_inf=2.0, coeffs=[(2.0, 4e14, 2e13)])
    for mon in mons:
        ms = []
        if mon == 1:
            ms = [td.FieldTimeMonitor(center=(0.3, 0.2, 0.1), size=(0, 0, 0), name="p", interval=1, colocate=False)]
        if mon == 2:
            ms = [td.FieldTimeMonitor(center=(0.3, 0.2, 0.1), size=(0, 0, 0), name="p", interval=1, colocate=False, fields=["Ez"])]
        sim = td.Simulation(size=(n * dl - 1e-6 * dl,) * 3, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                            structures=[] if kind == "vac" else [td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=100 * dl * n / 512), medium=med)],
                            sources=[td.PointDipole(center=(0, 0, 0), source_time=td.GaussianPulse(freq0=2e14, fwidth=2e13), polarization="Ez")],
                            monitors=ms, boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()), shutoff=0)
        d = discretize(sim, n_steps=120); d.spec.decay_every = 0
        for two in twos:
            with HipEngine(d.spec, variant=L.VARIANT_FUSED) as e:
                e.set_option(L.OPT_PLACEMENT_TRIES, 0)
                if two == 0: e.set_option(L.OPT_TWOSTEP, 0)
                e.run(10)
                t0 = time.perf_counter(); st = e.run(40); dt = time.perf_counter() - t0
                e.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
                st = e.run(20)
                print(json.dumps({"kind": kind, "mon": mon, "twostep": two, "ms_per_step": dt / 40 * 1e3, "pairs": int(st.fused2_pairs), "disp": int(st.disp_pairs),
                                  "fused_ms_per_launch": st.fused_kernel_ms / max(1, st.fused_kernel_launches), "launches": int(st.fused_kernel_launches), "shape": int(st.fused2_shape)}), flush=True)
