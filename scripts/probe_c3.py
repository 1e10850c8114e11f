"""Kernel breakdown of a BASELINE config-3-like workload: 400 x 200 x 800 cells + 12 PML layers per
face (424 x 224 x 824), Si strip in oxide, dipole source (the mode source / monitors add nothing per
step worth profiling)."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import tidy3d_amd.schema as td
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dl = 0.01
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
    sim = td.Simulation(
        size=(4.0 - 1e-6, 2.0 - 1e-6, 8.0 - 1e-6), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12, subpixel=False,
        medium=td.Medium(permittivity=1.44 ** 2),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.45, 0.22, td.inf)),
                                 medium=td.Medium(permittivity=3.48 ** 2))],
        sources=[td.PointDipole(center=(0, 0, -3.5), source_time=pulse, polarization="Ex")], monitors=[],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=12)), shutoff=0)
    sp = discretize(sim, n_steps=steps + 40).spec
    sp.decay_every = 0
    n = sp.shape[0] * sp.shape[1] * sp.shape[2]
    # optional: cyclic axis shifts to compare (each its own engine, all kept alive; the list is walked twice)
    shifts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [None]
    held = []
    for rnd in range(2 if len(shifts) > 1 else 1):
        for sh in shifts:
            e = HipEngine(sp, axis_shift=sh)
            held.append(e)
            import os
            from tidy3d_amd import lib as L
            for key, env in ((L.OPT_TWOSTEP, "C3_TWOSTEP"), (L.OPT_SHELL_PAIRS, "C3_SHELL"), (L.OPT_SHELL2, "C3_SHELL2"), (L.OPT_SHELL2_SHAPE, "C3_SHELL2_SHAPE")):
                if os.environ.get(env):
                    e.set_option(key, int(os.environ[env]))
            e.run(30)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                st = e.run(steps)
                ts.append(time.perf_counter() - t0)
            dt = sorted(ts)[1]
            print(json.dumps({"shape": sp.shape, "axis_shift": e.axis_shift, "device_shape": list(e.spec.shape), "ms_per_step": dt / steps * 1e3,
                              "mcells_per_s": n * steps / dt / 1e6,
                              "fused2_pairs": int(st.fused2_pairs), "shell_pairs": int(st.shell_pairs), "shell2_pairs": int(st.shell2_pairs), "why": int(st.fused2_off_reason)}), flush=True)
    for e in held:
        e.close()


if __name__ == "__main__":
    main()
