"""Launch-overhead probe: BASELINE config 2 (200^3 vacuum, PEC, dipole, point FieldTimeMonitor) and
smaller cubes — wall time per step vs the kernel time the profiler sees."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import tidy3d_amd.schema as td
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine


def main():
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    pml = len(sys.argv) > 3 and sys.argv[3] == "pml"
    split = int(sys.argv[4]) if len(sys.argv) > 4 else -1          # FDTD_OPT_PML_SPLIT
    for n in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "200,128,64").split(",")]:
        dl = 0.05
        pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
        sim = td.Simulation(size=(n * dl,) * 3, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                            sources=[td.PointDipole(center=(0.1, 0.2, 0.3), source_time=pulse, polarization="Ez")],
                            monitors=[td.FieldTimeMonitor(center=(0.4, 0.3, 0.2), size=(0, 0, 0), name="probe")],
                            boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=12) if pml else td.PECBoundary()),
                            shutoff=0)
        sp = discretize(sim, n_steps=steps + 100).spec
        sp.decay_every = 0
        with HipEngine(sp) as e:
            from tidy3d_amd import lib as L
            e.set_option(L.OPT_PML_SPLIT, split)
            e.run(100)
            t0 = time.perf_counter()
            e.run(steps)
            dt = time.perf_counter() - t0
        n = sp.shape[0]
        print(json.dumps({"n": n, "pml": pml, "split": split, "us_per_step": dt / steps * 1e6, "mcells_per_s": n ** 3 * steps / dt / 1e6}), flush=True)


if __name__ == "__main__":
    main()
