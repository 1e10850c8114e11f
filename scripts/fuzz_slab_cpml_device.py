"""CPML slab-rank pairs on the device: random boxes with layers on x / y and a periodic z, run as ONE z-slab rank whose RCCL exchange
is looped back to itself (HipEngine force_comm; FDTD_OPT_PML_FUSED = 7, step pairs forced) against the plain one-GPU run of the
same problem in single steps — fields and records, bit for bit.
    python scripts/fuzz_slab_cpml_device.py [n_cases] [seed] [FDTD_OPT_SLAB_BOXES_FIRST value]"""
import os
import sys

import numpy as np
import torch  # noqa: F401  (before the solver library: one HIP runtime per process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tidy3d_amd.schema as td  # noqa: E402
from cases import DL, PULSE  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402


def draw(rng):
    N = (int(rng.integers(20, 52)) * 4, int(rng.integers(40, 120)), int(rng.integers(40, 140)))
    layers = [int(rng.integers(3, 13)), int(rng.integers(3, 13))]
    if rng.random() < 0.3:
        layers[int(rng.integers(0, 2))] = 0

    def bnd(a):
        if layers[a]:
            return td.Boundary.pml(num_layers=layers[a]) if rng.random() < 0.7 else td.Boundary.stable_pml(num_layers=layers[a])
        return td.Boundary(minus=td.PMCBoundary() if rng.random() < 0.4 else td.PECBoundary(), plus=td.PECBoundary())
    bspec = td.BoundarySpec(x=bnd(0), y=bnd(1), z=td.Boundary.periodic())
    size = tuple(n * DL for n in N)
    structures = []
    for _ in range(int(rng.integers(0, 4))):
        c = tuple(float(rng.uniform(-0.4, 0.4) * s) for s in size)
        sz = tuple(float(rng.uniform(0.1, 0.5) * s) if rng.random() < 0.8 else td.inf for s in size)
        med = td.PEC if rng.random() < 0.25 else td.Medium(permittivity=float(rng.uniform(1.5, 4.0)), conductivity=float(rng.choice([0.0, 0.02])))
        structures.append(td.Structure(geometry=td.Box(center=c, size=sz), medium=med))
    sources = []
    wild = rng.random() < 0.25
    for _ in range(int(rng.integers(1, 4))):
        f = 0.45 if wild else 0.2
        c = tuple(float(rng.uniform(-f, f) * s) for s in size)
        sources.append(td.PointDipole(center=c, source_time=PULSE, polarization=str(rng.choice(["Ex", "Ey", "Ez", "Hx", "Hy", "Hz"]))))
    monitors = [td.FieldTimeMonitor(center=tuple(float(rng.uniform(-0.3, 0.3) * s) for s in size), size=(0, 0, 0), name="probe",
                                    interval=int(rng.integers(3, 12)), colocate=False),
                td.FieldMonitor(center=(0, 0, 0), size=(td.inf, 0, td.inf), name="plane", freqs=[3e14], fields=["Ex", "Hy"], colocate=False)]
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, structures=structures, sources=sources,
                        monitors=monitors, boundary_spec=bspec, shutoff=0)
    twostep = int(rng.choice([5, 6, 8, 12, 16])) + 64 * int(rng.integers(3, 33))
    return sim, twostep, int(rng.integers(30, 70))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    boxes = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    L.load_library()
    bad = in_pairs = 0
    for q in range(n_cases):
        sim, twostep, steps = draw(rng)
        disc = discretize(sim, n_steps=steps)
        disc.spec.decay_every = int(rng.choice([0, 10, 16]))
        with HipEngine(disc.spec, axis_shift=0) as e:
            e.set_option(L.OPT_TWOSTEP, 0)
            e.run()
            ref, ref_f = e.results(), [e.get_field(c) for c in range(6)]
        with HipEngine(disc.spec, force_comm=True) as e:
            e.comm_init(e.unique_id())
            e.set_option(L.OPT_PML_FUSED, 7)
            e.set_option(L.OPT_TWOSTEP, twostep)
            if boxes >= 0:
                e.set_option(L.OPT_SLAB_BOXES_FIRST, boxes)
            st = e.run()
            got, got_f = e.results(), [e.get_field(c) for c in range(6)]
        diff = [c for c in range(6) if not np.array_equal(got_f[c], ref_f[c])] + [k for k, v in ref.items() if not np.array_equal(np.asarray(got[k]), np.asarray(v))]
        in_pairs += int(st.shell2_pairs) > 0
        print(f"case {q}: shape={disc.spec.shape} twostep={twostep & 63}x{twostep >> 6} steps={steps} pairs={int(st.fused2_pairs)} shell2={int(st.shell2_pairs)} "
              f"why={int(st.fused2_off_reason)} max|F|={max(float(np.abs(f).max()) for f in ref_f):.2e} -> {'ok' if not diff else 'DIFFERS: ' + str(diff)}", flush=True)
        bad += bool(diff)
    print(f"fuzz_slab_cpml_device: {n_cases - bad} of {n_cases} cases bit-identical; {in_pairs} took CPML slab pairs")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
