#!/usr/bin/env python
"""Golden R / T / A of BASELINE config[4]'s unit cell from the fp64 oracle (TEST INFRASTRUCTURE: the generating script of
tests/golden/config5_unit_cell_oracle.json).  The 1024 x 1024 x 256 array of tests/test_gpu_parity.py::test_config5_* is 16 x 16
copies of one 64 x 64 x 256 cell (one Au disc on the glass half-space, periodic x / y, CPML z, plane wave): the oracle
(oracle/fdtd_numpy.py: its own coefficient derivation, Lorentz reciprocity to 1e-11) runs that cell in fp64 for the number of
steps the full-size run takes — the GPU test then holds the full-size R, T, A to these numbers to 1e-3.
  python scripts/make_config5_unit_cell_golden.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import tidy3d_amd.schema as td  # noqa: E402
from tidy3d_amd.data import assemble  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402


def config5_sim(n_cells_xy=64, shutoff=1e-4):
    """The stack of test_config5_au_nanoparticle_array_1024x1024x256 over n_cells_xy / 64 periods per side."""
    from cases import gold_johnson_christy
    dl = 0.005
    nz = 256 - 24
    pitch = 64 * dl
    au = gold_johnson_christy()
    f0 = 5e14
    pulse = td.GaussianPulse(freq0=f0, fwidth=1e14)
    L, Lz = n_cells_xy * dl, nz * dl
    m = n_cells_xy // 64
    discs = [td.Structure(geometry=td.Cylinder(center=(-L / 2 + (i + 0.5) * pitch, -L / 2 + (j + 0.5) * pitch, 0.0),
                                               radius=0.08, length=0.04, axis=2), medium=au)
             for i in range(m) for j in range(m)]
    slab = td.Structure(geometry=td.Box(center=(0, 0, -Lz / 4 - 0.02), size=(td.inf, td.inf, Lz / 2)), medium=td.Medium(permittivity=2.1))
    plane = (td.inf, td.inf, 0)
    return td.Simulation(
        size=(L, L, Lz), grid_spec=td.GridSpec.uniform(dl=dl), run_time=6e-14, structures=[slab] + discs,
        sources=[td.PlaneWave(center=(0, 0, Lz / 2 - 0.1), size=plane, source_time=pulse, direction="-")],
        monitors=[td.FluxMonitor(center=(0, 0, Lz / 2 - 0.05), size=plane, freqs=[f0], name="R"),
                  td.FluxMonitor(center=(0, 0, -Lz / 2 + 0.1), size=plane, freqs=[f0], name="T"),
                  td.FieldMonitor(center=(0, 0, 0.03), size=(td.inf, 0, 0), freqs=[f0], name="line", fields=["Ex"], colocate=False)],
        boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml()), shutoff=shutoff)


def rta(sd, L):
    area = L * L
    R = float(sd["R"].flux.values[0]) / area
    T = -float(sd["T"].flux.values[0]) / area
    return R, T, 1 - R - T


def main():
    from oracle.fdtd_numpy import OracleFdtd
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "config5_unit_cell_oracle.json")
    sim = config5_sim(64, shutoff=0)            # the whole run_time: the full-size run's shutoff (1e-4 of the peak energy) cuts what is below the tolerance
    disc = discretize(sim)
    assert disc.spec.shape == (64, 64, 256), disc.spec.shape
    t0 = time.time()
    o = OracleFdtd(disc.spec)
    raw = o.run()
    sd = assemble(disc, raw, log="")
    R, T, A = rta(sd, 64 * 0.005)
    rec = {"what": "BASELINE config[4] unit cell (64 x 64 x 256, one Au disc, periodic x / y, CPML z) through oracle/fdtd_numpy.py in fp64",
           "made_by": "scripts/make_config5_unit_cell_golden.py", "steps": int(disc.spec.n_steps), "dt": float(disc.spec.dt),
           "R": R, "T": T, "A": A, "oracle_seconds": time.time() - t0}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
