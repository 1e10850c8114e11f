#!/usr/bin/env python
"""BASELINE config 5's grid (1024 x 1024 x 256, periodic x / y, CPML z, plane wave, flux planes) WITHOUT the dispersive discs: what
step pairs could buy that configuration if dispersive bodies rode the shell (python scripts/probe_c5.py [steps])."""
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tidy3d_amd.schema as td  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dl = 0.005
nxy, nz = 1024, 256 - 24
f0 = 5e14
pulse = td.GaussianPulse(freq0=f0, fwidth=1e14)
Lx, Lz = nxy * dl, nz * dl
slab = td.Structure(geometry=td.Box(center=(0, 0, -Lz / 4 - 0.02), size=(td.inf, td.inf, Lz / 2)), medium=td.Medium(permittivity=2.1))
plane = (td.inf, td.inf, 0)
sim = td.Simulation(size=(Lx, Lx, Lz), grid_spec=td.GridSpec.uniform(dl=dl), run_time=6e-14, structures=[slab],
                    sources=[td.PlaneWave(center=(0, 0, Lz / 2 - 0.1), size=plane, source_time=pulse, direction="-")],
                    monitors=[td.FluxMonitor(center=(0, 0, Lz / 2 - 0.05), size=plane, freqs=[f0], name="R"),
                              td.FluxMonitor(center=(0, 0, -Lz / 2 + 0.1), size=plane, freqs=[f0], name="T")],
                    boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml()), shutoff=0)
disc = discretize(sim)
spec = disc.spec
n_src = max([len(s.wave_e) for s in spec.sources] + [len(t.wave) for t in spec.tfsf] + [0])
print(json.dumps({"shape": list(spec.shape), "n_steps": spec.n_steps, "sources_end_at": n_src, "tfsf": len(spec.tfsf)}), flush=True)
L.load_library()
with HipEngine(spec, axis_shift=0) as e:
    e.run(n_src + 20)                       # past the pulse
    for ts in (0, -1, 0, -1):
        e.set_option(L.OPT_TWOSTEP, ts)
        e.run(20)
        t0 = time.perf_counter()
        st = e.run(steps)
        dt = time.perf_counter() - t0
        print(json.dumps({"twostep": ts, "ms_per_step": dt / steps * 1e3, "gcells_per_s": np.prod(spec.shape) * steps / dt / 1e9,
                          "pairs": int(st.fused2_pairs), "shell_pairs": int(st.shell_pairs), "why": int(st.fused2_off_reason)}), flush=True)
