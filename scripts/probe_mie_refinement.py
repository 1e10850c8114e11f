#!/usr/bin/env python
"""sigma_sca / Mie series - 1 of the config-4 problem (same sphere, box, TFSF source, flux box) at several resolutions:
  python scripts/probe_mie_refinement.py 20 30 40 50"""
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tidy3d_amd.schema as td  # noqa: E402
from tidy3d_amd.analytic import mie_cross_sections  # noqa: E402
from tidy3d_amd.constants import C_0  # noqa: E402
from tidy3d_amd.data import assemble  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

lam0 = 1.0
f0 = C_0 / lam0
nfreq = int(os.environ.get("NFREQ", "7"))
freqs = [float(v) * f0 for v in np.linspace(0.85, 1.15, nfreq)]
r, eps = 56 * lam0 / 40, 2.56
_, ana = mie_cross_sections(r, eps, freqs)
subpixel = os.environ.get("SUBPIXEL", "1") != "0"
for ppw in [int(a) for a in sys.argv[1:]] or [20, 40]:
    dl = lam0 / ppw
    n = (512 - 24) * ppw // 40
    box = 2 * r + lam0
    sim = td.Simulation(
        size=(n * dl,) * 3, grid_spec=td.GridSpec.uniform(dl=dl), run_time=float(os.environ.get("RUN_PERIODS", "70")) / f0,
        structures=[td.Structure(geometry=td.Sphere(radius=r), medium=td.Medium(permittivity=eps))],
        sources=[td.TFSF(center=(0, 0, 0), size=(box,) * 3, source_time=td.GaussianPulse(freq0=f0, fwidth=f0 / 6), injection_axis=2, direction="+")],
        monitors=[td.FluxMonitor(center=(0, 0, 0), size=(box + lam0,) * 3, freqs=freqs, name="sca")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=12)), shutoff=1e-5, subpixel=subpixel)
    disc = discretize(sim)
    t0 = time.time()
    with HipEngine(disc.spec) as e:
        st = e.run()
        raw = e.results()
    got = assemble(disc, raw, log="")["sca"].flux.values
    print(json.dumps({"ppw": ppw, "shape": list(disc.spec.shape), "steps": int(st.steps_done), "solve_s": time.time() - t0, "subpixel": subpixel, "run_periods": float(os.environ.get("RUN_PERIODS", "70")),
                      "f_over_f0": [f / f0 for f in freqs], "sigma_sca": [float(v) for v in got], "mie": [float(v) for v in ana],
                      "sigma_over_mie_minus_1": [float(v) for v in got / ana - 1]}), flush=True)
