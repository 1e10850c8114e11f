#!/usr/bin/env python
"""BASELINE config 3 (Si strip 424 x 224 x 824 with its ModeSource) WHILE the mode plane injects: Gcells/s per layout (cyclic axis
renaming) with the plane's terms as paged source terms inside the pairs (round 6) and as in round 5 (z hole / single steps).
    python scripts/probe_c3_paged.py [steps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tidy3d_amd.schema as td  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.constants import C_0  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
f0 = C_0 / 1.55
pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 10)
plane = (td.inf, td.inf, 0)
sim = td.Simulation(
    size=(4.0, 2.0, 8.0), grid_spec=td.GridSpec.uniform(dl=0.01), run_time=2.6e-13, medium=td.Medium(permittivity=1.44 ** 2),
    structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.45, 0.22, td.inf)), medium=td.Medium(permittivity=3.48 ** 2))],
    sources=[td.ModeSource(center=(0, 0, -3.5), size=plane, source_time=pulse, direction="+", mode_spec=td.ModeSpec(num_modes=1), mode_index=0)],
    monitors=[td.FluxMonitor(center=(0, 0, 3.0), size=plane, freqs=[f0], name="fwd"), td.FluxMonitor(center=(0, 0, -3.8), size=plane, freqs=[f0], name="bwd"),
              td.ModeMonitor(center=(0, 0, 3.0), size=plane, freqs=[f0], mode_spec=td.ModeSpec(num_modes=1), name="mm")],
    boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=12)), shutoff=1e-5)
disc = discretize(sim)
disc.spec.decay_every = 0
for shift in (None,):
    for paged in (1, 0):
        with HipEngine(disc.spec, axis_shift=shift) as e:
            e.set_option(L.OPT_SRC_PAGED, paged)
            e.run(60)
            t0 = time.perf_counter()
            st = e.run(steps)
            dt = time.perf_counter() - t0
            e.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
            sk = e.run(20)
            e.set_option(L.OPT_FLAGS, 0)
            print(json.dumps({"bulk_ms_per_launch": sk.fused_kernel_ms / max(1, sk.fused_kernel_launches), "shell_ms_per_launch": sk.shell_kernel_ms / max(1, sk.shell_kernel_launches),
                              "axis_shift": shift, "used_shift": int(e.axis_shift), "device_shape": [int(v) for v in e.spec.shape], "paged": paged,
                              "gcells": disc.spec.n_cells * steps / dt / 1e9, "ms_per_step": dt / steps * 1e3, "pairs": int(st.fused2_pairs),
                              "paged_pairs": int(st.src_paged_pairs), "shell2_pairs": int(st.shell2_pairs), "shape": int(st.fused2_shape)}), flush=True)
            if shift is None and paged == 1:        # the same engine once the list is spent: what pairs without any source term run at on this box
                n_src = max(max(len(sc.wave_e), len(sc.wave_h)) for sc in disc.spec.sources)
                e.run(max(0, n_src + 20 - 60 - steps))
                t0 = time.perf_counter()
                st = e.run(steps)
                dt = time.perf_counter() - t0
                e.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
                sk = e.run(20)
                e.set_option(L.OPT_FLAGS, 0)
                print(json.dumps({"bulk_ms_per_launch": sk.fused_kernel_ms / max(1, sk.fused_kernel_launches), "shell_ms_per_launch": sk.shell_kernel_ms / max(1, sk.shell_kernel_launches),
                                  "axis_shift": shift, "used_shift": int(e.axis_shift), "phase": "list spent", "gcells": disc.spec.n_cells * steps / dt / 1e9,
                                  "ms_per_step": dt / steps * 1e3, "pairs": int(st.fused2_pairs), "paged_pairs": int(st.src_paged_pairs)}), flush=True)
