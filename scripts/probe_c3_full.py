"""BASELINE config 3 itself (Si strip waveguide, ModeSource, flux / mode monitors, 424 x 224 x 824 cells with CPML) under a forced cyclic
axis renaming: whole-run throughput and how many of its steps went out in pairs.
  python scripts/probe_c3_full.py [shift,shift,...]      (None = the engine's own choice)"""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import tidy3d_amd.schema as td
from tidy3d_amd.constants import C_0
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

lam = 1.55
f0 = C_0 / lam
pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 10)
plane = (td.inf, td.inf, 0)
sim = td.Simulation(
    size=(4.0, 2.0, 8.0), grid_spec=td.GridSpec.uniform(dl=0.01), run_time=2.6e-13, medium=td.Medium(permittivity=1.44 ** 2),
    structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.45, 0.22, td.inf)), medium=td.Medium(permittivity=3.48 ** 2))],
    sources=[td.ModeSource(center=(0, 0, -3.5), size=plane, source_time=pulse, direction="+", mode_spec=td.ModeSpec(num_modes=1), mode_index=0)],
    monitors=[td.FluxMonitor(center=(0, 0, 3.0), size=plane, freqs=[f0], name="fwd"), td.FluxMonitor(center=(0, 0, -3.8), size=plane, freqs=[f0], name="bwd"),
              td.ModeMonitor(center=(0, 0, 3.0), size=plane, freqs=[f0], mode_spec=td.ModeSpec(num_modes=1), name="mm"),
              td.FieldMonitor(center=(0, 0, 1.0), size=(0, 0, 0), freqs=[f0], name="p1", fields=["Ex"]),
              td.FieldMonitor(center=(0, 0, 2.0), size=(0, 0, 0), freqs=[f0], name="p2", fields=["Ex"])],
    boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=12)), shutoff=1e-5)
disc = discretize(sim)
shifts = [None if v == "None" else int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["None", "0"])]
for sh in shifts:
    with HipEngine(disc.spec, axis_shift=sh) as e:
        t0 = time.perf_counter()
        st = e.run()
        dt = time.perf_counter() - t0
        print(json.dumps({"axis_shift": e.axis_shift, "device_shape": list(e.spec.shape), "steps": int(st.steps_done), "fused2_pairs": int(st.fused2_pairs),
                          "shell2_pairs": int(st.shell2_pairs), "why": int(st.fused2_off_reason), "run_ms": st.run_ms, "wall_s": dt,
                          "mcells_per_s": disc.spec.n_cells * st.steps_done / (st.run_ms * 1e-3) / 1e6}), flush=True)
