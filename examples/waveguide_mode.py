"""The classic first simulation: launch the fundamental mode of a silicon strip waveguide and read it back
further down — modal transmission |a0|^2, total flux, nothing going backwards.

    python examples/waveguide_mode.py             # needs an MI355X and the built library
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # run from a checkout
import tidy3d_amd
import tidy3d_amd.schema as td

f0 = 299792458e6 / 1.55                                # 1.55 um
pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 15)
strip = td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.45, 0.22)), medium=td.Medium(permittivity=3.48 ** 2))
plane = dict(size=(0, 2.0, 1.6))
mode_spec = td.ModeSpec(num_modes=2)
sim = td.Simulation(
    size=(6.0, 2.6, 2.2), grid_spec=td.GridSpec.auto(min_steps_per_wvl=14), run_time=2.5e-13,
    medium=td.Medium(permittivity=1.44 ** 2), structures=[strip],
    sources=[td.ModeSource(center=(-2.0, 0, 0), source_time=pulse, mode_spec=mode_spec, mode_index=0, direction="+", **plane)],
    monitors=[td.ModeMonitor(center=(2.0, 0, 0), freqs=[f0], mode_spec=mode_spec, name="modes", **plane),
              td.FluxMonitor(center=(2.0, 0, 0), freqs=[f0], name="through", **plane),
              td.FluxMonitor(center=(-2.5, 0, 0), freqs=[f0], name="behind", **plane)],
    boundary_spec=td.BoundarySpec.all_sides(td.PML()))

data = tidy3d_amd.run(sim, task_name="waveguide", verbose=False)
amps = data["modes"].amps
a_fwd = amps.values[list(amps.coords["direction"]).index("+"), 0, :]
print("n_eff of the two modes:", np.round(np.real(data["modes"].n_complex.values[0]), 4))
print("forward modal powers |a|^2:", np.round(np.abs(a_fwd) ** 2, 4))
print("flux through the far plane: %.4f W,  behind the source: %.2e W" % (float(data["through"].flux.values[0]),
                                                                       float(data["behind"].flux.values[0])))
