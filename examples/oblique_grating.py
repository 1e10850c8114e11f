"""A dielectric grating under oblique incidence: Bloch boundaries taken from the source, diffraction orders on both
sides, energy balance.  Same script with `import tidy3d as td` + `tidy3d_amd.adapter.install(td)` where tidy3d is
installed; here the tidy3d-free mirror classes are used.

    python examples/oblique_grating.py            # needs an MI355X and the built library
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # run from a checkout
import tidy3d_amd
import tidy3d_amd.schema as td

f0, period = 2e14, 2.4                                   # 1.5 um light on a 2.4 um period
pulse = td.GaussianPulse(freq0=f0, fwidth=1e13)
source = td.PlaneWave(center=(0, 0, -1.5), size=(td.inf, td.inf, 0), source_time=pulse, direction="+",
                      angle_theta=np.deg2rad(20), angle_phi=0.0, pol_angle=np.pi / 2)          # s polarisation
bar = td.Structure(geometry=td.Box(center=(0.3, 0, 0), size=(1.0, td.inf, 0.6)), medium=td.Medium(permittivity=4.0))
plane = dict(size=(td.inf, td.inf, 0), freqs=[f0])
sim = td.Simulation(
    size=(period, 0, 4.0), grid_spec=td.GridSpec.auto(min_steps_per_wvl=30), run_time=6e-13, structures=[bar],
    sources=[source],
    monitors=[td.DiffractionMonitor(center=(0, 0, 1.2), name="T", **plane),
              td.DiffractionMonitor(center=(0, 0, -1.8), name="R", normal_dir="-", **plane)],
    boundary_spec=td.BoundarySpec(x=td.Boundary.bloch_from_source(source, period, 0), y=td.Boundary.periodic(),
                                  z=td.Boundary.pml()))

data = tidy3d_amd.run(sim, task_name="oblique_grating", verbose=False)
incident = np.cos(source.angle_theta) * period            # 1 W/um^2 along the beam, cell area = period x 1
for name in ("T", "R"):
    d = data[name]
    for m, p, th in zip(d.orders_x, d.power.values[:, 0, 0] / incident, np.degrees(d.angles[0].values[:, 0, 0])):
        print(f"{name} order {int(m):+d}: {100 * p:6.2f} %  at {th:6.2f} deg")
print("R + T =", float((data["T"].power.values.sum() + data["R"].power.values.sum()) / incident))
