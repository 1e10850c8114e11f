"""A tilted Gaussian beam focused 2 um in front of its source plane: beam radius through the focus against
w(z) = w0 sqrt(1 + (z / zR)^2), carried power against pi w0^2 / 2 (peak intensity 1 W/um^2 at the waist).

    python examples/gaussian_beam_focus.py        # needs an MI355X and the built library
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # run from a checkout
import tidy3d_amd
import tidy3d_amd.schema as td

f0 = 3e14                                            # 1 um
w0, focus, theta = 1.5, 2.0, np.deg2rad(8)
pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 10)
beam = td.GaussianBeam(center=(-0.3, 0, -2.0), size=(td.inf, td.inf, 0), source_time=pulse, direction="+",
                       waist_radius=w0, waist_distance=-focus, angle_theta=theta, angle_phi=0.0, pol_angle=np.pi / 2)
planes = [-1.0, 0.0, 1.0, 2.0]
sim = td.Simulation(
    size=(10, 10, 5), grid_spec=td.GridSpec.uniform(dl=1 / 16), run_time=1.2e-13, sources=[beam],
    monitors=[td.FieldMonitor(center=(0, 0, z), size=(td.inf, td.inf, 0), freqs=[f0], name=f"z{i}")
              for i, z in enumerate(planes)] + [td.FluxMonitor(center=(0, 0, 2.2), size=(td.inf, td.inf, 0), freqs=[f0], name="P")],
    boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=10)))

data = tidy3d_amd.run(sim, task_name="beam", verbose=False)
zr = np.pi * w0 ** 2                                  # Rayleigh range at 1 um
for i, z in enumerate(planes):
    fd = data[f"z{i}"]
    inten = np.abs(fd.Ex.values[:, :, 0, 0]) ** 2 + np.abs(fd.Ey.values[:, :, 0, 0]) ** 2
    X, Y = np.meshgrid(np.asarray(fd.Ex.coords["x"]), np.asarray(fd.Ex.coords["y"]), indexing="ij")
    cx = (inten * X).sum() / inten.sum()
    wy = 2 * np.sqrt((inten * Y ** 2).sum() / inten.sum())
    along = (z + 2.0) / np.cos(theta) - focus            # distance from the waist along the beam axis
    print(f"z = {z:+.1f}: beam centre x = {cx:+.3f} (axis: {-0.3 + np.tan(theta) * (z + 2.0):+.3f}),"
          f"  radius w_y = {wy:.3f} (paraxial: {w0 * np.sqrt(1 + (along / zr) ** 2):.3f})")
print(f"power through z = 2.2: {float(data['P'].flux.values[0]):.3f} W  (pi w0^2 / 2 = {np.pi * w0 ** 2 / 2:.3f})")
