"""An open scattering problem without CPML: a dipole next to a dielectric sphere, `Absorber` boundaries on all faces, two nested
closed flux boxes.  Non-dispersive media + absorber layers + point source + DFT monitors is the class of open problems the
two-steps-per-sweep kernel covers: the log says how many step pairs were taken.  Checks: the power through
the two boxes agrees (nothing is absorbed between them), and it differs from the dipole's free-space power by the sphere's
back-action (Purcell factor), which is printed.

    python examples/dipole_sphere_absorber.py        # needs an MI355X and the built library
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # run from a checkout
import tidy3d_amd
import tidy3d_amd.schema as td

f0 = 3e14                                            # 1 um
pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 8)
dipole = td.PointDipole(center=(0, 0, 0.6), source_time=pulse, polarization="Ez")
sphere = td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=0.35), medium=td.Medium(permittivity=6.0))


def flux_box(name, half):
    return td.FluxMonitor(center=(0, 0, 0.15), size=(2 * half,) * 3, freqs=[f0], name=name)


def run(structures):
    sim = td.Simulation(size=(4.0, 4.0, 4.0), grid_spec=td.GridSpec.uniform(dl=1 / 40), run_time=1.2e-13, structures=structures,
                        sources=[dipole], monitors=[flux_box("inner", 0.9), flux_box("outer", 1.3)],
                        boundary_spec=td.BoundarySpec.all_sides(td.Absorber(num_layers=40)), shutoff=1e-5)
    data = tidy3d_amd.run(sim, task_name="dipole", verbose=False)
    for line in data.log.splitlines():
        if "Two time steps" in line or "Time-stepping speed" in line or "grid points" in line:
            print("   ", line)
    return float(data["inner"].flux.values[0]), float(data["outer"].flux.values[0])


print("vacuum:")
p_in0, p_out0 = run([])
print(f"    power through the boxes: {p_in0:.5f} / {p_out0:.5f}  (ratio {p_out0 / p_in0:.4f})")
print("with the sphere:")
p_in, p_out = run([sphere])
print(f"    power through the boxes: {p_in:.5f} / {p_out:.5f}  (ratio {p_out / p_in:.4f})")
print(f"Purcell factor of the dipole 0.25 um above a r = 0.35 um, eps = 6 sphere at 1 um: {p_out / p_out0:.3f}")
assert abs(p_out0 / p_in0 - 1) < 0.01 and abs(p_out / p_in - 1) < 0.01
