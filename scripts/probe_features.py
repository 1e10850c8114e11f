"""Throughput of the paths added after the headline kernels: Bloch boundaries (complex fields = a (Re, Im)
pair of solvers on the two-pass kernels + fix-up kernels) and Absorber layers (damping kernel), on a
256 x 256 x 320-cell periodic-array set-up (Bloch / periodic in x, y; 40-layer absorber or 12-layer CPML in z).
One JSON line per case: ms per step and Mcells/s of REAL cells (a complex cell counts once)."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import tidy3d_amd.schema as td
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine


def case(name, bx, by, bz, steps, lxy=2.56, lz=2.4, axis_shift=None, zchunk=0):
    dl = 0.01
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
    sim = td.Simulation(size=(lxy - 1e-6, lxy - 1e-6, lz - 1e-6), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                        subpixel=False,
                        structures=[td.Structure(geometry=td.Cylinder(radius=0.6, length=0.3, axis=2),
                                                 medium=td.Medium(permittivity=6.0))],
                        sources=[td.PointDipole(center=(0.1, 0.2, -0.8), source_time=pulse, polarization="Ex")], monitors=[],
                        boundary_spec=td.BoundarySpec(x=bx, y=by, z=bz), shutoff=0)
    sp = discretize(sim, n_steps=steps + 40).spec
    sp.decay_every = 0
    with HipEngine(sp, axis_shift=axis_shift) as e:
        shift = e.axis_shift
        if zchunk:
            from tidy3d_amd import lib as L
            e.set_option(L.OPT_ZCHUNK, zchunk)
        e.run(20)
        t0 = time.perf_counter()
        e.run(steps)
        dt = time.perf_counter() - t0
    n = sp.shape[0] * sp.shape[1] * sp.shape[2]
    print(json.dumps({"case": name, "shape": sp.shape, "complex": sp.bloch is not None, "axis_shift": shift, "zchunk": zchunk, "ms_per_step": dt / steps * 1e3,
                      "mcells_per_s": n * steps / dt / 1e6}), flush=True)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    per, pml, ab = td.Boundary.periodic(), td.Boundary.pml(), td.Boundary.absorber()
    case("periodic_xy_pml_z (fused sweep)", per, per, pml, steps)
    case("periodic_xy_absorber_z (fused sweep + damp_kernel)", per, per, ab, steps)
    case("bloch_xy_pml_z (solver pair on the ghost-cell layout)", td.Boundary.bloch(0.21), td.Boundary.bloch(0.13), pml, steps)
    case("bloch_xy_absorber_z", td.Boundary.bloch(0.21), td.Boundary.bloch(0.13), ab, steps)
    # 248 real cells + 2 ghost cells stay inside one 256-cell x tile of the fused sweep
    case("periodic_xy_pml_z 248", per, per, pml, steps, 2.48)
    case("bloch_xy_pml_z 248", td.Boundary.bloch(0.21), td.Boundary.bloch(0.13), pml, steps, 2.48)


def narrow(steps):
    """A unit cell of 64 x 64 cells, 1024 + 24 planes tall: the axes as given (x = 64 cells: a quarter of the
    lanes busy) against the cyclic renaming the engine picks by default (z along x)."""
    per, pml = td.Boundary.periodic(), td.Boundary.pml()
    case("narrow 64x64x1048 as given", per, per, pml, steps, lxy=0.64, lz=10.24, axis_shift=0)
    case("narrow 64x64x1048 renamed (default)", per, per, pml, steps, lxy=0.64, lz=10.24)
    for zc in (8, 4, 2):
        case("narrow renamed, z-chunk %d" % zc, per, per, pml, steps, lxy=0.64, lz=10.24, zchunk=zc)
        case("narrow as given, z-chunk %d" % zc, per, per, pml, steps, lxy=0.64, lz=10.24, axis_shift=0, zchunk=zc)
    for zc in (16, 8, 4):
        case("128^3-ish cube, z-chunk %d" % zc, per, per, pml, steps, lxy=1.28, lz=1.04, zchunk=zc)
    case("narrow bloch as given", td.Boundary.bloch(0.21), td.Boundary.bloch(0.13), pml, steps, lxy=0.64, lz=10.24, axis_shift=0)
    case("narrow bloch renamed (default)", td.Boundary.bloch(0.21), td.Boundary.bloch(0.13), pml, steps, lxy=0.64, lz=10.24)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "cube":
        per, pml = td.Boundary.periodic(), td.Boundary.pml()
        case("128^3-ish cube, default shape choice", per, per, pml, int(sys.argv[1]), lxy=1.28, lz=1.04)
        case("128^3-ish cube, z-chunk 16", per, per, pml, int(sys.argv[1]), lxy=1.28, lz=1.04, zchunk=16)
        case("200^3-ish cube, default shape choice", per, per, pml, int(sys.argv[1]), lxy=2.0, lz=1.76)
    elif len(sys.argv) > 2 and sys.argv[2] == "narrow":
        narrow(int(sys.argv[1]))
    else:
        main()
