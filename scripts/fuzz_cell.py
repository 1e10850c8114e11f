#!/usr/bin/env python
"""Randomised check of shell2 pairs on metasurface / grating cells — BASELINE config 5's kind of problem in small: periodic x and / or y,
CPML on z (and on the non-periodic transverse axis), a dispersive body (Drude / Lorentz: its planes are z holes of the bulk, the ADE
memory term goes into the middle step), a lossy substrate, a plane wave or a current sheet (one more hole while it injects), flux
planes that span the wrap (their record steps end pairs), a probe — against single steps of the same library, fields and records, bit
for bit.     python scripts/fuzz_cell.py [cases] [seed] [emu]      (emu: the CPU emulator of tests/hipemu instead of the device)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import tidy3d_amd.schema as td  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

DL = 0.05


def case(rng):
    nx, ny, nz = int(rng.integers(9, 20)) * 4, int(rng.integers(20, 40)), int(rng.integers(56, 80))
    size = tuple((n - 1e-6) * DL for n in (nx, ny, nz))
    hz = 0.5 * size[2]
    pulse = td.GaussianPulse(freq0=3e14, fwidth=float(rng.uniform(1.8e14, 2.6e14)))
    per = td.Boundary.periodic()
    kind = int(rng.integers(0, 3))          # 0: periodic x and y (plane wave), 1: periodic x + layers on y, 2: periodic y + layers on x
    lay = lambda: td.Boundary.pml(num_layers=int(rng.integers(2, 6)))      # noqa: E731
    bspec = td.BoundarySpec(x=per if kind in (0, 1) else lay(), y=per if kind in (0, 2) else lay(), z=td.Boundary.pml(num_layers=int(rng.integers(3, 7))))
    zc = float(rng.uniform(-0.25, 0.05) * hz)
    med = td.Drude(eps_inf=2.0, coeffs=[(1.2e15, 9e13)]) if rng.random() < 0.6 else td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])
    body = td.Cylinder(center=(0.1, 0.05, zc), radius=float(rng.uniform(0.2, 0.4)), length=float(rng.uniform(0.1, 0.35)), axis=2) if rng.random() < 0.7 \
        else td.Box(center=(0, 0, zc), size=(td.inf, 0.4, 0.2))
    structures = [td.Structure(geometry=td.Box(center=(0, 0, -0.75 * hz), size=(td.inf, td.inf, 0.5 * hz)), medium=td.Medium(permittivity=2.1, conductivity=float(rng.choice([0.0, 0.01])))),
                  td.Structure(geometry=body, medium=med)]
    zs = float(rng.uniform(0.45, 0.55) * hz)
    if kind == 0 and rng.random() < 0.7:
        srcs = [td.PlaneWave(center=(0, 0, zs), size=(td.inf, td.inf, 0), source_time=pulse, direction="-")]
    else:
        srcs = [td.UniformCurrentSource(center=(0, 0, zs), size=(td.inf, td.inf, 0), source_time=pulse, polarization=str(rng.choice(["Ex", "Ey"])))]
    mons = [td.FieldTimeMonitor(center=(0.1, 0.05, 0.2 * hz), size=(0, 0, 0), name="probe", interval=int(rng.integers(2, 6)), colocate=False),
            td.FluxMonitor(center=(0, 0, 0.3 * hz), size=(td.inf, td.inf, 0), freqs=[2.5e14, 3e14], name="R"),
            td.FluxMonitor(center=(0, 0, -0.55 * hz), size=(td.inf, td.inf, 0), freqs=[3e14], name="T")]
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=float(rng.uniform(0.9e-14, 1.3e-14)), structures=structures, sources=srcs,
                        monitors=mons, boundary_spec=bspec, shutoff=0)
    disc = discretize(sim)
    disc.spec.decay_every = 0
    return disc.spec, f"kind={kind} {type(med).__name__} {type(srcs[0]).__name__}"


def run(spec, lib, twostep, split):
    with HipEngine(spec, lib=lib, variant=L.VARIANT_FUSED, axis_shift=0) as e:
        e.set_option(L.OPT_TWOSTEP, twostep)
        e.set_option(L.OPT_SHELL_PAIRS, 1)
        e.set_option(L.OPT_SHELL2, 1 if twostep else 0)
        pairs = s2 = 0
        for r in (split, spec.n_steps - split):
            st = e.run(r)
            pairs += int(st.fused2_pairs)
            s2 += int(st.shell2_pairs)
        return [e.get_field(c) for c in range(6)], e.results(), pairs, s2


def run_cases(n_cases, seed=1, lib=None, quiet=False):
    """-> (cases that differ, cases that took shell2 pairs)"""
    rng = np.random.default_rng(seed)
    bad = taken = 0
    for q in range(n_cases):
        spec, desc = case(rng)
        split = int(rng.integers(5, 60))
        twostep = int(rng.choice([5, 6, 8, 16])) + 64 * int(rng.integers(3, 12))
        ref_f, ref_m, p0, _ = run(spec, lib, 0, split)
        got_f, got_m, p1, s2 = run(spec, lib, twostep, split)
        ok = p0 == 0 and all(np.array_equal(a, b) for a, b in zip(ref_f, got_f)) and all(np.array_equal(np.asarray(ref_m[k]), np.asarray(got_m[k])) for k in ref_m)
        if not quiet or not ok:
            print(f"case {q}: N={spec.shape} steps={spec.n_steps} {desc} twostep={twostep & 63}x{twostep >> 6} pairs={p1} shell2={s2} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
        taken += s2 > 0
    return bad, taken


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    lib = None
    if len(sys.argv) > 3 and sys.argv[3] == "emu":
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
        import build_emu
        from tidy3d_amd.lib import load_library
        lib = load_library(build_emu.build())
    bad, taken = run_cases(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1, lib)
    print("fuzz_cell:", n_cases - bad, "of", n_cases, "cases bit-identical;", taken, "took shell2 pairs")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
