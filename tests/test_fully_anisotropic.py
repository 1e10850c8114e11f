"""FullyAnisotropicMedium (ref medium.py:5058): a full symmetric permittivity tensor.  The sweep advances every E component with the
diagonal of eps^-1; the off-diagonal coupling follows as a list update (spec.AnisoSet, csrc/fdtd_aniso.hpp) built from the
symmetric average of the tensor over the two nodes of each pair.  Pins: a wave plate against the Jones calculus, discrete
reciprocity and long-run stability (what a non-symmetric coupling would break), the HIP kernels against the oracle."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd.constants import C_0
from tidy3d_amd.data import assemble
from tidy3d_amd.discretize import discretize
from tidy3d_amd.exceptions import Tidy3dNotImplementedError

from oracle.fdtd_numpy import OracleFdtd

PULSE = td.GaussianPulse(freq0=2e14, fwidth=4e13)


def rot(axis, angle):
    c, s = np.cos(angle), np.sin(angle)
    m = {2: [[c, -s, 0], [s, c, 0], [0, 0, 1]], 1: [[c, 0, s], [0, 1, 0], [-s, 0, c]], 0: [[1, 0, 0], [0, c, -s], [0, s, c]]}[axis]
    return np.array(m)


def test_wave_plate_matches_the_jones_calculus():
    """A uniaxial slab (n_o = 1.5, n_e = 1.8) with its optic axis in the plane at 45 degrees, x-polarised light at normal
    incidence: T_xx = (t_e + t_o) / 2, T_yx = (t_e - t_o) / 2 with the Airy coefficients of the two indices — amplitude AND
    phase, seven frequencies, on the fp64 oracle (numerical dispersion of lambda / 50 cells: 3e-3)."""
    no, ne, d = 1.5, 1.8, 0.6
    med = td.FullyAnisotropicMedium.from_diagonal(ne ** 2, no ** 2, no ** 2, rot(2, np.pi / 4))
    freqs = np.linspace(1.7e14, 2.3e14, 7)
    per = td.Boundary.periodic()

    def run(structures, dl=0.02):
        sim = td.Simulation(size=(4 * dl, 4 * dl, 4.4), grid_spec=td.GridSpec.uniform(dl=dl), run_time=4e-13, structures=structures, subpixel=False,
                            sources=[td.UniformCurrentSource(center=(0, 0, -1.4), size=(td.inf, td.inf, 0), source_time=PULSE, polarization="Ex")],
                            monitors=[td.FieldMonitor(center=(0, 0, 1.3), size=(0, 0, 0), freqs=list(freqs), name="p", fields=["Ex", "Ey"], colocate=False)],
                            boundary_spec=td.BoundarySpec(x=per, y=per, z=td.Boundary.pml(num_layers=12)), shutoff=0)
        disc = discretize(sim)
        sd = assemble(disc, OracleFdtd(disc.spec).run(), log="")
        return sd["p"].Ex.values.ravel(), sd["p"].Ey.values.ravel()
    ex0, ey0 = run([])
    ex1, ey1 = run([td.Structure(geometry=td.Box(center=(0, 0, 0.3), size=(td.inf, td.inf, d)), medium=med)])
    assert np.abs(ey0).max() == 0

    def airy(n):
        k = 2 * np.pi * freqs / C_0 * n
        r = (1 - n) / (1 + n)
        return (1 - r ** 2) * np.exp(1j * k * d) / (1 - r ** 2 * np.exp(2j * k * d)) * np.exp(-1j * 2 * np.pi * freqs / C_0 * d)
    te, to = airy(ne), airy(no)
    assert np.abs(ex1 / ex0 - 0.5 * (te + to)).max() < 5e-3
    assert np.abs(ey1 / ex0 - 0.5 * (te - to)).max() < 3e-3
    assert np.abs(0.5 * (te - to)).min() > 0.3                     # (a real polarisation conversion, not a small number)


def _body_sim(N=(18, 16, 14), n_steps=80):
    from cases import _sim
    med = td.FullyAnisotropicMedium.from_diagonal(2.0, 5.0, 3.2, rot(2, 0.7) @ rot(1, 0.4))
    structures = [td.Structure(geometry=td.Sphere(center=(0.05, 0, 0.02), radius=0.28), medium=med),
                  td.Structure(geometry=td.Box(center=(-0.2, 0.1, 0), size=(0.2, 0.3, td.inf)), medium=td.Medium(permittivity=2.5, conductivity=0.02)),
                  td.Structure(geometry=td.Box(center=(0.2, -0.15, 0.1), size=(0.15, 0.15, 0.15)), medium=td.PEC)]
    bspec = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary(minus=td.PECBoundary(), plus=td.PML(num_layers=3)), z=td.Boundary.periodic())
    return discretize(_sim(N, bspec, structures), n_steps=n_steps)


@pytest.mark.parametrize("variant", ["fused", "two_pass"])
def test_hip_kernels_match_the_oracle(variant, emu_lib):
    """A rotated biaxial sphere next to a lossy block and a PEC box, CPML / PEC / periodic walls: the sweep + the coupling
    lists of csrc/fdtd_aniso.hpp against the oracle's own statement of the same update (fields and every record <= 2e-5);
    step pairs stay off (the coupling follows every single step)."""
    from cases import rel_err
    from tidy3d_amd import lib as L
    from tidy3d_amd.engine import HipEngine
    disc = _body_sim()
    assert len(disc.spec.aniso) == 3 and min(len(a.ijk) for a in disc.spec.aniso) > 500
    o = OracleFdtd(disc.spec)
    ref = o.run()
    with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED if variant == "fused" else L.VARIANT_ZMARCH) as e:
        e.set_option(L.OPT_TWOSTEP, 5 + 64 * 4)
        st = e.run()
        got = e.results()
        f = [e.get_field(c) for c in range(6)]
    assert int(st.fused2_pairs) == 0
    for k in ref:
        assert rel_err(got[k], ref[k]) < 2e-5, k
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    assert max(float(np.linalg.norm(f[c] - o.E[c]) / en) for c in range(3)) < 2e-5


def test_reciprocity_and_stability_with_a_fully_anisotropic_body():
    """What the SYMMETRIC average of the tensor over the two nodes of a coupled pair buys: the discrete operator stays
    self-adjoint — Lorentz reciprocity to 1e-11 across the body's faces, and a lossless cavity that holds the body rings for
    6000 steps without gaining energy."""
    from test_reciprocity import reciprocity_sims, series
    med = td.FullyAnisotropicMedium.from_diagonal(2.0, 6.0, 3.0, rot(2, 0.5) @ rot(0, 0.8))
    cfg = dict(N=(20, 18, 16), A=(5, 6, 5), ca=0, B=(14, 11, 10), cb=2,
               bspec=td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()), z=td.Boundary.periodic()),
               structures=[td.Structure(geometry=td.Box(center=(0.05, 0, 0), size=(0.45, 0.5, 0.4)), medium=med)])
    (d1, at1), (d2, at2) = reciprocity_sims(n_steps=260, **cfg)
    assert d1.spec.aniso
    e1, e2 = series(OracleFdtd(d1.spec).run(), at1), series(OracleFdtd(d2.spec).run(), at2)
    assert np.abs(e1 - e2).max() < 1e-11 * np.abs(e1).max()
    # stability: PEC box, the body inside, a short pulse, then 6000 free steps
    sim = td.Simulation(size=(0.8, 0.7, 0.6), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=1e-12, subpixel=False, shutoff=0,
                        structures=[td.Structure(geometry=td.Sphere(center=(0.05, 0, 0), radius=0.25), medium=med)],
                        sources=[td.PointDipole(center=(-0.2, 0.1, 0.1), source_time=td.GaussianPulse(freq0=3e14, fwidth=1.5e14), polarization="Ez")],
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    disc = discretize(sim, n_steps=6400)
    o = OracleFdtd(disc.spec)
    o.run(400)
    amp0 = max(float(np.abs(x).max()) for x in o.E)
    o.run(6000)
    amp1 = max(float(np.abs(x).max()) for x in o.E)
    assert 0 < amp1 < 3 * amp0, (amp0, amp1)


def test_what_is_refused(emu_lib):
    from tidy3d_amd.engine import HipEngine
    with pytest.raises(Tidy3dNotImplementedError, match="conductivity"):
        lossy = td.FullyAnisotropicMedium(permittivity=[[2, 0, 0], [0, 3, 0], [0, 0, 4]], conductivity=[[0.1, 0, 0], [0, 0, 0], [0, 0, 0]])
        discretize(td.Simulation(size=(1, 1, 1), grid_spec=td.GridSpec.uniform(dl=0.1), run_time=1e-14,
                                 structures=[td.Structure(geometry=td.Box(size=(0.4, 0.4, 0.4)), medium=lossy)],
                                 sources=[td.PointDipole(source_time=PULSE, polarization="Ez")]), n_steps=2)
    disc = _body_sim()
    with pytest.raises(Tidy3dNotImplementedError, match="z-slab"):
        HipEngine(disc.spec, lib=emu_lib, force_comm=True)
    # the JSON form of the reference parses; a body that a later structure covers loses those nodes
    m = td.parse({"type": "FullyAnisotropicMedium", "permittivity": [[2, 0.5, 0], [0.5, 3, 0], [0, 0, 4]]})
    assert isinstance(m, td.FullyAnisotropicMedium) and m.n_cfl == pytest.approx(np.sqrt(np.min(np.linalg.eigvalsh(m.eps_tensor))))
    sim = td.Simulation(size=(1, 1, 1), grid_spec=td.GridSpec.uniform(dl=0.1), run_time=1e-14, subpixel=False,
                        structures=[td.Structure(geometry=td.Box(size=(0.6, 0.6, 0.6)), medium=m),
                                    td.Structure(geometry=td.Box(size=(0.6, 0.6, 0.6)), medium=td.Medium(permittivity=2.0))],
                        sources=[td.PointDipole(source_time=PULSE, polarization="Ez")], boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    assert discretize(sim, n_steps=2).spec.aniso == []


@pytest.mark.parametrize("seed", range(4))
def test_random_bodies_fused_two_pass_and_oracle_agree(seed, emu_lib):
    """Random tensors (random principal values and rotations), random bodies overlapping each other and ordinary media, random
    walls and cell sizes: fused sweep == two-pass kernels bit for bit (the coupling lists are the same launches), both <= 2e-5
    from the oracle."""
    from cases import DL, PULSE as P3, rel_err
    from tidy3d_amd import lib as L
    from tidy3d_amd.engine import HipEngine
    rng = np.random.default_rng(40 + seed)
    N = [int(rng.integers(12, 20)) for _ in range(3)]
    size = tuple(n * DL for n in N)

    def pos():
        return tuple(float(rng.uniform(-0.3, 0.3) * s) for s in size)
    structures = []
    for _ in range(int(rng.integers(1, 4))):
        R = rot(2, float(rng.uniform(0, 3))) @ rot(1, float(rng.uniform(0, 3))) @ rot(0, float(rng.uniform(0, 3)))
        med = td.FullyAnisotropicMedium.from_diagonal(*[float(v) for v in rng.uniform(1.0, 8.0, 3)], R)
        geo = td.Sphere(center=pos(), radius=float(rng.uniform(0.15, 0.3))) if rng.integers(0, 2) else \
            td.Box(center=pos(), size=tuple(float(rng.uniform(0.2, 0.6) * s) for s in size))
        structures.append(td.Structure(geometry=geo, medium=med))
    structures.insert(int(rng.integers(0, len(structures) + 1)),
                      td.Structure(geometry=td.Box(center=pos(), size=tuple(float(rng.uniform(0.2, 0.5) * s) for s in size)),
                                   medium=[td.Medium(permittivity=3.0, conductivity=0.02), td.PEC, td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])][int(rng.integers(0, 3))]))
    faces = [td.PML(num_layers=3), td.PECBoundary(), td.PMCBoundary(), td.StablePML(num_layers=4)]
    edges = [td.Boundary.periodic() if rng.integers(0, 4) == 0 else
             td.Boundary(minus=faces[int(rng.integers(0, 4))], plus=faces[int(rng.choice([0, 1, 3]))]) for _ in range(3)]
    grid = td.GridSpec.uniform(dl=DL)
    if rng.integers(0, 2):
        def coords(n, s_):
            d = rng.uniform(0.8, 1.2, n)
            return tuple(np.concatenate(([0.0], np.cumsum(d))) * (s_ / d.sum()) - 0.5 * s_)
        grid = td.GridSpec(grid_x=td.CustomGridBoundaries(coords=coords(N[0], size[0])), grid_y=td.CustomGridBoundaries(coords=coords(N[1], size[1])),
                           grid_z=td.CustomGridBoundaries(coords=coords(N[2], size[2])))
    sim = td.Simulation(size=size, grid_spec=grid, run_time=1e-12, shutoff=0, subpixel=False, structures=structures,
                        sources=[td.PointDipole(center=pos(), source_time=P3, polarization=str(rng.choice(["Ex", "Ey", "Ez", "Hy"])))],
                        monitors=[td.FieldTimeMonitor(center=pos(), size=(0.2, 0.2, 0.2), name="t", interval=3, colocate=False),
                                  td.FieldMonitor(center=pos(), size=(td.inf, td.inf, 0), freqs=[3e14], name="f")],
                        boundary_spec=td.BoundarySpec(x=edges[0], y=edges[1], z=edges[2]))
    disc = discretize(sim, n_steps=70)
    assert disc.spec.aniso
    ref = OracleFdtd(disc.spec).run()
    out = []
    for variant in (L.VARIANT_FUSED, L.VARIANT_ZMARCH):
        with HipEngine(disc.spec, lib=emu_lib, variant=variant) as e:
            e.run(30)
            e.run(40)
            out.append((e.results(), [e.get_field(c) for c in range(6)]))
    for k in ref:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k
        assert rel_err(out[0][0][k], ref[k]) < 2e-5, (k, rel_err(out[0][0][k], ref[k]))
    assert all(np.array_equal(a, b) for a, b in zip(out[0][1], out[1][1]))
