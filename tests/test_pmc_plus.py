"""``PMCBoundary`` on PLUS faces (ref boundary.py:45; VERDICT round 2, missing 5).  The kernels know PMC on min faces only;
a plus-face PMC wall is laid out with two ghost cells beyond it whose contents are the mirror image of the inside, refreshed
at the start of every step (discretize._discretize_pmc_plus, SolverSpec.mirror_plus, kernel mirror_fill_kernel).

Pin: a mirror-symmetric problem solved on its lower half behind a PMC plus wall IS the full problem there — identical numbers
(the image arithmetic is the arithmetic of the other half), on every axis, with PML, a dispersive sphere cut by the wall and a
source on the wall node.  Emulator: fused sweep == two-pass kernels == oracle."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.data import assemble
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine
from oracle.fdtd_numpy import OracleFdtd

DL = 0.0625
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1e14)


def _sim(half, axis):
    n = [24, 16, 14]
    size, cen = [v * DL for v in n], [0.0, 0.0, 0.0]
    edges = [[td.PML(num_layers=4), td.PML(num_layers=4)] for _ in range(3)]
    if half:
        size[axis] /= 2
        cen[axis] = -size[axis] / 2
        edges[axis][1] = td.PMCBoundary()
    pol = "E" + "xyz"[(axis + 1) % 3]                     # tangential to the mirror plane: an even source
    src_c, struct_c, mon_c, msize = [0.0] * 3, [0.0] * 3, [0.0] * 3, [0.5] * 3
    src_c[(axis + 1) % 3], src_c[(axis + 2) % 3] = 0.13, -0.07          # ON the wall along `axis`
    struct_c[(axis + 1) % 3] = 0.2
    mon_c[axis], msize[axis] = -0.2, 0.3
    return td.Simulation(
        size=tuple(size), center=tuple(cen), grid_spec=td.GridSpec.uniform(dl=DL), run_time=6e-14, shutoff=0,
        structures=[td.Structure(geometry=td.Sphere(center=tuple(struct_c), radius=0.22),
                                 medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)]))],
        sources=[td.PointDipole(center=tuple(src_c), source_time=PULSE, polarization=pol)],
        monitors=[td.FieldMonitor(center=tuple(mon_c), size=tuple(msize), freqs=[2.5e14, 3e14], name="f", colocate=False),
                  td.FieldMonitor(center=(0, 0, 0), size=tuple(0 if a == (axis + 1) % 3 else td.inf for a in range(3)), freqs=[3e14], name="cut"),
                  td.FieldTimeMonitor(center=tuple(mon_c), size=(0, 0, 0), name="t", interval=2)],
        boundary_spec=td.BoundarySpec(**{"xyz"[a]: td.Boundary(minus=edges[a][0], plus=edges[a][1]) for a in range(3)}))


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_half_domain_behind_a_pmc_plus_wall_is_the_symmetric_problem(axis):
    out = []
    for half in (False, True):
        d = discretize(_sim(half, axis), n_steps=160)
        if half:
            assert d.spec.mirror_plus[axis] == d.spec.shape[axis] - 2 and sum(w >= 0 for w in d.spec.mirror_plus) == 1
        out.append(assemble(d, OracleFdtd(d.spec).run()))
    full, half = out
    for name in ("Ex", "Ey", "Ez", "Hx", "Hy", "Hz"):
        a, b = np.asarray(full["f"][name].values), np.asarray(half["f"][name].values)
        assert a.shape == b.shape and np.abs(a).max() > 0
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-12 * np.abs(a).max())
    # the plane through the wall: the half run returns the user's half of it, ending ON the wall — where the normal E and the
    # tangential H of the colocated data vanish
    n_ax = "xyz"[axis]
    cf, ch = full["cut"], half["cut"]
    xs_h = np.asarray(ch["E" + n_ax].coords[n_ax])
    assert xs_h.max() == pytest.approx(0.0, abs=1e-9)
    e_norm = np.asarray(ch["E" + n_ax].values)
    wall = [slice(None)] * 4
    wall[axis] = -1
    assert np.abs(e_norm[tuple(wall)]).max() < 1e-9 * max(np.abs(np.asarray(ch["E" + "xyz"[(axis + 1) % 3]].values)).max(), 1e-30)
    for c in ("Ex", "Ey", "Ez"):
        a, b = cf[c], ch[c]
        sel = np.asarray(a.coords[n_ax]) <= 1e-9
        av = np.compress(sel, np.asarray(a.values), axis=axis)
        np.testing.assert_allclose(np.asarray(b.values), av, rtol=0, atol=1e-9 * max(np.abs(av).max(), 1e-30))


@pytest.mark.parametrize("variant,rows,zc", [(L.VARIANT_FUSED, 3, 4), (L.VARIANT_FUSED, 4, 16), (L.VARIANT_ZMARCH, 4, 2)])
def test_pmc_plus_kernels_match_the_oracle(emu_lib, variant, rows, zc):
    from cases import pmc_plus_mix, rel_err
    disc = discretize(pmc_plus_mix(), n_steps=60)
    assert disc.spec.mirror_plus[0] >= 0 and disc.spec.mirror_plus[2] >= 0
    ref = OracleFdtd(disc.spec).run()
    with HipEngine(disc.spec, lib=emu_lib, variant=variant, z_chunk=zc, axis_shift=0) as e:
        e.set_option(L.OPT_ROWS, rows)
        e.run()
        got = e.results()
    for k in ref:
        assert rel_err(got[k], ref[k]) < 2e-5, (k, rel_err(got[k], ref[k]))


def test_pmc_plus_survives_the_axis_renaming(emu_lib):
    from cases import pmc_plus_mix
    disc = discretize(pmc_plus_mix((20, 16, 14)), n_steps=40)
    outs = []
    for shift in (0, 1, 2):
        with HipEngine(disc.spec, lib=emu_lib, axis_shift=shift) as e:
            e.run()
            outs.append(e.results())
    for k in outs[0]:
        for o in outs[1:]:
            assert np.abs(o[k] - outs[0][k]).max() <= 2e-5 * np.abs(outs[0][k]).max(), k


def _pmc_plus_x_periodic_z(N=(20, 14, 16)):
    from cases import DL, PULSE
    size = tuple(n * DL for n in N)
    return td.Simulation(
        size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, shutoff=0,
        structures=[td.Structure(geometry=td.Sphere(center=(0.5 * size[0] - 0.1, 0, 0.1), radius=0.3), medium=td.Medium(permittivity=2.5, conductivity=0.01))],
        sources=[td.PointDipole(center=(0.5 * size[0] - 0.16, -0.07, 0.01), source_time=PULSE, polarization="Ez"),
                 td.PointDipole(center=(-0.1, 0.07, 0.5 * size[2] - 0.08), source_time=PULSE, polarization="Hx")],
        monitors=[td.FieldTimeMonitor(center=(0.3, 0.05, 0.2), size=(0.3, 0.2, 0.3), name="t", colocate=False, interval=7)],
        boundary_spec=td.BoundarySpec(x=td.Boundary(minus=td.PML(num_layers=4), plus=td.PMCBoundary()), y=td.Boundary.pml(num_layers=3),
                                      z=td.Boundary.periodic()))


def test_pmc_plus_wall_across_a_periodic_z(emu_lib):
    """A PMC plus wall (x) crossing a periodic z: the wrapped copies of the end planes carry image cells too.  The fused sweep
    took them from the end of the last step — in front of the refresh of the images, which a dipole next to the wall makes
    more than a no-op — where the two-pass kernels, the z-slab ranks and the oracle use the refreshed ones (found by the
    slab-rank test of round 4; the fused step now refreshes the ghost planes with the planes they copy).  Fused == two-pass
    == a slab rank exchanging with itself, bit for bit; <= 2e-5 from the oracle."""
    from cases import rel_err
    disc = discretize(_pmc_plus_x_periodic_z(), n_steps=40)
    disc.spec.decay_every = 16
    assert disc.spec.mirror_plus[0] >= 0
    outs = {}
    for name, variant, comm in (("fused", L.VARIANT_FUSED, False), ("two_pass", L.VARIANT_ZMARCH, False),
                                ("fused_slab", L.VARIANT_FUSED, True), ("two_pass_slab", L.VARIANT_ZMARCH, True)):
        with HipEngine(disc.spec, lib=emu_lib, variant=variant, axis_shift=0, force_comm=comm) as e:
            if comm:
                e.comm_init(e.unique_id())
            e.run(15)
            e.run(25)
            outs[name] = ([e.get_field(c) for c in range(6)], e.results())
    N = disc.spec.mirror_plus[0]
    for name in ("two_pass", "fused_slab", "two_pass_slab"):
        for c in range(6):
            # (up to the wall: what the image cells hold after the last step is refreshed before it is used)
            assert np.array_equal(outs[name][0][c][:, :, :N], outs["fused"][0][c][:, :, :N]), (name, c)
        for k, v in outs["fused"][1].items():
            assert np.array_equal(outs[name][1][k], v), (name, k)
    ref = OracleFdtd(disc.spec).run()
    for k in ref:
        assert rel_err(outs["fused"][1][k], ref[k]) < 2e-5, (k, rel_err(outs["fused"][1][k], ref[k]))
