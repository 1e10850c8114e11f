"""Media of the mirror schema against the LIVE reference's own objects on random parameters: eps_model over frequency, the
pole-residue form the ADE kernels integrate (eps_inf, poles, residues), n_cfl — Medium (with conductivity), Lorentz (under- and
over-damped), Drude, Debye, Sellmeier, PoleResidue, diagonal AnisotropicMedium, Medium2D's volumetric equivalent.  Skipped where
the reference checkout is absent."""
import json
import os

import numpy as np
import pytest

import tidy3d_amd.schema as td

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")


@pytest.fixture(scope="module")
def td_ref():
    from oracle.tidy3d_ref_loader import load_tidy3d
    return load_tidy3d()


def _random_medium(tdr, rng):
    k = int(rng.integers(0, 6))
    if k == 0:
        return tdr.Medium(permittivity=float(rng.uniform(1, 12)), conductivity=float(rng.choice([0.0, rng.uniform(0, 0.5)])))
    if k == 1:
        return tdr.Lorentz(eps_inf=float(rng.uniform(1, 4)), coeffs=[(float(rng.uniform(0.2, 3)), float(rng.uniform(1e14, 8e14)),
                                                                       float(rng.uniform(1e12, 1.2e15))) for _ in range(int(rng.integers(1, 4)))])
    if k == 2:
        return tdr.Drude(eps_inf=float(rng.uniform(1, 4)), coeffs=[(float(rng.uniform(2e14, 3e15)), float(rng.uniform(1e12, 2e14))) for _ in range(int(rng.integers(1, 3)))])
    if k == 3:
        return tdr.Debye(eps_inf=float(rng.uniform(1, 4)), coeffs=[(float(rng.uniform(0.2, 3)), float(rng.uniform(1e-16, 1e-13))) for _ in range(int(rng.integers(1, 3)))])
    if k == 4:
        return tdr.Sellmeier(coeffs=[(float(rng.uniform(0.1, 1.5)), float(rng.uniform(0.002, 0.2))) for _ in range(int(rng.integers(1, 4)))])
    return tdr.PoleResidue(eps_inf=float(rng.uniform(1, 4)), poles=[(complex(-rng.uniform(1e12, 1e14), -rng.uniform(1e14, 4e15)),
                                                                      complex(rng.uniform(-1e15, 1e15), rng.uniform(1e13, 2e15))) for _ in range(int(rng.integers(1, 4)))])


@pytest.mark.parametrize("seed", range(8))
def test_random_media_model_the_permittivity_the_reference_does(td_ref, seed):
    rng = np.random.default_rng(700 + seed)
    freqs = np.linspace(1e14, 6e14, 9)
    for q in range(40):
        if rng.integers(0, 5) == 0:
            comps = [_random_medium(td_ref, rng) for _ in range(3)]
            m_ref = td_ref.AnisotropicMedium(xx=comps[0], yy=comps[1], zz=comps[2])
        else:
            m_ref = _random_medium(td_ref, rng)
        m = td.parse(json.loads(m_ref.json()))
        assert type(m).__name__ == m_ref.type
        np.testing.assert_allclose(m.eps_model(freqs), m_ref.eps_model(freqs), rtol=1e-11, err_msg=f"{seed}/{q} {m_ref.type}")
        assert m.n_cfl == pytest.approx(float(m_ref.n_cfl), rel=1e-12), (seed, q, m_ref.type)
        parts = m.components if isinstance(m, td.AnisotropicMedium) else [m]
        parts_ref = [m_ref.xx, m_ref.yy, m_ref.zz] if isinstance(m, td.AnisotropicMedium) else [m_ref]
        for a, a_ref in zip(parts, parts_ref):
            eps_inf, sigma, poles = a.pole_residue()
            # eps(w) rebuilt from what the kernels integrate: eps_inf + i sigma / (w eps0) - sum [c / (i w + a) + c* / (i w + a*)]
            w = 2 * np.pi * freqs
            e = eps_inf + 1j * sigma / (w * 8.8541878128e-18) + 0j
            for pa, pc in poles:
                e = e - (pc / (1j * w + pa) + np.conj(pc) / (1j * w + np.conj(pa)))
            np.testing.assert_allclose(e, a_ref.eps_model(freqs), rtol=1e-9, err_msg=f"{seed}/{q} {a_ref.type} from poles")
            if hasattr(a_ref, "pole_residue") and a_ref.type != "Medium":
                pr = a_ref.pole_residue
                assert eps_inf == pytest.approx(float(pr.eps_inf), rel=1e-12)
                assert sorted(poles, key=lambda t: (t[0].real, t[0].imag)) == pytest.approx(
                    sorted(((complex(a_), complex(c_)) for a_, c_ in pr.poles), key=lambda t: (t[0].real, t[0].imag)), rel=1e-10)


@pytest.mark.parametrize("seed", range(4))
def test_random_sheets_have_the_reference_volumetric_equivalent(td_ref, seed):
    """Medium2D.volumetric_equivalent (ref medium.py:6170-6238) for random sheets, neighbours and cell sizes against the table entry
    discretize.rasterize builds (the same arithmetic on MediumCoeffs)."""
    from tidy3d_amd.data import medium_eps_table
    from tidy3d_amd.discretize import discretize
    rng = np.random.default_rng(900 + seed)
    freqs = np.array([1.5e14, 2.5e14, 4e14])
    for q in range(6):
        ss, tt = _random_medium(td_ref, rng), _random_medium(td_ref, rng)
        below, above = _random_medium(td_ref, rng), _random_medium(td_ref, rng)
        sheet_ref = td_ref.Medium2D(ss=ss, tt=tt)
        # the neighbours are those at the sheet's OWN plane, before it is snapped to the grid plane z = 0 (ref simulation.py:1317,
        # geometry/utils_2d.py:61-70): on the interface -> (below, above); 4 nm above it -> the upper medium on both sides
        z_sheet = 0.0 if q % 2 == 0 else 0.004
        dz = np.concatenate([np.full(10, 0.05), rng.uniform(0.03, 0.07, 10)])
        zb = np.concatenate(([0.0], np.cumsum(dz))) - 0.5
        conv = lambda m_: td.parse(json.loads(m_.json()))      # noqa: E731
        sim = td.Simulation(
            size=(0.5, 0.5, float(zb[-1] - zb[0])), center=(0, 0, float(0.5 * (zb[0] + zb[-1]))), run_time=1e-14, subpixel=False,
            grid_spec=td.GridSpec(grid_x=td.UniformGrid(dl=0.05), grid_y=td.UniformGrid(dl=0.05), grid_z=td.CustomGridBoundaries(coords=tuple(zb))),
            structures=[td.Structure(geometry=td.Box(center=(0, 0, -5.0), size=(td.inf, td.inf, 10.0)), medium=conv(below)),
                        td.Structure(geometry=td.Box(center=(0, 0, 5.0), size=(td.inf, td.inf, 10.0)), medium=conv(above)),
                        td.Structure(geometry=td.Box(center=(0, 0, z_sheet), size=(td.inf, td.inf, 0)), medium=conv(sheet_ref))],
            sources=[td.PointDipole(center=(0, 0, 0.2), source_time=td.GaussianPulse(freq0=2e14, fwidth=4e13), polarization="Ex")],
            boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
        spec = discretize(sim, n_steps=2).spec
        b = np.asarray(spec.boundaries[2])
        k = int(np.argmin(np.abs(b)))
        assert abs(b[k]) < 1e-12
        vol = sheet_ref.volumetric_equivalent(axis=2, adjacent_media=(below if z_sheet == 0.0 else above, above), adjacent_dls=(b[k] - b[k - 1], b[k + 1] - b[k]))
        for c, name in ((0, "xx"), (1, "yy")):
            idx = spec.mat_idx[c][k, 3, 3]
            got = np.array([medium_eps_table(spec, f)[idx] for f in freqs])
            np.testing.assert_allclose(got, np.asarray(getattr(vol, name).eps_model(freqs)), rtol=1e-9, err_msg=f"{seed}/{q} {name} {ss.type}/{tt.type} on {below.type}|{above.type}")
