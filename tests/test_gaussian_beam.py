"""GaussianBeam / AstigmaticGaussianBeam (ref source.py:1109-1201), launched as a sheet of currents carrying
the paraxial beam profile (tidy3d_amd/planewave.py).  The reference computes the beam profile server-side
(parity unpinned); pinned physically on the oracle: one-way injection, beam radius w(z) along the axis,
carried power pi w0x w0y / 2 (peak intensity 1 W/um^2 at the waist), focusing with a negative waist
distance, tilt by angle_theta, two different waists of the astigmatic beam."""
import numpy as np
import pytest

import tidy3d_amd.schema as td

from test_physics_oracle import solve

F0 = 3e14          # lambda = 1 um
PULSE = td.GaussianPulse(freq0=F0, fwidth=F0 / 5)


def _sim(beam, planes):
    mons = [td.FluxMonitor(center=(0, 0, 0.9), size=(td.inf, td.inf, 0), freqs=[F0], name="fwd"),
            td.FluxMonitor(center=(0, 0, -1.05), size=(td.inf, td.inf, 0), freqs=[F0], name="back")]
    mons += [td.FieldMonitor(center=(0, 0, z), size=(td.inf, td.inf, 0), freqs=[F0], name=f"z{i}")
             for i, z in enumerate(planes)]
    return td.Simulation(size=(5.4, 5.4, 2.4), grid_spec=td.GridSpec.uniform(dl=0.1), run_time=5.5e-14, sources=[beam],
                         monitors=mons, shutoff=0, boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=6)))


def _moments(fd):
    """Centroid and 1/e^2 radii along x and y of |E_t|^2 (for exp(-2 q^2 / w^2): <q^2> = w^2 / 4)."""
    inten = np.abs(fd.Ex.values[:, :, 0, 0]) ** 2 + np.abs(fd.Ey.values[:, :, 0, 0]) ** 2
    X, Y = np.meshgrid(np.asarray(fd.Ex.coords["x"]), np.asarray(fd.Ex.coords["y"]), indexing="ij")
    tot = inten.sum()
    cx, cy = (inten * X).sum() / tot, (inten * Y).sum() / tot
    wx = 2 * np.sqrt((inten * (X - cx) ** 2).sum() / tot)
    wy = 2 * np.sqrt((inten * (Y - cy) ** 2).sum() / tot)
    return cx, cy, wx, wy


def test_gaussian_beam_radius_power_and_direction():
    w0, wd = 0.9, 0.8
    beam = td.GaussianBeam(center=(0, 0, -0.8), size=(td.inf, td.inf, 0), source_time=PULSE, direction="+",
                           waist_radius=w0, waist_distance=wd)
    sd, _, _ = solve(_sim(beam, (-0.4, 0.6)))
    assert sd["fwd"].flux.values[0] == pytest.approx(np.pi * w0 ** 2 / 2, rel=0.08)
    assert abs(sd["back"].flux.values[0]) < 2e-3 * sd["fwd"].flux.values[0]
    zr = np.pi * w0 ** 2                         # k w0^2 / 2 at lambda = 1
    for i, z in enumerate((-0.4, 0.6)):
        cx, cy, wx, wy = _moments(sd[f"z{i}"])
        want = w0 * np.sqrt(1 + ((z + 0.8 + wd) / zr) ** 2)
        assert wx == pytest.approx(want, rel=0.06) and wy == pytest.approx(want, rel=0.06)    # paraxial formula at w0 = 0.9 lambda
        assert abs(cx) < 0.02 and abs(cy) < 0.02


def test_tilted_astigmatic_beam_focuses_in_front_of_the_source():
    """waist_distances < 0: the waists lie in front of the source (ref source.py:1191-1200); two different
    waists along the beam's x' (P direction) and y'; the beam axis follows angle_theta / angle_phi."""
    th = 0.2
    beam = td.AstigmaticGaussianBeam(center=(-0.2, 0, -0.8), size=(td.inf, td.inf, 0), source_time=PULSE, direction="+",
                                     waist_sizes=(1.0, 1.4), waist_distances=(-0.8, -0.8), angle_theta=th, angle_phi=0.0,
                                     pol_angle=np.pi / 2)
    sd, _, _ = solve(_sim(beam, (-0.03, 0.6)))
    cx0, cy0, wx0, wy0 = _moments(sd["z0"])
    cx1, cy1, wx1, wy1 = _moments(sd["z1"])
    # beam axis: x = -0.2 + tan(theta) (z + 0.8)
    assert cx0 == pytest.approx(-0.2 + np.tan(th) * 0.77, abs=0.04) and cx1 == pytest.approx(-0.2 + np.tan(th) * 1.4, abs=0.06)
    assert abs(cy0) < 0.02 and abs(cy1) < 0.02
    # near the waists (0.8 along the tilted axis = z close to 0): x' radius seen on a z-plane is w / cos(theta)
    assert wx0 == pytest.approx(1.0 / np.cos(th), rel=0.06) and wy0 == pytest.approx(1.4, rel=0.04)
    assert wx1 > wx0 * 1.01                                               # the tighter waist diverges faster
    assert sd["fwd"].flux.values[0] == pytest.approx(np.pi * 1.0 * 1.4 / 2, rel=0.08)
    assert abs(sd["back"].flux.values[0]) < 3e-3 * sd["fwd"].flux.values[0]


def test_parses_from_the_reference_json_form():
    d = {"type": "GaussianBeam", "center": [0, 0, 0], "size": [0, 3, 3], "source_time": {"type": "GaussianPulse", "freq0": 2e14, "fwidth": 2e13},
         "direction": "-", "angle_theta": 0.1, "angle_phi": 0.2, "pol_angle": 1.0, "waist_radius": 1.5, "waist_distance": -2.0,
         "num_freqs": 1, "name": None}
    g = td.parse(d)
    assert isinstance(g, td.GaussianBeam) and g.injection_axis == 0 and g.waist_distance == -2.0
    a = td.parse({**d, "type": "AstigmaticGaussianBeam", "waist_sizes": [1.0, 2.0], "waist_distances": [3.0, 4.0]})
    assert isinstance(a, td.AstigmaticGaussianBeam) and a.waist_sizes == (1.0, 2.0)
