"""Generate tests/golden/geometry_golden.json by RUNNING THE REFERENCE'S OWN geometry code (imported
from /root/reference through oracle/tidy3d_ref_loader.py): ``inside`` and ``bounds`` of slanted
cylinders and of ``Transformed`` geometries at random points.  Run once in the build container:

    python tests/golden/make_geometry_golden.py

Nothing here is solver logic; tests/test_golden_schema.py reads the committed fixture."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle.tidy3d_ref_loader import load_tidy3d  # noqa: E402

td = load_tidy3d()


def cases():
    T = td.Transformed
    rot = T.rotation(0.7, (1.0, 2.0, -0.5))
    geos = [
        td.Cylinder(center=(0, 0.1, 0), radius=0.5, length=0.9, axis=1, sidewall_angle=0.3),
        td.Cylinder(center=(-0.2, 0, 0.1), radius=0.3, length=1.1, axis=2, sidewall_angle=-0.25, reference_plane="bottom"),
        td.Cylinder(center=(0.1, 0, 0), radius=0.6, length=0.8, axis=0, sidewall_angle=0.5, reference_plane="top"),
        td.Cylinder(center=(0, 0, 0), radius=0.2, length=1.4, axis=2, sidewall_angle=0.4),          # tip inside: r <= 0
        T(geometry=td.Box(center=(0.1, 0, -0.1), size=(0.8, 0.4, 1.0)), transform=T.rotation(0.5, 2)),
        T(geometry=td.Cylinder(center=(0, 0.1, 0), radius=0.3, length=0.9, axis=1),
          transform=np.dot(T.translation(0.2, -0.1, 0.05), rot)),
        T(geometry=td.Sphere(center=(0.1, 0.2, 0), radius=0.4), transform=T.scaling(1.5, 0.6, 1.1)),
        T(geometry=T(geometry=td.Box(center=(0, 0, 0), size=(0.6, 0.5, 0.4)), transform=T.rotation(-0.3, 0)),
          transform=np.dot(T.scaling(1.2, 1.0, 0.8), T.translation(0.1, 0.1, -0.2))),
        T(geometry=td.GeometryGroup(geometries=[td.Box(center=(0.3, 0, 0), size=(0.3, 0.3, 0.3)),
                                                td.Sphere(center=(-0.3, 0, 0), radius=0.25)]),
          transform=T.rotation(1.1, 1)),
    ]
    rng = np.random.default_rng(1)
    pts = np.round(rng.uniform(-1, 1, (600, 3)), 6)
    out = []
    for g in geos:
        ins = g.inside(pts[:, 0], pts[:, 1], pts[:, 2])
        out.append({"json": json.loads(g.json()),
                    "inside": "".join("1" if b else "0" for b in np.asarray(ins).reshape(-1)),
                    "bounds": [list(map(float, b)) for b in g.bounds]})
    return {"points": np.round(pts, 6).tolist(), "geometry": out, "polyslab_helpers": polyslab_helpers()}


def polyslab_helpers():
    """The reference's own static polygon helpers (a PolySlab instance cannot be built here: its
    validators need shapely): _proper_vertices, _shift_vertices, _maximal_erosion, _area."""
    P = td.PolySlab
    rng = np.random.default_rng(2)
    polys = [[(0, 0), (1, 0), (1, 1), (0, 1)],
             [(0, 0), (0, 1), (1, 1), (1, 0)],                       # clockwise
             [(0, 0), (2, 0), (2, 0), (2, 1), (1, 0.4), (0, 1)],       # duplicate vertex, concave
             [(-1, -0.5), (0.2, -0.7), (1.1, 0.1), (0.4, 0.2), (0.6, 0.9), (-0.3, 0.5), (-0.9, 0.8)],
             [(np.cos(t), 0.6 * np.sin(t)) for t in np.linspace(0, 2 * np.pi, 9)[:-1]],
             [(0, 0), (1, 0), (2, 0), (2, 1), (0, 1)]]                 # collinear vertex
    out = []
    for v in polys:
        prop = P._proper_vertices(v)
        rec = {"vertices": [list(map(float, q)) for q in v], "proper": prop.tolist(), "area": float(P._area(prop)),
               "max_erosion": float(P._maximal_erosion(prop)), "shifts": []}
        for d in (0.05, -0.03, 0.2, float(-0.5 * P._maximal_erosion(prop))):
            sv, par, _ = P._shift_vertices(prop, d)
            rec["shifts"].append({"dist": d, "vertices": np.asarray(sv).tolist(), "parallel": np.asarray(par).tolist()})
        out.append(rec)
    return out


if __name__ == "__main__":
    path = os.path.join(HERE, "geometry_golden.json")
    with open(path, "w") as f:
        json.dump({"_generator": "tests/golden/make_geometry_golden.py (reference tidy3d v%s)" % td.__version__,
                   **cases()}, f)
    print("wrote", os.path.getsize(path), "bytes")
