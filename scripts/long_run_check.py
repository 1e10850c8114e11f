"""A long run of a complete open scattering problem (bench workload v4a: dielectric sphere, absorber layers on six faces, a closed
flux box with a running DFT) on the two-step sweep against single sweeps: fields and spectra after N steps, bit for bit.
    python scripts/long_run_check.py [n] [steps] [workload: v4a (default), v2, v3, v4 — CPML walls: shell2 pairs, dispersive cells inside them]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_spec  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10001
wl = sys.argv[3] if len(sys.argv) > 3 else "v4a"
spec = build_spec(n, steps + 4, wl)
spec.decay_every = 500


def run(twostep):
    with HipEngine(spec, axis_shift=0) as e:
        e.set_option(L.OPT_TWOSTEP, twostep)
        e.set_option(L.OPT_PLACEMENT_TRIES, 0)
        t0 = time.perf_counter()
        pairs = 0
        for r in (steps // 3, steps - steps // 3):          # two calls: the second starts on an odd step
            pairs += int(e.run(r).fused2_pairs)
        dt = time.perf_counter() - t0
        return [e.get_field(c) for c in range(6)], e.results(), pairs, dt


ref_f, ref_m, p0, t0 = run(0)
got_f, got_m, p1, t1 = run(-1)
ok = p0 == 0 and all(np.array_equal(a, b) for a, b in zip(ref_f, got_f)) and all(np.array_equal(ref_m[k], got_m[k]) for k in ref_m)
print(f"{wl} {n}^3, {steps} steps: single sweeps {t0:.2f} s, two steps per sweep {t1:.2f} s ({p1} pairs); max|E| {max(float(np.abs(f).max()) for f in ref_f[:3]):.3g}; "
      f"monitors {sorted(ref_m)}; bit-identical: {ok}")
sys.exit(0 if ok else 1)
