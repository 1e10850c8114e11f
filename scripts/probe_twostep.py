"""A/B of the two-steps-per-sweep kernel inside ONE engine (same placement of the field arrays): option values of
FDTD_OPT_TWOSTEP (0 = single steps) timed in turn on the bench workload, several rounds.
    python scripts/probe_twostep.py [--lib path.so] [--n 512] [--steps 60] [--rounds 3] v0 v1 ...
v = waves + 64 * planes per chunk; `auto` = the library's default (shape by grid size)"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_spec  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("values", nargs="+", help="option values; `auto` = -1 (library default)")
    args = ap.parse_args()
    args.values = [-1 if v == "auto" else int(v) for v in args.values]
    lib = L.load_library(args.lib)
    n = args.n
    spec = build_spec(n, args.steps * (args.rounds * len(args.values) + 2) + 64, "v0")
    with HipEngine(spec, lib=lib, axis_shift=0) as e:
        for kv in args.opt:
            name, val = kv.split("=")
            e.set_option(getattr(L, name), int(val))
        rng = np.random.default_rng(0)
        for c in range(6):
            e.set_field(c, rng.uniform(-1e-3, 1e-3, (n, n, n)).astype(np.float32))
        e.run(args.steps)            # placement probe, warm-up
        res = {v: [] for v in args.values}
        shape = {}
        for r in range(args.rounds):
            for v in args.values:
                e.set_option(L.OPT_TWOSTEP, v)
                st = e.run(args.steps)
                res[v].append(st.run_ms / args.steps)
                shape[v] = int(st.fused2_shape)
        for v in args.values:
            ms = sorted(res[v])[len(res[v]) // 2]
            print(json.dumps({"lib": os.path.basename(args.lib or "default"), "n": n, "twostep": v, "waves": shape[v] & 63,
                              "zchunk": shape[v] >> 6, "ms_per_step": round(ms, 5),
                              "all": [round(x, 5) for x in res[v]], "gcells_per_s": round(n ** 3 / ms / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
