"""GPU probe (round 2): the fused sweep across workloads (V0 plain, V1 materials, V2 materials + CPML), tile
shapes, CPML placements and — for same-box A/B of two builds — libraries.  One process per workload so a
launch failure cannot poison the next.  One JSON line per configuration.

    python scripts/probe_r02.py [n] [workloads]          configurations: $PROBE_CFGS = {"v2": [{...}, ...], "*": [...]}
    keys of a configuration: rows, zc, pml (-1 = default), remap (absent = library default),
                             lib (tag: tidy3d_amd/libfdtd_<tag>.so instead of the product library),
                             lds_pad (bytes of extra LDS per workgroup: 0 -> 4, 30000 -> 3, 60000 -> 2, 100000 -> 1 workgroups per CU)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n, wl, cfgs):
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    from tidy3d_amd import lib as L
    from tidy3d_amd.engine import HipEngine
    spec = bench.build_spec(n, 4000, wl)
    engines = {}

    def engine(tag):
        if tag not in engines:
            lib = None if tag is None else L.load_library(os.path.join(ROOT, "tidy3d_amd", "libfdtd_%s.so" % tag))
            e = HipEngine(spec, lib=lib)
            for c in range(6):
                arr = np.empty((n, n, n), dtype=np.float32)
                for k in range(n):
                    arr[k] = np.random.default_rng(c * 100003 + k).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
                e.set_field(c, arr)
            engines[tag] = e
        return engines[tag]
    for cfg in cfgs:
        try:
            eng = engine(cfg.get("lib"))
            eng.set_option(L.OPT_ROWS, cfg.get("rows", 3))
            eng.set_option(L.OPT_ZCHUNK, cfg.get("zc", 16))
            eng.set_option(L.OPT_PML_FUSED, cfg.get("pml", -1))
            if cfg.get("lds_pad"):
                eng.set_option(L.OPT_LDS_PAD, cfg["lds_pad"])
            if "remap" in cfg:
                eng.set_option(L.OPT_XCD_REMAP, cfg["remap"])
            eng.set_option(L.OPT_FLAGS, 0)
            eng.run(5)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); eng.run(20); ts.append(time.perf_counter() - t0)
            t = sorted(ts)[1]
            eng.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
            st = eng.run(10)
            print(json.dumps({"wl": wl, "n": n, **{k: v for k, v in cfg.items()}, "ms_per_step": t / 20 * 1e3, "gcells": n**3 * 20 / t / 1e9,
                              "fused_ms_per_step": st.fused_kernel_ms / 10}), flush=True)
        except Exception as e:
            print(json.dumps({"wl": wl, **cfg, "error": str(e)[:200]}), flush=True)
    for e in engines.values():
        e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3], json.loads(sys.argv[4]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    wls = sys.argv[2].split(",") if len(sys.argv) > 2 else ["v0", "v1", "v2"]
    custom = json.loads(os.environ["PROBE_CFGS"]) if os.environ.get("PROBE_CFGS") else {}
    default = [dict(rows=3, zc=16), dict(rows=7, zc=16)]
    for wl in wls:
        cfgs = custom.get(wl, custom.get("*", default))
        # one process per configuration that names a library or an environment ("env": {...}): device allocations of
        # engines created one after the other in ONE process land differently, and the sweep is sensitive to that
        # (profiles/r02m: the first engine of a process ran a byte-identical kernel 12 % faster than the third)
        groups = [(None, [cfg]) for cfg in cfgs]
        for _, grp in groups:
            env = dict(os.environ)
            for c in grp:
                env.update({k: str(v) for k, v in (c.get("env") or {}).items()})
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), wl, json.dumps(grp)],
                               capture_output=True, text=True, env=env)
            sys.stdout.write(r.stdout)
            if r.returncode != 0:
                print(json.dumps({"wl": wl, "error": r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else "rc %d" % r.returncode}))
            sys.stdout.flush()
