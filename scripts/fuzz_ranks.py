"""Random simulations of scripts/fuzz_variants.py on 2 and 3 z-slab ranks (gloo processes, the library on the CPU emulator) against
the single-slab run of the same library: fields inside the walls and every record, bit for bit.  The unattended form of
tests/test_dist_gloo.py::test_random_simulations_on_two_and_three_ranks:
    python scripts/fuzz_ranks.py [n_cases] [seed]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import build_emu  # noqa: E402
import fuzz_variants  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402
from tidy3d_amd.lib import load_library  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = load_library(build_emu.build())
    rng = np.random.default_rng(seed)
    bad = skipped = 0
    for q in range(n_cases):
        disc, steps, per_z, desc, *_ = fuzz_variants.draw(rng, False)
        world = 2 + q % 2
        if disc.spec.shape[2] < 4 * world:
            skipped += 1
            continue
        out = os.path.join(tempfile.mkdtemp(prefix="fuzz_ranks_"), "dist.npz")
        port = 29800 + q % 100
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), f"fuzz:{seed}:{q}", str(steps), out]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(f"case {q}: {desc} world={world} -> WORKER FAILED\n{r.stderr[-1500:]}", flush=True)
            bad += 1
            continue
        got = np.load(out)
        if "refused" in got:
            print(f"case {q}: {desc} world={world} -> refused on every rank ({got['refused']})", flush=True)
            skipped += 1
            continue
        sl = fuzz_variants.inside(disc)
        with HipEngine(disc.spec, lib=lib, variant=L.VARIANT_AUTO, axis_shift=0) as e:
            e.run()
            ref, fields = e.results(), [e.get_field(c) for c in range(6)]
        diff = [c for c in range(6) if not np.array_equal(got[f"field{c}"][sl], fields[c][sl])] + \
               [k for k, v in ref.items() if not np.array_equal(got[f"mon_{k}"], v)]
        print(f"case {q}: {desc} world={world} -> {'ok' if not diff else 'DIFFERS: ' + str(diff)}", flush=True)
        bad += bool(diff)
    print(f"fuzz_ranks: {n_cases - bad - skipped} of {n_cases - skipped} cases bit-identical on 2 / 3 ranks ({skipped} skipped)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
