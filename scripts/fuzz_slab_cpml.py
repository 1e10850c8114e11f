"""Random CPML-walled boxes on 2 - 4 z-slab ranks with step pairs forced (gloo processes, the library on the CPU emulator) against the
single-slab run of the same library: every field and record, bit for bit — the unattended form of
tests/test_dist_gloo.py::test_step_pairs_on_slab_ranks_that_carry_cpml:
    python scripts/fuzz_slab_cpml.py [n_cases] [seed]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import build_emu  # noqa: E402
import cases  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402
from tidy3d_amd.lib import load_library  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = load_library(build_emu.build())
    bad = in_pairs = 0
    for q in range(n_cases):
        sim, world, twostep, steps = cases.random_slab_pml_box(seed, q)
        out = os.path.join(tempfile.mkdtemp(prefix="fuzz_slab_"), "dist.npz")
        port = 29850 + q % 100
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TWOSTEP=str(twostep), PML_FUSED="7")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), f"slabfuzz:{seed}:{q}", str(steps), out]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
        desc = f"{sim.grid_shape if hasattr(sim, 'grid_shape') else ''} world={world} twostep={twostep & 63}x{twostep >> 6} steps={steps}"
        if r.returncode != 0:
            print(f"case {q}: {desc} -> WORKER FAILED\n{r.stderr[-1500:]}", flush=True)
            bad += 1
            continue
        got = np.load(out)
        disc = discretize(sim, n_steps=steps)
        disc.spec.decay_every = 10
        with HipEngine(disc.spec, lib=lib) as e:
            e.run()
            ref, fields = e.results(), [e.get_field(c) for c in range(6)]
        diff = [c for c in range(6) if not np.array_equal(got[f"field{c}"], fields[c])] + \
               [k for k, v in ref.items() if not np.array_equal(got[f"mon_{k}"], v)]
        in_pairs += bool(got["pairs"].max() > 0)
        print(f"case {q}: shape={disc.spec.shape} {desc} pairs={got['pairs'].tolist()} -> {'ok' if not diff else 'DIFFERS: ' + str(diff)}", flush=True)
        bad += bool(diff)
    print(f"fuzz_slab_cpml: {n_cases - bad} of {n_cases} cases bit-identical ({in_pairs} with a rank in step pairs)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
