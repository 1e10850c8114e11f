"""bench.py's N > 1 branch (what the driver launches for the scaling runs: one process per GPU under
torch.distributed.run) exercised on CPU: BENCH_EMULATE=1 swaps RCCL/CUDA for gloo and the emulated library and
leaves the rank logic — slab split, communicator set-up, barrier-bracketed timing, max over ranks, the single
JSON line from rank 0 — exactly as on the GPU node."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("world", [1, 2, 4])
def test_bench_contract_line(world, emu_lib):
    env = dict(os.environ, BENCH_EMULATE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29641 + world))
    args = ["--gpus", str(world), "--steps", "3", "--warmup", "1", "--size", "16", "--no-cpu"]
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(29641 + world), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only, ONE line
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == world and out["steps"] == 3 and out["warmup"] == 1 and out["unit"] == "Mcells/s"
    assert out["value"] == pytest.approx(16 ** 3 * 3 / (out["ms_per_step"] * 3e-3) / 1e6, rel=1e-6)
    assert out["config"]["parallelism"] == f"z-slab x{world}" and out["roofline"]["bound"] == "hbm"
    assert out["roofline"]["kernel"] == "fused_step_kernel"     # every rank takes the fused z-slab schedule
    if world > 1:
        # the line itself proves the halo communicator spans `world` ranks (ncclCommCount / ncclCommUserRank per rank)
        rc = out["rccl"]
        assert rc["rccl_ranks"] == [world] and rc["comm_user_ranks"] == list(range(world))
        assert sum(rc["planes_per_rank"]) == 16 and len(rc["ms_per_step_per_rank"]) == world
        assert rc["ms_per_step_rank_min"] <= rc["ms_per_step_rank_max"] <= out["ms_per_step"] * 1.0001
    else:
        assert "rccl" not in out
