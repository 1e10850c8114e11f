"""Worker of the world_size-2 gloo test: runs the library's multi-rank path on CPU.  The HIP
sources run under the emulator; its RCCL shim hands every grouped Send/Recv to the callback
below, which performs them with torch.distributed (gloo)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))


class P2POp(C.Structure):
    _fields_ = [("kind", C.c_int), ("peer", C.c_int), ("buf", C.c_void_p), ("bytes", C.c_size_t)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(P2POp), C.c_int, C.c_void_p)


def _exchange(ops, n, _user):
    reqs, keep = [], []
    for i in range(n):
        op = ops[i]
        if op.kind in (0, 1):
            arr = np.ctypeslib.as_array(C.cast(op.buf, C.POINTER(C.c_float)), shape=(op.bytes // 4,))
            t = torch.from_numpy(arr)
            keep.append(t)
            reqs.append(dist.isend(t, op.peer) if op.kind == 1 else dist.irecv(t, op.peer))
        else:
            ct, dt = (C.c_float, 4) if op.kind == 2 else (C.c_double, 8)
            arr = np.ctypeslib.as_array(C.cast(op.buf, C.POINTER(ct)), shape=(op.bytes // dt,))
            t = torch.from_numpy(arr)
            dist.all_reduce(t)
    for r in reqs:
        r.wait()
    return 0


def main():
    case, n_steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import build_emu
    from tidy3d_amd.lib import load_library
    from tidy3d_amd import dist as tdist
    from tidy3d_amd.discretize import discretize
    import cases
    lib = load_library(build_emu.build())
    cb = EXCHANGE_FN(_exchange)
    lib.dll.hipemu_set_exchange(cb, None)
    if case.startswith("fuzz:"):            # "fuzz:<seed>:<index>": a random simulation of scripts/fuzz_variants.py (every rank draws the same)
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import fuzz_variants
        _, seed, index = case.split(":")
        rng = np.random.default_rng(int(seed))
        for _ in range(int(index) + 1):
            disc = fuzz_variants.draw(rng, False)[0]
    elif case.startswith("slabfuzz:"):      # "slabfuzz:<seed>:<index>": a random CPML-walled box (cases.random_slab_pml_box; every rank draws the same)
        _, seed, index = case.split(":")
        disc = discretize(cases.random_slab_pml_box(int(seed), int(index))[0], n_steps=n_steps)
        disc.spec.decay_every = 10
    else:
        sim = cases.CASES[case]() if case in cases.CASES else getattr(cases, case)()
        disc = discretize(sim, n_steps=n_steps)
        disc.spec.decay_every = 10          # exercise the scalar all-reduce too
    shift = os.environ.get("SLAB_SHIFT")                    # force a slab-axis renaming (tests)
    from tidy3d_amd.exceptions import SetupError
    try:
        eng = tdist.make_engine(disc.spec, lib=lib, device=0,   # the emulator exposes one device
                                axis_shift=None if shift is None else int(shift))
    except SetupError as err:               # a split the front end refuses — on every rank alike (decided from the split alone)
        if rank == 0:
            np.savez(out, refused=str(err))
        dist.barrier()
        dist.destroy_process_group()
        return
    if os.environ.get("PML_FUSED"):                         # CPML recursions inside the sweeps of the slab ranks (tests)
        from tidy3d_amd import lib as L
        eng.set_option(L.OPT_PML_FUSED, int(os.environ["PML_FUSED"]))
    if os.environ.get("TWOSTEP"):                           # step pairs on the slab ranks, whatever the grid size (tests)
        from tidy3d_amd import lib as L
        eng.set_option(L.OPT_TWOSTEP, int(os.environ["TWOSTEP"]))
    if os.environ.get("SLAB_BOXES"):                        # where the shell's boxes of a CPML slab pair go (tests: 2 = a third stream)
        from tidy3d_amd import lib as L
        eng.set_option(L.OPT_SLAB_BOXES_FIRST, int(os.environ["SLAB_BOXES"]))
    if os.environ.get("PLACEMENT_TRIES"):                   # force the placement probe of the library on (tests)
        from tidy3d_amd import lib as L
        eng.set_option(L.OPT_PLACEMENT_TRIES, int(os.environ["PLACEMENT_TRIES"]))
    st = eng.run()
    pairs = torch.tensor([int(st.fused2_pairs)], dtype=torch.int64)
    allp = [torch.zeros_like(pairs) for _ in range(world)]
    dist.all_gather(allp, pairs)
    raw = tdist.gather_results(eng)
    s_ax = getattr(eng, "slab_shift", 0)
    fields = [eng.get_field((c % 3 - s_ax) % 3 + 3 * (c // 3)) for c in range(6)]     # device component of user c
    allf = [None] * world if rank == 0 else None
    dist.gather_object((eng.z0, fields), allf, dst=0)
    eng.close()
    if rank == 0:
        allf.sort(key=lambda p: p[0])
        from tidy3d_amd.engine import unpermute_array
        full = {f"field{c}": unpermute_array(np.concatenate([p[1][c] for p in allf], axis=0), s_ax) for c in range(6)}
        np.savez(out, decay=st.field_decay, pairs=np.asarray([int(x.item()) for x in allp]), **{f"mon_{k}": v for k, v in raw.items()}, **full)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
