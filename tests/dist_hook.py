"""TEST INFRASTRUCTURE: hook for ``python -m tidy3d_amd.dist_main --hook dist_hook:setup`` — connects the emulated
library's RCCL shim to torch.distributed (gloo), as tests/dist_worker.py does for the hand-rolled workers."""
import dist_worker

_keep = []


def setup(lib):
    cb = dist_worker.EXCHANGE_FN(dist_worker._exchange)
    _keep.append(cb)
    lib.dll.hipemu_set_exchange(cb, None)


def setup_rank1_dies_noisily(lib):
    """Rank 1 writes far more than a pipe buffer to stderr and exits non-zero while rank 0 is already waiting in its first
    ghost-plane exchange: web.run must notice, take rank 0 down and raise (tests/test_dist_gloo.py)."""
    import os
    import sys
    setup(lib)
    if os.environ.get("RANK") == "1":
        sys.stderr.write("rccl warning line\n" * 20000)          # ~360 KiB: a pipe nobody drained would block here
        sys.stderr.flush()
        os._exit(7)
