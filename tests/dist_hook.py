"""TEST INFRASTRUCTURE: hook for ``python -m tidy3d_amd.dist_main --hook dist_hook:setup`` — connects the emulated
library's RCCL shim to torch.distributed (gloo), as tests/dist_worker.py does for the hand-rolled workers."""
import dist_worker

_keep = []


def setup(lib):
    cb = dist_worker.EXCHANGE_FN(dist_worker._exchange)
    _keep.append(cb)
    lib.dll.hipemu_set_exchange(cb, None)
