"""Worker of the ``tidy3d_amd.dist.run`` test: every rank calls the distributed counterpart of ``web.run`` with the
same Simulation (emulated library, gloo); rank 0 gets the SimulationData and stores its monitor values."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))


def main():
    case, n_steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dist.init_process_group(backend="gloo")
    import build_emu
    import cases
    import dist_worker
    from tidy3d_amd import dist as tdist
    from tidy3d_amd.lib import load_library
    lib = load_library(build_emu.build())
    cb = dist_worker.EXCHANGE_FN(dist_worker._exchange)
    lib.dll.hipemu_set_exchange(cb, None)
    sd = tdist.run(cases.CASES[case](), verbose=False, n_steps=n_steps, lib=lib, device=0)
    if dist.get_rank() == 0:
        vals = {}
        for d in sd.data:
            comps = getattr(d, "field_components", None) or {"flux": d.flux}
            for k, v in comps.items():
                vals[f"{d.monitor.name}__{k}"] = np.asarray(v.values)
        np.savez(out, **vals)
    else:
        assert sd is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
