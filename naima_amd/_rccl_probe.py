"""``python -m naima_amd._rccl_probe``: does an RCCL all-gather survive hipGraph capture and
replay on THIS set of ranks?  Run by ``RcclComm.graph_capture_ok`` in a throw-away process
per rank (own rendezvous port), so that a collective that hangs under capture takes the
probe down, not the run.  Prints ``captured + replayed ok: True`` and exits 0 on success."""
import os
import sys

import numpy as np


def main():
    from . import _lib
    from .dist import RcclComm
    ctx = _lib.get_context()
    comm = RcclComm(ctx)
    n = 256
    a = ctx.array(np.full(n, float(comm.rank)))
    b = ctx.empty((n * comm.size,))
    comm.allgather_device(ctx, a.ptr, b, n)
    ctx.sync()
    want = np.repeat(np.arange(comm.size, dtype=float), n)
    print("eager ok:", bool(np.array_equal(b.get(), want)), flush=True)
    ctx.graph_begin()
    comm.allgather_device(ctx, a.ptr, b, n)
    g = ctx.graph_end()
    ok = True
    for rep in range(1, 4):
        a.set(np.full(n, float(comm.rank) + 10.0 * rep))
        ctx.graph_launch(g)
        ctx.sync()
        ok = ok and bool(np.array_equal(b.get(), want + 10.0 * rep))
    print("captured + replayed ok:", ok, flush=True)
    comm.barrier()
    return 0 if ok else 1


if __name__ == "__main__":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    sys.exit(main())
