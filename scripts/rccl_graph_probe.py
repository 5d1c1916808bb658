"""does an RCCL all-gather survive hipGraph capture and replay?  (1-rank communicator: an
indication only; the sharded loop keeps the collective BETWEEN two graphs)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
from naima_amd import _lib
from naima_amd.dist import RcclComm
ctx = _lib.get_context()
comm = RcclComm(ctx)
a = ctx.array(np.arange(256.0)); b = ctx.empty((256,))
comm.allgather_device(ctx, a.ptr, b, 256); ctx.sync()
print("eager ok:", np.array_equal(b.get(), np.arange(256.0)))
try:
    ctx.graph_begin()
    comm.allgather_device(ctx, a.ptr, b, 256)
    g = ctx.graph_end()
    a.set(np.arange(256.0) * 2)
    for _ in range(3):
        ctx.graph_launch(g)
    ctx.sync()
    print("captured + replayed ok:", np.array_equal(b.get(), np.arange(256.0) * 2))
except Exception as e:
    print("capture failed:", e)
