"""TEST-ONLY communicator over torch.distributed/gloo (CPU tensors): the world_size-2 CPU
tests run the sharded sampler over it as well as over naima_amd.dist.HostComm.  The product
(naima_amd/, bench.py) never imports torch."""
import datetime
import os

import numpy as np


class GlooComm:
    in_stream = False

    def __init__(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        self.dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()

    def allgather(self, x):
        import torch
        x = np.ascontiguousarray(x, dtype=float)
        outs = [torch.zeros(x.shape, dtype=torch.float64) for _ in range(self.size)]
        self.dist.all_gather(outs, torch.from_numpy(x))
        return np.concatenate([o.numpy() for o in outs], axis=0)

    def barrier(self):
        self.dist.barrier()

    def max(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather_device(self, ctx, send_ptr, recv, n):
        from naima_amd import _lib
        host = np.empty(n)
        ctx.join()
        _lib._chk(_lib._lib.nh_download(ctx.h, host.ctypes.data, send_ptr, host.nbytes))
        recv.set(self.allgather(host))
