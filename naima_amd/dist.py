"""Walker sharding across the GPUs of one node.

One process per GPU.  Within a half-step every proposal needs only its own
position, one partner from the inactive half and two random numbers, so each rank
evaluates a contiguous block of the proposed walkers and the ensemble meets in ONE
all-gather of the new log-probabilities per half-step (a few KB: latency-bound,
xGMI bandwidth is irrelevant).  The random stream is replicated (same seed on
every rank), so proposals and accept decisions are identical everywhere and no
coordinates have to move.

Backends
  * ``RcclComm``  -- the product path: RCCL straight from libnaima_hip
    (nh_comm_*), device buffers, the context's stream.  torch.distributed (gloo)
    is used once, to hand the RCCL unique id to the other ranks.
  * ``GlooComm``  -- CPU-only, for the world_size-2 tests that run without a GPU.
  * ``LocalComm`` -- a single process.
"""
import os

import numpy as np


class LocalComm:
    rank, size = 0, 1

    def allgather(self, x):
        return np.asarray(x, dtype=float)

    def barrier(self):
        pass

    def max(self, v):
        return float(v)

    in_stream = True  # allgather_device enqueues on the context's stream

    def allgather_device(self, ctx, send_ptr, recv, n):
        ctx.call("nh_copy", recv, send_ptr, 8 * n)


def shard_bounds(n, rank, size):
    """contiguous block [lo, hi) of n items owned by ``rank``; blocks differ by <= 1"""
    base, rem = divmod(n, size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_counts(n, size):
    return [shard_bounds(n, r, size)[1] - shard_bounds(n, r, size)[0] for r in range(size)]


def _ensure_gloo():
    import datetime

    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
    return dist


class GlooComm:
    """torch.distributed/gloo on CPU tensors (tests; also the control plane)"""

    def __init__(self):
        self.dist = _ensure_gloo()
        self.rank = self.dist.get_rank()
        self.size = self.dist.get_world_size()

    def allgather(self, x):
        """ranks contribute equal-sized blocks (the sampler pads to the largest shard)"""
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

    in_stream = False  # host-staged: synchronises the stream (tests only)

    def allgather_device(self, ctx, send_ptr, recv, n):
        host = np.empty(n)
        from . import _lib
        ctx.join()
        _lib._chk(_lib._lib.nh_download(ctx.h, host.ctypes.data, send_ptr, host.nbytes))
        recv.set(self.allgather(host))


class RcclComm:
    """RCCL all-gather through the C ABI (nh_comm_allgather) on device buffers"""

    def __init__(self, ctx=None):
        import ctypes as C

        from . import _lib
        self.ctx = ctx if ctx is not None else _lib.get_context()
        dist = _ensure_gloo()
        self.dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        lib = _lib.load()
        # NCCL_DEBUG=VERSION makes RCCL print a banner on STDOUT, where bench.py owes the
        # driver exactly one JSON line
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            _lib._chk(lib.nh_comm_unique_id(buf))
        box = [buf.raw]
        dist.broadcast_object_list(box, src=0)
        _lib._chk(lib.nh_comm_init(self.ctx.h, self.rank, self.size, box[0]))
        self._send = self._recv = None
        self._graph_ok = None

    def graph_capture_ok(self, timeout=120.0):
        """True when an all-gather of this communicator's ranks has been seen to survive
        hipGraph capture and replay.  Probed once, in a throw-away process per rank with
        its own rendezvous (``python -m naima_amd._rccl_probe``): a collective that hangs
        under capture then costs the probe its timeout, not the run.  Every rank gets the
        same answer (minimum over ranks)."""
        if self._graph_ok is None:
            import subprocess
            import sys

            import torch
            ok = False
            proc = None
            try:
                env = dict(os.environ)
                env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 23)
                env["NAIMA_AMD_DEVICE"] = str(self.ctx.device)
                env["RANK"], env["WORLD_SIZE"] = str(self.rank), str(self.size)
                env.pop("NAIMA_AMD_FORCE_SHARDED", None)
                # under torchrun the workers are clients of the AGENT's store on
                # MASTER_PORT; the probes make their own store on the shifted port
                env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
                root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
                proc = subprocess.Popen([sys.executable, "-m", "naima_amd._rccl_probe"], env=env,
                                        cwd=root, stdout=subprocess.PIPE,
                                        stderr=subprocess.DEVNULL)
                out, _ = proc.communicate(timeout=timeout)
                ok = proc.returncode == 0 and b"captured + replayed ok: True" in out
            except Exception:
                ok = False
                if proc is not None and proc.poll() is None:
                    proc.kill()  # this probe only: the PID we started
                    proc.wait()
            t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
            self._graph_ok = float(t.item()) == 1.0
        return self._graph_ok

    def allgather(self, x):
        """ranks contribute equal-sized blocks (the sampler pads to the largest shard)"""
        x = np.ascontiguousarray(x, dtype=float)
        n = x.size
        if self._send is None or self._send.size != n:
            self._send = self.ctx.empty((n,))
            self._recv = self.ctx.empty((n * self.size,))
        self._send.set(x.ravel())
        self.ctx.call("nh_comm_allgather", self._send, self._recv, n)
        return self._recv.get().reshape((self.size * x.shape[0],) + x.shape[1:])

    in_stream = True  # RCCL on the context's stream: no host synchronisation

    def allgather_device(self, ctx, send_ptr, recv, n):
        ctx.call("nh_comm_allgather", send_ptr, recv, n)

    def barrier(self):
        self.ctx.sync()
        self.dist.barrier()

    def max(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def from_env(prefer="rccl"):
    """LocalComm for a single process, else RCCL (GPU) or gloo (CPU tests)"""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and \
            os.environ.get("NAIMA_AMD_FORCE_SHARDED", "0") != "1":
        return LocalComm()
    if prefer != "rccl":
        return GlooComm()
    comm = err = None
    try:
        comm = RcclComm()
    except Exception as e:  # librccl missing / communicator refused
        err = e
    import torch
    dist = _ensure_gloo()
    ok = torch.tensor([1.0 if comm is not None else 0.0], dtype=torch.float64)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank takes the same path
    if float(ok.item()) == 1.0:
        return comm
    import warnings
    warnings.warn("RCCL communicator unavailable (%s); all-gathers are staged through the host "
                  "with gloo: correct, slower" % (err,))
    return GlooComm()
