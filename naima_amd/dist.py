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
    (nh_comm_*), device buffers, the context's stream.
  * ``HostComm``  -- all-gathers staged through the host over the control plane
    (CPU-only tests; the fallback when no RCCL communicator can be built).
  * ``LocalComm`` -- a single process.

Control plane: ``SocketGroup``, a stdlib-``socket`` rendezvous (rank 0 is the hub of a
star over TCP on MASTER_ADDR).  It carries the 128-byte RCCL unique id, barriers and the
max/min of a scalar -- nothing on the data path.  No PyTorch anywhere: the launcher only
has to provide RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (``torch.distributed.run``
does, so does any other launcher).
"""
import hashlib
import os
import socket
import struct
import time

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


# ----------------------------------------------------------------------------------------
# control plane
# ----------------------------------------------------------------------------------------
_MAGIC = b"NHRV1"
_PORT_OFFSET = 101   # the launcher's own store usually sits on MASTER_PORT itself
_PORT_STRIDE = 7
_PORT_TRIES = 24


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("control-plane peer closed the connection")
        buf.extend(chunk)
    return bytes(buf)


_MAX_MSG = 64 << 20  # nothing on the control plane comes near this (ids, scalars, staged rows)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > _MAX_MSG:
        raise ConnectionError("control-plane message of %d bytes refused (cap %d; a gathered message "
                              "is every rank's payload together)" % (n, _MAX_MSG))
    return _recv_exact(sock, n)


class SocketGroup:
    """rank 0 listens, everyone else connects; every collective is a gather to rank 0
    followed by a broadcast of the gathered list (payloads are tens of bytes).

    The port is ``MASTER_PORT + 101 + 7 k`` for the first k whose bind succeeds; clients
    try the same candidates and recognise THEIR hub by a token derived from the job's
    rendezvous variables (another service, or another job's hub, fails the handshake and
    the client moves on).  ``salt`` separates several groups of one job (the RCCL capture
    probes make their own)."""

    def __init__(self, rank, size, addr=None, port=None, salt="", timeout=300.0):
        self.rank, self.size, self.timeout = int(rank), int(size), float(timeout)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
        # (NAIMA_AMD_GROUP_SECRET: a value only the launcher's processes know, for hosts where
        # the rendezvous variables alone -- all public -- are not credential enough)
        ident = "%s:%d:%d:%s:%s:%s" % (addr, base, self.size,
                                       os.environ.get("TORCHELASTIC_RUN_ID", ""), salt,
                                       os.environ.get("NAIMA_AMD_GROUP_SECRET", ""))
        self.token = hashlib.sha256(ident.encode()).digest()[:16]
        # the hub's answer is NOT the hello's prefix: a client that connects to a loopback port
        # nobody listens on yet can be handed that very port as its own source port (TCP
        # simultaneous open: the socket is connected to ITSELF), reads its own hello back and
        # -- with a symmetric handshake -- takes itself for the hub; its first collective then
        # reads the four bytes of its own rank and half of its own header as a length (round 6:
        # "control-plane message of 17592186044417 bytes", one run in ~10 of two self-started ranks)
        self.reply = hashlib.sha256(b"hub:" + self.token).digest()[:16]
        self.peers = {}     # rank 0: rank -> socket
        self.sock = None    # other ranks: the socket to rank 0
        self._srv = None
        ports = [base + _PORT_OFFSET + _PORT_STRIDE * k for k in range(_PORT_TRIES)]
        ports = [p for p in ports if p < 65536] or [base]
        if self.size == 1:
            return
        if self.rank == 0:
            self._serve(addr, ports)
        else:
            self._connect(addr, ports)

    def _serve(self, addr, ports):
        srv = None
        # the hub listens on the interface the ranks are told to connect to (MASTER_ADDR),
        # not on every interface of the host; a name that does not resolve to a local address
        # (some launchers pass the node's public name) falls back to all interfaces
        bind_addr = "0.0.0.0"
        try:
            cand = socket.gethostbyname(addr)
            probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                probe.bind((cand, 0))
                # (a node's own NAME may resolve to a loopback address on the node itself --
                # Debian's 127.0.1.1 -- while the other nodes resolve it to the real one: only a
                # literal loopback address, or "localhost", is taken as "this host only")
                literal = addr == "localhost" or addr.startswith("127.")
                if not cand.startswith("127.") or literal:
                    bind_addr = cand
            finally:
                probe.close()
        except OSError:
            pass
        for p in ports:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((bind_addr, p))
            except OSError:
                s.close()
                continue
            srv = s
            break
        if srv is None:
            raise RuntimeError("no free control-plane port among %s" % (ports,))
        srv.listen(self.size + 8)
        srv.settimeout(1.0)
        self._srv = srv
        deadline = time.time() + self.timeout
        while len(self.peers) < self.size - 1:
            if time.time() > deadline:
                raise TimeoutError("control plane: %d of %d ranks joined within %.0f s"
                                   % (len(self.peers) + 1, self.size, self.timeout))
            try:
                c, _ = srv.accept()
            except socket.timeout:
                continue
            try:
                c.settimeout(10.0)
                hello = _recv_exact(c, len(_MAGIC) + 16 + 4)
                r = struct.unpack("<i", hello[-4:])[0]
                if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC):-4] != self.token or \
                        not (0 < r < self.size) or r in self.peers:
                    c.close()
                    continue
                c.sendall(_MAGIC + self.reply)
                c.settimeout(self.timeout)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self.peers[r] = c
            except (OSError, ConnectionError, struct.error):
                c.close()

    def _client_handshake(self, s):
        """True when the socket's other end is THIS job's hub"""
        if s.getsockname() == s.getpeername():  # (connected to itself: see self.reply)
            return False
        s.settimeout(10.0)
        s.sendall(_MAGIC + self.token + struct.pack("<i", self.rank))
        return _recv_exact(s, len(_MAGIC) + 16) == _MAGIC + self.reply

    def _connect(self, addr, ports):
        deadline = time.time() + self.timeout
        while True:
            for p in ports:
                try:
                    s = socket.create_connection((addr, p), timeout=2.0)
                except OSError:
                    continue
                try:
                    if self._client_handshake(s):
                        s.settimeout(self.timeout)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self.sock = s
                        return
                except (OSError, ConnectionError):
                    pass
                s.close()
            if time.time() > deadline:
                raise TimeoutError("control plane: rank %d found no hub on %s ports %s"
                                   % (self.rank, addr, ports[:4]))
            time.sleep(0.05)

    # -- collectives (every rank calls them in the same order) -------------------------
    def allgather_bytes(self, payload):
        payload = bytes(payload)
        if self.size == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(self.peers[r]) for r in range(1, self.size)]
            blob = b"".join(struct.pack("<Q", len(x)) + x for x in parts)
            for r in range(1, self.size):
                _send_msg(self.peers[r], blob)
            return parts
        _send_msg(self.sock, payload)
        blob, parts, off = _recv_msg(self.sock), [], 0
        for _ in range(self.size):
            (n,) = struct.unpack_from("<Q", blob, off)
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def bcast(self, payload, src=0):
        return self.allgather_bytes(payload if self.rank == src else b"")[src]

    def barrier(self):
        self.allgather_bytes(b"")

    def reduce_scalar(self, v, op):
        vals = [struct.unpack("<d", x)[0] for x in self.allgather_bytes(struct.pack("<d", float(v)))]
        return max(vals) if op == "max" else min(vals)

    def free_port(self):
        """a TCP port that is free on rank 0 right now, agreed by every rank.  Rank 0 keeps
        the probing socket (SO_REUSEADDR, never listening) open until every rank holds the
        number, so nothing else on the host is handed the port in between; whoever binds it
        next with SO_REUSEADDR -- the hub of the group that asked -- succeeds."""
        p, s = b"", None
        if self.rank == 0:
            s = socket.socket()
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            s.bind(("", 0))
            p = struct.pack("<i", s.getsockname()[1])
        try:
            return struct.unpack("<i", self.bcast(p))[0]
        finally:
            if s is not None:
                s.close()

    def close(self):
        for c in list(self.peers.values()) + [self.sock, self._srv]:
            try:
                if c is not None:
                    c.close()
            except OSError:
                pass
        self.peers, self.sock, self._srv = {}, None, None


_group = None


def default_group():
    """the process-wide control plane built from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT"""
    global _group
    if _group is None:
        _group = SocketGroup(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                             salt=os.environ.get("NAIMA_AMD_GROUP_SALT", ""))
    return _group


class HostComm:
    """all-gathers staged through the host over the control plane: the CPU tests, and the
    (slow, correct) fallback when RCCL is unavailable"""

    in_stream = False  # synchronises the stream

    def __init__(self, group=None):
        self.group = group if group is not None else default_group()
        self.rank, self.size = self.group.rank, self.group.size

    def allgather(self, x):
        """ranks contribute equal-sized blocks (the sampler pads to the largest shard)"""
        x = np.ascontiguousarray(x, dtype=float)
        parts = self.group.allgather_bytes(x.tobytes())
        return np.concatenate([np.frombuffer(p, dtype=float).reshape(x.shape) for p in parts],
                              axis=0)

    def barrier(self):
        self.group.barrier()

    def max(self, v):
        return self.group.reduce_scalar(v, "max")

    def allgather_device(self, ctx, send_ptr, recv, n):
        from . import _lib
        host = np.empty(n)
        ctx.join()
        _lib._chk(_lib._lib.nh_download(ctx.h, host.ctypes.data, send_ptr, host.nbytes))
        recv.set(self.allgather(host))


class RcclComm:
    """RCCL all-gather through the C ABI (nh_comm_allgather) on device buffers.  Building
    one is a collective over ``group``: every rank takes the same branch at every step,
    so a rank that cannot load librccl (checked and agreed BEFORE the blocking
    ncclCommInitRank) or a hub that cannot make the unique id makes ALL ranks raise
    ``RcclUnavailable`` together instead of leaving the others waiting; a control-plane
    time-out or a lost peer during construction is reported the same way."""

    def __init__(self, ctx=None, group=None):
        import ctypes as C

        from . import _lib
        self.ctx = ctx if ctx is not None else _lib.get_context()
        self.group = group if group is not None else default_group()
        self.rank, self.size = self.group.rank, self.group.size
        lib = _lib.load()
        # NCCL_DEBUG=VERSION makes RCCL print a banner on STDOUT, where bench.py owes the
        # driver exactly one JSON line
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]
        buf = C.create_string_buffer(128)
        err = ""
        try:
            # 0. can EVERY rank load librccl and resolve its entry points?  Agreed before anybody
            #    enters ncclCommInitRank: that call blocks until all ranks have made it, so a
            #    rank that fails earlier would leave the others inside it
            have = 1.0
            if lib.nh_comm_available() != 0:
                have, err = 0.0, lib.nh_last_error().decode()
            if self.group.reduce_scalar(have, "min") != 1.0:
                raise RcclUnavailable("librccl cannot be loaded on some rank %s" % err)
            # 1. rank 0 makes the id; ALWAYS broadcasts: the id, or an empty message that says
            #    "no id"
            if self.rank == 0:
                try:
                    _lib._chk(lib.nh_comm_unique_id(buf))
                    msg = buf.raw
                except Exception as e:  # ncclGetUniqueId refused
                    msg, err = b"", str(e)
            else:
                msg = b""
            uid = self.group.bcast(msg)
            if len(uid) != 128:
                raise RcclUnavailable("rank 0 could not create an RCCL unique id %s" % err)
            # 2. the communicator; whoever fails says so before anybody uses it
            ok = 1.0
            try:
                _lib._chk(lib.nh_comm_init(self.ctx.h, self.rank, self.size, uid))
            except Exception as e:
                ok, err = 0.0, str(e)
            if self.group.reduce_scalar(ok, "min") != 1.0:
                raise RcclUnavailable("ncclCommInitRank failed on some rank %s" % err)
        except (socket.timeout, ConnectionError, OSError) as e:
            # a peer died or never answered: the same fallback as "no RCCL" (the host-staged
            # path will say so itself if the control plane is really gone)
            raise RcclUnavailable("control plane failed while building the communicator: %r" % (e,))
        self._send = self._recv = None
        self._graph_ok = None

    def graph_capture_ok(self, timeout=120.0):
        """True when an all-gather of this communicator's ranks has been seen to survive
        hipGraph capture and replay.  Probed once, in a throw-away process per rank with
        its own rendezvous (``python -m naima_amd._rccl_probe``): a collective that hangs
        under capture then costs the probe its timeout, not the run.  Every rank gets the
        same answer (minimum over ranks).  (from_env has normally run the probe already,
        ahead of the communicator itself.)"""
        if self._graph_ok is None:
            self._graph_ok = run_rccl_probe(self.group, self.ctx.device, timeout)["graph"]
        return self._graph_ok

    def live_info(self):
        """what the LIVE communicator says of itself: ncclCommCount / UserRank / CuDevice"""
        import ctypes as C

        from . import _lib
        n, r, d = C.c_int(), C.c_int(), C.c_int()
        _lib._chk(_lib.load().nh_comm_info(self.ctx.h, C.byref(n), C.byref(r), C.byref(d)))
        return dict(nranks=n.value, rank=r.value, device=d.value)

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
        self.group.barrier()

    def max(self, v):
        return self.group.reduce_scalar(v, "max")


class RcclUnavailable(RuntimeError):
    pass


def run_rccl_probe(group, device, timeout=None):
    """Collective over ``group``: every rank starts ``python -m naima_amd._rccl_probe`` -- a
    throw-away process with its own rendezvous that builds an RCCL communicator of the same ranks
    on the same devices, all-gathers eagerly, then under hipGraph capture and replay -- and the
    ranks agree on what it saw: {"init": communicator + eager all-gather fine on EVERY rank,
    "graph": ... and inside a graph, "why": the first complaint}.  ncclCommInitRank blocks until
    every rank has called it and cannot be interrupted: a set of ranks on which it hangs (or
    fails, or an all-gather hangs under capture) costs the probe its timeout, not the run."""
    import subprocess
    import sys
    if timeout is None:
        timeout = float(os.environ.get("NAIMA_AMD_RCCL_PROBE_TIMEOUT", "120"))
    eager = graph = False
    why, proc = "", None
    port = group.free_port()  # the probes' own rendezvous (agreed by all ranks)
    try:
        env = dict(os.environ)
        env["MASTER_PORT"] = str(port)
        env["NAIMA_AMD_DEVICE"] = str(device)
        env["RANK"], env["WORLD_SIZE"] = str(group.rank), str(group.size)
        env["NAIMA_AMD_GROUP_SALT"] = "rccl-probe"
        env.pop("NAIMA_AMD_FORCE_SHARDED", None)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        proc = subprocess.Popen([sys.executable, "-m", "naima_amd._rccl_probe"], env=env,
                                cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        out, err = proc.communicate(timeout=timeout)
        eager = b"eager ok: True" in out
        graph = proc.returncode == 0 and b"captured + replayed ok: True" in out
        if not eager:
            tail = [ln for ln in err.decode(errors="replace").splitlines() if ln.strip()]
            why = "rank %d's probe: %s" % (group.rank, tail[-1][:300] if tail else
                                           "exit code %s" % proc.returncode)
    except subprocess.TimeoutExpired:
        why = "rank %d's probe did not finish within %.0f s (ncclCommInitRank or an all-gather hangs)" \
              % (group.rank, timeout)
    except Exception as e:
        why = "rank %d's probe could not run: %r" % (group.rank, e)
    finally:
        if proc is not None and proc.poll() is None:
            proc.kill()  # this probe only: the PID we started
            proc.wait()
    parts = group.allgather_bytes(struct.pack("<??", eager, graph) + why.encode())
    flags = [struct.unpack("<??", p_[:2]) for p_ in parts]
    whys = [p_[2:].decode(errors="replace") for p_ in parts if len(p_) > 2]
    return dict(init=all(f[0] for f in flags), graph=all(f[1] for f in flags),
                why=whys[0] if whys else "")


class _Watchdog:
    """``with _Watchdog(what, seconds):`` -- a step that can block for ever inside a library call
    (ncclCommInitRank waits for every rank; a rank that never comes leaves the others inside it,
    where no Python exception can reach them).  When the time is up the process says which rank
    was stuck in what and exits with code 17: its launcher (bench.py's own, torch.distributed.run)
    then ends the other ranks.  The driver's limit for a bench line is 1 800 s; this is 120."""

    def __init__(self, what, seconds):
        import threading
        self.what, self.seconds = what, float(seconds)
        self._done = threading.Event()
        self._t = threading.Thread(target=self._watch, daemon=True)

    def _watch(self):
        if not self._done.wait(self.seconds):
            import sys
            sys.stderr.write("naima_amd: rank %s stuck in %s for %.0f s -- giving up (exit 17)\n"
                             % (os.environ.get("RANK", "0"), self.what, self.seconds))
            sys.stderr.flush()
            os._exit(17)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._done.set()
        return False


def from_env(prefer="rccl"):
    """LocalComm for a single process, else RCCL (GPU) or the host-staged control plane
    (``prefer="host"``: CPU tests).  The returned communicator says how it came about:
    ``comm.ladder`` = the rungs tried, in order, each {"rung", "taken", "why"}."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and \
            os.environ.get("NAIMA_AMD_FORCE_SHARDED", "0") != "1":
        return LocalComm()
    if prefer != "rccl":
        c = HostComm()
        c.ladder = [dict(rung="RCCL all-gather", taken=False,
                         why="NAIMA_AMD_COMM=%s asks for the host-staged control plane" % prefer),
                    dict(rung="host-staged all-gather", taken=True, why="")]
        return c
    import warnings
    group = default_group()
    limit = float(os.environ.get("NAIMA_AMD_COMM_TIMEOUT", "120"))
    why, graph = None, None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("NAIMA_AMD_RCCL_PROBE", "1") != "0":
        # the communicator is built in a throw-away process FIRST (collective: every rank the same)
        from . import _lib
        pr = run_rccl_probe(group, _lib.default_device())
        graph = pr["graph"]
        if not pr["init"]:
            why = "the probe processes could not build an RCCL communicator of these ranks (%s)" % pr["why"]
    if why is None:
        try:
            with _Watchdog("RCCL communicator creation (ncclCommInitRank)", limit):
                c = RcclComm()  # collective: succeeds or raises on every rank alike
            c._graph_ok = graph
            c.ladder = [dict(rung="RCCL all-gather", taken=True, why="")]
            return c
        except RcclUnavailable as e:
            why = str(e)
    warnings.warn("RCCL communicator unavailable (%s); all-gathers are staged through the "
                  "host over the control plane: correct, slower" % (why,))
    c = HostComm()
    c.ladder = [dict(rung="RCCL all-gather", taken=False, why=why),
                dict(rung="host-staged all-gather", taken=True, why="")]
    return c
