"""ctypes binding of libnaima_hip.so (include/naima_hip.h) + device-array plumbing.

There is NO CPU fallback: if the shared library or a GPU is missing, the first
compute call raises.  Importing this module does not load the library, so the
host-side pieces (units, data ingest, sampler bookkeeping) import on any box.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnaima_hip.so")
if os.environ.get("NAIMA_AMD_LIB"):  # (experiments: a variant of the library built with other flags)
    LIB_PATH = os.path.abspath(os.environ["NAIMA_AMD_LIB"])

NH_PD_NPAR = 8
NH_K_NAMES = ("particle_weights", "integrate_tables", "synchrotron", "tables", "lnprob",
              "ic_seed_walkers", "glue", "integrate_rows", "half_step")
PD_KIND = {"PowerLaw": 0, "ExponentialCutoffPowerLaw": 1, "BrokenPowerLaw": 2,
           "ExponentialCutoffBrokenPowerLaw": 3, "LogParabola": 4}
PP_MODEL = {"Geant4": 0, "Pythia8": 1, "SIBYLL": 2, "QGSJET": 3}

_dp = C.c_void_p
_ITEMSIZE = {np.float64: 8, np.int64: 8, np.int32: 4, np.float32: 4, np.uint8: 1}
_i = C.c_int
_d = C.c_double
_ll = C.c_longlong

# name -> argtypes (restype is always int unless noted); mirrors include/naima_hip.h
_SIGS = {
    "nh_create": [_i, C.POINTER(_dp)],
    "nh_destroy": [_dp],
    "nh_device_count": [C.POINTER(_i)],
    "nh_device_info": [_dp, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_d), C.POINTER(_i)],
    "nh_device_pci_bus_id": [_dp, C.c_char_p, _i],
    "nh_alloc": [_dp, _ll, C.POINTER(_dp)],
    "nh_free": [_dp, _dp],
    "nh_upload": [_dp, _dp, _dp, _ll],
    "nh_download": [_dp, _dp, _dp, _ll],
    "nh_upload_ahead": [_dp, _dp, _dp, _ll, _dp, _dp],
    "nh_stream_wait_marker": [_dp, _dp],
    "nh_memset": [_dp, _dp, _i, _ll],
    "nh_sync": [_dp],
    "nh_timer_start": [_dp],
    "nh_timer_stop": [_dp, C.POINTER(_d)],
    "nh_profile_enable": [_dp, _i],
    "nh_profile_read": [_dp, C.POINTER(_d), C.POINTER(_ll), _i],
    "nh_profile_calibrate": [_dp, _i, C.POINTER(_d)],
    "nh_trapz_loglog": [_dp, _dp, _dp, _i, _i, _dp],
    "nh_trapz_loglog_intervals": [_dp, _dp, _dp, _i, _i, _dp],
    "nh_particle_weights": [_dp, _i, _dp, _i, _dp, _dp, _i, _d, _dp, _dp, _dp],
    "nh_particle_weights_multi": [_dp, _i, _dp, _i, _dp, _i],
    "nh_grid_logratio": [_dp, _dp, _i, _dp],
    "nh_integrate_tables": [_dp, _dp, _dp, _i, _i, _dp, _dp, _dp, _i, _dp, _dp, _i, _i, _i],
    "nh_synchrotron": [_dp, _dp, _dp, _dp, _i, _i, _dp, _dp, _i, _dp, _i, _dp, _i],
    "nh_table_ic_planck": [_dp, _dp, _i, _dp, _i, _d, _d, _dp, _dp, _i],
    "nh_table_ic_seed": [_dp, _dp, _i, _dp, _i, _dp, _dp, _i, _dp, _dp, _i],
    "nh_ic_seed_walkers": [_dp, _dp, _dp, _i, _dp, _dp, _i, _dp, _i, _dp, _dp, _i, _dp, _i],
    "nh_ssc_table": [_dp, _dp, _i, _dp, _i, _dp, _i, _dp],
    "nh_ic_seed_walkers_tab": [_dp, _dp, _dp, _i, _dp, _dp, _i, _dp, _i, _dp, _dp, _i, _dp, _dp,
                               _i],
    "nh_table_brems": [_dp, _dp, _i, _dp, _i, _dp, _dp, _dp, _dp, _i],
    "nh_table_pion_analytic": [_dp, _dp, _i, _dp, _i, _i, _i, _dp, _dp, _i],
    "nh_table_pion_lut": [_dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _dp, _dp, _i],
    "nh_lnprobmodel": [_dp, C.POINTER(_dp), C.POINTER(_d), _i, _i, _i, _i, _dp, _dp, _dp, _dp,
                       _dp, _dp, _dp, _dp],
    "nh_host_alloc": [_dp, _ll, C.POINTER(_dp)],
    "nh_host_free": [_dp, _dp],
    "nh_marker_create": [_dp, C.POINTER(_dp)],
    "nh_marker_record": [_dp, _dp],
    "nh_marker_wait": [_dp, _dp],
    "nh_marker_destroy": [_dp, _dp],
    "nh_pack_rows": [_dp, _dp, _i, _i, _dp, _i],
    "nh_ew_binary": [_dp, _i, _dp, _dp, _i, _dp],
    "nh_lincomb": [_dp, _dp, _i, _dp, _dp, _i, _i, _dp, _i],
    "nh_priors": [_dp, _dp, _i, _i, _dp],
    "nh_integrate_tables_nsplit": [_i, _i, _i],
    "nh_lnprob": [_dp, _dp, _i, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _i, _dp, _dp],
    "nh_move_propose": [_dp, _dp, _dp, _dp, _i, _i, _i, _i, _dp, _dp],
    "nh_move_accept": [_dp, _dp, _dp, _dp, _dp, _dp, _i, _i, _dp, _dp, _dp, _i],
    "nh_move_accept_rows": [_dp, _dp, _dp, _dp, _dp, _dp, _i, _i, _i, _dp, _dp, _dp, _i, _i, _dp, _dp],
    "nh_step_front": [_dp, _dp, _dp, _dp, _dp, _dp, _i, _i, _i, _i, _dp, _dp, _dp, _i, _i, _dp, _dp,
                      _i, _dp, _i, _dp],
    "nh_pion_kelner06": [_dp, _i, _dp, _i, _dp, _i, _d, _i, _dp, _i, _dp, _dp],
    "nh_synchrotron_lnprob": [_dp, _dp, _dp, _dp, _i, _i, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i,
                              _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _i, _dp, _dp],
    "nh_integrate_tables_lnprob": [_dp, _dp, _dp, _i, _i, _dp, _dp, _dp, _i, _dp, _dp, _i, _i,
                                   _dp, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _i, _dp,
                                   _dp],
    "nh_lnprob_accept": [_dp, _dp, _i, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _i, _dp, _dp,
                         _dp],
    "nh_scatter_rows": [_dp, _dp, _i, _dp, _i, _dp, _dp, _i, _i, _i],
    "nh_copy": [_dp, _dp, _dp, _ll],
    "nh_stream_fork": [_dp, _i],
    "nh_stream_fork_at": [_dp, _i, _dp],
    "nh_stream_switch": [_dp, _i],
    "nh_stream_wait": [_dp, _i, _i],
    "nh_stream_join": [_dp],
    "nh_moves_create": [C.c_ulonglong, _i, _d, _i, _i, _i, C.POINTER(_dp)],
    "nh_moves_take": [_dp, _i, C.POINTER(_dp), C.POINTER(_i)],
    "nh_moves_destroy": [_dp],
    "nh_graph_begin": [_dp],
    "nh_graph_end": [_dp, C.POINTER(_dp)],
    "nh_graph_launch": [_dp, _dp],
    "nh_graph_destroy": [_dp, _dp],
    "nh_comm_available": [],
    "nh_comm_unique_id": [C.c_char_p],
    "nh_comm_init": [_dp, _i, _i, C.c_char_p],
    "nh_comm_destroy": [_dp],
    "nh_comm_info": [_dp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "nh_comm_allgather": [_dp, _dp, _dp, _ll],
    "nh_half_step_create": [_dp, _dp, C.POINTER(_dp)],
    "nh_half_step_begin_block": [_dp, _dp, _i, _i],
    "nh_half_step_launch": [_dp, _dp, _i],
    "nh_half_step_info": [_dp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_ll)],
    "nh_half_step_split": [_dp, C.POINTER(_i)],
    "nh_half_step_syn_form": [_dp, C.POINTER(_i)],
    "nh_half_step_destroy": [_dp, _dp],
    "nh_nan_count": [_dp, _i, C.POINTER(_i)],
    "nh_clock_read": [_dp, _i, C.POINTER(C.c_longlong)],
    "nh_half_step_span": [_dp, _i, _i],
    "nh_half_step_nan_count": [_dp, _dp, _i, C.POINTER(_i)],
    "nh_half_step_counts": [_dp, _dp, _i, C.POINTER(_i), C.POINTER(_i)],
    "nh_half_step_stamps": [_dp, _dp, _dp],
    "nh_half_step_run_create": [_dp, _dp, C.POINTER(_dp)],
    "nh_half_step_run": [_dp, _dp, _dp, _i, _i, _dp, _dp, _dp, _ll, _ll],
    "nh_half_step_run_status": [_dp, _dp, C.POINTER(_i)],
    "nh_half_step_run_report": [_dp, _dp, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                                C.POINTER(_i)],
    "nh_half_step_run_tables": [_dp, _dp, _dp, _dp, _i],
    "nh_half_step_run_create_shared": [_dp, _dp, _i, _i, C.POINTER(_dp)],
    "nh_half_step_run_export": [_dp, _dp, _dp],
    "nh_half_step_run_attach": [_dp, _dp, _i, _dp],
    "nh_half_step_run_probe": [_dp, _dp, _i, C.POINTER(_i), C.POINTER(C.c_double)],
    "nh_half_step_run_hist_flags": [_dp, _dp],
    "nh_half_step_run_counters": [_dp, _dp, _dp, _dp, _i],
    "nh_half_step_run_info": [_dp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_ll)],
    "nh_half_step_run_syn_info": [_dp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "nh_half_step_run_split_info": [_dp, C.POINTER(_i), C.POINTER(_i)],
    "nh_half_step_run_pipeline_info": [_dp, C.POINTER(_i)],
    "nh_half_step_run_table_info": [_dp, C.POINTER(_i), C.POINTER(_i)],
    "nh_half_step_run_stamps": [_dp, _dp, _dp],
    "nh_half_step_run_destroy": [_dp, _dp],
    "nh_general_electron": [_dp, _i, _dp, _i, _dp, _d, _dp, _d, _dp, _i, _dp, _dp, _dp, _i, _dp, _i,
                            _dp, _i, _i, _dp],
    "nh_general_electron_seed": [_dp, _i, _dp, _i, _dp, _d, _dp, _d, _dp, _dp, _dp, _i, _dp, _i, _dp,
                                 _i, _i, _dp],
    "nh_general_electron_seed_rows": [_dp, _i, _dp, _i, _dp, _d, _dp, _d, _dp, _dp, _dp, _ll, _i, _dp,
                                      _i, _dp, _i, _i, _dp],
    "nh_general_proton": [_dp, _i, _dp, _i, _dp, _d, _dp, _d, _dp, _i, _i, _i, _i, _dp, _i, _dp, _i,
                          _dp, _dp, _i, _dp, _i, _i, _dp],
    "nh_table_interleave": [_dp, _dp, _dp, _dp, _i, _i, _dp],
    "nh_hist_append": [_dp, _dp, _dp, _ll, _i, _dp, _ll],
    "nh_set_words": [_dp, _dp, _dp, _i],
    "nh_half_step_append_blobs": [_dp, _dp, _ll],
}
EXPORTS = tuple(_SIGS) + ("nh_last_error", "nh_version", "nh_ssc_table_bytes")

_lib = None


class NaimaHipError(RuntimeError):
    pass


def under_rocprofiler():
    """True when this process was started under rocprofv3 / a rocprofiler-sdk tool"""
    return bool(os.environ.get("ROCPROFILER_LIBRARY_CTOR") or os.environ.get("ROCP_TOOL_LIBRARIES")
                or "rocprofiler-sdk" in os.environ.get("LD_PRELOAD", ""))


def _profiler_workaround():
    """rocprofv3 (ROCm 7.2) + hipGraph replay: with rocprofiler-sdk's queue interception active,
    the HIP runtime's replay of a graph from PRE-BUILT AQL packets ("graph packet capture", its
    default) dies inside hipGraphLaunch after a few hundred replays of the step graphs -- a
    SIGSEGV in the runtime's packet copy, or "AQL packet is malformed" / a hung queue -- for
    graphs of sixteen kernel nodes with 3 KB by-value arguments.  The same graphs, binary and
    command run through under the profiler when the runtime dispatches the nodes one by one
    (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0), and always when no profiler is attached (round 3:
    profiles/README.md has the four runs).  Set here, before the HIP runtime reads its flags,
    and only when a profiler is attached; an explicit setting of the user wins."""
    if under_rocprofiler():
        os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    # device memory shared between processes (RCCL's buffers; the rings of an ensemble shared by
    # several GPUs, hipIpcGetMemHandle): this driver stack supports dmabuf handles only -- the
    # legacy mode fails with "hipIpcGetMemHandle: invalid argument".  Read when the HSA runtime
    # starts, i.e. before the library's first HIP call; an explicit setting of the user wins.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def sorted_columns(K):
    """column order and first rows for Context.sorted_tables: K [nG][nK] (host) -> (perm, row0)
    with perm = the columns ordered by the first row in which they are non-zero (stable: equal
    columns keep their order; a column of zeros counts as nG) and row0[t] = the smallest such
    row among the 64 columns perm[64 t : 64 t + 64] -- the segments [0, row0[t]) of that tile
    have a zero node at both ends or at one (trapz_loglog: exactly 0, utils.py:347-348)"""
    K = np.asarray(K)
    nG, nK = K.shape
    live = K != 0.0
    first = np.where(live.any(axis=0), live.argmax(axis=0), nG)
    perm = np.argsort(first, kind="stable")
    row0 = [int(first[perm][q:q + 64].min()) for q in range(0, nK, 64)]
    return perm, row0


def load():
    """dlopen libnaima_hip.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or naima_amd/csrc/build.sh.  naima_amd has no CPU fallback." % LIB_PATH)
    _profiler_workaround()
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _i
    lib.nh_last_error.restype = C.c_char_p
    lib.nh_last_error.argtypes = []
    lib.nh_version.restype = _i
    lib.nh_ssc_table_bytes.restype = _ll
    lib.nh_ssc_table_bytes.argtypes = [_i, _i, _i]
    _lib = lib
    return lib


def _chk(rc):
    if rc != 0:
        raise NaimaHipError("libnaima_hip error %d: %s" % (rc, _lib.nh_last_error().decode()))


_POISON = int(os.environ.get("NAIMA_AMD_POISON", "0"), 0)


class Moves:
    """the stretch-move random stream (nh_moves_*): a C++ worker thread draws ahead.
    ``take(k)`` -> (address, got, S, P, Z, L views for `got` consecutive steps)"""

    def __init__(self, seed, N, a=2.0, ksteps=32, depth=4, pinned=False):
        load()
        h = _dp()
        _chk(_lib.nh_moves_create(int(seed) & (2 ** 64 - 1), int(N), float(a), int(ksteps),
                                  int(depth), int(bool(pinned)), C.byref(h)))
        self.h, self.N, self.ns, self.ksteps = h, int(N), int(N) // 2, int(ksteps)

    def take(self, want):
        p, got = _dp(), _i()
        _chk(_lib.nh_moves_take(self.h, int(want), C.byref(p), C.byref(got)))
        return p.value, got.value

    def view(self, addr, got):
        """numpy views (no copy) of `got` steps at `addr`: S, P int32 and Z, L float64,
        each [got, 2, ns]"""
        ns = self.ns
        n = got * 2 * 3 * ns
        f = np.frombuffer((C.c_double * n).from_address(addr), dtype=np.float64).reshape(
            got, 2, 3 * ns)
        iv = f[:, :, 2 * ns:].view(np.int32)
        return iv[:, :, :ns], iv[:, :, ns:], f[:, :, :ns], f[:, :, ns:2 * ns]

    def close(self):
        if self.h:
            _lib.nh_moves_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceArray:
    """A float64 (or int32) array in HBM owned by a Context's pool."""
    __slots__ = ("ctx", "ptr", "shape", "dtype", "nbytes", "_cap", "stream", "anchor",
                 "graph_owned", "pending", "__weakref__")

    def __init__(self, ctx, ptr, shape, dtype, cap, nbytes=None):
        self.ctx, self.ptr, self.shape, self.dtype, self._cap = ctx, ptr, tuple(shape), dtype, cap
        self.stream = ctx.cur_stream  # the stream whose work produces this buffer
        self.anchor = None
        # allocated while a hipGraph was being captured: the graph keeps using the address
        # on every replay, so the buffer never returns to the general pool
        self.graph_owned = bool(ctx.capturing)
        # a deferred launch that will fill this buffer (see Context.defer)
        self.pending = None
        self.nbytes = nbytes if nbytes is not None else \
            math.prod(int(x) for x in shape) * _ITEMSIZE.get(dtype, 0) or \
            int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def get(self):
        if self.ctx.capturing:
            raise RuntimeError("device->host download while a hipGraph is being captured")
        self.ctx.join()
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            _chk(_lib.nh_download(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def set(self, host):
        if self.ctx.capturing:
            raise RuntimeError("host->device upload while a hipGraph is being captured: the "
                               "value must come from a device-resident input")
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes == self.nbytes, (host.shape, self.shape)
        if host.nbytes:
            _chk(_lib.nh_upload(self.ctx.h, self.ptr, host.ctypes.data, host.nbytes))
        return self

    def __del__(self):
        try:
            if self._cap > 0 and self.ctx is not None and self.ctx.h:
                self.ctx._release(self.ptr, self._cap, self.graph_owned)
        except Exception:
            pass


class Context:
    """One HIP device + stream; owns a size-bucketed buffer pool (hipMalloc is far
    too slow for per-step scratch) and a content-addressed cache of small
    constant arrays (grids, photon energies, data columns) resident in HBM."""

    def __init__(self, device=0):
        load()
        h = _dp()
        _chk(_lib.nh_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self._pool = {}
        self._const = {}
        self._keep = {}
        self._lx = {}
        self._lne = {}
        self._plan = None
        self._in_eval = False
        self._accept_hook = None
        self._cap_pool = {}
        self._retained = []
        self._release_later = []  # device loops collected during a stream capture
        self._deferred = []
        self._pinned = set()
        self._pinned_ptrs = set()
        self._anchors = []
        self._nanchor = 0
        self.side_small = os.environ.get("NAIMA_AMD_SIDE_SMALL", "0") != "0"
        self._tables = {}
        self._big_tables = {}
        self.capturing = False
        # side streams (nh_stream_fork/join): -1 = main
        self.cur_stream = -1
        self.multistream = os.environ.get("NAIMA_AMD_MULTISTREAM", "0") != "0"
        self._next_side = 0
        self._forked = False
        self._limbo = []
        self._waited = set()
        # particle grids recently asked of each distribution kind: a fresh distribution
        # evaluates all of them in one launch (nh_particle_weights_multi)
        self._wgrids = {}
        self._weval = 0

    # -- memory -------------------------------------------------------------
    @staticmethod
    def _bucket(nbytes):
        # (the next power of two, 256 at least; on the sampler's per-call path)
        return 256 if nbytes <= 256 else 1 << (int(nbytes) - 1).bit_length()

    def empty(self, shape, dtype=np.float64):
        # (on the sampler's per-call path: no numpy reductions for the product of three ints)
        shape = (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(int(s) for s in shape)
        isz = _ITEMSIZE.get(dtype)
        if isz is None:
            isz = np.dtype(dtype).itemsize
        nbytes = math.prod(shape) * isz
        cap = self._bucket(nbytes)
        # during a capture, buffers released earlier in the SAME capture may be reused
        # (graph order is stream order); nothing else may ever alias them
        free = self._cap_pool.get(cap) if self.capturing else None
        if not free:
            free = self._pool.get(cap)
        if free:
            ptr = free.pop()
        else:
            p = _dp()
            _chk(_lib.nh_alloc(self.h, cap, C.byref(p)))
            ptr = p.value
        if _POISON and not self.capturing:
            # NAIMA_AMD_POISON=<byte>: every buffer handed out is filled with that byte first (a
            # debugging aid: a launch that reads what nobody wrote shows up as a changed result)
            _chk(_lib.nh_memset(self.h, ptr, _POISON & 0xFF, cap))
        return DeviceArray(self, ptr, shape, dtype, cap, nbytes)

    def _release(self, ptr, cap, graph_owned=False):
        if graph_owned:
            if self.capturing:
                self._cap_pool.setdefault(cap, []).append(ptr)
            else:
                self._retained.append((ptr, cap))  # a captured graph still writes here
        elif self._forked:
            # side streams are in flight: the buffer may still be read or written by a
            # stream other than the one that will reuse it -> park it until the join
            self._limbo.append((ptr, cap))
        else:
            self._pool.setdefault(cap, []).append(ptr)

    # -- deferred launches --------------------------------------------------------------
    # Inside the device step loop the producer of the LAST spectrum the likelihood needs
    # (the synchrotron kernel) may be held back so that the likelihood can ride on its
    # launch (nh_synchrotron_lnprob).  Any other reader of the buffer flushes it first.
    def defer(self, arr, name, args, keep=()):
        arr.pending = (name, args, keep)  # keep: buffers the arguments point into
        self._deferred.append(arr)

    def flush(self, *arrs):
        """launch what is still deferred for these buffers (all of them when none given)"""
        todo = arrs if arrs else tuple(self._deferred)
        for a in todo:
            p = getattr(a, "pending", None)
            if p is not None:
                a.pending = None
                self.call(p[0], *p[1])
        self._deferred = [a for a in self._deferred if a.pending is not None]

    # -- launches that the device step loop folds into nh_step_front ------------------
    # The loop first RECORDS what a model evaluation asks for (parameter packs, the
    # particle-weights launch, single-row reductions such as We; persistent output
    # buffers), then has nh_step_front produce all of it right after the proposal; in
    # REPLAY mode a request only checks that it is the recorded one and returns its
    # buffers.
    def plan_begin(self):
        self._plan = dict(mode="record", packs=[], weights=[], moments=[], i=[0, 0, 0, 0, 0],
                          emit=[], calls=[], mega=False, hs=None, bufs=[])
        return self._plan

    def _replayed(self, kind, slot, key):
        plan = self._plan
        if plan is None or plan["mode"] != "replay":
            return None
        i = plan["i"][slot]
        if i >= len(plan[kind]) or plan[kind][i][0] != key:
            raise NaimaHipError("the model's launch sequence changed between evaluations (%s); "
                                "run the sampler with use_graph=False" % kind)
        plan["i"][slot] = i + 1
        return plan[kind][i][1]

    def _recorded(self, kind, key, value):
        if self._plan is not None and self._plan["mode"] == "record":
            self._plan[kind].append((key, value))
        return value

    def plan_buffer(self, key, shape):
        """an output buffer of a launch that is NOT one of the step loop's recorded kinds (the SSC
        seed integral) but whose result the likelihood reads: the same buffer at every
        evaluation of a recorded plan, so that the plan's component pointers stay what they
        were; a fresh one outside a plan"""
        plan = self._plan
        if plan is None:
            return self.empty(shape)
        if plan["mode"] == "record":
            buf = self.empty(shape)
            plan["bufs"].append((key, buf))
            return buf
        i = plan["i"][4]
        if i >= len(plan["bufs"]) or plan["bufs"][i][0] != key:
            raise NaimaHipError("the model's launch sequence changed between evaluations (%s); "
                                "run the sampler with use_graph=False" % (key[0],))
        plan["i"][4] = i + 1
        return plan["bufs"][i][1]

    def pack_rows(self, cols, ncols, N):
        """out[N][ncols] from lazy columns (one nh_pack_rows launch unless replayed)"""
        import ctypes as C
        key = (bytes(C.string_at(C.addressof(cols), C.sizeof(cols))), ncols, N)
        hit = self._replayed("packs", 0, key)
        if hit is not None:
            return hit
        out = self.empty((N, ncols))
        self.call("nh_pack_rows", cols, ncols, N, out, ncols)
        return self._recorded("packs", key, out)

    def weights_multi(self, kind, rows, N, grids):
        """particle weights of N walkers on several grids: ``grids`` is a list of
        (e_eV, xg, ln_e, lx, unit_scale, nG) with device arrays; returns [(w, dlw)]"""
        from .darray import nh_grid
        key = (kind, rows.ptr, N, tuple((g[0].ptr, g[1].ptr, g[2].ptr, g[3].ptr, g[4], g[5])
                                        for g in grids))
        hit = self._replayed("weights", 1, key)
        if hit is not None:
            return hit
        desc = (nh_grid * len(grids))()
        out = []
        for j, (ed, xd, lne, lx, scale, nG) in enumerate(grids):
            wk, lwk = self.empty((N, nG)), self.empty((N, nG))
            desc[j] = nh_grid(ed.ptr, xd.ptr, wk.ptr, lwk.ptr, scale, nG, 0, lne.ptr, lx.ptr)
            out.append((wk, lwk))
        self.call("nh_particle_weights_multi", kind, rows, N, desc, len(grids))
        return self._recorded("weights", key, out)

    def weights_replay(self, kind, rows, N):
        """REPLAY mode: the grids and buffers of the recorded weights launch (the grid
        bookkeeping of the radiative classes is shared state and is bypassed), else None"""
        plan = self._plan
        if plan is None or plan["mode"] != "replay":
            return None
        i = plan["i"][1]
        if i >= len(plan["weights"]) or plan["weights"][i][0][:3] != (kind, rows.ptr, N):
            raise NaimaHipError("the model's launch sequence changed between evaluations "
                                "(weights); run the sampler with use_graph=False")
        plan["i"][1] = i + 1
        return plan["weights"][i][0][3], plan["weights"][i][1]

    def moment(self, w, lw, N, nG, lx, Kt, dlnKt):
        """out[N] = trapz_loglog(n * K, x) for ONE table row K (We, Wp)"""
        key = (w.ptr, lw.ptr, N, nG, lx.ptr, Kt.ptr, dlnKt.ptr)
        hit = self._replayed("moments", 2, key)
        if hit is not None:
            return hit
        out = self.empty((N, 1))
        self.call("nh_integrate_tables", w, lw, N, nG, lx, Kt, dlnKt, 1, None, out, 1, 0, 1)
        return self._recorded("moments", key, out)

    # -- the general path (a particle grid per walker) -------------------------------------
    general_nmax = 2048  # grid nodes a workgroup's LDS is sized for (4 doubles each)

    def general_status(self):
        """device ints of the general kernel: [0] the largest node count asked for when a walker's
        grid exceeds general_nmax, [1] evaluations whose node count sat on an int() boundary"""
        if getattr(self, "_gen_status", None) is None:
            self._gen_status = self.array(np.zeros(2, dtype=np.int32), dtype=np.int32)
        return self._gen_status

    general_boundary_hits = 0  # evaluations of the general kernel that sat on an int() boundary

    def check_general(self):
        """raise if a general-path launch since the last check met a grid longer than
        general_nmax (those walkers were given NaN); warn if a walker's node count
        int(nEed * decades) was decided within rounding of an integer (the device's log10 and
        numpy's may then disagree on it in the last place)"""
        st = getattr(self, "_gen_status", None)
        if st is None or self.capturing:
            return
        n, amb = (int(v) for v in st.get())
        if n or amb:
            st.set(np.zeros(2, dtype=np.int32))
        if amb:
            import warnings
            self.general_boundary_hits += amb
            warnings.warn("%d evaluation(s) of a per-walker particle grid had nEed * decades within "
                          "1e-9 of an integer: int() of it (radiative.py:152-154) may differ between "
                          "the device's log10 and numpy's there" % amb, RuntimeWarning)
        if n > self.general_nmax:
            raise NaimaHipError("a walker's particle grid has %d nodes, more than the %d the "
                                "general kernel's LDS is sized for: raise Context.general_nmax "
                                "(at most ~4600) or lower nEed" % (n, self.general_nmax))

    # -- emission launches of a model evaluation ------------------------------------------
    # In the one-launch-per-half-step mode of the device loop (plan["mega"]) these are not
    # launched at all: nh_half_step produces every spectrum of the model, into buffers that
    # belong to the plan.  While a half-step is RECORDED they run as usual and leave their
    # arguments in plan["emit"]; outside the step loop they are plain launches.
    def _emit_replayed(self, kind, key, shape):
        plan = self._plan
        i = plan["i"][3]
        if i >= len(plan["emit"]) or plan["emit"][i]["key"] != key:
            raise NaimaHipError("the model's launch sequence changed between evaluations (%s); "
                                "run the sampler with use_graph=False" % kind)
        plan["i"][3] = i + 1
        ent = plan["emit"][i]
        if ent["out"] is None:
            ent["out"] = self.empty(shape)
        return ent["out"]

    def emit_tables(self, w, lw, N, nG, lx, Kt, dlnKt, nK, scale, nonneg, may_split=True):
        """out[N][nK] = scale[k] * trapz_loglog(n K_k, x) for every walker (nh_integrate_tables).
        Returns (out, nsplit): out has nsplit planes of N rows that the consumer adds."""
        plan = self._plan
        key = ("tab", w.ptr, lw.ptr, N, nG, lx.ptr, Kt.ptr, dlnKt.ptr, nK,
               scale.ptr if scale is not None else 0, int(nonneg))
        if plan is not None and plan["mega"] and plan["mode"] == "replay":
            return self._emit_replayed("tables", key, (N, nK)), 1
        ns = _lib.nh_integrate_tables_nsplit(N, nG, nK) if may_split else 1
        out = self.empty((ns * N, nK))
        args = (w, lw, N, nG, lx, Kt, dlnKt, nK, scale, out, nK, int(nonneg), ns)
        if plan is not None and plan["mode"] == "record":
            plan["emit"].append(dict(kind="tab", key=key, out=None, N=N,
                                     keep=(w, lw, lx, Kt, dlnKt, scale)))
        hook = self._accept_hook
        if hook is not None and not hook["used"] and hook["N"] == N and nK <= 64 and ns == 1 \
                and not self._deferred:
            # device step loop: hold the launch back, the likelihood may ride on it
            self.defer(out, "nh_integrate_tables", args)
        else:
            self.call("nh_integrate_tables", *args)
        return out, ns

    def emit_synchrotron(self, w, lw, Bp, ldB, N, gd, lx, nG, Ed, nE, keep=(), E_host=None):
        """out[N][nE] = Synchrotron._spectrum of every walker (nh_synchrotron)"""
        plan = self._plan
        key = ("syn", w.ptr, lw.ptr, int(Bp), int(ldB), N, gd.ptr, lx.ptr, nG, Ed.ptr, nE)
        if plan is not None and plan["mega"] and plan["mode"] == "replay":
            if plan.get("staged"):
                return self._stage_a(plan, key, N, nE)
            return self._emit_replayed("synchrotron", key, (N, nE))
        out = self.empty((N, nE))
        args = (w, lw, Bp, ldB, N, gd, lx, nG, Ed, nE, out, nE)
        if plan is not None and plan["mode"] == "record":
            plan["emit"].append(dict(kind="syn", key=key, out=None, N=N, keep=(w, lw, gd, lx, Ed) + tuple(keep),
                                     E_host=None if E_host is None else np.array(E_host, dtype=float)))
        hook = self._accept_hook
        if hook is not None and not hook["used"] and hook["N"] == N and nE <= 64 \
                and not self._deferred:
            self.defer(out, "nh_synchrotron", args, keep=keep)
        else:
            self.call("nh_synchrotron", *args)
        return out

    def _hs_front(self, d, f):
        """the part of an nh_hs_desc every plan of a device loop shares: the ensemble, the block of
        moves, the parameter packs, the grids (returns {weights pointer: grid index})"""
        for name in ("coords", "logp", "blk", "cursor", "qT", "factors"):
            setattr(d, name, f[name])
        d.ns, d.ndim, d.lo, d.nloc = f["ns"], f["ndim"], f["lo"], f["nloc"]
        pk, npk, kind, rows_ptr, gd, ngr, mm, nmm = f["front_args"]
        for q in range(npk):
            d.packs[q] = pk[q]
        d.npacks, d.kind, d.params = npk, kind, rows_ptr
        wgrid = {}
        for g in range(ngr):
            d.grids[g] = gd[g]
            wgrid[gd[g].w] = g
        d.ngrids = ngr
        return wgrid

    def _stage_a(self, plan, key, N, nE):
        """A model that asks for its synchrotron spectrum TWICE -- at the energies of a seed photon
        field it then builds from it, and at the data's (examples/CrabNebula_SynSSC.py:29-45) --
        with launches of other kernels in between (the SSC seed integral batches sixteen WALKERS
        per wave: nothing a one-workgroup-per-walker launch can absorb): the half-step is two
        nh_half_step launches around them.  Stage A, launched where the model asks for the first
        spectrum: proposal -> packs -> weights (written to HBM for the kernels in between) ->
        ONE synchrotron component over both sets of energies, no accept.  Stage C is the plan's
        own launch (Context.half_step): proposal, packs and weights again (a few microseconds),
        the table reductions, the spectra of the launches in between and stage A's from HBM,
        likelihood, accept."""
        import ctypes as C

        from . import darray as D
        i = plan["i"][3]
        ent = plan["emit"][i] if i < len(plan["emit"]) else None
        if ent is None or ent["key"] != key:
            raise NaimaHipError("the model's launch sequence changed between evaluations "
                                "(synchrotron); run the sampler with use_graph=False")
        plan["i"][3] = i + 1
        syn = [e for e in plan["emit"] if e["kind"] == "syn"]
        st = plan.get("stage")
        if st is None:
            e1, e2 = syn
            n1, n2 = e1["key"][10], e2["key"][10]
            base = self.empty((N * (n1 + n2),))
            e1["out"] = DeviceArray(self, base.ptr, (N, n1), np.float64, 0)
            e2["out"] = DeviceArray(self, base.ptr + 8 * N * n1, (N, n2), np.float64, 0)
            Ecat = self.array(np.concatenate([e1["E_host"], e2["E_host"]]))
            f = plan["front"]
            d = D.nh_hs_desc()
            wgrid = self._hs_front(d, f)
            rows_ptr = f["front_args"][3]
            d.hist = None
            d.do_accept, d.write_weights = 0, 1
            d.nmoms, d.ntab = 0, 0
            _, w, lw, Bp, ldB, _, gdp, lx, nG, _, _ = e1["key"]
            in_rows = ldB == NH_PD_NPAR and 0 <= Bp - rows_ptr < 8 * NH_PD_NPAR
            d.syn = D.nh_hs_syn(wgrid[w], n1 + n2, n1, (Bp - rows_ptr) // 8 if in_rows else -1, ldB,
                                n1, Ecat.ptr, None if in_rows else Bp, base.ptr,
                                base.ptr + 8 * N * n1, n2, 0)
            # (a launch has a likelihood: this one's is of the first spectrum against columns of
            # ones and zeros, into a buffer nobody reads)
            ones, zeros = self.array(np.ones(n1)), self.array(np.zeros(n1))
            izero = self.array(np.zeros(n1, dtype=np.int32), dtype=np.int32)
            half = self.array(np.full(n1, 0.5))
            dummy = self.empty((N,))
            d.comps[0] = D.nh_comp(base.ptr, n1, 1.0)
            d.ncomp, d.nE = 1, n1
            d.conv, d.flux, d.err_lo, d.err_hi = ones.ptr, zeros.ptr, ones.ptr, ones.ptr
            d.ul, d.cl, d.lp, d.nterms = izero.ptr, half.ptr, None, 0
            # the prior of the recorded evaluation: a proposal it forbids is integrated by nobody
            # (its synchrotron spectrum is written as zeros, the seed field made of it is empty and
            # the SSC kernel packs such walkers out of its groups: k_ssc_order) -- as the plan's
            # own launch does for it
            pt = plan.get("prior_terms")
            if pt is not None:
                for q in range(pt[1]):
                    d.terms[q] = pt[0][q]
                d.nterms = pt[1]
            d.model_out, d.total, d.nblobs, d.send_width = None, dummy.ptr, 0, 0
            h = _dp()
            _chk(_lib.nh_half_step_create(self.h, C.addressof(d), C.byref(h)))
            st = plan["stage"] = dict(plan=h, keep=(base, Ecat, ones, zeros, izero, half, dummy))
            # the span clock: this launch opens the half-step's span, the plan's own closes it
            _chk(_lib.nh_half_step_span(h, 1, 0))
            self.call("nh_half_step_begin_block", h, f["pos"]["slice"], 0)
        if ent is syn[0]:
            pos = plan["front"]["pos"]
            self.call("nh_half_step_launch", st["plan"], pos["slice"] if pos["bake"] else -1)
        return ent["out"]

    def half_step(self, hook, comps, ncomp, nE, conv, dd, lpd, terms, nterms, total, blobs=()):
        """the plan's nh_half_step launch: everything the recorded model evaluation asked
        for plus the likelihood of ``comps`` (created on first use, then checked and reused)"""
        import ctypes as C

        from . import darray as D
        plan = self._plan
        if plan["i"][3] != len(plan["emit"]):
            raise NaimaHipError("the model's launch sequence changed between evaluations "
                                "(fewer emission components); run with use_graph=False")
        key = (bytes(C.string_at(C.addressof(comps), C.sizeof(comps))), ncomp, nE, conv.ptr,
               lpd.ptr if lpd is not None else 0,
               bytes(C.string_at(C.addressof(terms), C.sizeof(terms))) if nterms else b"",
               total.ptr)
        hs = plan["hs"]
        if hs is not None:
            if hs["key"] != key:
                raise NaimaHipError("the model's likelihood inputs changed between evaluations; "
                                    "run the sampler with use_graph=False")
            self.call("nh_half_step_launch", hs["plan"],
                      plan["front"]["pos"]["slice"] if plan["front"]["pos"]["bake"] else -1)
            return
        f = plan["front"]  # filled in by the device loop when it chose this mode
        d = D.nh_hs_desc()
        for name in ("coords", "logp", "blk", "cursor", "qT", "factors", "hist",
                     "accepted", "naccepted", "sel"):
            setattr(d, name, f[name])
        d.ns, d.ndim, d.lo, d.nloc = f["ns"], f["ndim"], f["lo"], f["nloc"]
        d.do_accept, d.write_weights = int(hook["mv"] is not None), 0
        pk, npk, kind, rows_ptr, gd, ngr, mm, nmm = f["front_args"]
        for q in range(npk):
            d.packs[q] = pk[q]
        d.npacks, d.kind, d.params = npk, kind, rows_ptr
        wgrid = {}
        for g in range(ngr):
            d.grids[g] = gd[g]
            wgrid[gd[g].w] = g
        d.ngrids = ngr
        for q in range(nmm):
            d.moms[q] = mm[q]
        d.nmoms = nmm
        d.syn.grid = -1
        nt = 0
        tabs = []  # (what the resident loop sorts the columns of: sorted_tables)
        for ent in plan["emit"]:
            k = ent["key"]
            if ent["kind"] == "tab":
                _, w, lw, N, nG, lx, Kt, dKt, nK, sc, nonneg = k

                def interleaved(Kt=Kt, dKt=dKt, nG=nG, nK=nK, lx=lx, nonneg=nonneg):
                    kd = self.empty((2 * nG * nK,))
                    # (a non-negative table carries its log-ratios in units of lx)
                    self.call("nh_table_interleave", Kt, dKt, lx if nonneg else None, nG, nK, kd)
                    return kd

                kd = self.table(("kd", Kt, dKt, nG * nK, bool(nonneg)), interleaved)
                self._pinned.add(("kd", Kt, dKt, nG * nK, bool(nonneg)))  # the plan points into it
                d.tab[nt] = D.nh_hs_table(wgrid[w], nK, nK, nonneg, kd.ptr, None, sc or None,
                                          ent["out"].ptr)
                tabs.append((Kt, dKt, nG, nK, lx, bool(nonneg)))
                nt += 1
            elif plan.get("staged"):
                pass  # (stage A's launch has produced it: the likelihood reads it from HBM)
            else:
                _, w, lw, Bp, ldB, N, gdp, lx, nG, Ed, nEs = k
                in_rows = ldB == NH_PD_NPAR and 0 <= Bp - rows_ptr < 8 * NH_PD_NPAR
                d.syn = D.nh_hs_syn(wgrid[w], nEs, nEs, (Bp - rows_ptr) // 8 if in_rows else -1,
                                    ldB, 0, Ed, None if in_rows else Bp, ent["out"].ptr, None, 0, 0)
        d.ntab = nt
        for q in range(ncomp):
            d.comps[q] = comps[q]
        d.ncomp, d.nE = ncomp, nE
        d.conv, d.flux, d.err_lo, d.err_hi = conv.ptr, dd.flux.ptr, dd.elo.ptr, dd.ehi.ptr
        d.ul, d.cl = dd.ul.ptr, dd.cl.ptr
        d.lp = lpd.ptr if lpd is not None else None
        for q in range(nterms):
            d.terms[q] = terms[q]
        d.nterms = nterms
        d.model_out, d.total = None, total.ptr
        # blobs the launch keeps itself (the device loop says where: hook["blobs"]); anything
        # it cannot express leaves them to the separate staging / scatter launches
        hook["blobs_in_kernel"] = False
        blobs = [b for b in blobs if not isinstance(b, (float, int))]  # (lnprob's constant NaN)
        dest = hook.get("blobs")
        if dest and (hook["mv"] is not None or hook.get("send_width")) and \
                len(dest) == len(blobs) <= 4:
            from . import units as u
            model_terms = [(int(comps[q].ptr), int(comps[q].ld), float(comps[q].scale))
                           for q in range(ncomp)]
            mouts = [mm[q].out for q in range(nmm)]
            ent = []
            for (cur, m, hist_word), b in zip(dest, blobs):
                v = b.value if isinstance(b, u.Quantity) else b
                if isinstance(v, D.DMat) and v.colfac is None and v.shape[1] == nE == m and \
                        [(int(t[1]), int(t[2]), float(t[3])) for t in v.terms] == model_terms:
                    ent.append(D.nh_hs_blob(0, 0, m, 0, D.lazy_const(1.0), cur, hist_word))
                elif isinstance(v, D.DVec) and m == 1 and v.stride == 1 and v.ptr in mouts:
                    ent.append(D.nh_hs_blob(1, mouts.index(v.ptr), 1, 0, v.lazy(), cur, hist_word))
                else:
                    ent = None
                    break
            if ent is not None:
                for q, e in enumerate(ent):
                    d.blobs[q] = e
                d.nblobs = len(ent)
                d.send_width = int(hook.get("send_width") or 0) if hook["mv"] is None else 0
                hook["rows_active"] = d.send_width
                hook["blobs_in_kernel"] = True
        h = _dp()
        _chk(_lib.nh_half_step_create(self.h, C.addressof(d), C.byref(h)))
        thr, blk, lds = _i(), _i(), _ll()
        _chk(_lib.nh_half_step_info(h, C.byref(thr), C.byref(blk), C.byref(lds)))
        spl = _i()
        _chk(_lib.nh_half_step_split(h, C.byref(spl)))
        if plan.get("staged"):  # (the span clock: stage A's launch has opened this half-step's span)
            _chk(_lib.nh_half_step_span(h, 0, 1))
        plan["hs"] = dict(key=key, plan=h, keep=(conv, lpd, total, dd), threads=thr.value,
                          blocks=blk.value, lds_bytes=lds.value, split=spl.value, tabs=tabs)
        # where the step loop stands in the current block of moves
        self.call("nh_half_step_begin_block", h, f["pos"]["slice"], f["pos"]["steps"])
        self.call("nh_half_step_launch", h, -1)

    def sorted_tables(self, hs, run):
        """the resident loop's own copies of the plan's tables, columns sorted by the first grid
        row in which they are non-zero (nh_half_step_run_tables): below the kinematic threshold
        an emission table is exactly zero -- half the grid for cfg3's TeV energies -- and once
        the highest-energy columns share a tile, the kernel does not walk those rows.  Built on
        the host from one download of each table (once per plan); a table without leading zero
        rows keeps the plan's copy."""
        import ctypes as C
        if os.environ.get("NAIMA_AMD_SORTED_TABLES", "1") == "0":
            return
        ptrs, keep = (C.c_void_p * 4)(), []
        for t, (Kt, dKt, nG, nK, lx, nonneg) in enumerate(hs.get("tabs", [])[:4]):
            tiles = (nK + 63) // 64
            if tiles > 8 or nG < 2:
                continue
            Kh, dKh = np.empty((nG, nK)), np.empty((nG, nK))
            self.join()
            _chk(_lib.nh_download(self.h, Kh.ctypes.data, Kt, Kh.nbytes))
            _chk(_lib.nh_download(self.h, dKh.ctypes.data, dKt, dKh.nbytes))
            perm, row0 = sorted_columns(Kh)
            if max(row0) < 32:  # (less than one work item's worth of rows to skip anywhere)
                continue
            Kp = self.array(np.ascontiguousarray(Kh[:, perm]))
            dKp = self.array(np.ascontiguousarray(dKh[:, perm]))
            ntrail = 8 + nK
            kd = self.empty((2 * nG * nK + (ntrail + 1) // 2,))
            self.call("nh_table_interleave", Kp, dKp, lx if nonneg else None, nG, nK, kd)
            trail = np.zeros(2 * ((ntrail + 1) // 2), dtype=np.int32)
            trail[:tiles] = row0
            trail[8:8 + nK] = perm
            th = self.array(trail, dtype=np.int32)
            self.call("nh_copy", kd.ptr + 16 * nG * nK, th, trail.nbytes)
            self.sync()
            ptrs[t] = kd.ptr
            keep.append(kd)
        if keep:
            _chk(_lib.nh_half_step_run_tables(self.h, hs["plan"], run, ptrs, 4))
            hs.setdefault("sorted", []).append(keep)  # (alive as long as the plan)

    # -- side streams ---------------------------------------------------------
    def branch(self):
        """context manager: run the enclosed launches on the next side stream (after
        everything issued so far on the main stream).  No-op when already inside a
        branch or when multistream is off."""
        return _Branch(self)

    def anchor(self):
        """a marker recorded now on the main stream (ring of 16): ``branch_at`` hangs a
        side stream off this point"""
        if self.cur_stream != -1:
            return None
        if not self._anchors:
            self._anchors = [self.marker() for _ in range(16)]
        m = self._anchors[self._nanchor % 16]
        self._nanchor += 1
        _chk(_lib.nh_marker_record(self.h, m))
        return m

    def branch_at(self, anchor):
        """context manager: the enclosed launches go to a side stream that waits only
        for ``anchor`` -- small reductions (We, Wp) then run beside the emission kernels
        issued before them.  No-op without an anchor or inside another branch."""
        return _Branch(self, anchor=anchor)

    def need(self, *objs):
        """make the current stream wait for the streams that produced ``objs``"""
        if not self._forked:
            return
        for o in objs:
            st = getattr(o, "stream", None)
            if st is None or st == self.cur_stream or st == -1:
                continue  # main-stream producers are ordered by the fork itself
            key = (self.cur_stream, st)
            if key not in self._waited:
                _chk(_lib.nh_stream_wait(self.h, self.cur_stream, st))
                self._waited.add(key)

    def join(self):
        """the main stream waits for every side stream and becomes current again"""
        if self._forked:
            _chk(_lib.nh_stream_join(self.h))
            self._forked = False
            self.cur_stream = -1
            self._waited.clear()
            for ptr, cap in self._limbo:
                self._pool.setdefault(cap, []).append(ptr)
            self._limbo = []

    def array(self, host, dtype=np.float64):
        host = np.ascontiguousarray(host, dtype=dtype)
        return self.empty(host.shape, dtype).set(host)

    def const(self, host, dtype=np.float64):
        """device copy of a small read-only array, cached by content"""
        host = np.ascontiguousarray(host, dtype=dtype)
        key = (host.shape, host.dtype.str, hash(host.tobytes()))
        hit = self._const.get(key)
        if hit is None:
            if len(self._const) > 256 + len(self._pinned):
                self._evict()
            hit = self.array(host, dtype)
            self._const[key] = hit
        return hit

    def pin_caches(self):
        """a captured graph (or a recorded step plan) holds raw pointers into the cached
        constants, grids and tables that exist now: they are never evicted"""
        self._pinned = set(self._const) | set(self._tables)
        self._pinned_ptrs = {v.ptr for v in self._const.values()}

    def _evict(self):
        keep = self._pinned
        self._const = {k: v for k, v in self._const.items() if k in keep}
        self._tables = {k: v for k, v in self._tables.items() if k in keep}
        self._lx = {k: v for k, v in self._lx.items() if k in self._pinned_ptrs}
        self._lne = {k: v for k, v in self._lne.items() if k in self._pinned_ptrs}

    def grid_logratio(self, grid_dev):
        """lx[i] = ln(x[i+1]/x[i]) for a cached grid (computed once on device)"""
        hit = self._lx.get(grid_dev.ptr)
        if hit is None:
            n = grid_dev.shape[0]
            hit = self.empty((n - 1,))
            _chk(_lib.nh_grid_logratio(self.h, grid_dev.ptr, n, hit.ptr))
            self._lx[grid_dev.ptr] = hit
        return hit

    def grid_ln(self, e_dev, e_host):
        """ln(e[i]) of a cached grid (walker-independent; lets the particle-weights
        kernel skip its per-node logarithms)"""
        hit = self._lne.get(e_dev.ptr)
        if hit is None:
            hit = self.array(np.log(np.asarray(e_host, dtype=float)))
            self._lne[e_dev.ptr] = hit
        return hit

    def table(self, key, build):
        """walker-independent emission tables, cached by what they depend on (device
        pointers of the content-addressed grid/energy arrays + scalar parameters)"""
        hit = self._tables.get(key)
        if hit is None:
            if len(self._tables) > 64 + len(self._pinned):
                self._tables = {k: v for k, v in self._tables.items() if k in self._pinned}
            hit = build()
            self._tables[key] = hit
        return hit

    def ssc_table(self, gd, nG, Ed, nE, sed, ns):
        """the tabulated SSC kernel of nh_ic_seed_walkers_tab for these three grids (cached; the
        big tables share NAIMA_AMD_TABLE_GB of HBM, default 16: the oldest ones no plan points
        into make room).  None when it does not fit: the caller evaluates the kernel per step."""
        if os.environ.get("NAIMA_AMD_SSC_TABLE", "1") == "0":
            return None
        key = ("ssc", gd.ptr, Ed.ptr, sed.ptr, nG, nE, ns)
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        nbytes = int(_lib.nh_ssc_table_bytes(nG, nE, ns))
        budget = float(os.environ.get("NAIMA_AMD_TABLE_GB", "16")) * 2.0 ** 30
        if nbytes <= 0 or nbytes > budget or self.capturing:
            return None
        big = self._big_tables = {k: v for k, v in self._big_tables.items() if k in self._tables}
        for k in list(big):
            if sum(big.values()) + nbytes <= budget:
                break
            if k not in self._pinned:
                self._tables.pop(k, None)
                del big[k]
        if sum(big.values()) + nbytes > budget:
            return None
        hit = self.empty(((nbytes + 7) // 8,))
        _chk(_lib.nh_ssc_table(self.h, gd.ptr, nG, Ed.ptr, nE, sed.ptr, ns, hit.ptr))
        self._tables[key] = hit
        big[key] = nbytes
        return hit

    # -- hipGraph capture -----------------------------------------------------
    def graph_begin(self):
        _chk(_lib.nh_graph_begin(self.h))
        self.capturing = True

    def graph_end(self):
        self.join()
        self.capturing = False
        for cap, ptrs in self._cap_pool.items():  # scratch of the captured launches
            self._retained.extend((p, cap) for p in ptrs)
        self._cap_pool = {}
        g = _dp()
        _chk(_lib.nh_graph_end(self.h, C.byref(g)))
        self._release_deferred()
        return g

    def _release_deferred(self):
        """device loops that were collected while a capture was in progress"""
        if self._release_later:
            from .device_sampler import _release_loop
            later, self._release_later = self._release_later, []
            for res in later:
                _release_loop(self, res)

    def graph_abort(self):
        if self.capturing:
            self.capturing = False
            for cap, ptrs in self._cap_pool.items():
                self._retained.extend((p, cap) for p in ptrs)
            self._cap_pool = {}
            g = _dp()
            _lib.nh_graph_end(self.h, C.byref(g))
            if g:
                _lib.nh_graph_destroy(self.h, g)
            self._release_deferred()

    def graph_launch(self, g):
        _chk(_lib.nh_graph_launch(self.h, g))

    def pinned(self, nbytes):
        """(numpy uint8 view, address) of a page-locked host buffer"""
        p = _dp()
        _chk(_lib.nh_host_alloc(self.h, int(nbytes), C.byref(p)))
        buf = (C.c_ubyte * int(nbytes)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8), p.value

    def marker(self):
        m = _dp()
        _chk(_lib.nh_marker_create(self.h, C.byref(m)))
        return m

    def sync(self):
        self.join()
        _chk(_lib.nh_sync(self.h))

    def info(self):
        name = C.create_string_buffer(256)
        cus, clk, hbm = _i(), _i(), _d()
        _chk(_lib.nh_device_info(self.h, name, 256, C.byref(cus), C.byref(hbm), C.byref(clk)))
        return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=hbm.value,
                    clock_khz=clk.value)

    def pci_bus_id(self):
        buf = C.create_string_buffer(64)
        _chk(_lib.nh_device_pci_bus_id(self.h, buf, 64))
        return buf.value.decode()

    # -- timing ---------------------------------------------------------------
    def timer_start(self):
        _chk(_lib.nh_timer_start(self.h))

    def timer_stop(self):
        ms = _d()
        _chk(_lib.nh_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile(self, on):
        _chk(_lib.nh_profile_enable(self.h, int(bool(on))))

    def profile_overhead_us(self, reps=200):
        v = _d()
        _chk(_lib.nh_profile_calibrate(self.h, int(reps), C.byref(v)))
        return v.value

    def clock_read(self, reset=False):
        """the device span clock (nh_clock_read): (microseconds the step loop's launches have
        spent on the device since the last reset, spans) -- measured on the launches themselves"""
        out = (_ll * 3)()
        _chk(_lib.nh_clock_read(self.h, int(reset), out))
        return out[0] * 1e3 / float(out[2]), int(out[1])

    def profile_read(self, reset=True):
        nk = len(NH_K_NAMES)
        ms = (_d * nk)()
        n = (_ll * nk)()
        _chk(_lib.nh_profile_read(self.h, ms, n, int(reset)))
        return {NH_K_NAMES[i]: dict(ms=ms[i], launches=int(n[i])) for i in range(nk) if n[i]}

    def call(self, name, *args):
        """invoke an entry point; DeviceArray arguments are passed as their pointers"""
        conv = [a.ptr if isinstance(a, DeviceArray) else a for a in args]
        plan = self._plan
        if plan is not None and plan["mode"] == "record" and self._in_eval:
            plan["calls"].append(name)
        _chk(getattr(_lib, name)(self.h, *conv))

    def close(self):
        if self.h:
            self._const.clear()
            self._lx.clear()
            self._tables.clear()
            _lib.nh_destroy(self.h)
            self.h = None


class _Branch:
    def __init__(self, ctx, anchor=None):
        self.ctx, self.active, self.anchor = ctx, False, anchor

    def __enter__(self):
        c = self.ctx
        if self.anchor is not None:
            if c.cur_stream == -1 and c.side_small:
                side = c._next_side
                c._next_side = (side + 1) % 4
                _chk(_lib.nh_stream_fork_at(c.h, side, self.anchor))
                c.cur_stream = side
                c._forked = True
                self.active = True
            return self
        # (not while a step graph is being captured: hipStreamEndCapture died on cfg4's graph
        # with forked side streams -- ROCm 7.2 -- and the fork never paid inside a graph)
        if c.multistream and c.cur_stream == -1 and not getattr(c, "capturing", False):
            side = c._next_side
            c._next_side = (side + 1) % 4
            _chk(_lib.nh_stream_fork(c.h, side))
            c.cur_stream = side
            c._forked = True
            self.active = True
        return self

    def __exit__(self, *exc):
        if self.active:
            _chk(_lib.nh_stream_switch(self.ctx.h, -1))
            self.ctx.cur_stream = -1
        return False


_default = {}


def device_count():
    """HIP devices this process sees (raises when the library or the HIP runtime is missing)"""
    load()
    n = _i(0)
    _chk(_lib.nh_device_count(C.byref(n)))
    return n.value


def default_device():
    """the device index a rank's default context takes: NAIMA_AMD_DEVICE, else LOCAL_RANK, else 0
    (no context is made, no GPU needed to ask)"""
    device = int(os.environ.get("NAIMA_AMD_DEVICE") or os.environ.get("LOCAL_RANK") or "0")
    if not os.environ.get("NAIMA_AMD_DEVICE") and device > 0:
        # a launcher that narrows every rank's view to its own GPU (HIP_VISIBLE_DEVICES /
        # ROCR_VISIBLE_DEVICES per rank) leaves LOCAL_RANK pointing past the one device the
        # rank sees: take what is visible (bench.py checks that the ranks' PCI bus ids differ)
        try:
            n = device_count()
        except Exception:
            n = 0
        if 0 < n <= device:
            device = device % n
    return device


def get_context(device=None):
    """process-wide default context (device from NAIMA_AMD_DEVICE / LOCAL_RANK, else 0)"""
    if device is None:
        device = default_device()
    ctx = _default.get(device)
    if ctx is None:
        ctx = Context(device)
        _default[device] = ctx
    return ctx
