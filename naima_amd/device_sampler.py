"""The device-resident step loop of ``EnsembleSampler(device=True)``.

State in HBM: coords[N][ndim], logp[N], the current blobs, per-walker acceptance
counters, and the chain history.  Per half-step the host only draws the random
numbers of the move (replicated numpy stream: identical on every rank and identical
to the host-path sampler), ships them as two small blocks, and launches ONE hipGraph:

    nh_move_propose -> [the launch sequence the naima model function produces:
    nh_pack_rows, nh_particle_weights, nh_integrate_tables, nh_synchrotron, ...]
    -> nh_priors -> nh_lnprob -> (RCCL all-gather when sharded) -> nh_move_accept
    -> nh_scatter_rows (blobs)

The graph is captured from the FIRST replay of the user's unchanged Python model
function on lazy device parameters (``naima_amd.darray``); walker-independent
pieces (grids, emission tables, data columns) are evaluated during a warm-up pass,
land in the context's caches, and are therefore not part of the graph.  After
capture no Python model code runs inside the loop.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from . import units as u
from .darray import DMat, DPars, DVec
from .dist import shard_bounds, shard_counts


class DeviceState:
    """what the loop yields: host views are fetched on first access"""

    def __init__(self, loop, rng):
        self._loop, self.random_state = loop, rng
        self._c = self._l = None

    @property
    def coords(self):
        if self._c is None:
            self._loop._flush_pending()
            self._loop._settle_before_read()  # (one GPU: the friendlier message, ahead of the status)
            self._loop.check_resident()
            self._loop.check_nan()
            self._c = self._loop.coords.get().reshape(self._loop.N, self._loop.ndim)
        return self._c

    @property
    def log_prob(self):
        if self._l is None:
            self._loop._flush_pending()
            self._loop._settle_before_read()
            self._loop.check_resident()
            self._loop.check_nan()
            self._l = self._loop.logp.get()
        return self._l

    @property
    def blobs(self):
        return self._loop.host_blobs()

    def __iter__(self):
        return iter((self.coords, self.log_prob, self.random_state))


def _release_loop(ctx, res):
    """finalizer of a DeviceLoop: its graph executables, then the half-step plan they launch
    (device packs, counters, partial-spectra buffers, the host descriptor)"""
    if ctx.h is None:
        return  # the context went first and took the device allocations with it
    if ctx.capturing:
        # (the collector may run this in the middle of ANOTHER loop's stream capture, which a
        # synchronisation would invalidate: the context releases it after the capture)
        ctx._release_later.append(res)
        return
    try:
        ctx.sync()
        for g in res["graphs"]:
            _lib._lib.nh_graph_destroy(ctx.h, g)
        for r in res.get("runs", []):
            _lib._lib.nh_half_step_run_destroy(ctx.h, r)
        res["runs"] = []
        for plan in res["plans"]:
            hs = plan.get("hs") if plan else None
            if hs is not None and hs.get("plan") is not None:
                _lib._lib.nh_half_step_destroy(ctx.h, hs["plan"])
                hs["plan"] = None
                plan["hs"] = None
            st = plan.get("stage") if plan else None
            if st is not None and st.get("plan") is not None:
                _lib._lib.nh_half_step_destroy(ctx.h, st["plan"])
                st["plan"] = None
                plan["stage"] = None
    except Exception:
        pass
    res["graphs"], res["plans"] = [], []


class DeviceLoop:
    def __init__(self, sampler):
        if not sampler.naima_style:
            raise ValueError("device=True needs naima_style=True (lnprob(pars, data, model, prior))")
        self.s = sampler
        self.ctx = _lib.get_context()
        self.N, self.ndim = sampler.nwalkers, sampler.ndim
        self.ns = self.N // 2
        comm = sampler.comm
        if self.ns < comm.size:
            raise ValueError("device=True needs at least one walker of every half-ensemble (%d) "
                             "per rank (%d)" % (self.ns, comm.size))
        # (a half-ensemble that does not divide evenly: the first ranks take one walker more,
        # as dist.shard_bounds says; all-gathers then travel padded to the largest block)
        self.lo, self.hi = shard_bounds(self.ns, comm.rank, comm.size)
        if comm.size > 1 and getattr(comm, "group", None) is not None:
            # ranks that sit on ONE device (a multi-GPU run rehearsed on a one-GPU box) each plan
            # for their share of its CUs -- or their resident launches cannot all be resident
            # together and the first shared launch gives up.  Who shares a device is read off the
            # ranks' PCI bus ids (collective: every rank, always); NAIMA_AMD_CU_SHARE overrides.
            import collections
            ids = [p_.decode() for p_ in comm.group.allgather_bytes(self.ctx.pci_bus_id().encode())]
            share = max(collections.Counter(ids).values())
            if share > 1 and not os.environ.get("NAIMA_AMD_CU_SHARE"):
                os.environ["NAIMA_AMD_CU_SHARE"] = str(share)
        self._pads = {}
        # the sharded code path (split graphs around the all-gather) even for ONE rank:
        # lets a 1-GPU box run everything but the multi-process part (tests)
        self.sharded = comm.size > 1 or os.environ.get("NAIMA_AMD_FORCE_SHARDED", "0") == "1"
        # the all-gather captured INTO the step graphs (the sharded loop then replays whole
        # steps like the single-GPU loop) when the communicator's ranks have been seen to
        # survive that in a throw-away probe process (RcclComm.graph_capture_ok); else the
        # collective sits between two graphs.  NAIMA_AMD_RCCL_IN_GRAPH = auto | 0 | 1
        mode = os.environ.get("NAIMA_AMD_RCCL_IN_GRAPH", "auto")
        # (several ranks normally take the resident loop over the shared ensemble after a few
        # steps, and these launches around an all-gather are only its run-in and its fallback:
        # "auto" then does not spend a probe process per rank on them)
        shared_expected = (comm.size > 1 and getattr(comm, "group", None) is not None and
                           os.environ.get("NAIMA_AMD_SHARED", "1") != "0" and
                           os.environ.get("NAIMA_AMD_RESIDENT", "1") != "0" and sampler.use_graph)
        self.coll_in_graph = self.sharded and getattr(comm, "in_stream", False) and (
            mode == "1" or (mode == "auto" and not shared_expected and
                            hasattr(comm, "graph_capture_ok") and comm.graph_capture_ok()))
        self.split = self.sharded and not self.coll_in_graph  # collective between two graphs
        self.nloc = self.hi - self.lo
        ctx = self.ctx
        self.coords = ctx.empty((self.N * self.ndim,))
        self.logp = ctx.empty((self.N,))
        # the random numbers of up to KSTEPS ensemble steps travel as ONE block; a device
        # cursor selects the current half-step's slice (advanced by the accept kernel)
        self.KSTEPS = 32
        # ensemble steps per hipGraph launch when the host has nothing to do
        self.GSTEPS = max(1, int(os.environ.get("NAIMA_AMD_GSTEPS", "8")))
        self.multi_graph = None
        # one spare slice: the fused move kernel proposes the half-step AFTER the one it
        # accepts, so the last one of a block reads (and discards) slice 2*KSTEPS
        # (room for MOVES_CAP steps: the resident loop keeps the move stream's next steps on the
        # device ahead of time -- uploaded on a copy stream while the current launch runs)
        self.MOVES_CAP = 4 * self.KSTEPS
        self.blk = ctx.empty((2 * self.MOVES_CAP + 1, 3 * self.ns))
        ctx.call("nh_memset", self.blk, 0, self.blk.nbytes)
        # steps on the device / consumed / last ahead marker / markers behind the last two launches
        self._mv = dict(have=0, used=0, ahead=None, prev=None, last=None)
        self._launch_marks = [ctx.marker() for _ in range(3)]
        self._nlaunch = 0
        self._blk_tmp = None
        # nh_step_front mode: set after a recording evaluation when the model's parameter
        # packs read the proposal buffer and it asks for ONE particle-weights launch
        self.fused = False
        # ... and, when every launch of the model evaluation is one nh_half_step can absorb
        # (table reductions, a synchrotron component, the likelihood), the whole half-step is
        # ONE launch: proposal -> ... -> accept (NAIMA_AMD_MEGA=0 keeps the three launches)
        self.mega = False
        # next slice of the current block of moves / history rows before that block: what a
        # half-step plan created in the middle of a block has to be told
        self._pos = dict(slice=0, steps=0, bake=False)
        self.multi_graphs = {}
        self._plan = None
        self._front_args = None
        self.done = ctx.empty((1,), dtype=np.int32)
        ctx.call("nh_memset", self.done, 0, 4)
        self._hook = None
        # nh_hist descriptor in HBM: the fused kernel appends the chain history itself
        # ... followed, in the one-launch mode, by the blobs' histories (device words holding
        # the base of each blob's history of the current call; 0: not kept).  Eight words in
        # one buffer: a call of sample() re-points all of them with ONE small launch
        # (nh_set_words) instead of two staged uploads
        self.histd = ctx.empty((8,), dtype=np.int64)
        ctx.call("nh_memset", self.histd, 0, self.histd.nbytes)
        self.blobhist_ptr = self.histd.ptr + 32
        self.blobs_in_kernel = False
        self.send_width = 0
        self.cursor = ctx.empty((1,), dtype=np.int32)
        self.sel = ctx.empty((self.ns,), dtype=np.int32)
        self.qT = ctx.empty((self.ndim * self.nloc,))
        self.factors = ctx.empty((self.nloc,))
        self.newlp = ctx.empty((self.ns,))
        self.mylp = ctx.empty((max(self.nloc, 1),))
        self.new_blobs = None
        self.all_blobs = None
        self.ident = ctx.array(np.arange(max(self.nloc, 1), dtype=np.int32), dtype=np.int32)
        self.graph2 = None
        self.graph21 = None
        self._pending = False
        self.step_graph = None
        self._markers = [ctx.marker() for _ in range(4)]
        self._inflight = []
        self._nmark = 0
        self.accepted = ctx.empty((self.ns,), dtype=np.int32)
        self.nacc = ctx.empty((self.N,), dtype=np.int32)
        ctx.call("nh_memset", self.nacc, 0, self.nacc.nbytes)
        self.graph = None
        self.cur_blobs = None      # list of (device buffer [N][m], m, unit, trailing shape)
        self.const_blobs = []      # (position, value) of blobs that are plain numbers
        self.hist = []             # pending device history blocks
        self.warm = 0
        self._have_state = False
        # what this loop owns on the device besides pooled buffers: released when it goes
        import weakref
        self._res = dict(graphs=[], plans=[], runs=[])
        # the resident loop (nh_half_step_run: a whole block of moves per launch, walkers handed
        # from half-step to half-step by per-walker records): None = not tried yet, False = this
        # plan / configuration cannot (or NAIMA_AMD_RESIDENT=0), else the handle
        self._run = None if os.environ.get("NAIMA_AMD_RESIDENT", "1") != "0" else False
        self._resident_trust = 0          # launches of the resident loop that were checked as they were made
        self.resident_failed_launches = 0
        # launches of a one-GPU resident loop that are not yet known to have ended well: the
        # sampler queues launch n + 1 only once launch n - 1 is (nh_half_step_run_report: page-
        # locked memory the launch's epilogue writes, no stream operation), and settles all of
        # them before a call returns -- a launch that gave up later in a run costs the replay of
        # two blocks of moves at most, by the per-launch kernel (sample)
        self._rl_count = 0     # launches queued so far (the library counts them the same way)
        self._rl_pending = []
        self._verify = os.environ.get("NAIMA_AMD_VERIFY_LAUNCHES", "1") != "0"
        # copies of the current blobs as a launch WITHOUT a history found them (such a launch
        # writes the blobs of accepted moves straight into the current-blob arrays): what a
        # launch that gave up -- or ran void behind one that did -- is rolled back to
        self._cur_snaps = [None] * 4
        self._last_snap = None
        self._nan_pending = self._forbidden_pending = 0
        _n = _lib._i(0)  # (the context's word may hold what an earlier sampler left uncounted)
        _lib._chk(_lib._lib.nh_nan_count(self.ctx.h, 1, C.byref(_n)))
        self.resident_launches = 0
        # ... and over an ensemble shared by several GPUs (nh_half_step_run_create_shared: movers
        # store their walkers' records into every rank's ring; no collective per half-step).
        # What a rank's launches leave on ITS device only -- the history rows and blobs of the
        # moves it made, their acceptance counts, the blobs' current values (stamped) -- is
        # merged over the control plane when somebody reads it
        self.shared = False
        self.shared_info = None
        self._cur_dirty = False   # current blobs differ between ranks (merge by stamp pending)
        self._cur_host = None     # host copies of the current blobs while nothing has changed them
        self._acc_dirty = False   # shared launches have accepted moves since the counts were summed
        self._acc_shared = 0.0
        self._acc_base = 0.0      # moves accepted by the launches of a shared loop that was given up
        # A shared ensemble's launches are not checked one by one (that would be a collective per
        # launch): every rank keeps the ensemble as it stood at the last point where ALL ranks were
        # known to be well (flush, reset: `verified`), and a journal of the sample() calls made
        # since.  A launch that gave up on any rank is found at the next such point -- the status
        # reduction every rank takes part in -- and the ranks TOGETHER go back to the kept ensemble
        # and make the journal's steps again with one launch and one all-gather per half-step
        # (the reference's Pool carries on when a worker is slow, core.py:523-536; this used to be
        # an exception on every rank)
        self._flushed = False  # (nothing has run since the last flush: see flush)
        self._sh_snap = None
        self._sh_want = False
        self._sh_journal = []
        self._replaying = False
        self._finalizer = weakref.finalize(self, _release_loop, self.ctx, self._res)

    # ------------------------------------------------------------------ pieces
    def reset(self):
        """forget the chain and the acceptance counts (the sampler clears its host copies): the
        history still on the device is dropped, not downloaded"""
        self._flush_pending()
        self.check_resident(collective=True)
        self.check_nan(collective=True)
        self.ctx.check_general()
        self._sync_cur_blobs()
        self.hist = []
        self.ctx.call("nh_memset", self.nacc, 0, self.nacc.nbytes)
        if self.shared:
            _lib._chk(_lib._lib.nh_half_step_run_counters(self.ctx.h, self._run, None, None, 1))
            self._acc_dirty, self._acc_shared = False, 0.0
        self._acc_base = 0.0
        self._flushed = False

    def _eval(self, qT_buf, n):
        """run the user's model on device parameters: (total DVec, blob list)"""
        pars = DPars(self.ctx, qT_buf, self.ndim, n)
        self.ctx._in_eval = True
        try:
            res = self.s.log_prob_fn(pars, *self.s.args)
        finally:
            self.ctx._in_eval = False
        self.s.n_lnprob_calls += 1
        self.s.n_walker_evals += n
        rows = bool(self._hook and self._hook.get("rows_active")) and self.ctx._accept_hook is self._hook
        total = res[0] if rows else res[0].dense()  # (rows: exchanged as they are)
        blobs = list(res[1:])
        # plain numbers among the blobs (lnprob's (model, nan) for a model that returns no
        # blob of its own, core.py:108-110) are the same for every walker and every step:
        # they are not kept in HBM but put back where they belong when the blobs are read
        self.const_blobs = [(j, float(b)) for j, b in enumerate(blobs)
                            if isinstance(b, (float, int))]
        return total, [b for b in blobs if not isinstance(b, (float, int))]

    def _blob_dense(self, b):
        """blob -> (owner, ptr, m, unit, trailing shape) with rows over walkers"""
        unit = None
        if isinstance(b, u.Quantity):
            b, unit = b.value, b.unit
        if isinstance(b, DMat):
            own, ptr = b.buffer()
            return own, ptr, b.shape[1], unit, (b.shape[1],)
        if isinstance(b, DVec):
            d = b.dense()
            return d, d.ptr, 1, unit, ()
        raise TypeError("blob of type %s cannot be kept on the device" % type(b).__name__)

    def _part_evaluate(self):
        """propose this rank's block and evaluate it: everything up to the new
        log-probabilities (the part that precedes the exchange)"""
        ctx = self.ctx
        if self.fused:
            # proposal, parameter rows, weights, We: written by the preceding nh_step_front
            self._plan["i"] = [0, 0, 0, 0, 0]
            ctx._plan = self._plan
            self._hook["used"] = False
            ctx._accept_hook = self._hook
        else:
            ctx.call("nh_move_propose", self.coords, self.blk, self.cursor, self.ns, self.ndim,
                     self.lo, self.nloc, self.qT, self.factors)
        try:
            total, blobs = self._eval(self.qT, self.nloc)
        finally:
            ctx._plan = None
            ctx._accept_hook = None
            ctx.flush()  # a held-back launch nobody consumed
        if self.sharded:
            # fixed hand-over buffer so that the two graphs and the collective between
            # them always see the same addresses
            if self._hook and self._hook.get("rows_active"):
                pass  # rows { lnprob | blobs } are exchanged as the launch wrote them
            elif total.ptr != self.mylp.ptr:  # (the fused likelihood wrote it there itself)
                ctx.call("nh_copy", self.mylp, total.ptr, 8 * self.nloc)
            self._newlp_ptr = self.newlp.ptr
        else:
            self._total = total  # single rank: the accept kernel reads it in place
            self._newlp_ptr = total.ptr
        self._stage_blobs(blobs)

    def _stage_blobs(self, blobs):
        """copy the proposal's blobs into fixed staging buffers (only when kept)"""
        if not (self.s.store_blobs and self.cur_blobs):
            return
        if self.mega and self._hook.get("blobs_in_kernel"):
            self.blobs_in_kernel = True  # the half-step launch keeps the blobs itself
            return
        if self.new_blobs is None:
            self.new_blobs = [self.ctx.empty((self.nloc, m)) for _, m, _, _ in self.cur_blobs]
            if self.sharded:  # every rank keeps every walker's blobs
                self.all_blobs = [self.ctx.empty((self.ns, m)) for _, m, _, _ in self.cur_blobs]
        for nb, (cur, m, _, _), b in zip(self.new_blobs, self.cur_blobs, blobs):
            own, ptr, mb, _, _ = self._blob_dense(b)
            self.ctx.call("nh_scatter_rows", nb, m, ptr, mb, self.ident, None, 0, self.nloc, m)
            del own

    def _part_accept(self):
        ctx = self.ctx
        if self.fused and self.sharded and self.blobs_in_kernel and self.send_width:
            import ctypes as C
            nb = len(self.cur_blobs)
            cur = (C.c_void_p * nb)(*[c_.ptr for c_, _, _, _ in self.cur_blobs])
            mm = (C.c_int * nb)(*[m for _, m, _, _ in self.cur_blobs])
            ctx.call("nh_move_accept_rows", self.coords, self.logp, self.blk, self.cursor,
                     self.recv_rows, self.send_width, self.ns, self.ndim, self.accepted, self.nacc,
                     self.sel, 0, nb, cur, mm)
        elif self.fused:
            if not (self._hook and self._hook["used"] and self._hook["mv"] is not None):
                # sharded (the accept waits for the all-gather), or a foreign likelihood
                ctx.call("nh_move_accept", self.coords, self.logp, self.blk, self.cursor,
                         self._newlp_ptr, self.ns, self.ndim, self.accepted, self.nacc,
                         self.sel, 0)
        else:
            ctx.call("nh_move_accept", self.coords, self.logp, self.blk, self.cursor,
                     self._newlp_ptr, self.ns, self.ndim, self.accepted, self.nacc, self.sel, 1)
        if self.s.store_blobs and self.cur_blobs and not self.blobs_in_kernel:
            if self.sharded:
                for ab, (cur, m, _, _) in zip(self.all_blobs, self.cur_blobs):
                    ctx.call("nh_scatter_rows", cur, m, ab, m, self.sel, self.accepted, 0,
                             self.ns, m)
            else:
                for nb, (cur, m, _, _) in zip(self.new_blobs, self.cur_blobs):
                    ctx.call("nh_scatter_rows", cur, m, nb, m, self.sel, self.accepted, self.lo,
                             self.nloc, m)
        if self.fused and not self.mega:
            self._front()

    def _front(self):
        """proposal of the next half-step + parameter packs + particle weights + We/Wp
        reductions (+ chain history, cursor advance): one launch"""
        self.ctx.call("nh_step_front", self.coords, self.logp, self.blk, self.cursor, self.done,
                      self.ns, self.ndim, self.lo, self.nloc, self.qT, self.factors,
                      *self._front_args, self.histd)

    def _record_half_step(self):
        """a half-step launched piece by piece while the model's parameter packs, weights
        launch and single-row reductions are recorded; if nh_step_front can produce them
        the loop switches to it"""
        import ctypes as C
        from .darray import nh_accept, nh_grid, nh_lazy, nh_moment, nh_pack
        ctx = self.ctx
        plan = ctx.plan_begin()
        try:
            self._half_step_body()
        finally:
            ctx._plan = None
        if not self.s.fuse_moves:
            return
        packs, weights, moments = plan["packs"], plan["weights"], plan["moments"]
        if not (1 <= len(packs) <= 4 and len(weights) == 1 and len(moments) <= 4):
            return
        lo_a = self.qT.ptr
        hi_a = lo_a + 8 * self.ndim * self.nloc
        pk = (nh_pack * 4)()
        for q, ((raw, ncols, N), out) in enumerate(packs):
            cols = (nh_lazy * 8).from_buffer_copy(raw.ljust(C.sizeof(nh_lazy) * 8, b"\0"))
            if N != self.nloc or ncols > 8:
                return
            for j in range(ncols):
                base = cols[j].base or 0
                if base and not (lo_a <= base < hi_a and (base - lo_a) % (8 * self.nloc) == 0
                                 and cols[j].stride == 1):
                    return
            pk[q].cols = cols
            pk[q].ncols, pk[q].ld, pk[q].out = ncols, ncols, out.ptr
        (kind, rows_ptr, N, grids), bufs = weights[0]
        if N != self.nloc or rows_ptr not in [out.ptr for _, out in packs] or len(grids) > 4:
            return
        gd = (nh_grid * 4)()
        wptr = {}
        for g, ((e, x, lne, lx, scale, nG), (wk, lwk)) in enumerate(zip(grids, bufs)):
            gd[g] = nh_grid(e, x, wk.ptr, lwk.ptr, scale, nG, 0, lne, lx)
            wptr[wk.ptr] = g
        mm = (nh_moment * 4)()
        lds_nodes = set()
        for k, ((w, lw, N, nG, lx, Kt, dlnKt), out) in enumerate(moments):
            if w not in wptr or N != self.nloc:
                return
            mm[k] = nh_moment(wptr[w], 0, Kt, dlnKt, out.ptr)
            lds_nodes.add(wptr[w])
        if 40 * sum(grids[g][5] for g in lds_nodes) * max(len(moments), 1) > 48 * 1024:
            return
        plan["mode"] = "replay"
        ctx.pin_caches()
        self._plan = plan
        self._res["plans"].append(plan)
        self._front_args = (pk, len(packs), kind, rows_ptr, gd, len(grids), mm, len(moments))
        if not self.sharded:
            self._hook = dict(N=self.nloc, used=False, total=None,
                              mv=nh_accept(self.coords.ptr, self.logp.ptr, self.blk.ptr,
                                           self.cursor.ptr, self.ns, self.ndim, self.lo, 0,
                                           self.accepted.ptr, self.nacc.ptr, self.sel.ptr))
        else:  # sharded: likelihood into the all-gather send buffer, accept afterwards
            self._hook = dict(N=self.nloc, used=False, total=self.mylp, mv=None)
        self.fused = True
        self.mega = self._can_be_one_launch(plan, grids, wptr, moments)
        if self.mega:
            plan["mega"] = True
            plan["front"] = dict(coords=self.coords.ptr, logp=self.logp.ptr, blk=self.blk.ptr,
                                 cursor=self.cursor.ptr, pos=self._pos, qT=self.qT.ptr,
                                 factors=self.factors.ptr, hist=self.histd.ptr,
                                 accepted=self.accepted.ptr, naccepted=self.nacc.ptr,
                                 sel=self.sel.ptr, ns=self.ns, ndim=self.ndim, lo=self.lo,
                                 nloc=self.nloc, front_args=self._front_args)
            if self._hook.get("total") is None:
                self._hook["total"] = ctx.empty((self.nloc,))  # persistent: the plan points at it
            if self.s.store_blobs and self.cur_blobs:
                self._hook["blobs"] = [(cur.ptr, m, self.blobhist_ptr + 8 * i)
                                       for i, (cur, m, _, _) in enumerate(self.cur_blobs)]
                if self.sharded:
                    # the blobs travel with the log-probabilities: ONE all-gather of rows
                    # { lnprob | blob 0 | blob 1 ... } per half-step
                    self.send_width = 1 + sum(m for _, m, _, _ in self.cur_blobs)
                    self._hook["send_width"] = self.send_width
                    self.send_rows = ctx.empty((max(self.nloc, 1), self.send_width))
                    self.recv_rows = ctx.empty((self.ns, self.send_width))
                    self._hook["total_rows"] = self.send_rows
        # new slice protocol: cursor = the slice accepted last.  The two piecewise
        # half-steps (slices 0 and 1 of the first block) left it at 2.
        self.cursor.set(np.array([1], dtype=np.int32))
        if not self.mega:
            self._front()

    def _can_be_one_launch(self, plan, grids, wptr, moments):
        """every launch the recorded model evaluation made is one nh_half_step absorbs, and
        its working set fits in one workgroup's LDS"""
        if os.environ.get("NAIMA_AMD_MEGA", "1") == "0":
            return False
        allowed = {"nh_pack_rows", "nh_particle_weights_multi", "nh_integrate_tables",
                   "nh_synchrotron", "nh_lnprob"}
        emit = plan["emit"]
        ntab = sum(1 for e in emit if e["kind"] == "tab")
        nsyn = len(emit) - ntab
        # A model that takes its synchrotron spectrum twice with launches of other kernels in
        # between -- the SSC seed of examples/CrabNebula_SynSSC.py:29-45: Synchrotron.flux at the
        # seed's energies, a linear combination, the seed integral (sixteen walkers per wave: not
        # a one-workgroup-per-walker job), Synchrotron.flux at the data's -- runs as TWO launches
        # of the half-step kernel around those (Context._stage_a): plan["staged"]
        between = {"nh_lincomb", "nh_ic_seed_walkers_tab", "nh_ic_seed_walkers"}
        staged = nsyn == 2 and bool(set(plan["calls"]) & between)
        if staged:
            syn = [e for e in emit if e["kind"] == "syn"]
            if os.environ.get("NAIMA_AMD_STAGED", "1") == "0" or emit[0]["kind"] != "syn" or \
                    syn[0]["key"][1:9] != syn[1]["key"][1:9] or \
                    any(e.get("E_host") is None for e in syn) or moments:
                return False
            allowed = allowed | between
            nsyn = 1
        if not set(plan["calls"]) <= allowed or not emit or ntab > 4 or nsyn > 1:
            return False
        if plan["calls"].count("nh_lnprob") != 1:
            return False
        # every integrate call is either a recorded single-row reduction or an emission table
        if plan["calls"].count("nh_integrate_tables") != ntab + len(moments) or \
                plan["calls"].count("nh_synchrotron") != (2 if staged else nsyn):
            return False
        lds = 88 + 3 * sum(g[5] for g in grids) + sum(2 * grids[wptr[m[0][0]]][5] for m in moments)
        items = nspec = 0
        for e in emit:
            k = e["key"]
            if e["N"] != self.nloc or k[1] not in wptr:
                return False
            if e["kind"] == "tab":
                nG, nK = k[4], k[8]
                items += ((nK + 63) // 64) * ((nG - 1 + 31) // 32)
                nspec += nK
            elif not staged:
                nG, nE = k[8], k[10]
                lds += 3 * nG + 4 * nE + 1 + 32 * nE
                nspec += nE
        lds += min(items, 96) * 64 + nspec
        if staged:  # its other launch: one synchrotron component over both sets of energies
            nG, nEa = syn[0]["key"][8], syn[0]["key"][10] + syn[1]["key"][10]
            cd = max(1, min(32, (40 * 1024) // (8 * nEa)))
            lds = max(lds, 88 + 6 * nG + (5 + cd) * nEa + 8 * syn[0]["key"][10])
        if 8 * lds > 140 * 1024:
            return False
        plan["staged"] = staged
        return True

    def _gather_rows(self, send_ptr, recv, n, width):
        """all-gather of rows of `width` doubles, rank r contributing its block of the n rows
        (dist.shard_counts); blocks of unequal length travel padded to the longest and are put
        side by side on arrival (one small copy per rank, in stream order)"""
        ctx, comm = self.ctx, self.s.comm
        counts = shard_counts(n, comm.size)
        if len(set(counts)) == 1:
            comm.allgather_device(ctx, send_ptr, recv, counts[0] * width)
            return
        m = max(counts)
        key = (m * width, comm.size)
        if key not in self._pads:
            self._pads[key] = (ctx.empty((m * width,)), ctx.empty((comm.size * m * width,)))
        ps, pr = self._pads[key]
        ctx.call("nh_copy", ps, send_ptr, 8 * counts[comm.rank] * width)
        comm.allgather_device(ctx, ps.ptr, pr, m * width)
        off = 0
        for r, cnt in enumerate(counts):
            ctx.call("nh_copy", recv.ptr + 8 * off * width, pr.ptr + 8 * r * m * width,
                     8 * cnt * width)
            off += cnt

    def _exchange(self):
        """the one collective of the path: every rank's new log-probabilities"""
        if self.sharded and self.blobs_in_kernel and self.send_width:
            self._gather_rows(self.send_rows.ptr, self.recv_rows, self.ns, self.send_width)
        elif self.sharded:
            self._gather_rows(self.mylp.ptr, self.newlp, self.ns, 1)
            if self.s.store_blobs and self.cur_blobs:
                # blobs of the proposals follow their log-probabilities: the walker a rank
                # evaluates changes every step, so every rank keeps all of them
                for nb, ab, (cur, m, _, _) in zip(self.new_blobs, self.all_blobs, self.cur_blobs):
                    self._gather_rows(nb.ptr, ab, self.ns, m)

    def _half_step_body(self):
        self._part_evaluate()
        self._exchange()
        self._part_accept()
        self._pos["slice"] += 1

    def _init_state(self, coords_host, logp_host):
        ctx, s = self.ctx, self.s
        self.coords.set(np.ascontiguousarray(coords_host, dtype=float).ravel())
        if logp_host is not None and self._have_state:
            self.logp.set(np.asarray(logp_host, dtype=float))
            return
        # initial log-probability (and blobs) of the whole ensemble: this rank's block
        lo, hi = shard_bounds(self.N, s.comm.rank, s.comm.size)
        qT = ctx.array(np.ascontiguousarray(coords_host[lo:hi].T))
        total, blobs = self._eval(qT, hi - lo)
        if self.sharded:
            self._gather_rows(total.ptr, self.logp, self.N, 1)
        else:
            ctx.call("nh_copy", self.logp, total.ptr, 8 * self.N)
        if s.store_blobs and blobs:
            self.cur_blobs, units = [], []
            ident = ctx.array(np.arange(lo, hi, dtype=np.int32), dtype=np.int32)
            for b in blobs:
                own, ptr, m, unit, trail = self._blob_dense(b)
                cur = ctx.empty((self.N, m))
                if self.sharded:
                    self._gather_rows(ptr, cur, self.N, m)
                else:
                    ctx.call("nh_scatter_rows", cur, m, ptr, m, ident, None, 0, hi - lo, m)
                self.cur_blobs.append((cur, m, unit, trail))
                units.append(unit)
            for j, _ in self.const_blobs:
                units.insert(j, None)
            s.blob_units = units
        lp = self.logp.get()
        if np.any(np.isnan(lp)):
            raise ValueError("Probability function returned NaN")
        self._have_state = True
        self._cur_host = None

    def host_blobs(self):
        if not self.cur_blobs:
            return None
        self._sync_cur_blobs()  # (a shared ensemble: collective -- every rank asks, or none)
        out = [cur.get().reshape((self.N,) + trail) for cur, m, _, trail in self.cur_blobs]
        for j, v in self.const_blobs:
            out.insert(j, np.full((self.N,), v))
        return out

    # ------------------------------------------------------------------- loop
    def sample(self, initial_state, iterations, store, yield_every=1):
        from .sampler import State
        s, ctx = self.s, self.ctx
        if isinstance(initial_state, DeviceState) and initial_state._loop is self:
            pass  # continue from the ensemble already in HBM
        else:
            st = State(initial_state)
            if st.coords.shape != (self.N, self.ndim):
                raise ValueError("incompatible input dimensions")
            self._init_state(st.coords, st.log_prob)
        rng, N, ns = s._rng, self.N, self.ns
        iterations = int(iterations)
        jentry = None
        self._flushed = False
        if self.sharded and s.comm.size > 1 and not self._replaying and self._run is not False:
            fresh = not (isinstance(initial_state, DeviceState) and initial_state._loop is self)
            if self._sh_snap is None or fresh:  # (normally kept at the verified point itself, below)
                self._shared_snapshot()
            # (the steps the call has MADE, kept up to date as it goes: a caller that breaks out of
            # the generator early must not have the full count replayed -- ADVICE r5)
            jentry = [0, bool(store), yield_every]
            self._sh_journal.append(jentry)
        block = None
        if store and iterations > 0:
            block = dict(n=0, coords=ctx.empty((iterations, N * self.ndim)),
                         logp=ctx.empty((iterations, N)),
                         blobs=[ctx.empty((iterations, N * m)) for _, m, _, _ in
                                (self.cur_blobs or [])] if s.store_blobs else [])
            if self.sharded and self.s.comm.size > 1:
                # a shared ensemble's launches: who moved (and accepted) what, per row; the
                # blobs as they stood before the first row (rows of rejected moves are filled
                # from the row before when the ranks' histories are merged)
                self._sync_cur_blobs()
                block["own"] = ctx.empty((iterations, N), dtype=np.int32)
                block["shared_rows"] = []  # [row0, row1) of every shared launch
                block["cur0"] = []
                if block["blobs"] and self._cur_host is not None:
                    block["cur0"] = self._cur_host  # (host copies the last merge left: no launch)
                elif block["blobs"]:
                    for cur, m, _, _ in self.cur_blobs or []:
                        c0 = ctx.empty((N, m))
                        ctx.call("nh_copy", c0, cur, 8 * N * m)
                        block["cur0"].append(c0)
            self.hist.append(block)
        if not (isinstance(initial_state, DeviceState) and initial_state._loop is self):
            self._sync_cur_blobs()
        # chain history: appended on the device by nh_move_cycle when it is active for
        # the whole call, else one pair of copies per step
        dev_hist = self.fused and block is not None and not (
            self.mega and self._plan["hs"] is None)  # (the launch's plan does not exist yet)
        blob_hist_ok = not (block is not None and block["blobs"] and not (dev_hist and self.blobs_in_kernel))
        # the whole call as launches of the resident loop, one per block of moves: nothing of the
        # per-launch kernel's bookkeeping (history descriptor words, slice counter) is needed
        fast = (self._resident_ok() and yield_every >= iterations and (block is None or dev_hist)
                and blob_hist_ok)
        def per_launch_history():
            # (the per-launch kernel's history descriptor: where its launches append the chain)
            words = [block["coords"].ptr, block["logp"].ptr, 0, iterations] if dev_hist \
                else [0, 0, 0, 0]
            if dev_hist and self.blobs_in_kernel:
                words = words + [hb.ptr for hb in block["blobs"]]
            words = (words + [0, 0, 0, 0])[:8]
            ctx.call("nh_set_words", self.histd, (C.c_longlong * 8)(*words), 8)

        if self.fused and not fast:
            per_launch_history()
        blob_dev_hist = dev_hist and self.blobs_in_kernel  # the launches append the blobs too
        moves = s.moves(pinned=True)
        it = 0
        mv = self._mv

        def rollback(rec):
            """a launch of the resident loop gave up waiting for a walker's record (its workgroups
            were not all resident: another process on the GPU, a profiler) -- `rec` is the first
            such launch, everything queued behind it found the same status and changed nothing
            either (k_run_epilogue leaves the ensemble, the counters and the blobs as a launch
            that gave up found them).  The books go back to where `rec` was queued, the move
            stream is made again up to that step (a function of the seed), and the per-launch
            loop takes over for good: -> (it, the new move stream)"""
            import warnings
            from ._lib import Moves
            void = [r for r in self._rl_pending if r["launch"] >= rec["launch"]]
            self._rl_pending = []
            ctx.sync()  # (the void launches behind it: nothing of theirs may still be running)
            self.resident_reason = ("a launch of the resident loop gave up waiting for a walker's "
                                    "record (status %d): its workgroups were not all resident"
                                    % rec["status"])
            warnings.warn(self.resident_reason + "; %d block(s) of moves are replayed and the run "
                          "continues with one launch per half-step" % len(void))
            self._run = False
            self.resident_failed_launches += len(void)
            self._read_counts(reset=False, set_to=rec["counts"])  # (the replay counts them again)
            self._restore_cur_blobs(rec.get("snap"))  # (a void launch behind it may have written some)
            s.iteration, s.steps_total = rec["iteration"], rec["steps_total"]
            s.n_lnprob_calls, s.n_walker_evals = rec["nlc"], rec["nwe"]
            self.resident_launches = rec["rl"]
            if block is not None:
                block["n"] = rec["block_n"]
            while self._inflight:
                ctx.call("nh_marker_wait", self._inflight.pop(0))
            mv["have"] = mv["used"] = 0
            mv["ahead"] = mv["prev"] = mv["last"] = None
            old = s._moves
            s._moves = Moves(s.seed, s.nwalkers, s.a, ksteps=32, depth=4, pinned=True)
            if old is not None:
                old.close()
            left = rec["steps_total"]
            while left > 0:  # (the steps every loop of this sampler has made so far)
                _, got = s._moves.take(min(32, left))
                left -= got
            if self.fused:
                per_launch_history()
            return rec["it"], s._moves
        while fast and it < iterations:
            # up to KSTEPS steps per launch.  The moves are a function of the seed only, so the
            # stream's next steps are already on the device (uploaded ahead, below) unless this
            # is the first call; the generator hands its stream out in pieces that end at its
            # own block boundaries: they land side by side in `blk` and run as ONE launch
            want = min(self.KSTEPS, iterations - it)
            if mv["have"] - mv["used"] < want:
                need = want - (mv["have"] - mv["used"])
                if mv["have"] + need > self.MOVES_CAP:
                    self._moves_to_front()
                self._moves_append(moves, need, ahead=False)
            if mv["ahead"] is not None:  # the launch waits for the copy stream's last upload
                ctx.call("nh_stream_wait_marker", mv["ahead"])
                mv["ahead"] = None
            verify = self._verify and not self.shared
            if verify and len(self._rl_pending) >= 2:
                # launch n + 1 is queued only once launch n - 1 is known to have ended well (launch
                # n is running meanwhile: the device does not wait for this)
                gave_up = self._rl_settle(keep=1)
                if gave_up is not None:
                    it, moves = rollback(gave_up)
                    if jentry is not None:
                        jentry[0] = it
                    fast = False
                    break
            rec = dict(launch=self._rl_count + 1, it=it, iteration=s.iteration, steps_total=s.steps_total,
                       block_n=block["n"] if block is not None else None, nlc=s.n_lnprob_calls,
                       nwe=s.n_walker_evals, rl=self.resident_launches)
            if not self._run_resident(2 * mv["used"], 2 * want, block):
                fast = False  # (gave up: the per-launch loop below replays these steps)
                if self.fused:
                    per_launch_history()
                break
            if verify:
                rec["snap"] = self._last_snap
                self._rl_pending.append(rec)
            mv["used"] += want
            # a marker behind this launch: `blk` is a ring, and the host runs launches ahead of
            # the device -- an upload on the copy stream may overlap THIS launch (whose steps lie
            # below `have`), but must wait for every earlier one, which may still be reading the
            # very bytes it is about to overwrite once the ring has wrapped
            mv["prev"] = mv["last"]
            mv["last"] = self._launch_marks[self._nlaunch % 3]
            self._nlaunch += 1
            ctx.call("nh_marker_record", mv["last"])
            if mv["have"] - mv["used"] < self.KSTEPS and \
                    mv["have"] + self.KSTEPS <= self.MOVES_CAP:
                self._moves_append(moves, self.KSTEPS, ahead=True)
            it += want
            if jentry is not None:
                jentry[0] = it
            s.iteration += want
            s.steps_total += want
            if block is not None:
                block["n"] += want
            if it >= iterations and self._rl_pending:
                # the call's last launches: known to have ended well before the state is handed out
                gave_up = self._rl_settle(keep=0)
                if gave_up is not None:
                    it, moves = rollback(gave_up)
                    if jentry is not None:
                        jentry[0] = it
                    fast = False
                    break
            yield DeviceState(self, rng)
        if it < iterations:
            mv["prev"] = mv["last"] = None  # (the per-launch loop takes over: see _moves_append)
        while it < iterations:
            self._flush_pending()  # (merged sharded mode) the block's last accept
            self._sync_cur_blobs()  # (launches per half-step keep every rank's blobs whole)
            self._cur_host = None
            # ---- ship the moves of the next K steps: ONE asynchronous upload from the
            # generator's page-locked ring (filled ahead by its worker thread) ----------
            # nh_moves_take's contract: only the copy of the MOST RECENT take may still be
            # queued when the next one is taken (the generator hands a used-up block back
            # one block late, so that copy's source is still intact); every earlier upload
            # has to be complete -- wait on the marker recorded after it
            while len(self._inflight) >= 2:
                ctx.call("nh_marker_wait", self._inflight.pop(0))
            if mv["have"] > mv["used"]:
                # steps of the move stream the resident loop uploaded ahead and did not use: they
                # come first (to the front of `blk`, where this loop expects its block)
                K = min(mv["have"] - mv["used"], self.KSTEPS, iterations - it)
                self._moves_to_front(K)
            else:
                mv["have"] = mv["used"] = 0
                addr, K = moves.take(min(self.KSTEPS, iterations - it))
                ctx.call("nh_upload", self.blk, addr, 8 * 2 * K * 3 * ns)
            if self.fused:
                if self.mega and self._plan["hs"] is not None:
                    # (the one-launch kernel writes the cursor itself, every launch)
                    ctx.call("nh_half_step_begin_block", self._plan["hs"]["plan"], 0,
                             block["n"] if dev_hist else 0)
                    if self._plan.get("stage") is not None:  # (its other launch: Context._stage_a)
                        ctx.call("nh_half_step_begin_block", self._plan["stage"]["plan"], 0, 0)
                else:
                    ctx.call("nh_memset", self.cursor, 0xFF, 4)  # -1: nothing accepted yet
                    if not self.mega:
                        self._front()  # slice 0 of the new block
            else:
                ctx.call("nh_memset", self.cursor, 0, 4)
            mark = self._markers[self._nmark % len(self._markers)]
            self._nmark += 1
            mark_pending = True  # recorded behind the block's first launch: the host gets
            self._inflight.append(mark)  # that one out first (any later point is as good)
            k = 0
            while k < K:
                self._pos["slice"], self._pos["steps"] = 2 * k, (block["n"] - k) if dev_hist else 0
                # several steps per graph launch when nothing has to happen on the host
                # between them (history is kept by the kernel, nobody reads the states)
                g = 1
                gmax = min(self.GSTEPS, K - k) if self.mega else self.GSTEPS
                resident = False
                if (self._resident_ok() and yield_every >= K - k and (block is None or dev_hist)
                        and not (block is not None and block["blobs"] and not blob_dev_hist)):
                    # the rest of this block of moves in ONE launch of resident workgroups
                    g = K - k
                    if k > 0 and dev_hist:
                        # the per-launch kernel leaves the history row of a closed step to the
                        # NEXT launch of the block: there is none, the resident loop takes over
                        ctx.call("nh_hist_append", self.coords, self.logp, N, self.ndim,
                                 self.histd, block["n"] - 1)
                        if blob_dev_hist:
                            ctx.call("nh_half_step_append_blobs", self._plan["hs"]["plan"],
                                     block["n"] - 1)
                    self._flush_pending()
                    if not self._run_resident(2 * k, 2 * g, block if dev_hist else None):
                        continue  # (gave up: _resident_ok() is false from now on)
                    resident = True
                elif (self.step_graph is not None and self.fused and K - k >= gmax >= 2 and
                        yield_every >= gmax and (block is None or dev_hist) and
                        not (block is not None and block["blobs"] and not blob_dev_hist)):
                    g = gmax
                    # one launch per half-step: the slices are baked into the graph (the
                    # proposal's chain of dependent reads is one trip shorter), so there is
                    # one graph per (starting step of the block of moves, number of steps):
                    # the tail of a block shorter than GSTEPS is a graph of its own
                    key = (k, g) if self.mega else 0
                    if key not in self.multi_graphs:
                        self.multi_graphs[key] = self.multi_graph = self._capture_steps(k, g)
                    ctx.graph_launch(self.multi_graphs[key])
                elif (self.split and self.graph2 is not None and self.fused and
                        yield_every >= 2 and (block is None or dev_hist) and
                        not (block is not None and block["blobs"])):
                    # sharded, nobody looks at the intermediate states: the accept of a
                    # half-step and the evaluation of the next are ONE graph, with the
                    # all-gather between two such graphs (two host calls per half-step
                    # instead of three)
                    self._run_half_step_merged()
                    self._run_half_step_merged()
                else:
                    self._flush_pending()
                    self._run_step()
                if mark_pending:
                    ctx.call("nh_marker_record", mark)
                    mark_pending = False
                k += g
                it += g
                if jentry is not None:
                    jentry[0] = it
                s.iteration += g
                s.steps_total += g
                if block is not None:
                    kk = block["n"]
                    if not dev_hist:
                        ctx.call("nh_copy", block["coords"].ptr + 8 * kk * N * self.ndim,
                                 self.coords, 8 * N * self.ndim)
                        ctx.call("nh_copy", block["logp"].ptr + 8 * kk * N, self.logp, 8 * N)
                    if not blob_dev_hist:
                        for hb, (cur, m, _, _) in zip(block["blobs"], self.cur_blobs or []):
                            ctx.call("nh_copy", hb.ptr + 8 * kk * N * m, cur, 8 * N * m)
                    block["n"] = kk + g
                if yield_every < 2:
                    self._flush_pending()
                if self.mega and dev_hist and not resident and (k >= K or yield_every <= iterations):
                    # one launch per half-step: the row of a closed step is written by the
                    # NEXT launch of the same block of moves; nothing follows the last one of
                    # a block, and whoever is handed this state may read the chain (nobody
                    # does between the graphs of one run_mcmc call: yield_every > iterations)
                    self._flush_pending()
                    ctx.call("nh_hist_append", self.coords, self.logp, N, self.ndim, self.histd,
                             block["n"] - 1)
                    if blob_dev_hist:
                        ctx.call("nh_half_step_append_blobs", self._plan["hs"]["plan"],
                                 block["n"] - 1)
                yield DeviceState(self, rng)
        self._flush_pending()

    # ------------------------------------------------------------- the move stream on the device
    def _moves_append(self, moves, n, ahead):
        """the next n steps of the move stream (fewer if the generator's piece ends earlier and
        `ahead`) behind the ones `blk` holds; ahead: on the copy stream, beside the running
        launch (nh_upload_ahead), else on the main stream"""
        ctx, mv, sb = self.ctx, self._mv, 8 * 2 * 3 * self.ns
        done = 0
        while done < n:
            # nh_moves_take's contract: only the copy of the MOST RECENT take may still be
            # queued when the next one is taken
            while len(self._inflight) >= 2:
                ctx.call("nh_marker_wait", self._inflight.pop(0))
            addr, got = moves.take(n - done)
            mark = self._markers[self._nmark % len(self._markers)]
            self._nmark += 1
            dst = self.blk.ptr + sb * mv["have"]
            if ahead:
                # (the first ahead-upload of a loop waits for the launch just queued: whatever
                # ran before it -- the per-launch loop's graphs read `blk` too -- is then done)
                ctx.call("nh_upload_ahead", dst, addr, sb * got, mark,
                         mv["prev"] if mv["prev"] is not None else mv["last"])
                mv["ahead"] = mark
            else:
                ctx.call("nh_upload", dst, addr, sb * got)
                ctx.call("nh_marker_record", mark)
            self._inflight.append(mark)
            mv["have"] += got
            done += got

    def _moves_to_front(self, k=None):
        """the unused steps `blk` holds (the first k of them) move to its front, in stream order
        behind whatever still reads the block; with k the loop that follows consumes them"""
        ctx, mv, sb = self.ctx, self._mv, 8 * 2 * 3 * self.ns
        avail = mv["have"] - mv["used"]
        n = avail if k is None else k
        if mv["ahead"] is not None:
            ctx.call("nh_stream_wait_marker", mv["ahead"])
            mv["ahead"] = None
        if n > 0 and mv["used"] > 0:
            if self._blk_tmp is None:
                self._blk_tmp = ctx.empty((2 * self.MOVES_CAP, 3 * self.ns))
            ctx.call("nh_copy", self._blk_tmp, self.blk.ptr + sb * mv["used"], sb * n)
            ctx.call("nh_copy", self.blk, self._blk_tmp, sb * n)
        if k is None:
            mv["have"], mv["used"] = avail, 0
        else:
            mv["used"] += k  # (what is left stays where it is, behind the part now at the front)
            if mv["used"] == mv["have"]:
                mv["have"] = mv["used"] = 0

    # ------------------------------------------------------------- resident loop
    def _resident_ok(self):
        """can the rest of a block of moves run as ONE launch (nh_half_step_run)?  Needs the
        one-launch plan, a single rank, blobs (if kept) kept by the launch; the library has
        the last word (LDS, occupancy, the plan's shape)"""
        if self._run is False or not self.mega or not self.s.use_graph or self._plan.get("staged"):
            return False
        if self.sharded and (self.s.comm.size < 2 or getattr(self.s.comm, "group", None) is None
                             or os.environ.get("NAIMA_AMD_SHARED", "1") == "0"):
            return False
        hs = self._plan["hs"] if self._plan else None
        if hs is None:
            return False
        if self.s.store_blobs and self.cur_blobs and not self.blobs_in_kernel:
            return False
        if self._run is None and self.sharded:
            self._run = self._create_shared_run(hs) or False
            self.shared = bool(self._run)
            return self.shared
        if self._run is None:
            h = _lib._dp()
            if _lib._lib.nh_half_step_run_create(self.ctx.h, hs["plan"], C.byref(h)) != 0:
                self._run = False
                self.resident_reason = _lib._lib.nh_last_error().decode()
                return False
            self._run = h
            self._rl_count, self._rl_pending = 0, []  # (the library numbers a loop's launches from 1)
            self._res["runs"].append(h)
            self.ctx.sorted_tables(hs, h)
            g, t, l = _lib._i(), _lib._i(), _lib._ll()
            _lib._chk(_lib._lib.nh_half_step_run_info(h, C.byref(g), C.byref(t), C.byref(l)))
            self.resident_info = dict(grid=g.value, threads=t.value, lds_bytes=l.value,
                                      **self._syn_info(h))
        return True

    @staticmethod
    def _syn_info(h):
        mode, m, pcs = _lib._i(), _lib._i(), _lib._i()
        _lib._chk(_lib._lib.nh_half_step_run_syn_info(h, C.byref(mode), C.byref(m), C.byref(pcs)))
        inreg, nmax = _lib._i(), _lib._i()
        _lib._chk(_lib._lib.nh_half_step_run_table_info(h, C.byref(inreg), C.byref(nmax)))
        spl, rows = _lib._i(), _lib._i()
        _lib._chk(_lib._lib.nh_half_step_run_split_info(h, C.byref(spl), C.byref(rows)))
        deep = _lib._i()
        _lib._chk(_lib._lib.nh_half_step_run_pipeline_info(h, C.byref(deep)))
        return dict(syn_log_domain=bool(mode.value), syn_nodes_per_piece=m.value, syn_pieces=pcs.value,
                    tables_in_registers=bool(inreg.value), workgroups_per_walker=spl.value,
                    rows_split=bool(rows.value), two_walkers_in_flight=bool(deep.value))

    def _create_shared_run(self, hs):
        """the resident loop over an ensemble shared with the other ranks' GPUs: rings in
        fine-grained memory, mapped by every other rank (hipIpc handles carried by the control
        plane), and a probe launch that exchanges tagged granules with every peer -- every step
        agreed by all ranks, so that all take the shared loop or none does"""
        lib, ctx, comm = _lib._lib, self.ctx, self.s.comm
        g = comm.group
        h = _lib._dp()
        why = ""

        def agreed(ok):
            return g.reduce_scalar(1.0 if ok else 0.0, "min") == 1.0

        def give_up(msg):
            if h.value:
                lib.nh_half_step_run_destroy(ctx.h, h)
            self.resident_reason = msg
            if comm.rank == 0:
                import warnings
                warnings.warn("the resident loop over a shared ensemble is not available (%s): "
                              "one launch and one all-gather per half-step instead" % msg)
            return None

        ok = lib.nh_half_step_run_create_shared(ctx.h, hs["plan"], comm.rank, comm.size,
                                                C.byref(h)) == 0
        if not ok:
            why = lib.nh_last_error().decode()
        if not agreed(ok):
            return give_up("create: " + why)
        handle = C.create_string_buffer(64)
        ok = lib.nh_half_step_run_export(ctx.h, h, handle) == 0
        parts = g.allgather_bytes(handle.raw if ok else b"")
        ok = all(len(p_) == 64 for p_ in parts)
        for r, p_ in enumerate(parts):
            if ok and r != comm.rank:
                ok = lib.nh_half_step_run_attach(ctx.h, h, r, p_) == 0
        if not ok:
            why = lib.nh_last_error().decode()
        if not agreed(ok):
            return give_up("hipIpc: " + why)
        ctx.sync()
        g.barrier()
        st, us = _lib._i(), C.c_double()
        ok = lib.nh_half_step_run_probe(ctx.h, h, 64, C.byref(st), C.byref(us)) == 0 and \
            st.value == 0
        why = "" if ok else lib.nh_last_error().decode()
        if "ring" in os.environ.get("NAIMA_AMD_LADDER_REFUSE", "").split(","):
            # (tests: this rung of the exchange ladder refused although the probe went well -- the
            # ranks then take the next one, as on a node whose GPUs cannot store into each other)
            ok, why = False, "refused on request (NAIMA_AMD_LADDER_REFUSE=ring: fault injection of the tests)"
        if not agreed(ok):
            return give_up("a record stored by another GPU did not reach a running kernel; %s" % why)
        self._res["runs"].append(h)
        ctx.sorted_tables(hs, h)
        gr, t, l = _lib._i(), _lib._i(), _lib._ll()
        _lib._chk(lib.nh_half_step_run_info(h, C.byref(gr), C.byref(t), C.byref(l)))
        self.resident_info = dict(grid=gr.value, threads=t.value, lds_bytes=l.value,
                                  **self._syn_info(h))
        self.shared_info = dict(ranks=comm.size, probe_us_per_exchange=us.value)
        return h

    def device_barrier(self):
        """ranks that share an ensemble meet on the device: one exchange of tagged granules
        with every peer (the probe launch), a few microseconds apart instead of the control
        plane's ~0.1 ms.  Synchronises the stream; a no-op for any other loop."""
        if self.shared:
            st = _lib._i()
            _lib._chk(_lib._lib.nh_half_step_run_probe(self.ctx.h, self._run, 1, C.byref(st), None))
            if st.value != 0:
                raise _lib.NaimaHipError("device barrier of the shared ensemble timed out")

    def _sync_cur_blobs(self):
        """a shared ensemble: every rank's current-blob arrays := the blobs of where each walker
        IS -- held by the rank that accepted its last move (largest stamp).  Collective."""
        if not self._cur_dirty:
            return
        self._cur_dirty = False
        ctx, comm = self.ctx, self.s.comm
        g, N = comm.group, self.N
        stamps = np.empty(N, dtype=np.int32)
        _lib._chk(_lib._lib.nh_half_step_run_counters(ctx.h, self._run, None,
                                                      stamps.ctypes.data, 2))
        allst = np.array([np.frombuffer(p_, dtype=np.int32) for p_ in
                          g.allgather_bytes(stamps.tobytes())])
        owner, has = allst.argmax(axis=0), allst.max(axis=0) >= 0
        mine = has & (owner == comm.rank)
        stash = []
        for cur, m, _, _ in self.cur_blobs or []:
            host = cur.get().reshape(N, m)
            # (in slabs of walkers: a gathered message is every rank's rows together, and the
            # control plane refuses messages beyond 64 MiB)
            step = max(1, (16 << 20) // (8 * m))
            idx = np.arange(N)
            for lo in range(0, N, step):
                slab = (idx >= lo) & (idx < lo + step)
                parts = g.allgather_bytes(np.ascontiguousarray(host[mine & slab]).tobytes())
                for r, p_ in enumerate(parts):
                    if r != comm.rank:
                        host[has & (owner == r) & slab] = np.frombuffer(p_, dtype=float).reshape(-1, m)
            cur.set(host.ravel())
            stash.append(host)
        self._cur_host = stash  # (valid until a launch changes the blobs again)

    def _merge_shared_block(self, block, n, c, l, per):
        """the rows a shared ensemble's launches wrote, gathered from the ranks that moved each
        walker (c [n][N][ndim], l [n][N], per[b] [n][N][m_b]: this rank's copies, completed in
        place), blob rows of rejected moves filled from the row before.  Collective."""
        comm = self.s.comm
        g, N, ndim = comm.group, self.N, self.ndim
        own = block["own"].get()[:n].reshape(n, N)
        valid = np.zeros(n, dtype=bool)  # (rows of launches per half-step are whole on every rank)
        for r0, r1 in block["shared_rows"]:
            valid[r0:r1] = True
        own[~valid] = -1
        ms = [p_.shape[2] for p_ in per]
        moved, acc = own >= 0, own > 0
        rows_per = max(1, (16 << 20) // (N * (8 * (ndim + 1 + sum(ms)) + 1)))
        for t0 in range(0, n, rows_per):
            t1 = min(n, t0 + rows_per)
            mv, ac = moved[t0:t1], acc[t0:t1]
            flags = own[t0:t1].astype(np.int8).tobytes()
            flags += b"\0" * (-len(flags) % 8)
            payload = flags + c[t0:t1][mv].tobytes() + l[t0:t1][mv].tobytes() + \
                b"".join(p_[t0:t1][ac].tobytes() for p_ in per)
            for r, blob in enumerate(g.allgather_bytes(payload)):
                if r == comm.rank:
                    continue
                nel = (t1 - t0) * N
                o = np.frombuffer(blob, dtype=np.int8, count=nel).reshape(t1 - t0, N)
                off = nel + (-nel % 8)
                mv_r, ac_r = o >= 0, o > 0
                cnt, cnta = int(mv_r.sum()), int(ac_r.sum())
                c[t0:t1][mv_r] = np.frombuffer(blob, dtype=float, count=cnt * ndim,
                                               offset=off).reshape(cnt, ndim)
                off += 8 * cnt * ndim
                l[t0:t1][mv_r] = np.frombuffer(blob, dtype=float, count=cnt, offset=off)
                off += 8 * cnt
                for p_, m in zip(per, ms):
                    p_[t0:t1][ac_r] = np.frombuffer(blob, dtype=float, count=cnta * m,
                                                    offset=off).reshape(cnta, m)
                    off += 8 * cnta * m
                moved[t0:t1] |= mv_r
                acc[t0:t1] |= ac_r
        if per:
            cur0 = [(b_ if isinstance(b_, np.ndarray) else b_.get()).reshape(N, m)
                    for b_, m in zip(block["cur0"], ms)]
            for t in range(n):
                rej = moved[t] & ~acc[t]
                if rej.any():
                    for p_, c0 in zip(per, cur0):
                        p_[t][rej] = (p_[t - 1] if t > 0 else c0)[rej]

    def _run_resident(self, slice0, nslices, block):
        ctx, hs = self.ctx, self._plan["hs"]
        if self.shared:
            self._acc_dirty = True
            own = block.get("own") if block is not None else None
            _lib._chk(_lib._lib.nh_half_step_run_hist_flags(self._run, own.ptr if own is not None
                                                            else None))
            if self.s.store_blobs and self.cur_blobs:
                self._cur_dirty = True
                self._cur_host = None
            if own is not None:
                block["shared_rows"].append((block["n"], block["n"] + nslices // 2))
        hc = hl = None
        hb, row0, cap = None, 0, 0
        if block is not None:
            hc, hl = block["coords"].ptr, block["logp"].ptr
            row0, cap = block["n"], block["coords"].shape[0]
            if block["blobs"]:
                hb = (C.c_void_p * 4)(*([b.ptr for b in block["blobs"]] + [None] * 4)[:4])
        probation = not self.shared and self._resident_trust < 2
        if probation:
            # The first launches of a loop are checked before anything builds on them: a launch
            # whose workgroups cannot all be resident (another process on the GPU, a profiler that
            # serialises workgroups) gives up on its first wait -- it leaves the ensemble, the
            # counters and the move stream as they were (k_run_epilogue does nothing then), so the
            # same block of moves is replayed by the per-launch kernel and the run carries on
            # with one launch per half-step.  (A shared ensemble: the ranks agree in
            # _create_shared_run / bench.py's rehearsal and raise together on a later time-out; a
            # one-GPU loop's later launches are checked a launch behind: sample, _rl_settle.)
            n0, f0 = self._read_counts(reset=False)
        self._last_snap = None
        if block is None and not self.shared and self.s.store_blobs and self.cur_blobs:
            self._last_snap = self._snapshot_cur_blobs()
        ctx.call("nh_half_step_run", hs["plan"], self._run, slice0, nslices, hc, hl, hb, row0, cap)
        self._rl_count += 1
        if probation:
            st = _lib._i()
            _lib._chk(_lib._lib.nh_half_step_run_status(ctx.h, self._run, C.byref(st)))
            if st.value != 0:
                import warnings
                self.resident_reason = ("a launch of the resident loop gave up waiting for a walker's "
                                        "record (status %d): its workgroups were not all resident"
                                        % st.value)
                warnings.warn(self.resident_reason + "; the block of moves is replayed and the run "
                              "continues with one launch per half-step")
                self._run = False
                self.resident_failed_launches += 1
                self._read_counts(reset=False, set_to=(n0, f0))  # (the replay counts them again)
                self._restore_cur_blobs(self._last_snap)  # (slices that finished before it gave up)
                return False
            self._resident_trust += 1
        self.resident_launches += 1
        self.s.n_lnprob_calls += nslices
        self.s.n_walker_evals += nslices * self.nloc
        return True

    def _snapshot_cur_blobs(self):
        """device copies of the current blobs, in stream order ahead of the launch about to be
        queued (one of four sets: at most three launches are ever unsettled) -> the set's index"""
        k = self._rl_count % len(self._cur_snaps)
        if self._cur_snaps[k] is None:
            self._cur_snaps[k] = [self.ctx.empty((self.N, m)) for _, m, _, _ in self.cur_blobs]
        for dst, (cur, m, _, _) in zip(self._cur_snaps[k], self.cur_blobs):
            self.ctx.call("nh_copy", dst, cur, 8 * self.N * m)
        return k

    def _restore_cur_blobs(self, k):
        if k is None or self._cur_snaps[k] is None:
            return
        for src, (cur, m, _, _) in zip(self._cur_snaps[k], self.cur_blobs):
            self.ctx.call("nh_copy", cur, src, 8 * self.N * m)
        self._cur_host = None

    def _settle_before_read(self):
        """somebody reads the state (or drains the counters) while launches of the resident loop
        are still unsettled -- a caller iterating sample() launch by launch: wait for them; one that
        gave up is replayed when the iteration goes on, and until then the state on the device is
        the one BEFORE it"""
        if self._rl_pending and self._run:
            rec = self._rl_settle(keep=0)
            if rec is not None:
                raise _lib.NaimaHipError(
                    "a launch of the resident loop gave up waiting for a walker's record (status %d); "
                    "its block of moves is replayed when the iteration continues -- read the state "
                    "after that" % rec["status"])

    # ------------------------------------------------ a shared ensemble: kept state and replay
    def _shared_snapshot(self):
        """device copies of the ensemble (positions, log-probabilities, acceptance counts, current
        blobs) and the host's books, at a point where every rank is known to be well"""
        ctx, s = self.ctx, self.s
        self._flush_pending()
        snap = dict(coords=ctx.empty((self.N * self.ndim,)), logp=ctx.empty((self.N,)),
                    nacc=ctx.empty((self.N,), dtype=np.int32), blobs=[],
                    iteration=s.iteration, steps_total=s.steps_total, nlc=s.n_lnprob_calls,
                    nwe=s.n_walker_evals, rl=self.resident_launches,
                    nan_pending=self._nan_pending, forbidden_pending=self._forbidden_pending,
                    # (chain rows of earlier calls that nobody has asked for yet: they stay)
                    hist_len=len(self.hist))
        ctx.call("nh_copy", snap["coords"], self.coords, 8 * self.N * self.ndim)
        ctx.call("nh_copy", snap["logp"], self.logp, 8 * self.N)
        ctx.call("nh_copy", snap["nacc"], self.nacc, 4 * self.N)
        if self.s.store_blobs and self.cur_blobs:
            self._sync_cur_blobs()  # (every rank's current blobs whole before they are kept)
            for cur, m, _, _ in self.cur_blobs:
                b = ctx.empty((self.N, m))
                ctx.call("nh_copy", b, cur, 8 * self.N * m)
                snap["blobs"].append(b)
        self._sh_snap = snap
        self._sh_journal = []

    def _shared_verified(self):
        """every rank's launches since the kept state ended well: the ensemble as it stands is the
        one to go back to from now on.  It is KEPT at the end of the flush / reset this was found in
        (keep_verified) -- after the NaN / forbidden counts have been drained, the acceptance
        counters zeroed and the sampler's iteration reset -- so that a replay starts from the
        books as they stand behind that point, not before it (kept here, a reset()'s
        snapshot held the burn-in's acceptance counts and iteration)."""
        self._sh_snap = None
        self._sh_journal = []
        self._sh_want = bool(self.shared and self._have_state)

    def keep_verified(self):
        """the last thing a flush does, and a reset (the sampler's, once ITS books are cleared):
        keep the ensemble every rank was found well at (here, where every rank is anyway, and
        not inside the next call: a 20-step call is a millisecond)"""
        if getattr(self, "_sh_want", False):
            self._sh_want = False
            if self.shared and self._have_state:
                self._shared_snapshot()

    def _replay_shared(self, bad, mine):
        """a launch of the shared loop gave up on some rank (status `bad`; this rank's own: `mine`).
        Every rank is here (the status was reduced over all of them): back to the kept ensemble,
        the move stream made again up to its step, and the journal's calls once more -- one launch
        and one all-gather per half-step from here on."""
        import warnings
        from ._lib import Moves
        ctx, s, snap = self.ctx, self.s, self._sh_snap
        journal = list(self._sh_journal)
        ctx.sync()
        self.resident_reason = ("a launch of the resident loop over the shared ensemble gave up waiting "
                                "for a record (status %d%s)" % (bad, "" if bad == mine else ", on another rank"))
        if s.comm.rank == 0:
            warnings.warn(self.resident_reason + "; every rank goes back %d step(s) to the last ensemble "
                          "all of them held and carries on with one launch and one all-gather per "
                          "half-step" % (s.steps_total - snap["steps_total"]))
        self.resident_failed_launches += max(1, self.resident_launches - snap["rl"])
        # what the shared launches accepted up to the kept state stays counted (flush folded it)
        self._acc_base = self._acc_base + self._acc_shared
        self._acc_shared, self._acc_dirty = 0.0, False
        self._run, self.shared = False, False
        self._cur_dirty, self._cur_host = False, None
        self._sh_snap, self._sh_journal = None, []
        ctx.call("nh_copy", self.coords, snap["coords"], 8 * self.N * self.ndim)
        ctx.call("nh_copy", self.logp, snap["logp"], 8 * self.N)
        ctx.call("nh_copy", self.nacc, snap["nacc"], 4 * self.N)
        for b, (cur, m, _, _) in zip(snap["blobs"], self.cur_blobs or []):
            ctx.call("nh_copy", cur, b, 8 * self.N * m)
        self.hist = self.hist[:snap.get("hist_len", 0)]  # (what predates the kept ensemble stays)
        s.iteration, s.steps_total = snap["iteration"], snap["steps_total"]
        s.n_lnprob_calls, s.n_walker_evals = snap["nlc"], snap["nwe"]
        self.resident_launches = snap["rl"]
        self._read_counts(reset=False, set_to=(0, 0))  # (drained at the kept state; the replay counts again)
        self._nan_pending, self._forbidden_pending = snap["nan_pending"], snap["forbidden_pending"]
        while self._inflight:
            ctx.call("nh_marker_wait", self._inflight.pop(0))
        mv = self._mv
        mv["have"] = mv["used"] = 0
        mv["ahead"] = mv["prev"] = mv["last"] = None
        old = s._moves
        s._moves = Moves(s.seed, s.nwalkers, s.a, ksteps=32, depth=4, pinned=True)
        if old is not None:
            old.close()
        left = snap["steps_total"]
        while left > 0:
            _, got = s._moves.take(min(32, left))
            left -= got
        self._replaying = True
        try:
            for made, store, yield_every in journal:
                if made <= 0:
                    continue
                for _ in self.sample(DeviceState(self, s._rng), made, store, yield_every=1 << 30):
                    pass
            self._flush_pending()
            ctx.sync()
        finally:
            self._replaying = False

    def _rl_settle(self, keep):
        """wait until all but the last `keep` queued launches are known to have ended; None, or
        the record of the first one that gave up (it and everything queued behind it are void)"""
        while len(self._rl_pending) > keep:
            rec = self._rl_pending[0]
            done, st, before = _lib._i(0), _lib._i(0), (C.c_int * 2)()
            _lib._chk(_lib._lib.nh_half_step_run_report(self.ctx.h, self._run, rec["launch"], 1,
                                                        C.byref(done), C.byref(st), None, None, before))
            if st.value != 0:
                rec["status"], rec["counts"] = st.value, (before[0], before[1])
                return rec
            self._rl_pending.pop(0)
        return None

    def _read_counts(self, reset, set_to=None):
        """(NaN log-probabilities, proposals forbidden by the prior) the one-launch kernels have
        counted on the device since the last reset"""
        hs = self._plan["hs"] if self._plan else None
        extra = 0
        if set_to is None:
            # launches of the separate kernels (cfg4; the first half-steps of any run): their
            # accepts count into the context's word
            n = _lib._i(0)
            _lib._chk(_lib._lib.nh_nan_count(self.ctx.h, 1 if reset else 0, C.byref(n)))
            extra = n.value
        if hs is None or hs.get("plan") is None:
            return extra, 0
        n, f = _lib._i(0), _lib._i(0)
        if set_to is not None:
            n, f = _lib._i(-set_to[0] - 1), _lib._i(-set_to[1] - 1)
        _lib._chk(_lib._lib.nh_half_step_counts(self.ctx.h, hs["plan"], 1 if reset else 0,
                                                C.byref(n), C.byref(f)))
        return n.value + (extra if reset else 0), f.value

    def check_nan(self, collective=False):
        """NaN log-probabilities the launches met since the last look (one-launch plans count
        them on the device): counted, or raised as emcee does, by the sampler's nan_policy.
        With several ranks the counts are per rank (a proposal is counted by the rank that moved
        the walker): where every rank is known to be here (``collective``: flush, reset, the merge
        of the current blobs) they are summed over the ranks first, so that every rank counts --
        and raises -- the same; elsewhere a rank only accumulates what it has seen, because a
        ValueError on one rank would leave the others waiting in their next collective."""
        self._settle_before_read()  # (the counters are not drained under an unsettled launch)
        n, f = self._read_counts(reset=True)
        self._nan_pending += n
        self._forbidden_pending += f
        multi = self.s.comm is not None and self.s.comm.size > 1
        if multi and not collective:
            return
        n, f = self._nan_pending, self._forbidden_pending
        self._nan_pending = self._forbidden_pending = 0
        if multi:
            import struct
            parts = self.s.comm.group.allgather_bytes(struct.pack("<qq", n, f))
            n = sum(struct.unpack("<qq", p_)[0] for p_ in parts)
            f = sum(struct.unpack("<qq", p_)[1] for p_ in parts)
        self.s.prior_forbidden_proposals += f
        if n:
            self.s.nan_proposals += n
            if self.s.nan_policy == "raise":
                raise ValueError("Probability function returned NaN (%d proposals of the device "
                                 "loop; emcee stops at the first one -- pass nan_policy='reject' "
                                 "to treat them as rejected proposals)" % n)

    def resident_status(self):
        """this rank's status word of the resident loop (0: every launch found its records);
        synchronises the stream, raises nothing -- for callers that reduce it over the ranks
        themselves (bench.py's rehearsal)"""
        st = _lib._i(0)
        if self._run:
            _lib._chk(_lib._lib.nh_half_step_run_status(self.ctx.h, self._run, C.byref(st)))
        return st.value

    def check_resident(self, collective=False):
        """raise if a launch of the resident loop gave up waiting for a walker's record (its
        workgroups were not all resident: another process on the GPU, a profiler that
        serialises workgroups); the ensemble is undefined from that launch on.  (A one-GPU
        loop's launches are checked as they are made, or a launch behind, and replayed by the
        per-launch kernel instead: _run_resident, sample -- what is left to find here is a
        shared ensemble's, or a loop run with NAIMA_AMD_VERIFY_LAUNCHES=0.)  ``collective``: every rank is here -- the worst status
        of all ranks decides, and all of them raise."""
        if self.shared and not collective and self.s.comm is not None and self.s.comm.size > 1:
            # a shared loop's status is acted on where EVERY rank is (flush, reset): an exception on
            # the one rank that reads its state here would leave the others waiting in their next
            # collective -- the launch that gave up is found there and replayed by all
            return
        st = _lib._i(0)
        if self._run:
            _lib._chk(_lib._lib.nh_half_step_run_status(self.ctx.h, self._run, C.byref(st)))
        bad = st.value
        if collective and self.s.comm is not None and self.s.comm.size > 1:
            bad = int(self.s.comm.group.reduce_scalar(float(bad), "max"))
            if bad == 0:
                self._shared_verified()
            elif self.shared and self._sh_snap is not None and not self._replaying:
                self._replay_shared(bad, st.value)
                return
        if bad != 0:
            raise _lib.NaimaHipError(
                "the resident half-step loop timed out waiting for a walker's record (status "
                "%d%s): its workgroups were not all resident%s.  Run with NAIMA_AMD_%s=0"
                % (bad, "" if bad == st.value else ", on another rank",
                   ", or a rank of the shared ensemble fell behind or failed"
                   if self.shared else "", "SHARED" if self.shared else "RESIDENT"))

    def _run_half_step_merged(self):
        ctx = self.ctx
        if self._pending:
            if self.graph21 is None:
                self.graph21 = self._capture(lambda: (self._part_accept(), self._part_evaluate()))
            ctx.graph_launch(self.graph21)  # accept + next proposal + evaluation
        else:
            ctx.graph_launch(self.graph)    # evaluation (the proposal is already there)
        self._exchange()
        self._pending = True

    def _flush_pending(self):
        if self._pending:
            self.ctx.graph_launch(self.graph2)
            self._pending = False

    def _capture_steps(self, k0, nsteps):
        """hipGraph of ``nsteps`` ensemble steps starting at step k0 of a block of moves"""
        def run():
            for i in range(2 * nsteps):
                self._pos["slice"] = 2 * k0 + i
                self._half_step_body()

        self._pos["bake"] = bool(self.mega)
        try:
            return self._capture(run)
        finally:
            self._pos["bake"] = False

    def _capture(self, fn):
        ctx = self.ctx
        ctx.pin_caches()
        ctx.sync()
        ctx.graph_begin()
        try:
            fn()
        except Exception:
            ctx.graph_abort()
            raise
        g = ctx.graph_end()
        self._res["graphs"].append(g)
        return g

    def _run_step(self):
        """one ensemble step = two half-steps.  Single GPU: after the eager warm-up and
        the per-half-step capture check, BOTH half-steps are one graph (the cursor
        advances inside it), i.e. one host call per step."""
        s, ctx = self.s, self.ctx
        if self.split or not s.use_graph:
            self._run_half_step()
            self._run_half_step()
            return
        if self.step_graph is not None:
            ctx.graph_launch(self.step_graph)
            return
        if self.warm < 1:
            # the first evaluation settles which grids share a weights launch, the
            # second one is recorded
            self._half_step_body()
            self._record_half_step()
            self.warm += 2
            return
        if self.mega and self._plan["hs"] is None:
            # the first one-launch half-step creates the kernel's descriptor (device
            # allocation + upload): not inside a stream capture
            self._first_one_launch_half_step()
            self._half_step_body()
            return

        def two():
            self._half_step_body()
            self._half_step_body()

        self.step_graph = self.graph = self._capture(two)
        ctx.graph_launch(self.step_graph)

    def _first_one_launch_half_step(self):
        """the half-step that creates the one-launch plan.  _can_be_one_launch is an estimate
        made from the recorded launches; nh_half_step_create has the last word (exact LDS
        layout, its admission rules).  If it turns the plan down, nothing has been launched
        yet for this half-step: the loop drops to the three-launch fused sequence for good."""
        try:
            self._half_step_body()
        except _lib.NaimaHipError:
            if self._plan.get("hs") is not None:
                raise  # the plan exists: this was a real failure of a launch
            import warnings
            warnings.warn("nh_half_step_create turned the one-launch plan down (%s); the device "
                          "loop keeps the three-launch half-step"
                          % _lib._lib.nh_last_error().decode())
            self.mega = False
            self._plan["mega"] = False
            self._plan["staged"] = False
            self.blobs_in_kernel = False
            self._hook.pop("blobs", None)
            self._hook.pop("blobs_in_kernel", None)
            self._front()  # proposal + packs + weights of this slice, as a launch of its own
            self._half_step_body()

    def _run_half_step(self):
        """eager warm-up (fills the static caches), then capture, then replay.  One
        graph on a single GPU; with walkers sharded over ranks the collective sits
        between two graphs (evaluate | all-gather | accept)."""
        s, ctx = self.s, self.ctx
        multi = self.split
        if self.graph is not None:
            ctx.graph_launch(self.graph)
            if multi:
                self._exchange()
                ctx.graph_launch(self.graph2)
            return
        if self.warm < 2:
            if self.warm == 0:
                self._half_step_body()
            else:
                self._record_half_step()
            self.warm += 1
            return
        if self.mega and self._plan["hs"] is None:
            self._first_one_launch_half_step()  # (never inside a capture)
            return
        if not s.use_graph:
            self._half_step_body()
            return
        if not multi:
            self.graph = self._capture(self._half_step_body)
            ctx.graph_launch(self.graph)
            return
        self.graph = self._capture(self._part_evaluate)
        ctx.graph_launch(self.graph)
        self._exchange()
        self.graph2 = self._capture(self._part_accept)
        ctx.graph_launch(self.graph2)

    def flush(self):
        """bring the pending chain history and acceptance counters to the host"""
        if self._flushed and not self.hist and not self._pending:
            # nothing has run since the last flush: the books are on the host already.  (With
            # several ranks a flush is collective; the ranks run the same program, so this early
            # return is taken by all of them or none -- and a rank that reads its results AGAIN,
            # alone, after the others have left -- rank 0 writing the run to disk, as the
            # reference's save_run does in its one process, analysis.py:366-471 -- is not left
            # waiting for peers that are gone.)
            return
        self._flush_pending()
        self.check_resident(collective=True)
        self.check_nan(collective=True)
        self.ctx.check_general()  # (a per-walker grid longer than the general kernel's LDS)
        self._sync_cur_blobs()
        s = self.s
        for block in self.hist:
            n = block["n"]
            if n == 0:
                continue
            c = block["coords"].get()[:n].reshape(n, self.N, self.ndim)
            l = block["logp"].get()[:n].reshape(n, self.N)
            per = [hb.get()[:n].reshape(n, self.N, m)
                   for hb, (cur, m, _, trail) in zip(block["blobs"], self.cur_blobs or [])]
            if self.shared and block.get("own") is not None:
                self._merge_shared_block(block, n, c, l, per)
            s._chain.extend(list(c))
            s._logp.extend(list(l))
            if block["blobs"]:
                per = [a.reshape((n, self.N) + trail)
                       for a, (cur, m, _, trail) in zip(per, self.cur_blobs)]
                for j, v in self.const_blobs:
                    per.insert(j, np.full((n, self.N), v))
                if s._blobs is None:
                    s._blobs = [[] for _ in per]
                for j, a in enumerate(per):
                    s._blobs[j].extend(list(a))
        self.hist = []
        s.naccepted = self.nacc.get().astype(float) + self._acc_base
        if self.shared:  # + the moves each rank's shared launches accepted
            if self._acc_dirty:  # (collective only when launches have run since the last look)
                self._acc_dirty = False
                own = np.empty(self.N, dtype=np.int32)
                _lib._chk(_lib._lib.nh_half_step_run_counters(self.ctx.h, self._run,
                                                              own.ctypes.data, None, 0))
                self._acc_shared = np.zeros(self.N)
                for p_ in s.comm.group.allgather_bytes(own.tobytes()):
                    self._acc_shared += np.frombuffer(p_, dtype=np.int32)
            s.naccepted += self._acc_shared
        self.keep_verified()
        self._flushed = True
