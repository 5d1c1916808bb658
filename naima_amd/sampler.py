"""The ensemble driver: an affine-invariant stretch-move sampler with the surface
naima reads from ``emcee.EnsembleSampler`` (core.py:127-160, 450-493, 529-530 of the
reference), but calling the log-probability ONCE per half-ensemble with all
proposed walkers, and sharding those walkers over the GPUs of a node.

emcee itself is third-party (``emcee>=3.0``, not in the reference tree, not
installed here): the move below restates its published algorithm
(``RedBlueMove.propose`` + ``StretchMove.get_proposal``) -- random split of the
ensemble into two halves; for the active half S with complement C:
z = ((a-1)U+1)^2/a, partner j ~ randint(|C|), q = C_j - (C_j - S) z,
accept if ln U' < (ndim-1) ln z + lnp(q) - lnp(S).  Parity with emcee is
statistical, not bitwise ("parity unpinned").
"""
import time

import os

import numpy as np

from . import units as u
from .dist import LocalComm, shard_bounds, shard_counts

__all__ = ["EnsembleSampler", "State", "get_sampler", "run_sampler"]


class State:
    """what ``sampler.sample`` yields (emcee.State): coords, log_prob, blobs"""

    def __init__(self, coords, log_prob=None, blobs=None, random_state=None):
        if isinstance(coords, State):
            self.coords, self.log_prob = coords.coords.copy(), coords.log_prob
            self.blobs, self.random_state = coords.blobs, coords.random_state
            return
        self.coords = np.atleast_2d(np.array(coords, dtype=float))
        self.log_prob = log_prob
        self.blobs = blobs
        self.random_state = random_state

    def __iter__(self):  # emcee allows  pos, lnp, rstate = state
        return iter((self.coords, self.log_prob, self.random_state))


def _split_blob(b):
    """a blob returned by the model -> (ndarray with leading walker axis, unit or None)"""
    if isinstance(b, u.Quantity):
        return np.asarray(b.value, dtype=float), b.unit
    return np.asarray(b, dtype=float), None


class EnsembleSampler:
    """Stretch-move ensemble sampler over a *batched* log-probability.

    log_prob_fn(coords[n, ndim], *args) -> lnp[n]  or  (lnp[n], blob0[n,...], ...)
    With ``naima_style=True`` (what ``get_sampler`` uses) the function is naima's
    ``lnprob(pars, data, model, prior)`` and is called with ``coords.T`` so that
    ``pars[i]`` is a vector over walkers.
    """

    def __init__(self, nwalkers, ndim, log_prob_fn, args=(), a=2.0, seed=None, comm=None,
                 naima_style=False, store_blobs=True, device=False, use_graph=True,
                 nan_policy="raise"):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("need an even number of walkers, at least twice the dimension")
        self.nwalkers, self.ndim, self.a = int(nwalkers), int(ndim), float(a)
        self.log_prob_fn, self.args = log_prob_fn, tuple(args)
        self.comm = comm if comm is not None else LocalComm()
        # the streams are replicated on every rank: proposals/accepts are identical.
        # _rng: numpy, for the initial ball; _moves: the C++ stretch-move stream
        self.seed = int(seed if seed is not None else 12345)
        self._rng = np.random.default_rng(self.seed)
        self._moves = None
        self.naima_style = naima_style
        self.store_blobs = store_blobs
        # a proposal whose log-probability is NaN: "raise" is emcee's behaviour (ValueError
        # "Probability function returned NaN", EnsembleSampler.compute_log_prob) -- the host loop
        # raises on the spot, the device loop (whose launches cannot) when the run's results
        # next reach the host; "reject" treats the proposal as one that is never accepted and
        # counts it in ``nan_proposals``
        if nan_policy not in ("raise", "reject"):
            raise ValueError("nan_policy must be 'raise' or 'reject'")
        self.nan_policy = nan_policy
        self.nan_proposals = 0
        # proposals the prior forbade (-inf): the device loop evaluates none of their integrals
        # (the reference evaluates the model and discards it, core.py:103-119); counted for the
        # bench line, which credits them as walker-steps like every other proposal
        self.prior_forbidden_proposals = 0
        self.steps_total = 0  # ensemble steps since the sampler was made (reset() keeps it)
        # device=True: ensemble, proposals, log-probabilities and blobs live in HBM; the
        # launch sequence of one half-step (propose -> model -> likelihood -> accept)
        # is captured into a hipGraph and replayed (needs naima_style log_prob_fn)
        self.device = bool(device)
        self.use_graph = bool(use_graph)
        self._device_ok = None
        # accept + next proposal + parameter packs as ONE launch (nh_move_cycle)
        self.fuse_moves = os.environ.get("NAIMA_AMD_FUSE_MOVES", "1") != "0"
        self._dev = None
        self.n_lnprob_calls = 0
        self.n_walker_evals = 0
        self.reset()

    def moves(self, pinned=False):
        """the stretch-move random stream (one per sampler, created on first use)"""
        if self._moves is None:
            from ._lib import Moves
            self._moves = Moves(self.seed, self.nwalkers, self.a, ksteps=32, depth=4,
                                pinned=pinned)
        return self._moves

    # ------------------------------------------------------------------ store
    def reset(self):
        if getattr(self, "_dev", None) is not None:
            self._dev.reset()
        self.iteration = 0
        self._chain, self._logp, self._blobs = [], [], None
        self.blob_units = None
        self.naccepted = np.zeros(self.nwalkers)
        if getattr(self, "_dev", None) is not None:
            # several ranks: the ensemble all of them were found well at is kept NOW, behind the
            # cleared books (a replay after the reference's burn-in -> reset -> run flow,
            # core.py:483-487, 529-530, must not bring the burn-in's iteration count and
            # acceptance counters back)
            self._dev.keep_verified()

    @property
    def acceptance_fraction(self):
        self._flush()
        return self.naccepted / max(1, self.iteration)

    def _flush(self):
        if self._dev is not None:
            self._dev.flush()

    def get_chain(self, flat=False, discard=0, thin=1):
        self._flush()
        c = np.array(self._chain).reshape(-1, self.nwalkers, self.ndim)[discard::thin]
        return c.reshape(-1, self.ndim) if flat else c

    def get_log_prob(self, flat=False, discard=0, thin=1):
        self._flush()
        c = np.array(self._logp).reshape(-1, self.nwalkers)[discard::thin]
        return c.reshape(-1) if flat else c

    def get_blobs(self, flat=False, discard=0, thin=1):
        """list (one entry per blob) of arrays [nsteps, nwalkers, ...]; dense, not
        emcee's object array"""
        self._flush()
        if self._blobs is None:
            return None
        out = []
        for per_step in self._blobs:
            a = np.array(per_step)[discard::thin]
            out.append(a.reshape((-1,) + a.shape[2:]) if flat else a)
        return out

    # legacy emcee-2 names that naima's analysis code touches
    @property
    def chain(self):
        return np.swapaxes(self.get_chain(), 0, 1)

    @property
    def flatchain(self):
        return self.get_chain(flat=True)

    @property
    def lnprobability(self):
        return self.get_log_prob().T

    # --------------------------------------------------------------- evaluate
    def compute_log_prob(self, coords):
        """lnprob of ``coords[n, ndim]``: this rank evaluates its block, the blocks
        meet in one all-gather.  Returns (lnp[n], [local blob arrays], (lo, hi))."""
        n = coords.shape[0]
        lo, hi = shard_bounds(n, self.comm.rank, self.comm.size)
        mine = coords[lo:hi]
        if self.naima_style:
            res = self.log_prob_fn(mine.T, *self.args)
        else:
            res = self.log_prob_fn(mine, *self.args)
        self.n_lnprob_calls += 1
        self.n_walker_evals += hi - lo
        if isinstance(res, tuple):
            lnp_local, blobs = np.asarray(res[0], dtype=float), list(res[1:])
        else:
            lnp_local, blobs = np.asarray(res, dtype=float), []
        lnp_local = np.broadcast_to(lnp_local, (hi - lo,)).astype(float)
        if np.any(np.isnan(lnp_local)):
            if self.nan_policy == "raise":
                raise ValueError("Probability function returned NaN")
            self.nan_proposals += int(np.isnan(lnp_local).sum())
            lnp_local = np.where(np.isnan(lnp_local), -np.inf, lnp_local)
        if self.comm.size > 1:
            m = max(shard_counts(n, self.comm.size))
            pad = np.full((m,), -np.inf)
            pad[:hi - lo] = lnp_local
            allp = self.comm.allgather(pad).reshape(self.comm.size, m)
            lnp = np.concatenate([allp[r, :c] for r, c in
                                  enumerate(shard_counts(n, self.comm.size))])
        else:
            lnp = lnp_local
        return lnp, blobs, (lo, hi)

    def _blob_arrays(self, blobs, nloc):
        arrs, units = [], []
        for b in blobs:
            v, un = _split_blob(b)
            if v.ndim == 0 or v.shape[0] != nloc:
                v = np.broadcast_to(v, (nloc,) + v.shape).copy()
            arrs.append(v)
            units.append(un)
        return arrs, units

    # ------------------------------------------------------------------ sample
    def sample(self, initial_state, iterations=1, store=True, log_prob0=None, yield_every=1):
        """emcee's generator: one State per ensemble step.  ``yield_every`` > 1 (what
        ``run_mcmc`` asks for) lets the device loop replay several steps as one hipGraph
        and yield only after each such group."""
        if self.device:
            if self._dev is None and self._device_ok is None:
                self._device_ok = self._probe_device(initial_state)
            if self._device_ok is not False:
                yield from self._sample_device(initial_state, iterations, store, yield_every)
                return
            # the model shapes a grid / table per walker (Eemin, a seed temperature ... as
            # fit parameters): the general path needs the values on the host
        state = State(initial_state)
        coords = state.coords.copy()
        if coords.shape != (self.nwalkers, self.ndim):
            raise ValueError("incompatible input dimensions")
        rng = self._rng
        N, ndim, a = self.nwalkers, self.ndim, self.a
        keep_blobs = self.store_blobs
        # every rank keeps the blobs of ALL walkers: the rank that evaluates a walker
        # changes from half-step to half-step (a random split, contiguous shards of it), so
        # the proposals' blobs follow their log-probabilities through one more all-gather
        # per blob (as the device loop does; this host-driven loop is the general /
        # fallback path and pays for it with the control plane's bandwidth)
        if state.log_prob is None:
            logp, blobs, (lo, hi) = self.compute_log_prob(coords)
            cur, units = self._blob_arrays(blobs, hi - lo) if keep_blobs else ([], [])
            self.blob_units = units
            self._cur_blobs = [np.array(self._gather_rows(b, N)) for b in cur]
        else:
            logp = np.array(state.log_prob, dtype=float)
            if not hasattr(self, "_cur_blobs"):
                self._cur_blobs = []
        logp = logp.copy()
        moves = self.moves()
        for _ in range(int(iterations)):
            addr, got = moves.take(1)
            Sm, Pm, Zm, Lm = moves.view(addr, got)
            for split in range(2):
                S, zz = Sm[0, split], Zm[0, split]
                s, cp = coords[S], coords[Pm[0, split]]
                factors = (ndim - 1.0) * np.log(zz)
                q = cp - (cp - s) * zz[:, None]
                newlp, blobs, (lo, hi) = self.compute_log_prob(q)
                lnpdiff = factors + newlp - logp[S]
                accepted = Lm[0, split] < lnpdiff
                acc_idx = S[accepted]
                coords[acc_idx] = q[accepted]
                logp[acc_idx] = newlp[accepted]
                self.naccepted[acc_idx] += 1
                if keep_blobs and blobs:
                    new, _ = self._blob_arrays(blobs, hi - lo)
                    for cur, nb in zip(self._cur_blobs, new):
                        cur[acc_idx] = self._gather_rows(nb, len(S))[accepted]
            self.iteration += 1
            self.steps_total += 1
            if store:
                self._chain.append(coords.copy())
                self._logp.append(logp.copy())
                if keep_blobs and self._cur_blobs:
                    if self._blobs is None:
                        self._blobs = [[] for _ in self._cur_blobs]
                    for j, b in enumerate(self._cur_blobs):
                        self._blobs[j].append(b.copy())
            yield State(coords, logp, self._cur_blobs if keep_blobs else None, rng)

    # ------------------------------------------------------------ device mode
    def _probe_device(self, initial_state):
        """one evaluation of two walkers on lazy device parameters: does the model run
        with its parameters in HBM?  Models that shape a particle grid or an emission
        table per walker do not (NotImplementedError from the radiative classes); they
        are sampled by the host-driven loop, which evaluates such walkers one by one."""
        import warnings

        from . import _lib
        from .darray import DPars
        if not self.naima_style:
            return True
        ctx = _lib.get_context()
        c = np.ascontiguousarray(State(initial_state).coords[:2].T, dtype=float)
        why = None
        try:
            res = self.log_prob_fn(DPars(ctx, ctx.array(c), self.ndim, c.shape[1]), *self.args)
            if self.store_blobs:
                from . import units as u
                from .darray import DMat, DVec
                for b in res[1:]:
                    v = b.value if isinstance(b, u.Quantity) else b
                    if not isinstance(v, (DMat, DVec, float, int)):
                        why = "a blob of type %s cannot be kept in HBM" % type(v).__name__
        except NotImplementedError as e:  # grid-/table-shaping parameters per walker
            why = str(e)
        except (TypeError, ValueError) as e:
            # a model that is not built from naima_amd's radiative classes: plain numpy
            # arithmetic on the parameters (ValueError from DVec), a functional model that
            # returns a host array (TypeError from the device likelihood) ...
            why = "%s: %s" % (type(e).__name__, e)
        finally:
            ctx.flush()
        if why is not None:
            warnings.warn("device=True is not possible for this model (%s); using the "
                          "host-driven loop (pass device=False to silence this)" % (why,))
            self.device = False
            return False
        return True

    def _sample_device(self, initial_state, iterations, store, yield_every=1):
        from .device_sampler import DeviceLoop
        if self._dev is None:
            self._dev = DeviceLoop(self)
        yield from self._dev.sample(initial_state, iterations, store, yield_every)

    def _gather_rows(self, rows, n):
        """this rank's block of an n-row array -> all n rows, on every rank"""
        size = self.comm.size
        if size == 1:
            return rows
        counts = shard_counts(n, size)
        pad = np.zeros((max(counts),) + rows.shape[1:])
        pad[:len(rows)] = rows
        allp = self.comm.allgather(pad).reshape((size, max(counts)) + rows.shape[1:])
        return np.concatenate([allp[r, :c] for r, c in enumerate(counts)], axis=0)

    def run_mcmc(self, initial_state, nsteps, **kw):
        state = None
        kw.setdefault("yield_every", 1 << 30)  # nobody looks at the intermediate states
        for state in self.sample(initial_state, iterations=nsteps, **kw):
            pass
        return state


# --------------------------------------------------------------------------
# naima's entry points (core.py:220-538)
# --------------------------------------------------------------------------
def _run_mcmc(sampler, pos, nrun, verbose=True):
    """core.py:127-160: the run with a progress printout every 5 %.  The steps between two
    printouts are one ``run_mcmc`` call, so the device loop replays whole groups of steps
    and the ensemble is only brought to the host for the printouts."""
    state = pos
    nrun = int(nrun)
    edges = sorted(set(int(round(x)) for x in np.linspace(0, nrun, 21)))
    for a, b in zip(edges[:-1], edges[1:]):
        if verbose and sampler.comm.rank == 0 and a > 0:
            _print_progress(sampler, state, a, nrun)
        state = sampler.run_mcmc(state, b - a, store=True)
    return sampler, state


def _print_progress(sampler, state, i, nrun):
    print("\nProgress of the run: {0:.0f} percent ({1} of {2} steps)".format(
        int(100.0 * i / nrun), i, nrun))
    coords = np.asarray(state.coords)
    npars = coords.shape[-1]
    print("                           " + (" ".join(
        ["{%i:-^15}" % k for k in range(npars)])).format(*sampler.labels))
    print("  Last ensemble median : " + (" ".join(
        ["{%i:^15.3g}" % k for k in range(npars)])).format(*np.median(coords, axis=0)))
    print("  Last ensemble std    : " + (" ".join(
        ["{%i:^15.3g}" % k for k in range(npars)])).format(*np.std(coords, axis=0)))
    lp = np.asarray(state.log_prob)
    print("  Last ensemble lnprob :  avg: {0:.3f}, max: {1:.3f}".format(np.average(lp), np.max(lp)))


def _prefit(p0, data, model, prior):
    """Nelder-Mead maximum-likelihood prefit (core.py:163-217).  Same algorithm, options
    (maxfev 500, relative xtol 0.1, ftol 1e-3) and acceptance rules as the reference; the
    candidate points of each simplex iteration are evaluated as one walker batch
    (naima_amd.neldermead)."""
    from .core import lnprob
    from .neldermead import minimize_batched
    P0_IS_ML = False

    def flat_prior(*args):
        return 0.0

    if prior is None:
        prior = flat_prior

    def nll(X):  # (m, ndim) points -> m values of -lnprob under the flat prior
        X = np.atleast_2d(np.asarray(X, dtype=float))
        return -np.asarray(lnprob(np.ascontiguousarray(X.T), data, model, flat_prior)[0],
                           dtype=float).reshape(-1)

    res = minimize_batched(nll, p0, maxfev=500, xtol=1e-1, ftol=1e-3)
    ll_prior = float(np.asarray(lnprob(res["x"], data, model, prior)[0]))
    if (res["success"] or res["status"] == 1) and not np.isinf(ll_prior):
        # also kept when maxfev was reached: likely better than p0 (core.py:193-195)
        P0_IS_ML = res["status"] != 1
        p0 = res["x"]
    return p0, P0_IS_ML


def get_sampler(data_table=None, p0=None, model=None, prior=None, nwalkers=500, nburn=100,
                guess=True, interactive=False, prefit=False, labels=None, threads=None,
                data_sed=None, seed=None, comm=None, verbose=True, store_blobs=True,
                device=True):
    """Generate a new MCMC sampler (signature of core.py:220-233; ``threads`` is
    accepted and ignored -- the walkers of a half-ensemble are one GPU batch;
    ``interactive`` is out of scope).  ``device=True`` (default): the ensemble and the step
    loop live on the GPU (models that cannot keep their parameters in HBM fall back to the
    host-driven loop with a warning).  Returns (sampler, state)."""
    from .core import lnprob, sed_conversion
    from .datatable import validate_data_table
    if data_table is None:
        raise TypeError("Data table is missing!")
    data = validate_data_table(data_table, sed=data_sed)
    if model is None:
        raise TypeError("Model function is missing!")
    p0 = np.array(p0, dtype=float)
    if labels is None:
        labels = ["norm"] + ["par{0}".format(i) for i in range(1, len(p0))]
    elif len(labels) < len(p0):
        labels = list(labels) + ["par{0}".format(i) for i in range(len(labels), len(p0))]

    modelout = model(p0, data)
    spec = modelout[0] if isinstance(modelout, (tuple, list)) else modelout
    try:  # core.py:352-376: model and data must be convertible to differential flux
        sed_conversion(data["energy"], spec.unit, False)
        sed_conversion(data["energy"], data["flux"].unit, False)
    except u.UnitsError:
        raise u.UnitsError(
            "The physical type of the model and data units are not compatible, please modify "
            "your model or data so they match:\n Model units: {0} [{1}]\n Data units: {2} [{3}]\n"
            .format(spec.unit, spec.unit.physical_type, data["flux"].unit,
                    data["flux"].unit.physical_type))

    if guess:  # core.py:378-419
        normNames = ["norm", "ampl", "we", "wp"]
        normNames += ["log({0}".format(n) for n in normNames[:4]] + \
                     ["log10({0}".format(n) for n in normNames[:4]]
        idxs = []
        for nn in normNames:
            for l2 in labels:
                if l2.lower().startswith(nn):
                    idxs.append(labels.index(l2))
        if len(idxs) == 1:
            e = data["energy"]
            nunit, sedf = sed_conversion(e, spec.unit, False)
            currFlux = np.trapezoid(e.value * (spec * sedf).to(nunit).value, e.value)
            nunit, sedf = sed_conversion(e, data["flux"].unit, False)
            dataFlux = np.trapezoid(e.value * (data["flux"] * sedf).to(nunit).value, e.value)
            ratio = dataFlux / currFlux
            if labels[idxs[0]].startswith("log("):
                p0[idxs[0]] += np.log(ratio)
            elif labels[idxs[0]].startswith("log10("):
                p0[idxs[0]] += np.log10(ratio)
            else:
                p0[idxs[0]] *= ratio

    P0_IS_ML = False
    if prefit:
        p0, P0_IS_ML = _prefit(p0, data, model, prior)

    sampler = EnsembleSampler(nwalkers, len(p0), lnprob, args=[data, model, prior], seed=seed,
                              comm=comm, naima_style=True, store_blobs=store_blobs,
                              device=device)
    sampler.data_table = data_table
    sampler.data = data
    sampler.labels = labels
    sampler.modelfn = model
    sampler.run_info = {"n_walkers": nwalkers, "n_burn": nburn,
                        "p0": [float(p) for p in p0], "guess": guess}
    # ball of 0.5 % (ML start) or 10 % around p0 (core.py:477-481), drawn from the
    # sampler's replicated stream so that every rank starts from the same ensemble
    spread = 0.005 if P0_IS_ML else 0.1
    p0var = spread * p0
    pos = p0 + p0var * sampler._rng.normal(size=(nwalkers, len(p0)))
    if nburn > 0:
        if verbose and sampler.comm.rank == 0:
            print("Burning in the {0} walkers with {1} steps...".format(nwalkers, nburn))
        sampler, state = _run_mcmc(sampler, pos, nburn, verbose)
    else:
        state = State(pos)
    sampler.run_info["p0_burn_median"] = [float(p) for p in np.median(state.coords, axis=0)]
    return sampler, state


def run_sampler(nrun=100, sampler=None, pos=None, verbose=True, **kwargs):
    """Run an MCMC sampler (core.py:496-538)."""
    if sampler is None or pos is None:
        sampler, pos = get_sampler(verbose=verbose, **kwargs)
    sampler.run_info["n_run"] = nrun
    if verbose and sampler.comm.rank == 0:
        print("\nWalker burn in finished, running {0} steps...".format(nrun))
    sampler.reset()
    t0 = time.time()
    if isinstance(pos, State):
        pos = State(pos.coords)
    elif not hasattr(pos, "_loop"):  # (a DeviceState continues from the ensemble in HBM)
        pos = State(pos)
    sampler, pos = _run_mcmc(sampler, pos, nrun, verbose)
    sampler.run_info["wall_s"] = time.time() - t0
    return sampler, pos
