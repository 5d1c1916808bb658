"""Likelihood, priors and the sampler entry points with naima's signatures
(core.py of the reference), batched over walkers and evaluated on the GPU.

``lnprob(pars, data, modelfunc, priorfunc)`` accepts ``pars[ndim]`` (one walker,
the reference's contract, core.py:97-121) or ``pars[ndim, N]`` -- every
``pars[i]`` is then a vector over walkers, the model function runs ONCE and
returns ``(N, n_E)`` fluxes, and the Gaussian/upper-limit log-likelihood of
core.py:64-94 is computed for all walkers by the ``nh_lnprobmodel`` kernel.
"""
import numpy as np

from . import units as u
from ._lib import get_context
from .darray import DMat, DPars, DVec, LazyPrior

__all__ = ["normal_prior", "uniform_prior", "log_uniform_prior", "lnprob", "lnprobmodel",
           "get_sampler", "run_sampler"]


# ---- priors (core.py:34-58), valid for scalars and for vectors over walkers ----
def uniform_prior(value, umin, umax):
    """Uniform prior distribution: 0 inside [umin, umax], -inf outside."""
    if isinstance(value, DVec):
        return LazyPrior(value.ctx, value.n, [(0, value, umin, umax)])
    value = np.asarray(value, dtype=float)
    out = np.where((umin <= value) & (value <= umax), 0.0, -np.inf)
    return float(out) if out.ndim == 0 else out


def normal_prior(value, mean, sigma):
    """Normal prior distribution (as the reference writes it, core.py:42-44)."""
    if isinstance(value, DVec):
        return LazyPrior(value.ctx, value.n, [(1, value, mean, sigma)])
    return -0.5 * (2 * np.pi * sigma) - (value - mean) ** 2 / (2.0 * sigma)


def log_uniform_prior(value, umin=0, umax=None):
    """Log-uniform prior distribution (returns 1/value as the reference, core.py:47-58)."""
    if isinstance(value, DVec):
        return LazyPrior(value.ctx, value.n,
                         [(2, value, umin, np.inf if umax is None else umax)])
    value = np.asarray(value, dtype=float)
    ok = (value > 0) & (value >= umin)
    if umax is not None:
        ok &= value <= umax
    with np.errstate(divide="ignore"):
        out = np.where(ok, 1.0 / value, -np.inf)
    return float(out) if out.ndim == 0 else out


# ---- SED <-> differential conversion (utils.py:219-282) ------------------------
_SED_TYPES = ("power", "flux", "energy")
_DIFF_TYPES = ("differential flux", "differential power", "differential energy",
               "differential number density")


def sed_conversion(energy, model_unit, sed):
    """(f_unit, sedf): the unit and per-energy factor that bring a model to SED
    (``sed=True``) or differential (``sed=False``) form."""
    pt = u.Unit(model_unit).physical_type
    is_integral = pt in _SED_TYPES
    is_differential = pt in _DIFF_TYPES
    ones = np.ones(np.shape(energy.value))
    if (sed and is_integral) or (not sed and is_differential):
        sedf = ones
    elif sed and is_differential:
        sedf = energy ** 2
    elif not sed and is_integral:
        sedf = 1 / (energy ** 2)
    else:
        raise u.UnitsError("Model physical type ({0}) is not supported".format(pt),
                           "Supported physical types are: power, flux, differential power, "
                           "differential flux")
    energy_like = pt in ("energy", "differential energy")
    particle_like = pt in ("flux", "differential flux")
    if sed:
        f_unit = u.erg if energy_like else (u.Unit("erg/(s cm2)") if particle_like
                                            else u.Unit("erg/s"))
    else:
        f_unit = u.Unit("1/TeV") if energy_like else (u.Unit("1/(s TeV cm2)") if particle_like
                                                      else u.Unit("1/(s TeV)"))
    return f_unit, sedf


def _conversion_to_data(model_unit, data):
    """per-energy factor taking model values (in model_unit) to data['flux'].unit"""
    dunit = data["flux"].unit
    model_is_sed = u.Unit(model_unit).physical_type in ["power", "flux"]
    data_is_sed = dunit.physical_type in ["power", "flux"]
    energy = data["energy"]
    n = np.size(energy.value)
    if model_is_sed != data_is_sed:
        _, sedf = sed_conversion(energy, model_unit, data_is_sed)
        f = (u.Quantity(np.ones(n), model_unit) * sedf).to(dunit).value
    else:
        f = np.full(n, u.Unit(model_unit)._factor_to(dunit))
    return np.ascontiguousarray(f, dtype=float)


def _data_on_device(ctx, data):
    """the _DataOnDevice of a table, created once per (table, context)"""
    cache = getattr(data, "_nh_dev", None)
    if cache is not None and cache[0] is ctx:
        return cache[1]
    dd = _DataOnDevice(ctx, data)
    try:
        data._nh_dev = (ctx, dd)
    except AttributeError:  # a plain dict: no place to remember it
        pass
    return dd


class _DataOnDevice:
    """the data columns the likelihood needs, resident in HBM (cached per table)"""

    def __init__(self, ctx, data):
        self._conv = {}
        self._data = data
        dunit = data["flux"].unit
        self.flux = ctx.const(data["flux"].value)
        self.elo = ctx.const(data["flux_error_lo"].to(dunit).value)
        self.ehi = ctx.const(data["flux_error_hi"].to(dunit).value)
        self.ul = ctx.const(np.asarray(data["ul"]).astype(np.int32), dtype=np.int32)
        cl = np.broadcast_to(np.asarray(data["cl"], dtype=float), np.shape(data["flux"].value))
        # cl is indexed by the violation count (core.py:92): pad so that index n_E is valid
        self.cl = ctx.const(np.concatenate([cl, cl[-1:]]))
        self.n = int(np.size(data["flux"].value))
        self.ctx = ctx

    def conv(self, model_unit, colfac=None):
        """device copy of the per-energy factor model unit -> data unit"""
        key = (model_unit.dims, float(model_unit.scale),
               None if colfac is None else hash(colfac.tobytes()))
        hit = self._conv.get(key)
        if hit is None:
            f = _conversion_to_data(model_unit, self._data)
            if colfac is not None:
                f = f * colfac
            hit = self.ctx.const(f)
            self._conv[key] = hit
        return hit


def lnprobmodel(model, data, lp=None, blobs=()):
    """Log-likelihood of ``model`` (Quantity, shape (n_E,) or (N, n_E)) given the data
    table: asymmetric Gaussian errors plus the upper-limit penalty (core.py:64-94).
    A device-resident model (``DMat``) gives a device-resident result."""
    ctx = get_context()
    dd = _data_on_device(ctx, data)
    if isinstance(model.value, DMat):
        m = model.value
        N, nE = m.shape
        if nE != dd.n:
            raise ValueError("model has %d energies, data table has %d" % (nE, dd.n))
        ctx.join()  # the emission components ran on side streams
        hook = ctx._accept_hook
        if hook is not None and (hook["N"] != N or hook["used"]):
            hook = None
        # the sharded step loop wants the result in its all-gather send buffer
        total = hook["total"] if hook is not None and hook.get("total") is not None \
            else ctx.empty((N,))
        if hook is not None and hook.get("total_rows") is not None:
            total = hook["total_rows"]  # rows { lnprob | blobs }: see Context.half_step
        lpd = terms = None
        nterms = 0
        if isinstance(lp, LazyPrior):
            terms, nterms = lp.packed()  # evaluated inside the likelihood kernel
            if terms is None:
                lpd = lp.evaluate()
            elif ctx._plan is not None and ctx._plan["mode"] == "record":
                # (a staged plan's first launch evaluates the prior too: a proposal it forbids
                # gets no synchrotron spectrum, hence no seed photons -- Context._stage_a)
                ctx._plan["prior_terms"] = (terms, nterms)
        elif lp is not None:
            lpd = lp.dense()
        args = (m.comps(), len(m.terms), N, nE, dd.conv(model.unit, m.colfac),
                dd.flux, dd.elo, dd.ehi, dd.ul, dd.cl, lpd.ptr if lpd is not None else None,
                terms, nterms, None, total)
        owners = [t[0] for t in m.terms]
        held = [j for j, o in enumerate(owners) if getattr(o, "pending", None) is not None]
        plan = ctx._plan
        if hook is not None and plan is not None and plan["mega"] and plan["mode"] == "replay":
            # ONE launch for the whole half-step: proposal, packs, weights, We/Wp, every
            # spectrum of the model, this likelihood, the priors and the accept
            ctx.half_step(hook, m.comps(), len(m.terms), nE, args[4], dd, lpd, terms, nterms, total,
                          blobs=blobs)
            hook["used"] = True
            del lpd
            return DVec(ctx, total, total.ptr, N, stride=int(hook.get("rows_active") or 1))
        if hook is not None:
            # device step loop: the stretch move's accept rides on this launch (single
            # rank; sharded, the accept has to wait for the all-gather: mv is None)
            import ctypes as C
            mv = C.addressof(hook["mv"]) if hook["mv"] is not None else None
            j = held[0] if len(held) == 1 else -1
            o = owners[j] if j >= 0 else None
            if o is not None and o.pending[0] == "nh_synchrotron" and m.terms[j][1] == o.ptr \
                    and m.terms[j][2] == nE and nE <= 64:
                # ... and both ride on the held-back synchrotron launch: its workgroups
                # finish their walker's likelihood themselves
                sa = o.pending[1]
                o.pending = None
                ctx._deferred = [a for a in ctx._deferred if a is not o]
                ctx.call("nh_synchrotron_lnprob", *sa, args[0], args[1], j, *args[4:13],
                         total, mv)
            elif o is not None and o.pending[0] == "nh_integrate_tables" \
                    and m.terms[j][1] == o.ptr and m.terms[j][2] == nE \
                    and o.pending[1][7] == nE:
                # ... or on a held-back table reduction that is the whole spectrum
                ia = o.pending[1][:12]  # without nsplit (one plane)
                o.pending = None
                ctx._deferred = [a for a in ctx._deferred if a is not o]
                ctx.call("nh_integrate_tables_lnprob", *ia, args[0], args[1], j, *args[4:13],
                         total, mv)
            else:
                ctx.flush(*owners)
                if mv is not None:
                    ctx.call("nh_lnprob_accept", *args, mv)
                else:
                    ctx.call("nh_lnprob", *args)
            hook["used"] = True
        else:
            ctx.flush(*owners)
            ctx.call("nh_lnprob", *args)
        del lpd
        return DVec(ctx, total, total.ptr, N)
    import ctypes as C
    m = np.asarray(model.value, dtype=float)
    batched = m.ndim == 2
    m2 = np.ascontiguousarray(m if batched else m[None, :])
    N, nE = m2.shape
    if nE != dd.n:
        raise ValueError("model has %d energies, data table has %d" % (nE, dd.n))
    md = ctx.array(m2)
    lnl = ctx.empty((N,))
    comps = (C.c_void_p * 1)(md.ptr)
    cscale = (C.c_double * 1)(1.0)
    ctx.call("nh_lnprobmodel", comps, cscale, 1, nE, N, nE, dd.conv(model.unit), dd.flux, dd.elo,
             dd.ehi, dd.ul, dd.cl, None, lnl)
    out = lnl.get()
    return out if batched else float(out[0])


def _lnprob_device(pars, data, modelfunc, priorfunc):
    """core.py:97-121 with the ensemble in HBM: nothing comes back to the host"""
    lp = None
    if priorfunc is not None:
        lp = priorfunc(pars)
        if not isinstance(lp, (LazyPrior, DVec)):
            lp = LazyPrior(pars.ctx, pars.n, [], float(lp))
    modelout = modelfunc(pars, data)
    if isinstance(modelout, (tuple, list)):
        model, blob = modelout[0], tuple(modelout)
    else:
        model, blob = modelout, (modelout, np.nan)  # core.py:108-110
    if not isinstance(model.value, DMat):
        raise TypeError("the model function returned a host array for device-resident "
                        "parameters; it must build its flux from naima_amd radiative models")
    total = lnprobmodel(model, data, lp=lp, blobs=blob)
    return (total, *blob)


def lnprob(pars, data, modelfunc, priorfunc):
    """(lnprob, *blobs) for one walker or for a batch (core.py:97-121)."""
    if isinstance(pars, DPars):
        return _lnprob_device(pars, data, modelfunc, priorfunc)
    pars = np.asarray(pars, dtype=float)
    if priorfunc is None:
        lnprob_priors = 0.0
    else:
        lnprob_priors = priorfunc(pars)
    modelout = modelfunc(pars, data)
    if isinstance(modelout, (tuple, list)):
        model = modelout[0]
        blob = tuple(modelout)
    else:
        model = modelout
        blob = (modelout, np.nan)
    lp = np.asarray(lnprob_priors, dtype=float)
    if np.all(np.isinf(lp)):
        total = lnprob_priors
    else:
        lnprob_model = lnprobmodel(model, data)
        # walkers forbidden by the prior keep the prior value (core.py:115-119)
        total = np.where(np.isinf(lp), lp, lnprob_model + lp)
        if total.ndim == 0:
            total = float(total)
    return (total, *blob)


def get_sampler(*args, **kwargs):
    from .sampler import get_sampler as _gs
    return _gs(*args, **kwargs)


def run_sampler(*args, **kwargs):
    from .sampler import run_sampler as _rs
    return _rs(*args, **kwargs)
