"""Radiative models with naima's class API (radiative.py of the reference), executed
by hand-written HIP kernels and vectorised over walkers.

    Synchrotron(pd, B=...)            radiative.py:239-342
    InverseCompton(pd, seed_photon_fields=[...])   radiative.py:370-791
    Bremsstrahlung(pd, n0=...)        radiative.py:794-989
    PionDecay(pd, nh=..., useLUT=..., hiEmodel=...) radiative.py:1099-1536

Same constructor arguments, same ``flux / sed / compute_We / set_We / We`` (``Wp``)
surface, same units in and out, same exceptions.  Differences, all additive:
the particle distribution's parameters and ``B`` may be 1-D arrays over walkers,
in which case every method returns a leading walker axis and runs ONE batched
launch sequence; the seed photon density of an array seed may be (N, n_s) (SSC).
Grids are generated on the host with the reference's exact expressions
(radiative.py:147-154, 1002-1009); everything else is in libnaima_hip.so.
There is no NumPy evaluation path.
"""
import logging
import os
from collections import OrderedDict

import numpy as np

from . import units as u
from . import _lib as _lib_mod
from ._lib import PD_KIND, PP_MODEL, get_context
from .darray import DMat, DVec
from .constants import (AR_CGS, ASTROPY_TO_ERG, ASTROPY_TO_GEV, C_CGS, ERG_TO_EV, MEC2_ERG,
                        MEC2_EV, M_P_GEV, T_TH_GEV, energy_ratio_to, mec2, mec2_unit)
from .models import _validate_ene
from .validator import (validate_array, validate_physical_type, validate_scalar,
                        validate_scalar_or_batch)

__all__ = ["Synchrotron", "InverseCompton", "PionDecay", "Bremsstrahlung"]

log = logging.getLogger("naima_amd.radiative")

_PER_EV = u.Unit("1/eV")
_SPEC_UNIT = u.Unit("1/(s eV)")


def _batch_of(x):
    v = x.value if isinstance(x, u.Quantity) else x
    return np.shape(v)[0] if np.ndim(v) > 0 else 1


def _is_batched(x):
    v = x.value if isinstance(x, u.Quantity) else x
    return np.ndim(v) > 0


def _is_dev(x):
    v = x.value if isinstance(x, u.Quantity) else x
    return getattr(v, "__array_priority__", 0) == 30000


def _per_walker(x):
    """True for a parameter given per walker: host vector or device value"""
    return _is_batched(x) or _is_dev(x)


def _as_dvec(ctx, rows, N):
    if isinstance(rows, DVec):
        return rows
    buf = ctx.array(np.broadcast_to(np.asarray(rows, dtype=float), (N,)))
    return DVec(ctx, buf, buf.ptr, N)


def _merge_batch(*sizes):
    n = 1
    for s in sizes:
        if s != 1:
            if n != 1 and s != n:
                raise ValueError("inconsistent walker-batch sizes: %d vs %d" % (n, s))
            n = s
    return n


class BaseRadiative:
    """flux/sed on top of a subclass ``_spectrum`` (radiative.py:61-134)."""

    def __init__(self, particle_distribution):
        self.particle_distribution = particle_distribution
        try:
            pd = self.particle_distribution.amplitude
            validate_physical_type("Particle distribution", pd,
                                   physical_type="differential energy")
        except (AttributeError, TypeError):
            pd = self.particle_distribution([0.1, 1, 10] * u.TeV)
            validate_physical_type("Particle distribution", pd,
                                   physical_type="differential energy")

    # -- batching -------------------------------------------------------------
    def _own_batch_sizes(self):
        return ()

    @property
    def batch_size(self):
        return _merge_batch(getattr(self.particle_distribution, "batch_size", 1),
                            *(self._own_batch_sizes() + self._structural_batch()))

    @property
    def is_batched(self):
        return bool(getattr(self.particle_distribution, "is_batched", False)) or any(
            s != 1 for s in self._own_batch_sizes() + self._structural_batch()) \
            or self._own_batched()

    def _own_batched(self):
        return False

    # -- the general path: parameters that shape the grids or the emission tables ----
    # Grid limits, grid densities, seed temperatures ... are walker-independent in the
    # batched launches (that is what lets tables be shared and cached).  When one of
    # them IS given per walker (a fit parameter), every walker is evaluated as a batch
    # of one through the same kernels: always correct, table caches do not help.
    _structural = ()

    def _structural_values(self):
        return [(n, getattr(self, n)) for n in self._structural if hasattr(self, n)]

    def _needs_walker_loop(self):
        for n, v in self._structural_values():
            if _is_dev(v):
                raise NotImplementedError(
                    "%s is a device-resident per-walker value: the particle grid / emission "
                    "tables would differ per walker.  Run the sampler with device=False (host "
                    "loop), which evaluates such walkers one by one" % n)
            if _is_batched(v):
                return True
        return False

    def _structural_batch(self):
        return tuple(_batch_of(v) for _, v in self._structural_values())

    def _walker(self, k):
        """this model for walker k alone (scalars everywhere)"""
        import copy
        one = copy.copy(self)
        one.__dict__ = dict(self.__dict__)
        for c in ("_cache", "_queue"):
            if c in one.__dict__:
                one.__dict__[c] = type(one.__dict__[c])()
        pd = copy.copy(self.particle_distribution)
        pd.__dict__ = {a: b for a, b in self.particle_distribution.__dict__.items()
                       if not a.startswith("_")}
        for name in getattr(pd, "param_names", ()):
            v = getattr(pd, name)
            if _is_batched(v):
                setattr(pd, name, v[k])
        one.particle_distribution = pd
        for name in tuple(self._structural) + tuple(self._walker_scalars):
            if hasattr(self, name):
                v = getattr(self, name)
                if _is_batched(v):
                    one.__dict__[name] = v[k]
        return one

    _walker_scalars = ()

    def _loop_walkers(self, what, *args, **kw):
        N = self.batch_size
        res = []
        for k in range(N):
            r = getattr(self._walker(k), what)
            res.append(r(*args, **kw) if callable(r) else r)
        unit = res[0].unit
        return u.Quantity(np.stack([np.asarray(r.to(unit).value, dtype=float) for r in res]), unit)

    def _finish(self, host, E):
        """(N, nE) ndarray -> Quantity with the reference's shape"""
        if E.isscalar:
            host = host[:, 0]
        if not self.is_batched:
            host = host[0]
        return host

    def _own_device_values(self):
        return ()

    @property
    def on_device(self):
        """True when a parameter is a device-resident value: results then stay in HBM
        as lazy ``DMat``/``DVec`` (no synchronising download)"""
        if getattr(self.particle_distribution, "on_device", False):
            return True
        for v in self._own_device_values():
            v = v.value if isinstance(v, u.Quantity) else v
            if getattr(v, "__array_priority__", 0) == 30000:
                return True
        return False

    def _result(self, ctx, out, N, nE, E, scale=1.0, rows=None):
        """spectra [N][nE] in a device buffer -> Quantity 1/(s eV) (lazy on device, or
        downloaded with the reference's shape).  ``rows``: per-walker factor (host
        vector or device scalar per walker) for physical parameters that the reference
        applies as one scalar and that may be fit parameters here (n0, nh, seed u)."""
        if self.on_device:
            m = DMat.from_buffer(ctx, out, N, nE, scale=scale)
            if rows is not None:
                m = m * _as_dvec(ctx, rows, N)
            return u.Quantity(m, _SPEC_UNIT)
        host = out.get()
        if scale != 1.0:
            host = host * scale
        if rows is not None:
            host = host * np.broadcast_to(np.asarray(rows, dtype=float), (N,))[:, None]
        return u.Quantity(self._finish(host, E), _SPEC_UNIT)

    def _weights(self, xg, e_eV, unit_scale):
        """device weights w = xg*n and log-ratios lw[i] = ln|w[i+1]/w[i]| of every walker.

        The components of one model evaluation share the particle distribution but use
        different grids (Synchrotron from 1 GeV, IC from Eemin, We above 1 TeV ...).
        The context remembers which grids were recently asked of a distribution kind; a
        fresh distribution evaluates all of them in ONE launch and later requests hit
        its cache."""
        pd = self.particle_distribution
        if getattr(pd, "kind", None) == "table":
            return self._weights_table(pd, xg, e_eV, unit_scale)
        if not hasattr(pd, "device_rows"):
            raise TypeError("naima_amd radiative models need a naima_amd.models particle "
                            "distribution (got %r)" % (type(pd).__name__,))
        ctx = get_context()
        N = self.batch_size
        nG = xg.size
        xd, ed = ctx.const(xg), ctx.const(e_eV)
        lx = ctx.grid_logratio(xd)
        key = (xd.ptr, ed.ptr, float(unit_scale), nG)
        cache = pd.__dict__.setdefault("_w_dev", {}).setdefault(N, {})
        reg = ctx._wgrids.setdefault((pd.kind, N), {})
        hit = cache.get(key)
        if hit is None and not cache and ctx._plan is not None:
            rows = pd.device_rows(ctx, N, amplitude_to=_PER_EV)
            rec = ctx.weights_replay(PD_KIND[pd.kind], rows, N)
            if rec is not None:  # produced by nh_step_front on the recorded grids
                for g, wl in zip(*rec):
                    cache[(g[1], g[0], g[4], g[5])] = wl
                hit = cache.get(key)
                if hit is None:
                    raise ValueError("the model asked for a particle grid it did not use when "
                                     "the step was recorded; run with use_graph=False")
        if hit is None:
            if not cache:
                ctx._weval += 1
            reg[key] = (ctx._weval, xd, ed, ctx.grid_ln(ed, e_eV), lx)
            # every grid used by the last two evaluations of this kind, this one first
            todo = [key] + [k for k, v in reg.items() if k != key and k not in cache
                            and ctx._weval - v[0] <= 2][:3]
            for k in [k for k, v in reg.items() if ctx._weval - v[0] > 2]:
                del reg[k]
            rows = pd.device_rows(ctx, N, amplitude_to=_PER_EV)
            bufs = ctx.weights_multi(PD_KIND[pd.kind], rows, N,
                                     [reg[k][1:3][::-1] + reg[k][3:5] + (k[2], k[3]) for k in todo])
            for k, wl in zip(todo, bufs):
                cache[k] = wl
            if self.on_device and ctx.side_small:
                # We/Wp reductions hang a side stream off this launch (ctx.branch_at)
                mark = ctx.anchor()
                for k in todo:
                    cache[k][0].anchor = mark
            hit = cache[key]
        else:
            reg[key] = (ctx._weval, xd, ed, ctx.grid_ln(ed, e_eV), lx)
        ctx.need(hit[0])
        return ctx, N, hit[0], hit[1], xd, lx

    def _weights_table(self, pd, xg, e_eV, unit_scale):
        """weights of a TableModel distribution: amplitude[w] times a walker-independent
        shape on the grid (host spline once, cached in HBM); rows by nh_lincomb"""
        import ctypes as C

        from .darray import lazy_const, nh_comp
        ctx = get_context()
        N = pd.batch_size
        nG = xg.size
        xd, ed = ctx.const(xg), ctx.const(e_eV)
        lx = ctx.grid_logratio(xd)
        cache = pd.__dict__.setdefault("_w_dev", {})
        key = (xd.ptr, ed.ptr, float(unit_scale), N)
        hit = cache.get(key)
        if hit is None:
            per_eV = u.Quantity(1.0, pd.unit).to("1/eV").value
            w0 = xg * (pd._shape(e_eV) * per_eV * unit_scale)
            with np.errstate(divide="ignore", invalid="ignore"):
                d0 = np.log(w0[1:] / w0[:-1])
            # a zero node ends the power-law segment (utils.py:347-348): same marker as in
            # the emission tables (NH_DL_ZERO)
            d0[(w0[1:] == 0.0) | (w0[:-1] == 0.0) | ~np.isfinite(d0)] = 1e300
            w0d, d0d = ctx.const(w0), ctx.const(np.append(d0, 0.0))
            amp = pd._amp
            if isinstance(amp, DVec):
                lz = amp.lazy()
            elif np.ndim(amp) > 0:
                keep = ctx.array(np.asarray(amp, dtype=float))
                lz = DVec(ctx, keep, keep.ptr, N).lazy()
            else:
                lz = lazy_const(float(amp))
            w, lw = ctx.empty((N, nG)), ctx.empty((N, nG))
            comp = (nh_comp * 1)()
            comp[0] = nh_comp(w0d.ptr, 0, 1.0)  # ld = 0: the same row for every walker
            ctx.call("nh_lincomb", comp, 1, None, C.addressof(lz), N, nG, w, nG)
            comp[0] = nh_comp(d0d.ptr, 0, 1.0)
            ctx.call("nh_lincomb", comp, 1, None, None, N, nG, lw, nG)
            hit = cache[key] = (w, lw)
        return ctx, N, hit[0], hit[1], xd, lx

    # -- public ---------------------------------------------------------------
    def _spectrum_branch(self, photon_energy):
        """``_spectrum`` on its own side stream when the values stay on the device:
        the emission components of a model evaluation are independent of each
        other, so their launch sequences run concurrently (graph branches)"""
        if self.on_device:
            ctx = get_context()
            self._prefork(ctx)
            with ctx.branch():
                return self._spectrum(photon_energy)
        return self._spectrum(photon_energy)

    def _prefork(self, ctx):
        """launch what several components share (the packed parameter rows) on the
        MAIN stream before forking, so that a side stream never has to wait for
        another side stream's whole queue"""
        pd = self.particle_distribution
        if hasattr(pd, "device_rows"):
            pd.device_rows(ctx, self.batch_size, amplitude_to=_PER_EV)

    def flux(self, photon_energy, distance=1 * u.kpc):
        """Differential flux at ``distance``; ``distance=0`` gives the intrinsic
        differential luminosity (radiative.py:88-111)."""
        if self._needs_walker_loop():
            return self._loop_walkers("flux", photon_energy, distance=distance)
        spec = self._spectrum_branch(photon_energy)
        if not _dist_is_zero(distance):
            distance = validate_scalar("distance", distance, physical_type="length")
            spec = spec / (4 * np.pi * distance.to("cm") ** 2)
            out_unit = "1/(s cm2 eV)"
        else:
            out_unit = "1/(s eV)"
        return spec.to(out_unit)

    def sed(self, photon_energy, distance=1 * u.kpc):
        """Spectral energy distribution (radiative.py:113-134)."""
        out_unit = "erg/s" if _dist_is_zero(distance) else "erg/(cm2 s)"
        photon_energy = _validate_ene(photon_energy)
        return (self.flux(photon_energy, distance) * photon_energy ** 2.0).to(out_unit)


def _dist_is_zero(distance):
    v = distance.value if isinstance(distance, u.Quantity) else distance
    return bool(np.all(np.asarray(v) == 0))


def _dlog(K):
    """ln(K[i+1]/K[i]) padded to len(K): the log-ratio column of a 1-row table"""
    return np.concatenate([np.log(K[1:] / K[:-1]), [0.0]])


def _log_grid(lo, hi, per_decade):
    l0, l1 = np.log10(lo), np.log10(hi)
    return np.logspace(l0, l1, max(10, int(per_decade * (l1 - l0))))


def _erg_factor(q):
    f = ASTROPY_TO_ERG.get(q.unit.name)
    return q.unit.to("erg") if f is None else f


def _to_GeV(q):
    return energy_ratio_to(q, ASTROPY_TO_GEV, "GeV")


def _scalar_energy(name, q):
    if _is_batched(q):
        raise NotImplementedError(
            "%s must be the same for every walker of a batch (the particle grid is shared); "
            "evaluate walkers with different %s in separate calls" % (name, name))
    return q


class BaseElectron(BaseRadiative):
    """electron grid, nelec, We (radiative.py:137-236)"""
    _structural = ("Eemin", "Eemax", "nEed")

    def __init__(self, particle_distribution):
        super().__init__(particle_distribution)
        self.param_names = ["Eemin", "Eemax", "nEed"]
        self._memoize = True
        self._cache = {}
        self._queue = []

    @staticmethod
    def _gam_between(Eemin, Eemax, nEed):
        # radiative.py:147-154: log10(E/mec2) with E and mec2 as astropy would
        # reduce them (value ratio times the unit ratio to erg)
        gmin = (Eemin.value / MEC2_ERG) * _erg_factor(Eemin)
        gmax = (Eemax.value / MEC2_ERG) * _erg_factor(Eemax)
        return _log_grid(gmin, gmax, nEed)

    @property
    def _gam(self):
        """Lorentz factor array"""
        return self._gam_between(_scalar_energy("Eemin", self.Eemin),
                                 _scalar_energy("Eemax", self.Eemax), self.nEed)

    # -- the general path on the device: Eemin / Eemax per walker -----------------------
    def _general_limits(self):
        """True when Eemin, Eemax or nEed is given per walker (host vector or device value):
        every walker then has its own grid (limits AND node count, radiative.py:147-154) and
        the spectrum comes from nh_general_electron"""
        return _per_walker(self.Eemin) or _per_walker(self.Eemax) or _per_walker(self.nEed)

    def _general_supported(self):
        return False

    def _general_needed(self):
        """True when the spectrum has to come from nh_general_electron (no table to share)"""
        return self._general_limits()

    _general_names = ("Eemin", "Eemax", "nEed")

    def _needs_walker_loop(self):
        if self._general_needed() and self._general_supported():
            others = [(n, v) for n, v in self._structural_values()
                      if n not in self._general_names and not n.endswith(("-T", "-theta"))]
            for n, v in others:
                if _per_walker(v):
                    return super()._needs_walker_loop()
            return False
        return super()._needs_walker_loop()

    def _general_launch(self, what, E_eV, B=None, seeds=(), Eemin=None, Eemax=None, seed_arrays=None,
                        seed_rows=False):
        """spectra [N][ncomp * nE] of nh_general_electron (device buffer); what = 2: We over
        each walker's grid between Eemin and Eemax (default: the object's own limits), [N][1];
        what = 4: InverseCompton on the shared monochromatic / tabulated seed
        ``seed_arrays = (energies [eV], densities)`` (nh_general_electron_seed)"""
        import ctypes as C

        from .darray import lazy_const
        ctx = get_context()
        pd = self.particle_distribution
        if not hasattr(pd, "device_rows"):
            raise NotImplementedError("per-walker Eemin / Eemax need a naima_amd.models "
                                      "particle distribution with analytic parameters")
        N = self.batch_size

        def lazy_of(q, unit):
            v = q.to(unit).value
            if isinstance(v, DVec):
                return v.lazy(), v
            if np.ndim(v) > 0:
                d = _as_dvec(ctx, v, N)
                return d.lazy(), d
            return lazy_const(float(v)), None

        # the limits travel in their own unit, with that unit in erg beside them: the kernel forms
        # (value / mec2[erg]) * unit_erg exactly as _gam_between does
        qmin = self.Eemin if Eemin is None else Eemin
        qmax = self.Eemax if Eemax is None else Eemax
        emin, k1 = lazy_of(qmin, qmin.unit)
        emax, k2 = lazy_of(qmax, qmax.unit)
        nv = self.nEed.value if isinstance(self.nEed, u.Quantity) else self.nEed
        if isinstance(nv, DVec):
            ned, k4 = nv.lazy(), nv
        elif np.ndim(nv) > 0:
            k4 = _as_dvec(ctx, np.asarray(nv, dtype=float), N)
            ned = k4.lazy()
        else:
            ned, k4 = lazy_const(float(nv)), None
        rows = pd.device_rows(ctx, N, amplitude_to=_PER_EV)
        nE = 1 if what == 2 else E_eV.size
        ncomp = len(seeds) if what == 1 else (2 if what == 3 else 1)
        out = ctx.empty((N, ncomp * nE))
        status = ctx.general_status()
        Bl, k3 = (lazy_of(B, "G") if what == 0 else (None, None))
        from .darray import nh_lazy
        T = (nh_lazy * max(ncomp, 1))()
        th = (nh_lazy * max(ncomp, 1))()
        keep = []
        for j, (Tq, thq) in enumerate(seeds):
            T[j], k = lazy_of(Tq, "K")
            keep.append(k)
            if thq is None:
                th[j] = lazy_const(-1.0)
            else:
                th[j], k = lazy_of(thq, "rad")
                keep.append(k)
        if what == 4 and seed_rows:
            se, sd = seed_arrays
            if isinstance(sd, DMat):
                sd_buf, sd_ptr = sd.buffer()
            else:
                sd_buf = ctx.array(np.ascontiguousarray(np.broadcast_to(np.asarray(sd, dtype=float),
                                                                        (N, se.size))))
                sd_ptr = sd_buf.ptr
            ctx.call("nh_general_electron_seed_rows", PD_KIND[pd.kind], rows, N, C.addressof(emin),
                     float(_erg_factor(qmin)), C.addressof(emax), float(_erg_factor(qmax)),
                     C.addressof(ned), ctx.const(se), sd_ptr, int(se.size), int(se.size),
                     ctx.const(E_eV), nE, out, nE, ctx.general_nmax, status)
            keep.append(sd_buf)
        elif what == 4:
            se, sd = seed_arrays
            ctx.call("nh_general_electron_seed", PD_KIND[pd.kind], rows, N, C.addressof(emin),
                     float(_erg_factor(qmin)), C.addressof(emax), float(_erg_factor(qmax)),
                     C.addressof(ned), ctx.const(se), ctx.const(sd), int(se.size), ctx.const(E_eV),
                     nE, out, nE, ctx.general_nmax, status)
        else:
            ctx.call("nh_general_electron", PD_KIND[pd.kind], rows, N, C.addressof(emin),
                     float(_erg_factor(qmin)), C.addressof(emax), float(_erg_factor(qmax)),
                     C.addressof(ned), what, C.addressof(Bl) if Bl is not None else None, T, th,
                     ncomp, ctx.const(E_eV) if what != 2 else None, nE, out, ncomp * nE,
                     ctx.general_nmax, status)
        del k1, k2, k3, k4, rows, keep
        if not self.on_device:
            ctx.check_general()
        return ctx, N, out

    def _electron_weights(self, gam=None):
        gam = self._gam if gam is None else gam
        e_eV = (gam * MEC2_ERG) * ERG_TO_EV
        return self._weights(gam, e_eV, MEC2_EV) + (gam,)

    @property
    def _nelec(self):
        """Particles per unit lorentz factor (radiative.py:156-160)"""
        ctx, N, w, lw, xd, lx, gam = self._electron_weights()
        n = w.get() / gam
        return n if self.is_batched else n[0]

    def _We_on(self, gam):
        ctx = get_context()
        if self.on_device and ctx.multistream:
            self._prefork(ctx)
            with ctx.branch():
                return self._We_on_impl(gam)
        if self.on_device:
            # the weights on the main stream, the small reduction beside what follows them
            w = self._electron_weights(gam)[2]
            with ctx.branch_at(getattr(w, "anchor", None)):
                return self._We_on_impl(gam)
        return self._We_on_impl(gam)

    def _We_on_impl(self, gam):
        ctx, N, w, lw, xd, lx, gam = self._electron_weights(gam)
        K = gam * MEC2_ERG  # u = x*y = (gam mec2)(gam nelec)
        Kt, dlnKt = ctx.const(K), ctx.const(_dlog(K))
        out = ctx.moment(w, lw, N, gam.size, lx, Kt, dlnKt)
        if self.on_device:
            return u.Quantity(DVec(ctx, out, out.ptr, N), u.erg)
        We = out.get()[:, 0]
        return u.Quantity(We if self.is_batched else We[0], u.erg)

    def _We_general(self, Eemin=None, Eemax=None):
        """We over every walker's own grid (limits / nEed per walker) by nh_general_electron"""
        ctx, N, out = self._general_launch(2, None, Eemin=Eemin, Eemax=Eemax)
        if self.on_device:
            return u.Quantity(DVec(ctx, out, out.ptr, N), u.erg)
        We = out.get()[:, 0]
        return u.Quantity(We if self.is_batched else We[0], u.erg)

    def _We_general_ok(self, lo, hi):
        """the device can integrate We over a grid per walker (limits lo, hi, self.nEed)"""
        if not hasattr(self.particle_distribution, "device_rows"):
            return False
        for q in (lo, hi, self.nEed):  # (a per-walker value has to match the object's own batch)
            v = q.value if isinstance(q, u.Quantity) else q
            if not isinstance(v, DVec) and np.ndim(v) > 0 and \
                    not (self.is_batched and len(v) == self.batch_size):
                return False
        return True

    def _We_between(self, lo, hi, what, **kw):
        if _per_walker(lo) or _per_walker(hi) or _per_walker(self.nEed):
            if self._We_general_ok(lo, hi):
                return self._We_general(lo, hi)
            return self._loop_walkers(what, **kw)  # (one walker at a time: host values only)
        return self._We_on(self._gam_between(_scalar_energy("Eemin", lo),
                                             _scalar_energy("Eemax", hi), self.nEed))

    @property
    def We(self):
        """Total energy in electrons used for the radiative calculation"""
        return self._We_between(self.Eemin, self.Eemax, "We")

    def compute_We(self, Eemin=None, Eemax=None):
        """Total energy in electrons between Eemin and Eemax (radiative.py:168-195).  The grid
        follows the limits in force: per walker only where THEY are (radiative.py:147-154)"""
        if Eemin is None and Eemax is None:
            return self.We
        lo = self.Eemin if Eemin is None else validate_scalar_or_batch(
            "Eemin", Eemin, physical_type="energy")
        hi = self.Eemax if Eemax is None else validate_scalar_or_batch(
            "Eemax", Eemax, physical_type="energy")
        return self._We_between(lo, hi, "compute_We", Eemin=Eemin, Eemax=Eemax)

    def set_We(self, We, Eemin=None, Eemax=None, amplitude_name=None):
        """Normalize the particle distribution so that the electron energy between
        Eemin and Eemax is We (radiative.py:197-236)"""
        We = validate_scalar_or_batch("We", We, physical_type="energy")
        oldWe = self.compute_We(Eemin=Eemin, Eemax=Eemax)
        ratio = (We / oldWe).decompose().value
        if amplitude_name is None:
            try:
                self.particle_distribution.amplitude = self.particle_distribution.amplitude * ratio
            except AttributeError:
                log.error("The particle distribution does not have an attribute called "
                          "amplitude to modify its normalization: you can set the name with "
                          "the amplitude_name parameter of set_We")
        else:
            oldampl = getattr(self.particle_distribution, amplitude_name)
            setattr(self.particle_distribution, amplitude_name, oldampl * ratio)


class Synchrotron(BaseElectron):
    """Synchrotron emission from an electron population in a random magnetic field
    (Aharonian, Kelner & Prosekin 2010); radiative.py:239-342.

    Parameters as in naima: ``particle_distribution``, ``B`` (default 3.24e-6 G),
    and the keyword overrides ``Eemin`` (1 GeV), ``Eemax`` (1e9 mec2), ``nEed`` (100).
    ``B`` may be a 1-D array over walkers.
    """

    _walker_scalars = ("B",)

    def __init__(self, particle_distribution, B=3.24e-6 * u.G, **kwargs):
        super().__init__(particle_distribution)
        self.B = validate_scalar_or_batch("B", B, physical_type="magnetic flux density")
        self.Eemin = 1 * u.GeV
        self.Eemax = 1e9 * mec2
        self.nEed = 100
        self.param_names += ["B"]
        self.__dict__.update(**kwargs)
        # a device-resident B rides in the free slot 7 of the distribution's packed rows
        Bv = self.B.to("G").value
        pd = particle_distribution
        if isinstance(Bv, DVec) and hasattr(pd, "device_rows") and "_slot7" not in pd.__dict__ \
                and not pd.__dict__.get("_rows_dev"):
            pd.__dict__["_slot7"] = Bv

    def _own_batch_sizes(self):
        return (_batch_of(self.B),)

    def _own_device_values(self):
        return (self.B,)

    def _prefork(self, ctx):
        super()._prefork(ctx)
        Bv = self.B.to("G").value
        if isinstance(Bv, DVec) and self.__dict__.get("_B_dense") is None \
                and not self._B_in_rows(Bv):
            self._B_dense = Bv.dense()

    def _B_in_rows(self, Bv):
        pd = self.particle_distribution
        rider = pd.__dict__.get("_slot7_packed")
        return rider is not None and rider.ptr == Bv.ptr and rider.stride == Bv.stride \
            and (rider.a, rider.b, rider.c, rider.tf) == (Bv.a, Bv.b, Bv.c, Bv.tf)

    def _general_supported(self):
        return True

    def _spectrum(self, photon_energy):
        E = _validate_ene(photon_energy)
        E_eV = np.atleast_1d(E.to("eV").value).astype(float)
        if self._general_limits():
            ctx, N, out = self._general_launch(0, E_eV, B=self.B)
            return self._result(ctx, out, N, E_eV.size, E)
        ctx, N, w, lw, gd, lx, gam = self._electron_weights()
        Bv = self.B.to("G").value
        ldB = 1
        if isinstance(Bv, DVec) and self._B_in_rows(Bv):
            Bd = self.particle_distribution.device_rows(ctx, N, amplitude_to=_PER_EV)
            Bp, ldB = Bd.ptr + 8 * 7, 8
        elif isinstance(Bv, DVec):
            Bd = self.__dict__.get("_B_dense") or Bv.dense()
            Bp = Bd.ptr
        else:
            Bd = ctx.array(np.broadcast_to(np.asarray(Bv, dtype=float), (N,)))
            Bp = Bd.ptr
        out = ctx.emit_synchrotron(w, lw, Bp, ldB, N, gd, lx, gam.size, ctx.const(E_eV),
                                   E_eV.size, keep=(Bd,), E_host=E_eV)
        del Bd
        return self._result(ctx, out, N, E_eV.size, E)


class InverseCompton(BaseElectron):
    """Inverse Compton emission (Khangulyan, Aharonian & Kelner 2014 for thermal seed
    fields; Aharonian & Atoyan 1981 for monochromatic / tabulated ones);
    radiative.py:370-791.  ``seed_photon_fields`` follows naima's grammar:
    'CMB' | 'FIR' | 'NIR' | [name, T, u] | [name, T, u, theta] | [name, E, density].
    """

    def __init__(self, particle_distribution, seed_photon_fields=["CMB"], **kwargs):
        super().__init__(particle_distribution)
        self.seed_photon_fields = self._process_input_seed(seed_photon_fields)
        self.Eemin = 1 * u.GeV
        self.Eemax = 1e9 * mec2
        self.nEed = 100
        self.param_names += ["seed_photon_fields"]
        self.__dict__.update(**kwargs)

    def _general_supported(self):
        """the general path on the device: thermal seed fields (their temperature, angle and
        energy density may all be per walker), monochromatic / tabulated seeds that every walker
        shares, and tabulated seeds with a photon density per walker (SSC:
        nh_general_electron_seed_rows)"""
        return True

    def _seed_shape_per_walker(self):
        """a thermal seed whose temperature or angle is given per walker: its Khangulyan kernel
        differs from walker to walker, no table can be shared (radiative.py:547-607)"""
        for seed in self.seed_photon_fields.values():
            if seed["type"] == "thermal" and (_per_walker(seed["T"]) or (
                    not seed["isotropic"] and _per_walker(seed["theta"]))):
                return True
        return False

    def _general_needed(self):
        if _per_walker(self.nEed):
            return False
        return self._general_limits() or (self._seed_shape_per_walker()
                                          and self._general_supported())

    def _spectrum_general(self, E, E_eV):
        """radiative.py:657-710 with a particle grid per walker (nh_general_electron): the
        Khangulyan kernel at every (node, energy) of every walker, no shared table"""
        nE = E_eV.size
        allnames = list(self.seed_photon_fields)
        names = [n for n in allnames if self.seed_photon_fields[n]["type"] == "thermal"]
        Eph = E_eV / MEC2_EV
        dev = self.on_device
        byname = {}
        for n in allnames:  # monochromatic / tabulated seeds: one launch each (what = 4)
            seed = self.seed_photon_fields[n]
            if seed["type"] == "thermal":
                continue
            se = np.atleast_1d(seed["energy"].to("eV").value).astype(float)
            if self._seed_per_walker(seed):
                # a photon density per walker (SSC: each walker's own synchrotron photons,
                # examples/CrabNebula_SynSSC.py:29-45) over each walker's own grid: rows
                # [N][ns] on the device, a host array or the lazy device expression the
                # model function built from SYN.flux(...)
                sdv = seed["photon_density"].to("1/(eV cm3)").value
                ctx, N, o1 = self._general_launch(4, E_eV, seed_arrays=(se, sdv), seed_rows=True)
            else:
                if se.size == 1:
                    sdv = np.atleast_1d(seed["photon_density"].to("eV/cm3").value).astype(float)
                else:
                    sdv = np.asarray(seed["photon_density"].to("1/(eV cm3)").value, dtype=float)
                ctx, N, o1 = self._general_launch(4, E_eV, seed_arrays=(se, sdv))
            colfac = Eph / E_eV  # radiative.py:684-687 (uf = 1)
            byname[n] = (DMat.from_buffer(ctx, o1, N, nE) * colfac) if dev else o1.get() * colfac
        seeds = []
        for n in names:
            sd = self.seed_photon_fields[n]
            seeds.append((sd["T"], None if sd["isotropic"] else sd["theta"]))
        if names:
            ctx, N, out = self._general_launch(1, E_eV, seeds=seeds)
        host = None if (dev or not names) else out.get()
        specs = []
        for j, n in enumerate(names):
            sd = self.seed_photon_fields[n]
            T = sd["T"].to("K").value
            uf = sd["u"].to("erg/cm3").value / (AR_CGS * T ** 4)  # radiative.py:684-687
            colfac = Eph / E_eV
            if dev:
                m = DMat(ctx, [(out, out.ptr + 8 * j * nE, len(names) * nE, 1.0)], (N, nE)) * colfac
                m = m * _as_dvec(ctx, uf, N) if (_per_walker(sd["u"]) or _per_walker(sd["T"])) \
                    else m * float(uf)
                specs.append(m)
            else:
                v = host[:, j * nE:(j + 1) * nE] * colfac
                specs.append(v * np.broadcast_to(np.asarray(uf, dtype=float), (N,))[:, None])
        byname.update(zip(names, specs))
        specs = [byname[n] for n in allnames]  # (in the order the seeds were given: specic)
        if dev:
            self.specic = [u.Quantity(m, _SPEC_UNIT) for m in specs]
            total = specs[0]
            for m in specs[1:]:
                total = total + m
            return u.Quantity(total, _SPEC_UNIT)
        self.specic = [u.Quantity(self._finish(v, E), _SPEC_UNIT) for v in specs]
        return u.Quantity(self._finish(np.sum(specs, axis=0), E), _SPEC_UNIT)

    def _structural_values(self):
        """grid parameters + the temperatures / angles of thermal seeds (their emission
        tables depend on them)"""
        vals = super()._structural_values()
        for name, seed in self.seed_photon_fields.items():
            if seed["type"] == "thermal":
                vals.append((name + "-T", seed["T"]))
                if not seed["isotropic"]:
                    vals.append((name + "-theta", seed["theta"]))
        return vals

    def _walker(self, k):
        one = super()._walker(k)
        seeds = OrderedDict()
        for name, seed in self.seed_photon_fields.items():
            sd = dict(seed)
            for key in ("T", "u", "theta"):
                if key in sd and _is_batched(sd[key]):
                    sd[key] = sd[key][k]
            if sd["type"] == "array" and np.ndim(sd["photon_density"].value) == 2:
                sd["photon_density"] = sd["photon_density"][k]
            seeds[name] = sd
        one.__dict__["seed_photon_fields"] = seeds
        return one

    def _own_batch_sizes(self):
        sizes = []
        for seed in self.seed_photon_fields.values():
            if seed["type"] == "array" and np.ndim(seed["photon_density"].value) == 2:
                sizes.append(seed["photon_density"].shape[0])
            if seed["type"] == "thermal":
                sizes.append(_batch_of(seed["u"]))
        return tuple(sizes)

    def _own_device_values(self):
        return tuple(s["photon_density"] for s in self.seed_photon_fields.values()
                     if s["type"] == "array") + tuple(
            s["u"] for s in self.seed_photon_fields.values() if s["type"] == "thermal")

    @staticmethod
    def _process_input_seed(seed_photon_fields):
        """radiative.py:432-545"""
        Tcmb = 2.72548 * u.K
        Tfir = 30 * u.K
        ufir = 0.5 * u.eV / u.cm ** 3
        Tnir = 3000 * u.K
        unir = 1.0 * u.eV / u.cm ** 3
        ar = AR_CGS * u.Unit("erg/(cm3 K4)")
        if type(seed_photon_fields) is not list:
            seed_photon_fields = seed_photon_fields.split("-")
        result = OrderedDict()
        for idx, inseed in enumerate(seed_photon_fields):
            seed = {}
            if isinstance(inseed, str):
                name = inseed
                seed["type"] = "thermal"
                if inseed == "CMB":
                    seed["T"], seed["u"], seed["isotropic"] = Tcmb, ar * Tcmb ** 4, True
                elif inseed == "FIR":
                    seed["T"], seed["u"], seed["isotropic"] = Tfir, ufir, True
                elif inseed == "NIR":
                    seed["T"], seed["u"], seed["isotropic"] = Tnir, unir, True
                else:
                    log.warning("Will not use seed {0} because it is not CMB, FIR or NIR".format(
                        inseed))
                    raise TypeError
            elif type(inseed) is list and (len(inseed) == 3 or len(inseed) == 4):
                isotropic = len(inseed) == 3
                if isotropic:
                    name, T, uu = inseed
                    seed["isotropic"] = True
                else:
                    name, T, uu, theta = inseed
                    seed["isotropic"] = False
                    if _is_batched(theta):
                        validate_physical_type("{0}-theta".format(name), theta, "angle")
                        seed["theta"] = theta
                    else:
                        seed["theta"] = validate_scalar("{0}-theta".format(name), theta,
                                                        physical_type="angle")
                thermal = T.unit.physical_type == "temperature"
                if thermal:
                    seed["type"] = "thermal"
                    if _is_batched(T):  # one temperature per walker: the general path
                        validate_physical_type("{0}-T".format(name), T, "temperature")
                    else:
                        validate_scalar("{0}-T".format(name), T, domain="positive",
                                        physical_type="temperature")
                    seed["T"] = T
                    if not isinstance(uu, u.Quantity) and uu == 0:
                        seed["u"] = ar * T ** 4
                    elif _per_walker(uu):  # one energy density per walker (fit parameter)
                        validate_physical_type("{0}-u".format(name), uu, "pressure")
                        seed["u"] = uu
                    else:
                        validate_scalar("{0}-u".format(name), uu, domain="positive",
                                        physical_type="pressure")
                        seed["u"] = uu
                else:
                    seed["type"] = "array"
                    T = u.Quantity(np.atleast_1d(T.value).ravel(), T.unit)
                    if not uu.on_device:
                        uv = np.asarray(uu.value, dtype=float)
                        uu = u.Quantity(uv if uv.ndim == 2 else np.atleast_1d(uv).ravel(), uu.unit)
                    seed["energy"] = validate_array("{0}-energy".format(name), T,
                                                    domain="positive", physical_type="energy")
                    if seed["energy"].size == 1:
                        validate_physical_type("{0}-density".format(name), uu, "pressure")
                        seed["photon_density"] = uu
                    else:
                        if uu.unit.physical_type == "pressure":
                            uu = uu / seed["energy"] ** 2
                        seed["photon_density"] = validate_array(
                            "{0}-density".format(name), uu, domain="positive", ndim=uu.ndim,
                            physical_type="differential number density")
            else:
                raise TypeError("Unable to process seed photon field: {0}".format(inseed))
            result[name] = seed
        return result

    def _static_seed_key(self, seed):
        if seed["type"] == "thermal":
            return ("T", float(seed["T"].to("K").value),
                    -1.0 if seed["isotropic"] else float(seed["theta"].to("rad").value))
        se = seed["energy"].to("eV").value
        if se.size == 1:
            sd = np.atleast_1d(seed["photon_density"].to("eV/cm3").value)
        else:
            sd = seed["photon_density"].to("1/(eV cm3)").value
        return ("A", hash(se.tobytes()), hash(np.ascontiguousarray(sd).tobytes()))

    def _build_static_tables(self, ctx, static, gd, nG, Ed, nE):
        """one transposed table [nG][S*nE]: the seeds sit side by side along k"""
        nK = len(static) * nE
        Kt, dlnKt = ctx.empty((nG, nK)), ctx.empty((nG, nK))
        for j, name in enumerate(static):
            seed = self.seed_photon_fields[name]
            kt, lkt = Kt.ptr + 8 * j * nE, dlnKt.ptr + 8 * j * nE
            if seed["type"] == "thermal":
                T = seed["T"].to("K").value
                theta = -1.0 if seed["isotropic"] else seed["theta"].to("rad").value
                ctx.call("nh_table_ic_planck", gd, nG, Ed, nE, float(T), float(theta), kt, lkt, nK)
            else:
                se = seed["energy"].to("eV").value
                if se.size == 1:
                    sd = np.atleast_1d(seed["photon_density"].to("eV/cm3").value)
                else:
                    sd = seed["photon_density"].to("1/(eV cm3)").value
                ctx.call("nh_table_ic_seed", gd, nG, Ed, nE, ctx.const(se), ctx.const(sd),
                         int(se.size), kt, lkt, nK)
        return Kt, dlnKt

    def _spectrum(self, photon_energy):
        """radiative.py:657-710: one table per walker-independent seed (built once and
        cached by what it depends on), ONE reduction launch over all of them, then
        the per-walker (SSC) seeds."""
        E = _validate_ene(photon_energy)
        E_eV = np.atleast_1d(E.to("eV").value).astype(float)
        nE = E_eV.size
        if self._general_needed():
            return self._spectrum_general(E, E_eV)
        ctx, N, w, lw, gd, lx, gam = self._electron_weights()
        nG = gam.size
        Ed = ctx.const(E_eV)
        Eph = E_eV / MEC2_EV
        names = list(self.seed_photon_fields)
        static = [n for n in names if not self._seed_per_walker(self.seed_photon_fields[n])]
        dev = self.on_device
        specs = {}
        if static:
            nK = len(static) * nE
            key = ("ic", gd.ptr, Ed.ptr) + tuple(
                self._static_seed_key(self.seed_photon_fields[n]) for n in static)
            Kt, dlnKt = ctx.table(
                key, lambda: self._build_static_tables(ctx, static, gd, nG, Ed, nE))
            scale = np.empty(nK)
            rowfac = {}
            for j, name in enumerate(static):
                seed = self.seed_photon_fields[name]
                if seed["type"] == "thermal":
                    T = seed["T"].to("K").value
                    uf = (seed["u"].to("erg/cm3").value / (AR_CGS * T ** 4))
                    if _per_walker(seed["u"]):  # energy density as a fit parameter
                        rowfac[name], uf = uf, 1.0
                else:
                    uf = 1.0
                scale[j * nE:(j + 1) * nE] = uf * Eph / E_eV  # radiative.py:684-687
            # the abscissa may be cut into planes that different workgroups reduce (evens out
            # the load per CU); the planes are summed by whoever consumes the spectrum.
            # IC kernels are >= 0; a user-supplied array seed is validated positive
            out, ns = ctx.emit_tables(w, lw, N, nG, lx, Kt, dlnKt, nK, ctx.const(scale), 1)
            if dev:
                for j, name in enumerate(static):
                    specs[name] = DMat(ctx, [(out, out.ptr + 8 * (h * N * nK + j * nE), nK, 1.0)
                                             for h in range(ns)], (N, nE))
                    if name in rowfac:
                        specs[name] = specs[name] * _as_dvec(ctx, rowfac[name], N)
            else:
                host = out.get().reshape(ns, N, nK).sum(axis=0)
                for j, name in enumerate(static):
                    specs[name] = host[:, j * nE:(j + 1) * nE]
                    if name in rowfac:
                        specs[name] = specs[name] * np.broadcast_to(
                            np.asarray(rowfac[name], dtype=float), (N,))[:, None]
        for name in names:
            if name in specs:
                continue
            seed = self.seed_photon_fields[name]
            se = seed["energy"].to("eV").value
            sdv = seed["photon_density"].to("1/(eV cm3)").value
            if isinstance(sdv, DMat):
                sd_buf, sd_ptr = sdv.buffer()
            else:
                sd_buf = ctx.array(np.broadcast_to(sdv, (N, se.size)))
                sd_ptr = sd_buf.ptr
            out = ctx.plan_buffer(("seed-integral", name, N, nE), (N, nE))
            sed, ns = ctx.const(se), int(se.size)
            # the Aharonian-Atoyan kernel on (seed energy, gamma, photon energy) does not depend
            # on the walker: tabulated once per set of grids when HBM has room for it
            tab = ctx.ssc_table(gd, nG, Ed, nE, sed, ns) if ns >= 2 else None
            if tab is not None:
                ctx.call("nh_ic_seed_walkers_tab", w, lw, N, gd, lx, nG, Ed, nE, sed, sd_ptr, ns,
                         tab, out, nE)
            else:
                ctx.call("nh_ic_seed_walkers", w, lw, N, gd, lx, nG, Ed, nE, sed, sd_ptr, ns,
                         out, nE)
            del sd_buf
            specs[name] = DMat.from_buffer(ctx, out, N, nE) if dev else out.get()
        if dev:
            self.specic = [u.Quantity(specs[n], _SPEC_UNIT) for n in names]
            total = specs[names[0]]
            for n in names[1:]:
                total = total + specs[n]
            return u.Quantity(total, _SPEC_UNIT)
        self.specic = [u.Quantity(self._finish(specs[n], E), _SPEC_UNIT) for n in names]
        total = np.sum([specs[n] for n in names], axis=0)
        return u.Quantity(self._finish(total, E), _SPEC_UNIT)

    @staticmethod
    def _seed_per_walker(seed):
        return seed["type"] == "array" and np.ndim(seed["photon_density"].value) == 2

    def _own_batched(self):
        return any(self._seed_per_walker(s) for s in self.seed_photon_fields.values())

    def _seed_index(self, seed):
        """radiative.py:732-747"""
        if not isinstance(seed, int):
            if seed not in self.seed_photon_fields:
                raise ValueError("Provided seed photon field name is not in the definition of "
                                 "the InverseCompton instance")
            return list(self.seed_photon_fields.keys()).index(seed)
        if seed > len(self.seed_photon_fields):
            raise ValueError("Provided seed photon field number is larger than the number of "
                             "seed photon fields defined in the InverseCompton instance")
        return seed

    def flux(self, photon_energy, distance=1 * u.kpc, seed=None):
        """Differential flux, optionally from a single seed photon field
        (radiative.py:712-759)"""
        if self._needs_walker_loop():
            return self._loop_walkers("flux", photon_energy, distance=distance, seed=seed)
        model = super().flux(photon_energy, distance=distance)
        if seed is not None:
            idx = self._seed_index(seed)
            if not _dist_is_zero(distance):
                distance = validate_scalar("distance", distance, physical_type="length")
                model = (self.specic[idx] / (4 * np.pi * distance.to("cm") ** 2)).to(
                    "1/(s cm2 eV)")
            else:
                model = self.specic[idx].to("1/(s eV)")
        return model

    def sed(self, photon_energy, distance=1 * u.kpc, seed=None):
        """radiative.py:761-791"""
        sed = super().sed(photon_energy, distance=distance)
        if seed is not None:
            out_unit = "erg/s" if _dist_is_zero(distance) else "erg/(cm2 s)"
            photon_energy = _validate_ene(photon_energy)
            sed = (self.flux(photon_energy, distance=distance, seed=seed)
                   * photon_energy ** 2.0).to(out_unit)
        return sed


class Bremsstrahlung(BaseElectron):
    """Bremsstrahlung on a completely ionised gas (Baring et al. 1999);
    radiative.py:794-989.  ``n0``: total ion number density; ``weight_ee`` /
    ``weight_ep`` default to ISM abundances."""

    _structural = BaseElectron._structural + ("weight_ee", "weight_ep")
    _walker_scalars = ("n0",)

    def _general_supported(self):
        return True

    def __init__(self, particle_distribution, n0=1 / u.cm ** 3, **kwargs):
        super().__init__(particle_distribution)
        self.n0 = n0
        self.Eemin = 100 * u.MeV
        self.Eemax = 1e9 * mec2
        self.nEed = 300
        Y = np.array([1.0, 9.59e-2])
        Z = np.array([1, 2])
        X = Y / np.sum(Y)
        self.weight_ee = np.sum(Z * X)
        self.weight_ep = np.sum(Z ** 2 * X)
        self.param_names += ["n0", "weight_ee", "weight_ep"]
        self.__dict__.update(**kwargs)

    def _spectrum(self, photon_energy):
        E = _validate_ene(photon_energy)
        E_eV = np.atleast_1d(E.to("eV").value).astype(float)
        nE = E_eV.size
        rows = None
        if _per_walker(self.n0):  # the target density is a fit parameter: one factor per walker
            validate_physical_type("n0", self.n0, "number density")
            rows, n0 = self.n0.to("1/cm3").value, 1.0
        else:
            n0 = validate_scalar("n0", self.n0, physical_type="number density").to("1/cm3").value
        if self._general_limits():
            # Eemin / Eemax / nEed per walker: every walker's own grid, the cross sections
            # evaluated at its nodes (nh_general_electron, what = 3: e-e | e-ion side by side)
            ctx, N, out = self._general_launch(3, E_eV)
            fee, fep = n0 * self.weight_ee * C_CGS, n0 * self.weight_ep * C_CGS
            if self.on_device:
                tot = DMat.from_buffer(ctx, out, N, nE, ld=2 * nE, scale=fee) + \
                    DMat.from_buffer(ctx, out, N, nE, ld=2 * nE, col0=nE, scale=fep)
                if rows is not None:
                    tot = tot * _as_dvec(ctx, rows, N)
                return u.Quantity(tot, _SPEC_UNIT)
            host = out.get()
            host = fee * host[:, :nE] + fep * host[:, nE:]
            if rows is not None:
                host = host * np.broadcast_to(np.asarray(rows, dtype=float), (N,))[:, None]
            return u.Quantity(self._finish(host, E), _SPEC_UNIT)
        ctx, N, w, lw, gd, lx, gam = self._electron_weights()
        nG = gam.size
        Ed = ctx.const(E_eV)

        def build():
            # [nG][2 nE]: sigma_ee | sigma_ep side by side
            Kt, dKt = ctx.empty((nG, 2 * nE)), ctx.empty((nG, 2 * nE))
            ctx.call("nh_table_brems", gd, nG, Ed, nE, Kt.ptr, dKt.ptr, Kt.ptr + 8 * nE,
                     dKt.ptr + 8 * nE, 2 * nE)
            return Kt, dKt

        Kt, dKt = ctx.table(("brems", gd.ptr, Ed.ptr), build)
        # spec = n0 (w_ee c int(n sigma_ee) + w_ep c int(n sigma_1)), radiative.py:949-987
        scale = np.concatenate([np.full(nE, n0 * self.weight_ee * C_CGS),
                                np.full(nE, n0 * self.weight_ep * C_CGS)])
        # (the Baring+99 fits go negative near their edges: signed reduction)
        out, _ = ctx.emit_tables(w, lw, N, nG, lx, Kt, dKt, 2 * nE, ctx.const(scale), 0,
                                 may_split=False)
        if self.on_device:
            tot = DMat.from_buffer(ctx, out, N, nE, ld=2 * nE) + \
                DMat.from_buffer(ctx, out, N, nE, ld=2 * nE, col0=nE)
            if rows is not None:
                tot = tot * _as_dvec(ctx, rows, N)
            return u.Quantity(tot, _SPEC_UNIT)
        host = out.get()
        host = host[:, :nE] + host[:, nE:]
        if rows is not None:
            host = host * np.broadcast_to(np.asarray(rows, dtype=float), (N,))[:, None]
        return u.Quantity(self._finish(host, E), _SPEC_UNIT)

    def _own_batch_sizes(self):
        return (_batch_of(self.n0),)

    def _own_device_values(self):
        return (self.n0,)


class BaseProton(BaseRadiative):
    """proton grid, J, Wp (radiative.py:992-1096)"""
    _structural = ("Epmin", "Epmax", "nEpd")

    def __init__(self, particle_distribution):
        super().__init__(particle_distribution)
        self.param_names = ["Epmin", "Epmax", "nEpd"]
        self._memoize = True
        self._cache = {}
        self._queue = []

    @property
    def _Ep(self):
        """Proton energy array in GeV (radiative.py:1002-1009)"""
        lo = _to_GeV(_scalar_energy("Epmin", self.Epmin))
        hi = _to_GeV(_scalar_energy("Epmax", self.Epmax))
        return np.logspace(np.log10(lo), np.log10(hi),
                           max(10, int(self.nEpd * (np.log10(hi / lo)))))

    # -- the general path on the device: Epmin / Epmax / nEpd per walker -------------------
    def _general_limits(self):
        """True when Epmin, Epmax or nEpd is given per walker: every walker then has its own
        proton grid (radiative.py:1002-1009) and nh_general_proton integrates over it"""
        return _per_walker(self.Epmin) or _per_walker(self.Epmax) or _per_walker(self.nEpd)

    def _general_ok(self, *values):
        if not hasattr(self.particle_distribution, "device_rows"):
            return False
        for q in values:  # (a per-walker value has to match the object's own batch)
            v = q.value if isinstance(q, u.Quantity) else q
            if not isinstance(v, DVec) and np.ndim(v) > 0 and \
                    not (self.is_batched and len(v) == self.batch_size):
                return False
        return True

    def _needs_walker_loop(self):
        if self._general_limits() and self._general_ok(self.Epmin, self.Epmax, self.nEpd):
            others = [v for n, v in self._structural_values()
                      if n not in ("Epmin", "Epmax", "nEpd")]
            if not any(_per_walker(v) for v in others):
                return False
        return super()._needs_walker_loop()

    def _general_proton(self, what, E_eV, lo, hi, count_mode, hiE=0, nuc=0, lut=None):
        """nh_general_proton over every walker's own grid between lo and hi: spectra [N][nE]
        (what 0: analytic cross section, 1: look-up table) or Wp [N][1] in GeV (what 2)"""
        import ctypes as C

        from .darray import lazy_const
        ctx = get_context()
        pd = self.particle_distribution
        N = self.batch_size

        def lazy_of(v):
            v = v.value if isinstance(v, u.Quantity) else v
            if isinstance(v, DVec):
                return v.lazy(), v
            if np.ndim(v) > 0:
                d = _as_dvec(ctx, np.asarray(v, dtype=float), N)
                return d.lazy(), d
            return lazy_const(float(v)), None

        def gev_factor(q):
            f = ASTROPY_TO_GEV.get(q.unit.name)
            return float(q.unit.to("GeV") if f is None else f)

        emin, k1 = lazy_of(lo)
        emax, k2 = lazy_of(hi)
        ned, k3 = lazy_of(self.nEpd)
        rows = pd.device_rows(ctx, N, amplitude_to=_PER_EV)
        nE = 1 if what == 2 else E_eV.size
        out = ctx.empty((N, nE))
        tx = ty = cf = None
        ntx = nty = 0
        if lut is not None:
            tx, ty, cf = (ctx.const(a) for a in lut)
            ntx, nty = int(lut[0].size), int(lut[1].size)
        ctx.call("nh_general_proton", PD_KIND[pd.kind], rows, N, C.addressof(emin), gev_factor(lo),
                 C.addressof(emax), gev_factor(hi), C.addressof(ned), count_mode, what, hiE, nuc,
                 tx, ntx, ty, nty, cf, ctx.const(E_eV) if what != 2 else None, nE, out, nE,
                 ctx.general_nmax, ctx.general_status())
        del k1, k2, k3, rows
        if not self.on_device:
            ctx.check_general()
        return ctx, N, out

    def _Wp_general(self, lo, hi, count_mode):
        ctx, N, out = self._general_proton(2, None, lo, hi, count_mode)
        if self.on_device:
            return u.Quantity(DVec(ctx, out, out.ptr, N), u.GeV).to("erg")
        Wp = out.get()[:, 0]
        return u.Quantity(Wp if self.is_batched else Wp[0], u.GeV).to("erg")

    def _proton_weights(self, Ep=None):
        Ep = self._Ep if Ep is None else Ep
        return self._weights(Ep, Ep * 1e9, 1e9) + (Ep,)

    @property
    def _J(self):
        """Particles per unit proton energy in particles per GeV"""
        ctx, N, w, lw, xd, lx, Ep = self._proton_weights()
        J = w.get() / Ep
        return J if self.is_batched else J[0]

    def _Wp_on(self, Ep):
        ctx = get_context()
        if self.on_device and ctx.multistream:
            self._prefork(ctx)
            with ctx.branch():
                return self._Wp_on_impl(Ep)
        if self.on_device:
            w = self._proton_weights(Ep)[2]
            with ctx.branch_at(getattr(w, "anchor", None)):
                return self._Wp_on_impl(Ep)
        return self._Wp_on_impl(Ep)

    def _Wp_on_impl(self, Ep):
        ctx, N, w, lw, xd, lx, Ep = self._proton_weights(Ep)
        Kt, dlnKt = ctx.const(Ep), ctx.const(_dlog(Ep))
        out = ctx.moment(w, lw, N, Ep.size, lx, Kt, dlnKt)
        if self.on_device:
            return u.Quantity(DVec(ctx, out, out.ptr, N), u.GeV).to("erg")
        Wp = out.get()[:, 0]
        return u.Quantity(Wp if self.is_batched else Wp[0], u.GeV).to("erg")

    @property
    def Wp(self):
        """Total energy in protons"""
        if self._general_limits() and not self._needs_walker_loop():
            return self._Wp_general(self.Epmin, self.Epmax, 0)
        if self._needs_walker_loop():
            return self._loop_walkers("Wp")
        return self._Wp_on(self._Ep)

    def compute_Wp(self, Epmin=None, Epmax=None):
        """Total energy in protons between Epmin and Epmax (radiative.py:1023-1055).  The grid
        follows the limits in force: per walker only where THEY are"""
        if Epmin is None and Epmax is None:
            return self.Wp
        lo = self.Epmin if Epmin is None else validate_scalar_or_batch(
            "Epmin", Epmin, physical_type="energy")
        hi = self.Epmax if Epmax is None else validate_scalar_or_batch(
            "Epmax", Epmax, physical_type="energy")
        if _per_walker(lo) or _per_walker(hi) or _per_walker(self.nEpd):
            if self._general_ok(lo, hi, self.nEpd):
                return self._Wp_general(lo, hi, 1)
            return self._loop_walkers("compute_Wp", Epmin=Epmin, Epmax=Epmax)
        Epmin, Epmax = lo, hi
        l0 = np.log10(_to_GeV(Epmin))
        l1 = np.log10(_to_GeV(Epmax))
        return self._Wp_on(np.logspace(l0, l1, max(10, int(self.nEpd * (l1 - l0)))))

    def set_Wp(self, Wp, Epmin=None, Epmax=None, amplitude_name=None):
        """radiative.py:1057-1096"""
        Wp = validate_scalar_or_batch("Wp", Wp, physical_type="energy")
        oldWp = self.compute_Wp(Epmin=Epmin, Epmax=Epmax)
        ratio = (Wp / oldWp).decompose().value
        if amplitude_name is None:
            try:
                self.particle_distribution.amplitude = self.particle_distribution.amplitude * ratio
            except AttributeError:
                log.error("The particle distribution does not have an attribute called "
                          "amplitude to modify its normalization: you can set the name with "
                          "the amplitude_name parameter of set_Wp")
        else:
            oldampl = getattr(self.particle_distribution, amplitude_name)
            setattr(self.particle_distribution, amplitude_name, oldampl * ratio)


_LUT_CACHE = {}


def _lut_spline(fname):
    """FITPACK bicubic spline through 10**lut (radiative.py:1786-1794), fitted once
    per process on the host; the knots and coefficients go to the device."""
    if fname not in _LUT_CACHE:
        from scipy.interpolate import RectBivariateSpline
        f = np.load(fname)
        with np.errstate(all="ignore"):
            spl = RectBivariateSpline(f["X"], f["Y"], 10 ** f["lut"], kx=3, ky=3, s=0)
        tx, ty = spl.get_knots()
        _LUT_CACHE[fname] = (np.ascontiguousarray(tx), np.ascontiguousarray(ty),
                             np.ascontiguousarray(spl.get_coeffs()))
    return _LUT_CACHE[fname]


class PionDecay(BaseProton):
    """Pion-decay gamma-ray emission, Kafexhiu et al. (2014) parametrisation;
    radiative.py:1099-1536.  ``nh``, ``nuclear_enhancement``, and the keyword
    overrides ``Epmin`` (threshold), ``Epmax`` (10 PeV), ``nEpd`` (100), ``hiEmodel``
    ('Pythia8' | 'Geant4' | 'SIBYLL' | 'QGSJET'), ``useLUT`` (True)."""

    _m_p = M_P_GEV
    _Tth = T_TH_GEV

    _walker_scalars = ("nh",)

    def __init__(self, particle_distribution, nh=1.0 / u.cm ** 3, nuclear_enhancement=True,
                 **kwargs):
        super().__init__(particle_distribution)
        if _per_walker(nh):  # a fit parameter: one value per walker
            validate_physical_type("nh", nh, "number density")
            self.nh = nh
        else:
            self.nh = validate_scalar("nh", nh, physical_type="number density")
        self.nuclear_enhancement = nuclear_enhancement
        self.useLUT = True
        self.hiEmodel = "Pythia8"
        self.Epmin = (self._m_p + self._Tth + 1e-4) * u.GeV
        self.Epmax = 10 * u.PeV
        self.nEpd = 100
        self.param_names += ["nh", "nuclear_enhancement", "useLUT", "hiEmodel"]
        self.__dict__.update(**kwargs)

    def _lut_file(self):
        base = "PionDecayKafexhiu14_LUT_" + ("NucEnh_" if self.nuclear_enhancement else "")
        return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data",
                            base + "{0}.npz".format(self.hiEmodel))

    def _spectrum(self, photon_energy):
        E = _validate_ene(photon_energy)
        E_eV = np.atleast_1d(E.to("eV").value).astype(float)
        nE = E_eV.size
        if self.hiEmodel not in PP_MODEL:
            raise KeyError(self.hiEmodel)
        if self._general_limits():
            return self._spectrum_general(E, E_eV)
        ctx, N, w, lw, xd, lx, Ep = self._proton_weights()
        nG = Ep.size
        Ed = ctx.const(E_eV)
        use_lut = bool(self.useLUT)
        if use_lut and not os.path.exists(self._lut_file()):
            # radiative.py:1484-1493: only the Pythia8+NucEnh table is packaged
            import warnings
            warnings.warn("LUT {0} not found, reverting to useLUT = False".format(
                os.path.basename(self._lut_file())))
            self.useLUT = use_lut = False

        def build():
            Kt, dKt = ctx.empty((nG, nE)), ctx.empty((nG, nE))
            if use_lut:
                tx, ty, cf = _lut_spline(self._lut_file())
                ctx.call("nh_table_pion_lut", xd, nG, Ed, nE, ctx.const(tx), int(tx.size),
                         ctx.const(ty), int(ty.size), ctx.const(cf), Kt, dKt, nE)
            else:
                ctx.call("nh_table_pion_analytic", xd, nG, Ed, nE, PP_MODEL[self.hiEmodel],
                         int(bool(self.nuclear_enhancement)), Kt, dKt, nE)
            return Kt, dKt

        Kt, dKt = ctx.table(("pp", xd.ptr, Ed.ptr, use_lut, self.hiEmodel,
                             bool(self.nuclear_enhancement)), build)
        # the FITPACK look-up table rings below zero; the analytic form does not
        out, _ = ctx.emit_tables(w, lw, N, nG, lx, Kt, dKt, nE, None, 0 if use_lut else 1,
                                 may_split=False)
        nh = self.nh.to("1/cm3").value
        fac = (nh * C_CGS) * 1e-9  # 1/(s GeV) -> 1/(s eV), radiative.py:1534-1536
        if _per_walker(self.nh):
            self.specpp = self._result(ctx, out, N, nE, E, rows=fac)
        else:
            self.specpp = self._result(ctx, out, N, nE, E, scale=fac)
        return self.specpp

    def _spectrum_general(self, E, E_eV):
        """radiative.py:1495-1536 with a proton grid per walker (nh_general_proton): no table
        can be shared, the cross section (or the look-up table's spline) is evaluated at every
        (node, photon energy) of every walker"""
        use_lut = bool(self.useLUT)
        if use_lut and not os.path.exists(self._lut_file()):
            import warnings
            warnings.warn("LUT {0} not found, reverting to useLUT = False".format(
                os.path.basename(self._lut_file())))
            self.useLUT = use_lut = False
        ctx, N, out = self._general_proton(
            1 if use_lut else 0, E_eV, self.Epmin, self.Epmax, 0, hiE=PP_MODEL[self.hiEmodel],
            nuc=int(bool(self.nuclear_enhancement)),
            lut=_lut_spline(self._lut_file()) if use_lut else None)
        nh = self.nh.to("1/cm3").value
        fac = (nh * C_CGS) * 1e-9  # 1/(s GeV) -> 1/(s eV), radiative.py:1534-1536
        if _per_walker(self.nh):
            self.specpp = self._result(ctx, out, N, E_eV.size, E, rows=fac)
        else:
            self.specpp = self._result(ctx, out, N, E_eV.size, E, scale=fac)
        return self.specpp

    def _own_batch_sizes(self):
        return (_batch_of(self.nh),)

    def _own_device_values(self):
        return (self.nh,)


class PionDecayKelner06(BaseRadiative):
    """Pion-decay gamma rays with the parametrisation of Kelner, Aharonian & Bugayov 2006
    (radiative.py:1543-1767): full calculation above ``Etrans`` (default 0.1 TeV),
    delta-functional approximation below, normalised to meet at ``Etrans``.

    The reference integrates adaptively (scipy ``quad``, epsrel = 1e-3) one photon energy
    at a time; here every (walker, energy) is one wave with a converged fixed rule
    (``nh_pion_kelner06``), so values agree with the reference within its own quadrature
    tolerance (measured 4e-5) and with the converged integral to 1e-9."""
    param_names = ["nh", "Etrans"]
    _walker_scalars = ("nh",)

    def __init__(self, particle_distribution, nh=1.0 / u.cm ** 3, Etrans=0.1 * u.TeV, **kwargs):
        self.particle_distribution = particle_distribution
        if _per_walker(nh):
            validate_physical_type("nh", nh, "number density")
            self.nh = nh
        else:
            self.nh = validate_scalar("nh", nh, physical_type="number density")
        self.Etrans = validate_scalar("Etrans", Etrans, domain="positive", physical_type="energy")
        self.__dict__.update(**kwargs)

    def _own_batch_sizes(self):
        return (_batch_of(self.nh),)

    def _own_device_values(self):
        return (self.nh,)

    def _launch(self, E_eV, want_wp=False):
        pd = self.particle_distribution
        if not hasattr(pd, "device_rows"):
            raise TypeError("PionDecayKelner06 needs an analytic naima_amd.models particle "
                            "distribution (got %r)" % (type(pd).__name__,))
        ctx = get_context()
        N = self.batch_size
        rows = pd.device_rows(ctx, N, amplitude_to=_PER_EV)
        nE = E_eV.size
        Etr = float(self.Etrans.to("eV").value)
        mixed = bool(np.any(E_eV < Etr) and np.any(E_eV >= Etr))  # radiative.py:1743
        out, nhat = ctx.empty((N, nE)), ctx.empty((N,))
        wp = ctx.empty((N,)) if want_wp else None
        ctx.call("nh_pion_kelner06", PD_KIND[pd.kind], rows, N, ctx.const(E_eV), nE, Etr,
                 int(mixed), out, nE, nhat, wp)
        return ctx, N, out, nhat, wp

    def _spectrum(self, photon_energy):
        E = _validate_ene(photon_energy)
        E_eV = np.atleast_1d(E.to("eV").value).astype(float)
        ctx, N, out, nhat, _ = self._launch(E_eV)
        if not self.on_device:
            nh_ = nhat.get()
            self.nhat = nh_ if self.is_batched else float(nh_[0])
        # density_factor = nh / (1 cm^-3), radiative.py:1765
        dens = self.nh.to("1/cm3").value
        if _per_walker(self.nh):
            self.specpp = self._result(ctx, out, N, E_eV.size, E, rows=dens)
        else:
            self.specpp = self._result(ctx, out, N, E_eV.size, E, scale=float(dens))
        return self.specpp

    @property
    def Wp(self):
        """Total energy in protons above the 1.22 GeV threshold (radiative.py:1716-1728)"""
        ctx, N, _, _, wp = self._launch(np.array([1e12]), want_wp=True)
        if self.on_device:
            return u.Quantity(DVec(ctx, wp, wp.ptr, N), u.TeV).to("erg")
        v = wp.get()
        return u.Quantity(v if self.is_batched else v[0], u.TeV).to("erg")
