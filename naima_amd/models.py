"""Particle-distribution models with naima's class API (models.py:49-422 of the
reference), vectorised over walkers.

Every parameter may be a scalar (one walker, the reference's usage) or a 1-D
array over walkers; a model called inside ``model(pars, data)`` with
``pars[ndim, N]`` then describes N particle spectra at once and the radiative
classes evaluate all of them in one batched HIP launch.  Evaluation itself
(``__call__``) runs the ``nh_particle_weights`` kernel; there is no NumPy path.
"""
import numpy as np

from . import units as u
from ._lib import NH_PD_NPAR, PD_KIND, get_context
from .validator import validate_physical_type, validate_scalar_or_batch

__all__ = ["PowerLaw", "ExponentialCutoffPowerLaw", "BrokenPowerLaw",
           "ExponentialCutoffBrokenPowerLaw", "LogParabola"]


def _validate_ene(ene):
    """models.py:33-46 / radiative.py:43-58: Quantity, or dict/table with 'energy'."""
    if isinstance(ene, dict) or (hasattr(ene, "keys") and not isinstance(ene, u.Quantity)):
        try:
            ene = ene["energy"]
        except KeyError:
            raise TypeError("Table or dict does not have 'energy' column")
    if not isinstance(ene, u.Quantity):
        ene = u.Quantity(ene)
    validate_physical_type("energy", ene, physical_type="energy")
    return ene


class _ParticleDistribution:
    """common machinery: parameter broadcasting and device evaluation"""
    _memoize = False
    kind = None
    # (attribute, slot in the NH_PD_NPAR row, is_energy)
    _slots = ()

    def __setattr__(self, name, value):
        # any parameter change invalidates the packed device rows and weights
        self.__dict__.pop("_rows_dev", None)
        self.__dict__.pop("_w_dev", None)
        self.__dict__.pop("_slot7_packed", None)
        object.__setattr__(self, name, value)

    @property
    def batch_size(self):
        """number of walkers this distribution describes (1 if all parameters are scalars)"""
        n = 1
        for name in self.param_names:
            v = getattr(self, name)
            v = v.value if isinstance(v, u.Quantity) else v
            if np.ndim(v) > 0:
                m = np.shape(v)[0]
                if n != 1 and m != 1 and m != n:
                    raise ValueError("inconsistent walker-batch sizes in %s: %d vs %d"
                                     % (type(self).__name__, n, m))
                n = max(n, m)
        return n

    @property
    def is_batched(self):
        for name in self.param_names:
            v = getattr(self, name)
            v = v.value if isinstance(v, u.Quantity) else v
            if np.ndim(v) > 0:
                return True
        return False

    def _amplitude_unit(self):
        a = self.amplitude
        return a.unit if isinstance(a, u.Quantity) else u.dimensionless_unscaled

    def param_rows(self, N, amplitude_to=None):
        """[N, NH_PD_NPAR] float64 rows (include/naima_hip.h); amplitude converted to
        ``amplitude_to`` (e.g. 1/eV) when given, energies in eV."""
        rows = np.zeros((N, NH_PD_NPAR))
        for name, slot, is_energy in self._slots:
            v = getattr(self, name)
            if name == "amplitude":
                if isinstance(v, u.Quantity):
                    v = v.to(amplitude_to).value if amplitude_to is not None else v.value
            elif is_energy:
                v = v.to("eV").value
            elif isinstance(v, u.Quantity):
                v = v.to(u.dimensionless_unscaled).value
            rows[:, slot] = np.asarray(v, dtype=float)
        if not any(s[0] == "beta" for s in self._slots):
            rows[:, 4] = 1.0
        return rows

    def device_rows(self, ctx, N, amplitude_to=None):
        """the [N][NH_PD_NPAR] parameter rows in HBM, packed by ``nh_pack_rows`` from
        host scalars, host vectors or lazy device scalars (no host round trip)"""
        from .darray import DVec, lazy_const, nh_lazy
        cache = self.__dict__.setdefault("_rows_dev", {})
        key = (N, None if amplitude_to is None else amplitude_to.name)
        hit = cache.get(key)
        if hit is not None:
            ctx.need(hit)  # packed on another side stream of this evaluation
            return hit
        cols = (nh_lazy * NH_PD_NPAR)()
        for j in range(NH_PD_NPAR):
            cols[j] = lazy_const(1.0 if j == 4 else 0.0)
        keep = []
        # slot 7 is free: a Synchrotron built on this distribution parks its (device)
        # magnetic field there so that one pack launch serves every component
        rider = self.__dict__.get("_slot7")
        if isinstance(rider, DVec) and rider.n == N:
            cols[7] = rider.lazy()
            keep.append(rider)
            self.__dict__["_slot7_packed"] = rider
        for name, slot, is_energy in self._slots:
            v = getattr(self, name)
            if name == "amplitude":
                if isinstance(v, u.Quantity):
                    v = v.to(amplitude_to).value if amplitude_to is not None else v.value
            elif is_energy:
                v = v.to("eV").value
            elif isinstance(v, u.Quantity):
                v = v.to(u.dimensionless_unscaled).value
            if isinstance(v, DVec):
                if v.n != N:
                    raise ValueError("parameter %s has %d walkers, batch has %d" % (name, v.n, N))
                cols[slot] = v.lazy()
                keep.append(v)
            elif np.ndim(v) == 0:
                cols[slot] = lazy_const(v)
            else:
                dev = ctx.array(np.broadcast_to(np.asarray(v, dtype=float), (N,)))
                cols[slot] = nh_lazy(dev.ptr, 1, 1.0, 1.0, 0.0, 0, 0)
                keep.append(dev)
        ctx.need(*[getattr(k, "owner", k) for k in keep])
        out = ctx.pack_rows(cols, NH_PD_NPAR, N)
        cache[key] = out
        return out

    @property
    def on_device(self):
        for name in self.param_names:
            v = getattr(self, name)
            v = v.value if isinstance(v, u.Quantity) else v
            if getattr(v, "__array_priority__", 0) == 30000:
                return True
        return False

    def __call__(self, e):
        """dN/dE at energies ``e`` -- shape (n_e,) or (N, n_e) for a walker batch.
        Runs on the GPU (nh_particle_weights, n_out)."""
        e = _validate_ene(e)
        scalar_in = e.isscalar
        e_eV = np.atleast_1d(e.to("eV").value).astype(float).ravel()
        N = self.batch_size
        ctx = get_context()
        rows = self.device_rows(ctx, N)
        ed = ctx.const(e_eV)
        nG = e_eV.size
        if nG < 2:  # the kernel wants a grid; pad a single energy
            ed = ctx.const(np.concatenate([e_eV, e_eV * 2.0]))
            nG2 = 2
        else:
            nG2 = nG
        w, lw, n = ctx.empty((N, nG2)), ctx.empty((N, nG2)), ctx.empty((N, nG2))
        ctx.call("nh_particle_weights", PD_KIND[self.kind], rows, N, ed, ed, nG2, 1.0, w, lw, n)
        out = n.get()[:, :nG].reshape((N,) + e.shape if not scalar_in else (N,))
        if not self.is_batched:
            out = out[0]
        return u.Quantity(out, self._amplitude_unit())


class PowerLaw(_ParticleDistribution):
    """f(E) = A (E/E0)^-alpha   (models.py:49-106)"""
    param_names = ["amplitude", "e_0", "alpha"]
    kind = "PowerLaw"
    _slots = (("amplitude", 0, False), ("e_0", 1, True), ("alpha", 2, False))

    def __init__(self, amplitude, e_0, alpha):
        self.amplitude = amplitude
        self.e_0 = validate_scalar_or_batch("e_0", e_0, domain="positive", physical_type="energy")
        self.alpha = alpha


class ExponentialCutoffPowerLaw(_ParticleDistribution):
    """f(E) = A (E/E0)^-alpha exp(-(E/Ecutoff)^beta)   (models.py:109-177)"""
    param_names = ["amplitude", "e_0", "alpha", "e_cutoff", "beta"]
    kind = "ExponentialCutoffPowerLaw"
    _slots = (("amplitude", 0, False), ("e_0", 1, True), ("alpha", 2, False),
              ("e_cutoff", 3, True), ("beta", 4, False))

    def __init__(self, amplitude, e_0, alpha, e_cutoff, beta=1.0):
        self.amplitude = amplitude
        self.e_0 = validate_scalar_or_batch("e_0", e_0, domain="positive", physical_type="energy")
        self.alpha = alpha
        self.e_cutoff = validate_scalar_or_batch("e_cutoff", e_cutoff, domain="positive",
                                                 physical_type="energy")
        self.beta = beta


class BrokenPowerLaw(_ParticleDistribution):
    """A (E/E0)^-alpha_1 below e_break; A (Eb/E0)^(a2-a1) (E/E0)^-alpha_2 above
    (models.py:180-254)"""
    param_names = ["amplitude", "e_0", "e_break", "alpha_1", "alpha_2"]
    kind = "BrokenPowerLaw"
    _slots = (("amplitude", 0, False), ("e_0", 1, True), ("e_break", 5, True),
              ("alpha_1", 2, False), ("alpha_2", 6, False))

    def __init__(self, amplitude, e_0, e_break, alpha_1, alpha_2):
        self.amplitude = amplitude
        self.e_0 = validate_scalar_or_batch("e_0", e_0, domain="positive", physical_type="energy")
        self.e_break = validate_scalar_or_batch("e_break", e_break, domain="positive",
                                                physical_type="energy")
        self.alpha_1 = alpha_1
        self.alpha_2 = alpha_2


class ExponentialCutoffBrokenPowerLaw(_ParticleDistribution):
    """broken power law times exp(-(E/Ecutoff)^beta)   (models.py:257-354)"""
    param_names = ["amplitude", "e_0", "e_break", "alpha_1", "alpha_2", "e_cutoff", "beta"]
    kind = "ExponentialCutoffBrokenPowerLaw"
    _slots = (("amplitude", 0, False), ("e_0", 1, True), ("e_break", 5, True),
              ("alpha_1", 2, False), ("alpha_2", 6, False), ("e_cutoff", 3, True),
              ("beta", 4, False))

    def __init__(self, amplitude, e_0, e_break, alpha_1, alpha_2, e_cutoff, beta=1.0):
        self.amplitude = amplitude
        self.e_0 = validate_scalar_or_batch("e_0", e_0, domain="positive", physical_type="energy")
        self.e_break = validate_scalar_or_batch("e_break", e_break, domain="positive",
                                                physical_type="energy")
        self.alpha_1 = alpha_1
        self.alpha_2 = alpha_2
        self.e_cutoff = validate_scalar_or_batch("e_cutoff", e_cutoff, domain="positive",
                                                 physical_type="energy")
        self.beta = beta


class LogParabola(_ParticleDistribution):
    """f(E) = A (E/E0)^(-alpha - beta ln(E/E0))   (models.py:357-422)"""
    param_names = ["amplitude", "e_0", "alpha", "beta"]
    kind = "LogParabola"
    _slots = (("amplitude", 0, False), ("e_0", 1, True), ("alpha", 2, False),
              ("beta", 4, False))

    def __init__(self, amplitude, e_0, alpha, beta):
        self.amplitude = amplitude
        self.e_0 = validate_scalar_or_batch("e_0", e_0, domain="positive", physical_type="energy")
        self.alpha = alpha
        self.beta = beta


class TableModel:
    """A model from a table of energies and values, interpolated with a cubic spline in
    log-log space; zero outside the table (models.py:425-467).  As the particle
    distribution of a radiative model its SHAPE on the particle grid is
    walker-independent (evaluated once on the host, as the reference does, and cached in
    HBM); only ``amplitude`` may vary per walker or live on the device."""
    param_names = ["amplitude"]
    kind = "table"

    def __init__(self, energy, values, amplitude=1):
        from scipy.interpolate import interp1d

        from .validator import validate_array
        self._energy = validate_array("energy", energy, domain="positive", physical_type="energy")
        self._values = values
        self.amplitude = amplitude
        loge = np.log10(self._energy.to("eV").value)
        if isinstance(values, u.Quantity):
            self.unit = values.unit
            with np.errstate(divide="ignore"):
                logy = np.log10(values.value)
        else:
            self.unit = u.dimensionless_unscaled
            with np.errstate(divide="ignore"):
                logy = np.log10(values)
        self._interplogy = interp1d(loge, logy, fill_value=-np.inf, bounds_error=False,
                                    kind="cubic")

    def __setattr__(self, name, value):
        self.__dict__[name] = value
        if name == "amplitude":
            self.__dict__.pop("_w_dev", None)

    def _shape(self, e_eV):
        """10**interp(log10 E): the table's values (in ``self.unit``) at E [eV]"""
        with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
            return np.power(10, self._interplogy(np.log10(np.asarray(e_eV, dtype=float))))

    @property
    def _amp(self):
        a = self.amplitude
        return a.value if isinstance(a, u.Quantity) else a

    @property
    def on_device(self):
        return getattr(self._amp, "__array_priority__", 0) == 30000

    @property
    def batch_size(self):
        a = self._amp
        return int(np.shape(a)[0]) if np.ndim(a) > 0 else 1

    @property
    def is_batched(self):
        return np.ndim(self._amp) > 0

    def __call__(self, e):
        e = _validate_ene(e)
        interpy = self._shape(e.to("eV").value)
        a = self._amp
        if np.ndim(a) > 0 and np.ndim(interpy) > 0:
            a = np.asarray(a, dtype=float)[:, None]
        return u.Quantity(a * interpy, self.unit)


class EblAbsorptionModel(TableModel):
    """Opacity of the extragalactic background light (Dominguez et al. 2011) at a given
    redshift as a TableModel; ``transmission(e)`` is the factor to multiply a flux with
    (models.py:470-552).  No interpolation in redshift: the closest tabulated z
    (step 0.01) is used, as in the reference."""

    def __init__(self, redshift, ebl_absorption_model="Dominguez"):
        import os

        from .validator import validate_scalar
        if not isinstance(redshift, u.Quantity):
            redshift = redshift * u.dimensionless_unscaled
        self.redshift = validate_scalar("redshift", redshift, domain="positive",
                                        physical_type="dimensionless")
        self.model = ebl_absorption_model
        if self.model != "Dominguez":
            raise ValueError('Model should be one of: ["Dominguez"]')
        fname = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data",
                             "tau_dominguez11.npz")
        tab = np.load(fname)
        energy = tab["energy_TeV"] * u.TeV
        z = float(self.redshift.value)
        if z >= 0.01:
            redshift_list = np.arange(0.01, 4, 0.01)
            col = int(np.abs(redshift_list - z).argmin())
            table_values = np.array(tab["table"][:, col], dtype=float)
            table_values[table_values > 150.0] = 150.0  # high enough; avoids overflow later
            taus = 10 ** table_values * u.dimensionless_unscaled
        else:
            taus = 10 ** np.zeros(len(tab["energy_TeV"])) * u.dimensionless_unscaled
        super().__init__(energy, taus)

    def transmission(self, e):
        e = _validate_ene(e)
        ev = np.atleast_1d(e.to("eV").value).astype(float)
        e_GeV = np.atleast_1d(e.to("GeV").value)
        e_TeV = np.atleast_1d(e.to("TeV").value)
        taus = np.zeros(len(ev))
        for i in range(len(ev)):
            if e_GeV[i] < 1.0:
                taus[i] = 0.0
            elif e_TeV[i] > 100.0:
                taus[i] = np.log10(6000.0)
            else:
                taus[i] = np.log10(float(np.asarray(self(ev[i] * u.eV).value)))
        return np.exp(-taus)
