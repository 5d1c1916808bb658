"""A small units layer with astropy.units' spelling for the subset naima's model
functions use (``u.eV .. u.PeV, u.erg, u.G/uG/mG, u.K, u.cm/pc/kpc, u.s, u.deg/rad``,
``*``, ``/``, ``**``, ``.to()``, ``.value``, ``.unit``, ``.unit.physical_type``,
``u.Unit("1/(s cm2 eV)")``, ``u.Quantity(...)``).

astropy is not installed in this image nor on the GPU box (hard part
1) and naima's API is expressed in Quantities (``B=pars[3]*u.uG``,
``flux(data, distance=1*u.kpc)``, examples/RXJ1713_SynIC.py:24-41), so "API
unchanged" needs this shim.  It is host-side bookkeeping only: every number that
reaches a kernel is a plain float64 in the fixed units of include/naima_hip.h.

Units are (scale to CGS-Gaussian base, integer exponents over
[cm, g, s, K, rad, G]).  The magnetic flux density is given its own base
dimension so that no fractional exponents are needed.  The physical types that
the reference registers as an import side effect of core.py:24-29 ("flux",
"differential flux", "differential power", "differential energy", "number
density", "differential number density") are built in.
"""
import re

import numpy as np

__all__ = ["Unit", "Quantity", "UnitsError", "UnitConversionError", "dimensionless_unscaled"]

_NB = 6  # cm g s K rad G


class UnitsError(ValueError):
    pass


class UnitConversionError(UnitsError):
    pass


_PHYS = {
    (0, 0, 0, 0, 0, 0): "dimensionless",
    (1, 0, 0, 0, 0, 0): "length",
    (2, 0, 0, 0, 0, 0): "area",
    (3, 0, 0, 0, 0, 0): "volume",
    (0, 1, 0, 0, 0, 0): "mass",
    (0, 0, 1, 0, 0, 0): "time",
    (0, 0, 0, 1, 0, 0): "temperature",
    (0, 0, 0, 0, 1, 0): "angle",
    (0, 0, 0, 0, 0, 1): "magnetic flux density",
    (1, 0, -1, 0, 0, 0): "speed",
    (2, 1, -2, 0, 0, 0): "energy",
    (-1, 1, -2, 0, 0, 0): "pressure",  # = energy density
    (-3, 0, 0, 0, 0, 0): "number density",
    (2, 1, -3, 0, 0, 0): "power",
    (0, 1, -3, 0, 0, 0): "flux",
    (-4, -1, 1, 0, 0, 0): "differential flux",
    (-2, -1, 1, 0, 0, 0): "differential power",
    (-2, -1, 2, 0, 0, 0): "differential energy",
    (-5, -1, 2, 0, 0, 0): "differential number density",
    (-2, 0, -1, 0, 0, 0): "particle flux",
}


class _QuantityBase:
    __slots__ = ()


class Unit:
    __slots__ = ("scale", "dims", "name")
    __array_ufunc__ = None  # ndarray <op> Unit defers to Unit.__r<op>__
    __array_priority__ = 20000

    def __new__(cls, spec=None, dims=None, name=None):
        if isinstance(spec, Unit) and dims is None:
            return spec
        if isinstance(spec, str):
            return _parse(spec)
        if isinstance(spec, _QuantityBase):
            v = float(spec.value)
            return Unit(spec.unit.scale * v, spec.unit.dims, "%g %s" % (v, spec.unit.name))
        self = object.__new__(cls)
        self.scale = 1.0 if spec is None else float(spec)
        self.dims = tuple(dims) if dims is not None else (0,) * _NB
        self.name = name if name is not None else ""
        return self

    # -- algebra ----------------------------------------------------------
    def _combine(self, other, sign):
        dims = tuple(a + sign * b for a, b in zip(self.dims, other.dims))
        if sign > 0:
            return Unit(self.scale * other.scale, dims, ("%s %s" % (self.name, other.name)).strip())
        return Unit(self.scale / other.scale, dims, "%s / (%s)" % (self.name or "1", other.name))

    def __mul__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, +1)
        if isinstance(other, _QuantityBase):
            return Quantity(other.value, other.unit * self)
        return Quantity(other, self)

    def __rmul__(self, other):
        if isinstance(other, _QuantityBase):
            return Quantity(other.value, other.unit * self)
        return Quantity(other, self)

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, -1)
        if isinstance(other, _QuantityBase):
            return Quantity(1.0 / other.value, self / other.unit)
        if getattr(other, "__array_priority__", 0) == 30000:
            return Quantity(1.0 / other, self)
        return Quantity(1.0 / np.asarray(other, dtype=float), self)

    def __rtruediv__(self, other):
        inv = self ** -1
        if isinstance(other, _QuantityBase):
            return Quantity(other.value, other.unit * inv)
        return Quantity(other, inv)

    def __pow__(self, p):
        if p != int(p):
            if any((d * p) != int(d * p) for d in self.dims):
                raise UnitsError("fractional unit powers are not supported: %s**%s" % (self.name, p))
        dims = tuple(int(round(d * p)) for d in self.dims)
        return Unit(self.scale ** p, dims, "(%s)**%g" % (self.name, p))

    def __eq__(self, other):
        try:
            other = Unit(other)
        except Exception:
            return False
        return self.dims == other.dims and np.isclose(self.scale, other.scale, rtol=1e-14, atol=0)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash((self.dims, float("%.13e" % self.scale)))

    def __repr__(self):
        return 'Unit("%s")' % (self.name or "dimensionless")

    def __str__(self):
        return self.name

    # -- conversions --------------------------------------------------------
    @property
    def physical_type(self):
        return _PHYS.get(self.dims, "unknown")

    def is_equivalent(self, other):
        return self.dims == Unit(other).dims

    def _factor_to(self, other):
        other = Unit(other)
        if self.dims != other.dims:
            raise UnitConversionError("'%s' (%s) and '%s' (%s) are not convertible" % (
                self.name, self.physical_type, other.name, other.physical_type))
        return self.scale / other.scale

    def to(self, other, value=1.0):
        return value * self._factor_to(other)

    def decompose(self):
        return Unit(self.scale, self.dims, _base_name(self.dims, self.scale))


def _base_name(dims, scale=1.0):
    names = ("cm", "g", "s", "K", "rad", "G")
    parts = ["%s%s" % (n, "" if d == 1 else d) for n, d in zip(names, dims) if d]
    s = " ".join(parts)
    return s if scale == 1.0 else ("%g %s" % (scale, s)).strip()


def _mk(scale, dims, name):
    return Unit(scale, dims, name)


_E = (2, 1, -2, 0, 0, 0)
_EV = 1.602176634e-12
_REG = {}


def _reg(name, scale, dims):
    _REG[name] = _mk(scale, dims, name)


for _p, _f in (("", 1.0), ("m", 1e-3), ("k", 1e3), ("M", 1e6), ("G", 1e9), ("T", 1e12), ("P", 1e15)):
    _reg(_p + "eV", _EV * _f, _E)
_reg("erg", 1.0, _E)
_reg("J", 1e7, _E)
for _n, _s in (("cm", 1.0), ("m", 100.0), ("km", 1e5), ("mm", 0.1), ("um", 1e-4), ("AA", 1e-8),
               ("pc", 3.0856775814913673e18), ("kpc", 3.0856775814913673e21),
               ("Mpc", 3.0856775814913673e24), ("lyr", 9.4607304725808e17),
               ("au", 1.495978707e13)):
    _reg(_n, _s, (1, 0, 0, 0, 0, 0))
_reg("cm2", 1.0, (2, 0, 0, 0, 0, 0))
_reg("cm3", 1.0, (3, 0, 0, 0, 0, 0))
_reg("m2", 1e4, (2, 0, 0, 0, 0, 0))
_reg("m3", 1e6, (3, 0, 0, 0, 0, 0))
_reg("g", 1.0, (0, 1, 0, 0, 0, 0))
_reg("kg", 1e3, (0, 1, 0, 0, 0, 0))
for _n, _s in (("s", 1.0), ("ms", 1e-3), ("h", 3600.0), ("d", 86400.0), ("yr", 31557600.0)):
    _reg(_n, _s, (0, 0, 1, 0, 0, 0))
_reg("K", 1.0, (0, 0, 0, 1, 0, 0))
_reg("rad", 1.0, (0, 0, 0, 0, 1, 0))
_reg("deg", np.pi / 180.0, (0, 0, 0, 0, 1, 0))
_reg("sr", 1.0, (0, 0, 0, 0, 2, 0))
for _n, _s in (("G", 1.0), ("uG", 1e-6), ("mG", 1e-3), ("nG", 1e-9), ("T", 1e4), ("Gauss", 1.0)):
    _reg(_n, _s, (0, 0, 0, 0, 0, 1))
_reg("W", 1e7, (2, 1, -3, 0, 0, 0))
_reg("Hz", 1.0, (0, 0, -1, 0, 0, 0))
_reg("", 1.0, (0,) * _NB)
dimensionless_unscaled = _mk(1.0, (0,) * _NB, "")
one = dimensionless_unscaled

_TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][+-]?\d+)?)|([A-Za-z_]+)(-?\d+)?|(\*\*|\^)|([()/*.]))")


def _parse(spec):
    """'1/(s cm2 eV)', 'erg / (cm2 s)', 'erg/(cm3 K4)', 'cm**2', '1 / (cm2 s TeV)' ..."""
    s = spec.strip()
    if s in _REG:
        return _REG[s]
    toks = []
    pos = 0
    while pos < len(s):
        m = _TOKEN.match(s, pos)
        if not m:
            raise ValueError("cannot parse unit %r at %r" % (spec, s[pos:]))
        pos = m.end()
        num, name, power, pw, op = m.groups()
        if num is not None:
            toks.append(("num", float(num)))
        elif name is not None:
            if name not in _REG:
                raise ValueError("unknown unit %r in %r" % (name, spec))
            uu = _REG[name]
            if power:
                uu = uu ** int(power)
            toks.append(("unit", uu))
        elif pw is not None:
            toks.append(("pow", None))
        else:
            toks.append((op, None))
    toks.append(("end", None))
    idx = [0]

    def peek():
        return toks[idx[0]][0]

    def take():
        t = toks[idx[0]]
        idx[0] += 1
        return t

    def factor():
        kind, val = take()
        if kind == "num":
            base = _mk(val, (0,) * _NB, "%g" % val)
        elif kind == "unit":
            base = val
        elif kind == "(":
            base = product()
            if take()[0] != ")":
                raise ValueError("unbalanced parentheses in %r" % spec)
        else:
            raise ValueError("cannot parse unit %r" % spec)
        if peek() == "pow":
            take()
            sign = 1
            k, v = take()
            if k == "num":
                base = base ** (sign * v)
            else:
                raise ValueError("bad exponent in %r" % spec)
        return base

    def product():
        res = factor()
        while True:
            k = peek()
            if k in ("*", "."):
                take()
                res = res * factor()
            elif k == "/":
                take()
                res = res / factor()
            elif k in ("unit", "num", "("):
                res = res * factor()  # juxtaposition: 'cm2 s'
            else:
                return res

    out = product()
    if peek() != "end":
        raise ValueError("cannot parse unit %r" % spec)
    return Unit(out.scale, out.dims, s)


def get_physical_type(x):
    return Unit(x).physical_type if not isinstance(x, str) or x in _REG else x


def def_physical_type(unit, name):  # the built-in table already has naima's types
    _PHYS.setdefault(Unit(unit).dims, name)


class Quantity(_QuantityBase):
    """value (float or ndarray) with a Unit.  Deliberately not an ndarray subclass:
    plain and predictable; numpy arrays defer to it for ``*``, ``/``, ``+`` ..."""
    __slots__ = ("value", "unit")
    __array_ufunc__ = None
    __array_priority__ = 10000

    def __init__(self, value, unit=None, dtype=float):
        if isinstance(value, Quantity):
            if unit is None:
                value, unit = value.value, value.unit
            else:
                value, unit = value.to(unit).value, Unit(unit)
        elif isinstance(value, (list, tuple)) and len(value) and isinstance(value[0], Quantity):
            u0 = value[0].unit if unit is None else Unit(unit)
            value, unit = np.array([q.to(u0).value for q in value]), u0
        elif isinstance(value, str):
            m = re.match(r"\s*([-+0-9.eE]+)\s*(.*)", value)
            value, unit = float(m.group(1)), Unit(m.group(2))
        if unit is None:
            unit = dimensionless_unscaled
        if getattr(value, "__array_priority__", 0) == 30000:
            self.value = value  # device-resident lazy value (naima_amd.darray)
        else:
            v = np.asarray(value, dtype=dtype)
            self.value = v if v.ndim else v[()]
        self.unit = Unit(unit)

    # -- conversion ---------------------------------------------------------
    def to(self, unit, equivalencies=None):
        unit = Unit(unit)
        f = self.unit._factor_to(unit)
        return Quantity(self.value if f == 1.0 else self.value * f, unit)

    def to_value(self, unit):
        return self.to(unit).value

    def decompose(self):
        d = self.unit.decompose()
        return Quantity(self.value * d.scale, Unit(1.0, d.dims, _base_name(d.dims)))

    @property
    def cgs(self):
        return self.decompose()

    @property
    def si(self):
        return self.decompose()

    # -- array protocol -----------------------------------------------------
    @property
    def shape(self):
        return np.shape(self.value)

    @property
    def ndim(self):
        return np.ndim(self.value)

    @property
    def size(self):
        return np.size(self.value)

    @property
    def isscalar(self):
        return np.ndim(self.value) == 0

    @property
    def on_device(self):
        return getattr(self.value, "__array_priority__", 0) == 30000

    @property
    def T(self):
        return Quantity(np.transpose(self.value), self.unit)

    def __len__(self):
        return len(self.value)

    def __iter__(self):
        for v in self.value:
            yield Quantity(v, self.unit)

    def __getitem__(self, k):
        return Quantity(self.value[k], self.unit)

    def __setitem__(self, k, v):
        self.value[k] = v.to(self.unit).value if isinstance(v, Quantity) else v

    def flatten(self):
        return Quantity(np.ravel(self.value), self.unit)

    def squeeze(self):
        return Quantity(np.squeeze(self.value), self.unit)

    def reshape(self, *a):
        return Quantity(np.reshape(self.value, *a), self.unit)

    def copy(self):
        return Quantity(np.copy(self.value), self.unit)

    def sum(self, axis=None):
        return Quantity(np.sum(self.value, axis=axis), self.unit)

    def item(self):
        return float(self.value)

    def __float__(self):
        return float(self.to(dimensionless_unscaled).value)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.value, dtype=dtype)

    # -- arithmetic ---------------------------------------------------------
    @staticmethod
    def _split(other):
        if isinstance(other, Quantity):
            return other.value, other.unit
        if isinstance(other, Unit):
            return 1.0, other
        if getattr(other, "__array_priority__", 0) == 30000:
            return other, dimensionless_unscaled
        return np.asarray(other, dtype=float), dimensionless_unscaled

    def __mul__(self, other):
        v, un = self._split(other)
        return Quantity(self.value * v, self.unit * un)

    __rmul__ = __mul__

    def __truediv__(self, other):
        v, un = self._split(other)
        return Quantity(self.value / v, self.unit / un)

    def __rtruediv__(self, other):
        v, un = self._split(other)
        return Quantity(v / self.value, un / self.unit)

    def __pow__(self, p):
        return Quantity(self.value ** p, self.unit ** p)

    def __neg__(self):
        return Quantity(-self.value, self.unit)

    def __abs__(self):
        return Quantity(np.abs(self.value), self.unit)

    def _same(self, other):
        if isinstance(other, Quantity):
            return other.to(self.unit).value
        if self.unit.dims == (0,) * _NB:
            return np.asarray(other, dtype=float) / self.unit.scale
        if np.all(np.asarray(other) == 0):
            return np.asarray(other, dtype=float)
        raise UnitConversionError("cannot combine '%s' with a dimensionless number" % self.unit.name)

    def __add__(self, other):
        return Quantity(self.value + self._same(other), self.unit)

    __radd__ = __add__

    def __sub__(self, other):
        return Quantity(self.value - self._same(other), self.unit)

    def __rsub__(self, other):
        return Quantity(self._same(other) - self.value, self.unit)

    def _cmp(self, other, op):
        try:
            o = self._same(other)
        except UnitConversionError:
            if op not in ("eq", "ne"):
                raise
            # astropy: quantities of incompatible units are simply "not equal"
            r = np.full(np.shape(self.value), op == "ne")
            return r if r.ndim else bool(r)
        return getattr(np, {"eq": "equal", "ne": "not_equal", "lt": "less", "le": "less_equal",
                            "gt": "greater", "ge": "greater_equal"}[op])(self.value, o)

    def __eq__(self, other):
        return self._cmp(other, "eq")

    def __ne__(self, other):
        return self._cmp(other, "ne")

    def __lt__(self, other):
        return self._cmp(other, "lt")

    def __le__(self, other):
        return self._cmp(other, "le")

    def __gt__(self, other):
        return self._cmp(other, "gt")

    def __ge__(self, other):
        return self._cmp(other, "ge")

    __hash__ = None

    def __bool__(self):
        return bool(np.all(self.value))

    def __repr__(self):
        return "<Quantity %s %s>" % (np.array2string(np.asarray(self.value), threshold=8), self.unit.name)

    __str__ = __repr__


# module-level unit names:  u.eV, u.TeV, u.uG, u.kpc ...
globals().update({k: v for k, v in _REG.items() if k and k.isidentifier()})


def isquantity(x):
    return isinstance(x, Quantity)
