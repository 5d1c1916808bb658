"""Device-resident values for the step loop.

When the sampler keeps the ensemble in HBM it hands the model function a ``DPars``
whose ``pars[i]`` are ``DVec`` -- lazy per-walker scalars
``a * tf(b * base[w*stride] + c)`` living on the device.  The arithmetic a naima
model function does on its parameters (``10 ** pars[0] / u.eV``, ``pars[3] * u.uG``)
folds into those five numbers without launching anything; the kernel that finally
consumes the value (``nh_pack_rows``, ``nh_priors``) evaluates it.  Anything the lazy
form cannot express falls back to an eager elementwise kernel.  ``DMat`` is the
matching (N, n_E) object: a lazy linear combination of spectra that the fused
likelihood kernel (``nh_lnprob``) consumes directly, so ``IC.flux(...) +
SYN.flux(...)`` never becomes a kernel or a host array.

``np.asarray(x)`` / ``x.get()`` bring a value to the host (and synchronise).
"""
import ctypes as C

import numpy as np

from . import _lib

TF_ID, TF_POW10, TF_EXP, TF_LOG, TF_LOG10, TF_SQRT, TF_SQUARE, TF_RECIP = range(8)
OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3, "pow": 4, "max": 5, "min": 6, "lt": 7, "le": 8,
       "gt": 9, "ge": 10}


class nh_lazy(C.Structure):
    _fields_ = [("base", C.c_void_p), ("stride", C.c_longlong), ("a", C.c_double),
                ("b", C.c_double), ("c", C.c_double), ("tf", C.c_int), ("pad", C.c_int)]


class nh_comp(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_longlong), ("scale", C.c_double)]


class nh_grid(C.Structure):
    _fields_ = [("e_eV", C.c_void_p), ("xg", C.c_void_p), ("w", C.c_void_p), ("dlw", C.c_void_p),
                ("unit_scale", C.c_double), ("nG", C.c_int), ("pad", C.c_int),
                ("ln_e", C.c_void_p), ("lx", C.c_void_p)]


class nh_pack(C.Structure):
    _fields_ = [("cols", nh_lazy * 8), ("ncols", C.c_int), ("ld", C.c_int), ("out", C.c_void_p)]


class nh_moment(C.Structure):
    _fields_ = [("grid", C.c_int), ("pad", C.c_int), ("Kt", C.c_void_p), ("dlnKt", C.c_void_p),
                ("out", C.c_void_p)]


class nh_accept(C.Structure):
    _fields_ = [("coords", C.c_void_p), ("logp", C.c_void_p), ("blk", C.c_void_p),
                ("cursor", C.c_void_p), ("ns", C.c_int), ("ndim", C.c_int), ("lo", C.c_int),
                ("pad", C.c_int), ("accepted", C.c_void_p), ("naccepted", C.c_void_p),
                ("sel", C.c_void_p)]


class nh_prior(C.Structure):
    _fields_ = [("x", nh_lazy), ("p0", C.c_double), ("p1", C.c_double), ("kind", C.c_int),
                ("pad", C.c_int)]


class nh_hs_table(C.Structure):
    _fields_ = [("grid", C.c_int), ("nK", C.c_int), ("ldo", C.c_int), ("nonnegative", C.c_int),
                ("KD", C.c_void_p), ("reserved", C.c_void_p), ("scale", C.c_void_p),
                ("out", C.c_void_p)]


class nh_hs_syn(C.Structure):
    _fields_ = [("grid", C.c_int), ("nE", C.c_int), ("ldo", C.c_int), ("bcol", C.c_int),
                ("ldB", C.c_int), ("n1", C.c_int), ("E_eV", C.c_void_p), ("B", C.c_void_p),
                ("out", C.c_void_p), ("out2", C.c_void_p), ("ldo2", C.c_int), ("pad2", C.c_int)]


class nh_hs_blob(C.Structure):
    _fields_ = [("kind", C.c_int), ("mom", C.c_int), ("m", C.c_int), ("pad", C.c_int),
                ("lazy", nh_lazy), ("cur", C.c_void_p), ("hist", C.c_void_p)]


class nh_hs_desc(C.Structure):
    """include/naima_hip.h: the descriptor of nh_half_step_create"""
    _fields_ = [("coords", C.c_void_p), ("logp", C.c_void_p), ("blk", C.c_void_p),
                ("cursor", C.c_void_p), ("reserved", C.c_void_p),
                ("ns", C.c_int), ("ndim", C.c_int), ("lo", C.c_int), ("nloc", C.c_int),
                ("qT", C.c_void_p), ("factors", C.c_void_p), ("hist", C.c_void_p),
                ("accepted", C.c_void_p), ("naccepted", C.c_void_p), ("sel", C.c_void_p),
                ("do_accept", C.c_int), ("write_weights", C.c_int),
                ("packs", nh_pack * 4), ("npacks", C.c_int),
                ("kind", C.c_int), ("params", C.c_void_p),
                ("grids", nh_grid * 4), ("ngrids", C.c_int),
                ("moms", nh_moment * 4), ("nmoms", C.c_int),
                ("tab", nh_hs_table * 4), ("ntab", C.c_int),
                ("syn", nh_hs_syn),
                ("comps", nh_comp * 8), ("ncomp", C.c_int), ("nE", C.c_int),
                ("conv", C.c_void_p), ("flux", C.c_void_p), ("err_lo", C.c_void_p),
                ("err_hi", C.c_void_p), ("ul", C.c_void_p), ("cl", C.c_void_p),
                ("lp", C.c_void_p),
                ("terms", nh_prior * 16), ("nterms", C.c_int),
                ("model_out", C.c_void_p), ("total", C.c_void_p),
                ("blobs", nh_hs_blob * 4), ("nblobs", C.c_int), ("send_width", C.c_int)]


def lazy_const(v):
    return nh_lazy(None, 0, float(v), 0.0, 0.0, TF_ID, 0)


def _defer(x):
    """operands that own the operation: units and quantities (their reflected methods
    wrap the device value), so DVec/DMat return NotImplemented for them"""
    return type(x).__name__ in ("Unit", "Quantity")


def _is_number(x):
    return isinstance(x, (int, float, np.integer, np.floating)) or (
        isinstance(x, np.ndarray) and x.ndim == 0)


class DVec:
    """lazy per-walker scalar on the device (see module docstring)"""
    __array_priority__ = 30000
    ndim = 1

    def __init__(self, ctx, owner, ptr, n, stride=1, a=1.0, b=1.0, c=0.0, tf=TF_ID):
        self.ctx, self.owner, self.ptr, self.n, self.stride = ctx, owner, ptr, int(n), int(stride)
        self.a, self.b, self.c, self.tf = float(a), float(b), float(c), tf

    # ---- array protocol ----------------------------------------------------
    @property
    def shape(self):
        return (self.n,)

    @property
    def size(self):
        return self.n

    def __len__(self):
        return self.n

    def _with(self, **kw):
        d = dict(a=self.a, b=self.b, c=self.c, tf=self.tf)
        d.update(kw)
        return DVec(self.ctx, self.owner, self.ptr, self.n, self.stride, **d)

    def lazy(self):
        return nh_lazy(self.ptr, self.stride, self.a, self.b, self.c, self.tf, 0)

    def is_plain(self):
        return self.tf == TF_ID and self.a == 1.0 and self.b == 1.0 and self.c == 0.0 \
            and self.stride == 1

    def dense(self):
        """a contiguous device vector holding the values (one tiny launch unless plain)"""
        if self.is_plain():
            return self
        self.ctx.need(self.owner)
        out = self.ctx.empty((self.n,))
        cols = (nh_lazy * 1)(self.lazy())
        self.ctx.call("nh_pack_rows", cols, 1, self.n, out, 1)
        return DVec(self.ctx, out, out.ptr, self.n)

    def get(self):
        self.ctx.join()
        d = self.dense()
        host = np.empty(self.n)
        _lib._chk(_lib._lib.nh_download(self.ctx.h, host.ctypes.data, d.ptr, host.nbytes))
        return host

    def __array__(self, dtype=None, copy=None):
        return self.get()

    # ---- lazy arithmetic -----------------------------------------------------
    def _scale(self, k):
        return self._with(a=self.a * k)

    def _shift(self, k):
        if k == 0:
            return self
        if self.tf == TF_ID:  # a*(b x + c) + k
            return self._with(a=1.0, b=self.a * self.b, c=self.a * self.c + k)
        return self._binary("add", k)

    def _apply(self, tf):
        """tf(value) for a value that is still affine in x"""
        if self.tf == TF_ID:
            return self._with(a=1.0, b=self.a * self.b, c=self.a * self.c, tf=tf)
        d = self.dense()
        return d._with(tf=tf)

    def _binary(self, op, other, reverse=False):
        x = self.lazy()
        if isinstance(other, DVec):
            y = other.lazy()
            keep = (self.owner, other.owner)
        elif _is_number(other):
            y = lazy_const(other)
            keep = (self.owner,)
        else:
            o = np.asarray(other, dtype=float)
            if o.shape != (self.n,):
                raise ValueError("cannot combine a device vector of %d walkers with shape %r"
                                 % (self.n, o.shape))
            dev = self.ctx.array(o)
            y = nh_lazy(dev.ptr, 1, 1.0, 1.0, 0.0, TF_ID, 0)
            keep = (self.owner, dev)
        if reverse:
            x, y = y, x
        self.ctx.need(*keep)
        out = self.ctx.empty((self.n,))
        self.ctx.call("nh_ew_binary", OPS[op], C.byref(x), C.byref(y), self.n, out)
        del keep
        return DVec(self.ctx, out, out.ptr, self.n)

    def __mul__(self, o):
        if _defer(o):
            return NotImplemented
        return self._scale(float(o)) if _is_number(o) else self._binary("mul", o)

    __rmul__ = __mul__

    def __truediv__(self, o):
        if _defer(o):
            return NotImplemented
        return self._scale(1.0 / float(o)) if _is_number(o) else self._binary("div", o)

    def __rtruediv__(self, o):
        if _defer(o):
            return NotImplemented
        if _is_number(o) and self.tf == TF_ID and self.a != 0.0:
            return self._with(a=float(o) / self.a, tf=TF_RECIP)
        return self._binary("div", o, reverse=True)

    def __add__(self, o):
        if _defer(o) or isinstance(o, LazyPrior):
            return NotImplemented
        return self._shift(float(o)) if _is_number(o) else self._binary("add", o)

    __radd__ = __add__

    def __sub__(self, o):
        if _defer(o):
            return NotImplemented
        return self._shift(-float(o)) if _is_number(o) else self._binary("sub", o)

    def __rsub__(self, o):
        return (-self).__add__(o)

    def __neg__(self):
        return self._scale(-1.0)

    def __pow__(self, p):
        if _is_number(p):
            if p == 1:
                return self
            if p == 2 and self.tf == TF_ID:
                return self._with(a=self.a * self.a, tf=TF_SQUARE)
            if p == 0.5 and self.tf == TF_ID and self.a >= 0:
                return self._with(a=np.sqrt(self.a), tf=TF_SQRT)
            if p == -1:
                return self.__rtruediv__(1.0)
        return self._binary("pow", p)

    def __rpow__(self, base):
        if _is_number(base) and base > 0:
            if base == 10:
                return self._apply(TF_POW10)
            return (self * float(np.log(base)))._apply(TF_EXP)
        return self._binary("pow", base, reverse=True)

    def __lt__(self, o):
        return self._binary("lt", o)

    def __le__(self, o):
        return self._binary("le", o)

    def __gt__(self, o):
        return self._binary("gt", o)

    def __ge__(self, o):
        return self._binary("ge", o)

    __hash__ = None

    _UFUNCS = {"exp": TF_EXP, "log": TF_LOG, "log10": TF_LOG10, "sqrt": TF_SQRT,
               "square": TF_SQUARE, "reciprocal": TF_RECIP}
    _BIN = {"add": ("__add__", "__radd__"), "subtract": ("__sub__", "__rsub__"),
            "multiply": ("__mul__", "__rmul__"), "true_divide": ("__truediv__", "__rtruediv__"),
            "divide": ("__truediv__", "__rtruediv__"), "power": ("__pow__", "__rpow__"),
            "less": ("__lt__", "__gt__"), "less_equal": ("__le__", "__ge__"),
            "greater": ("__gt__", "__lt__"), "greater_equal": ("__ge__", "__le__")}

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        """np.log10(pars[0]), ndarray * pars[1] ... stay on the device"""
        if method != "__call__" or kwargs:
            return NotImplemented
        name = ufunc.__name__
        if name in self._UFUNCS and len(inputs) == 1:
            return self._apply(self._UFUNCS[name])
        if name == "negative":
            return -self
        if name in self._BIN and len(inputs) == 2:
            fwd, rev = self._BIN[name]
            if inputs[0] is self:
                return getattr(self, fwd)(inputs[1])
            return getattr(self, rev)(inputs[0])
        return NotImplemented


class DPars:
    """the parameter block a device-resident sampler passes to the model function:
    qT[ndim][n] in HBM; ``pars[i]`` is the DVec of parameter i over the n walkers"""

    def __init__(self, ctx, buf, ndim, n):
        self.ctx, self.buf, self.ndim, self.n = ctx, buf, int(ndim), int(n)

    def __len__(self):
        return self.ndim

    @property
    def shape(self):
        return (self.ndim, self.n)

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            if i < 0:
                i += self.ndim
            if not 0 <= i < self.ndim:
                raise IndexError(i)
            return DVec(self.ctx, self.buf, self.buf.ptr + 8 * i * self.n, self.n)
        raise TypeError("a device parameter block is indexed by parameter number")

    def __iter__(self):
        return (self[i] for i in range(self.ndim))

    def get(self):
        return self.buf.get().reshape(self.ndim, self.n)

    def __array__(self, dtype=None, copy=None):
        return self.get()


class DMat:
    """lazy (N, m) device matrix: colfac[k] * sum_j scale_j * buf_j[w*ld_j + k]"""
    __array_priority__ = 30000
    __array_ufunc__ = None
    ndim = 2

    def __init__(self, ctx, terms, shape, colfac=None):
        self.ctx, self.terms, self.shape, self.colfac = ctx, list(terms), tuple(shape), colfac

    @classmethod
    def from_buffer(cls, ctx, buf, N, m, ld=None, col0=0, scale=1.0):
        return cls(ctx, [(buf, buf.ptr + 8 * col0, ld if ld is not None else m, float(scale))],
                   (N, m))

    @property
    def size(self):
        return self.shape[0] * self.shape[1]

    def __len__(self):
        return self.shape[0]

    def _scaled(self, k):
        return DMat(self.ctx, [(b, p, ld, s * k) for b, p, ld, s in self.terms], self.shape,
                    self.colfac)

    def __mul__(self, o):
        if _defer(o):
            return NotImplemented
        if _is_number(o):
            return self._scaled(float(o))
        if isinstance(o, DVec):  # per-walker factor (a target density that is a fit parameter)
            if o.n != self.shape[0]:
                raise ValueError("factor has %d walkers, matrix has %d" % (o.n, self.shape[0]))
            return self._rows_scaled(o)
        o = np.asarray(o, dtype=float)
        if o.shape == (self.shape[1],):  # per-energy factor (e.g. E**2 of sed())
            cf = o if self.colfac is None else self.colfac * o
            return DMat(self.ctx, self.terms, self.shape, cf)
        raise ValueError("unsupported operand shape %r for a device matrix" % (o.shape,))

    __rmul__ = __mul__

    def __truediv__(self, o):
        if _defer(o):
            return NotImplemented
        if _is_number(o):
            return self._scaled(1.0 / float(o))
        return self.__mul__(1.0 / np.asarray(o, dtype=float))

    def __neg__(self):
        return self._scaled(-1.0)

    def __add__(self, o):
        if isinstance(o, DMat):
            if o.shape != self.shape:
                raise ValueError("shape mismatch %r vs %r" % (self.shape, o.shape))
            same_cf = (self.colfac is None and o.colfac is None) or (
                self.colfac is not None and o.colfac is not None
                and np.array_equal(self.colfac, o.colfac))
            if same_cf and len(self.terms) + len(o.terms) <= 8:
                return DMat(self.ctx, self.terms + o.terms, self.shape, self.colfac)
            a, b = self.dense(), o.dense()
            return DMat(self.ctx, a.terms + b.terms, self.shape)
        if _is_number(o) and o == 0:
            return self
        raise TypeError("cannot add %r to a device matrix" % type(o).__name__)

    __radd__ = __add__

    def __sub__(self, o):
        return self + (-o)

    def comps(self):
        arr = (nh_comp * len(self.terms))()
        for j, (_, p, ld, s) in enumerate(self.terms):
            arr[j] = nh_comp(p, ld, s)
        return arr

    def _rows_scaled(self, rf):
        import ctypes as C
        self.ctx.flush(*[t[0] for t in self.terms])
        N, m = self.shape
        self.ctx.need(*[t[0] for t in self.terms])
        out = self.ctx.empty((N, m))
        cf = self.ctx.const(self.colfac) if self.colfac is not None else None
        lz = rf.lazy()
        self.ctx.call("nh_lincomb", self.comps(), len(self.terms), cf, C.addressof(lz), N, m, out,
                      m)
        return DMat.from_buffer(self.ctx, out, N, m)

    def dense(self):
        """one contiguous [N][m] buffer (nh_lincomb) unless already so"""
        self.ctx.flush(*[t[0] for t in self.terms])
        N, m = self.shape
        if len(self.terms) == 1 and self.colfac is None and self.terms[0][3] == 1.0 \
                and self.terms[0][2] == m:
            return self
        self.ctx.need(*[t[0] for t in self.terms])
        out = self.ctx.empty((N, m))
        cf = self.ctx.const(self.colfac) if self.colfac is not None else None
        self.ctx.call("nh_lincomb", self.comps(), len(self.terms), cf, None, N, m, out, m)
        return DMat.from_buffer(self.ctx, out, N, m)

    def buffer(self):
        d = self.dense()
        return d.terms[0][0], d.terms[0][1]

    def get(self):
        self.ctx.join()
        d = self.dense()
        host = np.empty(self.shape)
        _lib._chk(_lib._lib.nh_download(self.ctx.h, host.ctypes.data, d.terms[0][1], host.nbytes))
        return host

    def __array__(self, dtype=None, copy=None):
        return self.get()

    def __getitem__(self, k):
        return self.get()[k]


class LazyPrior:
    """sum of prior terms on device scalars, evaluated by nh_priors"""
    __array_priority__ = 30000
    __array_ufunc__ = None

    def __init__(self, ctx, n, terms=(), const=0.0):
        self.ctx, self.n, self.terms, self.const = ctx, n, list(terms), float(const)

    def __add__(self, o):
        if isinstance(o, LazyPrior):
            return LazyPrior(self.ctx, self.n, self.terms + o.terms, self.const + o.const)
        if isinstance(o, DVec):
            return LazyPrior(self.ctx, self.n, self.terms + [(3, o, 0.0, 0.0)], self.const)
        if _is_number(o):
            return LazyPrior(self.ctx, self.n, self.terms, self.const + float(o))
        return NotImplemented

    __radd__ = __add__

    def packed(self):
        """(ctypes nh_prior array, n) when the sum fits one kernel argument block,
        else (None, 0)"""
        terms = list(self.terms)
        if self.const != 0.0:
            terms.append((3, None, self.const, 0.0))
        if len(terms) > 16:
            return None, 0
        if not terms:
            return None, 0
        arr = (nh_prior * len(terms))()
        for j, (kind, x, p0, p1) in enumerate(terms):
            lz = x.lazy() if x is not None else lazy_const(p0)
            arr[j] = nh_prior(lz, float(p0), float(p1), int(kind), 0)
        self._keep = arr
        return arr, len(terms)

    def evaluate(self):
        """dense device vector lp[n]"""
        terms = list(self.terms)
        if self.const != 0.0 or not terms:
            terms.append((3, None, self.const, 0.0))
        out = self.ctx.empty((self.n,))
        for i in range(0, len(terms), 15):  # NH_MAX_PRIOR = 16; chain through a VALUE term
            chunk = terms[i:i + 15]
            if i > 0:
                chunk = [(3, DVec(self.ctx, out, out.ptr, self.n), 0.0, 0.0)] + chunk
                nxt = self.ctx.empty((self.n,))
            else:
                nxt = out
            arr = (nh_prior * len(chunk))()
            for j, (kind, x, p0, p1) in enumerate(chunk):
                lz = x.lazy() if x is not None else lazy_const(p0)
                arr[j] = nh_prior(lz, float(p0), float(p1), int(kind), 0)
            self.ctx.call("nh_priors", arr, len(chunk), self.n, nxt)
            out = nxt
        return DVec(self.ctx, out, out.ptr, self.n)


def is_device(x):
    return isinstance(x, (DVec, DMat, DPars, LazyPrior))
