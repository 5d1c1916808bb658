"""Convenience functions with naima.utils' names: ``trapz_loglog`` (utils.py:285-355 of the
reference) evaluated by the ``nh_trapz_loglog`` kernel, and ``sed_conversion``."""
import numpy as np

from . import units as u
from ._lib import get_context
from .core import sed_conversion  # noqa: F401  (re-export, utils.py:219-282)

__all__ = ["trapz_loglog", "sed_conversion"]


def trapz_loglog(y, x, axis=-1, intervals=False):
    """Integrate ``y(x)`` along ``axis`` with the composite trapezoid rule in log-log
    space (exact for power laws).  Quantity-aware like the reference."""
    y_unit = x_unit = u.dimensionless_unscaled
    if isinstance(y, u.Quantity):
        y, y_unit = y.value, y.unit
    if isinstance(x, u.Quantity):
        x, x_unit = x.value, x.unit
    y = np.asarray(y, dtype=float)
    x = np.asarray(x, dtype=float)
    if x.ndim != 1:
        raise ValueError("x must be one-dimensional")
    ym = np.ascontiguousarray(np.moveaxis(y, axis, -1))
    n = ym.shape[-1]
    if n != x.size:
        raise ValueError("x and y have different lengths along the integration axis")
    rows = ym.reshape(-1, n)
    ctx = get_context()
    if intervals:  # the per-segment terms, along the integration axis (utils.py:350-351)
        out = ctx.empty((rows.shape[0], n - 1))
        ctx.call("nh_trapz_loglog_intervals", ctx.array(rows), ctx.array(x), rows.shape[0], n, out)
        res = np.moveaxis(out.get().reshape(ym.shape[:-1] + (n - 1,)), -1, axis)
    else:
        out = ctx.empty((rows.shape[0],))
        ctx.call("nh_trapz_loglog", ctx.array(rows), ctx.array(x), rows.shape[0], n, out)
        res = out.get().reshape(ym.shape[:-1])
    if res.ndim == 0:
        res = float(res)
    unit = y_unit * x_unit
    if unit.dims == u.dimensionless_unscaled.dims and unit.scale == 1.0:
        return res
    return u.Quantity(res, unit)
