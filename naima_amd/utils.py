"""Convenience functions with naima.utils' names: ``trapz_loglog`` (utils.py:285-355 of the
reference) evaluated by the ``nh_trapz_loglog`` kernel, and ``sed_conversion``."""
import numpy as np

from . import units as u
from ._lib import get_context
from .core import sed_conversion  # noqa: F401  (re-export, utils.py:219-282)
from .datatable import (build_data_table, generate_energy_edges,  # noqa: F401
                        validate_data_table)

__all__ = ["trapz_loglog", "sed_conversion", "estimate_B", "build_data_table",
           "generate_energy_edges", "validate_data_table"]


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


def estimate_B(xray_table, vhe_table, photon_energy_density=0.261 * u.eV / u.cm ** 3):
    """Magnetic field from the ratio of X-ray to gamma-ray luminosity,
    L_x / L_gamma = u_B / u_ph = B^2 / (8 pi u_ph) (utils.py:484-542 of the reference;
    Thomson regime, both bands assumed to hold the bulk of the emission): a starting
    value of B for joint X-ray / gamma-ray fits.  Tables as for ``get_sampler``."""
    from .datatable import validate_data_table
    xray = validate_data_table(xray_table, sed=False)
    vhe = validate_data_table(vhe_table, sed=False)
    lum = []
    for t in (xray, vhe):
        e = t["energy"].to("erg")
        f = t["flux"].to("1/(s cm2 erg)")
        lum.append(float(trapz_loglog(f.value * e.value, e.value)))  # erg / (cm2 s)
    uph = photon_energy_density.to("erg/cm3").value
    return u.Quantity(np.sqrt(lum[0] / lum[1] * 8 * np.pi * uph) * 1e6, u.uG)
