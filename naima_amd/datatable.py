"""Spectral data tables without astropy: the dict-of-Quantities that naima's
``validate_data_table`` (utils.py:38-213 of the reference) hands to model functions and
to ``lnprob`` -- columns ``energy, flux, flux_error_lo, flux_error_hi, ul, cl``.
"""
import numpy as np

from . import units as u


class DataTable(dict):
    """a dict of columns with a ``meta`` attribute (what the path reads of a QTable)"""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.meta = {}

    def __len__(self):
        return int(np.size(self["energy"].value)) if "energy" in self else 0


def make_data(raw):
    """plain arrays + unit strings (naima_amd.workloads.build_data) -> DataTable"""
    eu, fu = str(raw["energy_unit"]), str(raw["flux_unit"])
    t = DataTable()
    t["energy"] = np.asarray(raw["energy"], dtype=float) * u.Unit(eu)
    t["flux"] = np.asarray(raw["flux"], dtype=float) * u.Unit(fu)
    t["flux_error_lo"] = np.asarray(raw["flux_error_lo"], dtype=float) * u.Unit(fu)
    t["flux_error_hi"] = np.asarray(raw["flux_error_hi"], dtype=float) * u.Unit(fu)
    t["ul"] = np.asarray(raw["ul"], dtype=bool)
    t["cl"] = np.broadcast_to(np.asarray(raw["cl"], dtype=float), t["ul"].shape).copy()
    return t
