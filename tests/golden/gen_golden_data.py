#!/opt/conda/bin/python3.9
"""Golden vectors for the data-ingest row: the reference's own
``validate_data_table`` (utils.py:38-213) run over the reference's own table fixtures
(tests/data/*.dat, copied as data to tests/golden/data/) -- single tables, and lists of
tables in both orders with sed = None / True / False.  Run in the build container:

    /opt/conda/bin/python3.9 tests/golden/gen_golden_data.py

Writes tests/golden/data_tables.npz: per case the validated columns as plain arrays in the
unit the reference gave them, plus that unit as a string."""
import glob
import importlib
import os
import sys
import types
import warnings

import numpy as np

for n, f in (("asscalar", lambda a: a.item()), ("alen", len), ("rank", np.ndim)):
    if not hasattr(np, n):
        setattr(np, n, f)
SRC = "/root/reference/src/naima"
pkg = types.ModuleType("naima")
pkg.__path__ = [SRC]
pkg.__file__ = SRC + "/__init__.py"
pkg.__package__ = "naima"
sys.modules["naima"] = pkg
sys.modules.setdefault("emcee", types.ModuleType("emcee"))
for m in ("extern", "extern.validator", "utils", "model_utils", "radiative", "models", "core"):
    importlib.import_module("naima." + m)
warnings.simplefilter("ignore")
from astropy.io import ascii  # noqa: E402

from naima.utils import validate_data_table  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")
out = {}
names = []


def store(tag, t):
    names.append(tag)
    for col in t.colnames:
        v = t[col]
        if hasattr(v, "unit") and v.unit is not None:
            out["%s__%s" % (tag, col)] = np.asarray(v.value, dtype=float)
            out["%s__%s__unit" % (tag, col)] = str(v.unit)
        else:
            out["%s__%s" % (tag, col)] = np.asarray(v)


tables = {}
for path in sorted(glob.glob(os.path.join(DATA, "*.dat"))):
    name = os.path.basename(path)[:-4]
    tables[name] = ascii.read(path)
    store("single:" + name, validate_data_table(tables[name]))
xr, tev, sedt = tables["CrabNebula_Fake_Xray"], tables["CrabNebula_HESS_ipac"], tables["Fake_ipac_sed"]
for tag, lst in (("xray+tev", [xr, tev]), ("tev+xray", [tev, xr]), ("sed+tev+xray", [sedt, tev, xr])):
    for sed in (None, True, False):
        store("list:%s:sed=%s" % (tag, sed), validate_data_table(lst, sed=sed))
out["cases"] = np.array(names)
np.savez_compressed(os.path.join(HERE, "data_tables.npz"), **out)
print("wrote %d cases, %d arrays" % (len(names), len(out)))
