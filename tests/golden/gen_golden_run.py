#!/opt/conda/bin/python3.9
"""Golden schema for the results-on-disk row.  Run in the build container:

    /opt/conda/bin/python3.9 tests/golden/gen_golden_run.py

1. The REFERENCE's save_run (analysis.py:366-471) writes a small mock run (inputs stored in
   run_inputs.npz) to HDF5; every object of the file -- names, shapes, dtypes, attributes,
   the astropy column-meta lines -- goes to save_run_schema.json.
2. naima_amd.analysis.save_run writes the same run to HDF5 (h5py exists in this
   interpreter only) and the REFERENCE's read_run reads it back: chain, log-prob, blobs
   (values and units), labels, run info and the data table must come out as they went in.
   The outcome is recorded in the schema ("reference_read_run_reads_naima_amd_file")."""
import importlib
import json
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
em = types.ModuleType("emcee")
em.autocorr = types.ModuleType("emcee.autocorr")
sys.modules.setdefault("emcee", em)
sys.modules.setdefault("corner", types.ModuleType("corner"))
import matplotlib  # noqa: E402
matplotlib.use("Agg")
for m in ("extern", "extern.validator", "utils", "model_utils", "radiative", "models", "core"):
    importlib.import_module("naima." + m)
warnings.simplefilter("ignore")
ana = importlib.import_module("naima.analysis")
import astropy.units as u  # noqa: E402
import h5py  # noqa: E402
from astropy.io import ascii  # noqa: E402

from naima.utils import validate_data_table  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
data = validate_data_table(ascii.read(os.path.join(HERE, "data", "CrabNebula_HESS_ipac.dat")))
rng = np.random.default_rng(0)
nsteps, nw, nd, nE = 3, 4, 2, len(data)
chain, lp = rng.normal(size=(nsteps, nw, nd)), rng.normal(size=(nsteps, nw))
b0, b1 = rng.random((nsteps, nw, nE)), rng.random((nsteps, nw))
run_info = {"n_walkers": nw, "n_burn": 0, "n_run": nsteps, "p0": [1.0, 2.0], "guess": True}
labels = ["norm", "index"]
np.savez(os.path.join(HERE, "run_inputs.npz"), chain=chain, log_prob=lp, blob0=b0, blob1=b1,
         labels=np.array(labels), run_info=np.array(json.dumps(run_info)),
         blob_units=np.array(["1/(cm2 s TeV)", "erg"]), acceptance=np.full(nw, 0.4))


class S:
    pass


s = S()
s.get_chain = lambda **k: chain
s.get_log_prob = lambda **k: lp
bl = np.empty((nsteps, nw), dtype=object)
for i in range(nsteps):
    for j in range(nw):
        bl[i, j] = (b0[i, j] * u.Unit("1/(cm2 s TeV)"), b1[i, j] * u.erg)
s.get_blobs = lambda **k: bl
s.data, s.labels, s.run_info = data, labels, run_info
s.acceptance_fraction = np.full(nw, 0.4)
ref_file = "/tmp/ref_run.h5"
ana.save_run(ref_file, s, clobber=True)


def jsonable(v):
    if isinstance(v, bytes):
        return v.decode()
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, np.generic):
        return v.item()
    return v


schema = {"_source": "naima.analysis.save_run of the reference on tests/golden/run_inputs.npz",
          "objects": {}}
with h5py.File(ref_file, "r") as f:
    g = f["mcmc"]
    schema["group_attrs"] = {k: jsonable(v) for k, v in g.attrs.items()}
    for k, v in g.items():
        ent = {"shape": list(v.shape), "dtype": str(v.dtype) if v.dtype.names is None else
               [[n, str(v.dtype[n])] for n in v.dtype.names],
               "attrs": {a: jsonable(x) for a, x in v.attrs.items()}}
        if k.endswith("__table_column_meta__"):
            ent["lines"] = [x.decode() for x in v[()]]
        schema["objects"][k] = ent

# ---- 2. our writer, the reference's reader ---------------------------------------------
import naima_amd as na  # noqa: E402
from naima_amd import analysis as A  # noqa: E402
from naima_amd import datatable as DT  # noqa: E402


class M:
    pass


m = M()
m.get_chain, m.get_log_prob = (lambda **k: chain), (lambda **k: lp)
m.get_blobs = lambda **k: [b0, b1]
m.blob_units = [na.u.Unit("1/(cm2 s TeV)"), na.u.erg]
m.data = DT.validate_data_table(DT.read(os.path.join(HERE, "data", "CrabNebula_HESS_ipac.dat")))
m.labels, m.run_info, m.acceptance_fraction = labels, run_info, np.full(nw, 0.4)
ours = "/tmp/ours_run.h5"
A.save_run(ours, m, clobber=True)
r = ana.read_run(ours)
ok = np.array_equal(r.chain, chain) and np.array_equal(r.log_prob, lp) and r.labels == labels
for i in range(nsteps):
    for j in range(nw):
        q0, q1 = r._blobs[i][j]
        ok = ok and np.array_equal(q0.to("1/(cm2 s TeV)").value, b0[i, j]) and q1.to("erg").value == b1[i, j]
for col in data.colnames:
    a, b = data[col], r.data[col]
    ok = ok and np.array_equal(np.asarray(getattr(a, "value", a)), np.asarray(getattr(b, "value", b)))
    ok = ok and str(getattr(a, "unit", "")) == str(getattr(b, "unit", ""))
ok = ok and abs(r.acceptance_fraction - 0.4) < 1e-15 and int(r.run_info["n_run"]) == nsteps
schema["reference_read_run_reads_naima_amd_file"] = bool(ok)
json.dump(schema, open(os.path.join(HERE, "save_run_schema.json"), "w"), indent=1)
print("reference read_run reads the naima_amd file:", ok)
