"""Results on disk with the layout of naima's ``save_run`` (analysis.py:366-471 of the
reference: ``mcmc/{chain, log_prob, blobN, data}`` + attributes).  h5py is not installed
on the target image, so the same keys go into a ``.npz``; ``read_run`` rebuilds a
read-only result object with the attributes naima's post-processing reads."""
import json

import numpy as np

from . import units as u
from .datatable import DataTable

__all__ = ["save_run", "read_run", "save_results_table", "find_ML"]


def save_run(filename, sampler, compression=True, clobber=False):
    """Save the sampler chain, log-probabilities, blobs, data table, parameter labels and
    run info to ``filename`` (``.npz`` appended if missing)."""
    import os
    if not filename.endswith(".npz"):
        filename += ".npz"
    if os.path.exists(filename) and not clobber:
        raise OSError("{0} exists; pass clobber=True to overwrite".format(filename))
    out = {"mcmc/chain": sampler.get_chain(), "mcmc/log_prob": sampler.get_log_prob()}
    blobs = sampler.get_blobs() or []
    units = list(getattr(sampler, "blob_units", None) or [None] * len(blobs))
    for j, b in enumerate(blobs):
        out["mcmc/blob%d" % j] = np.asarray(b)
    meta = {"labels": list(sampler.labels), "run_info": dict(getattr(sampler, "run_info", {})),
            "blob_units": [None if un is None else un.name for un in units],
            "acceptance_fraction": float(np.mean(sampler.acceptance_fraction)),
            "data_units": {}}
    data = getattr(sampler, "data", None)
    if data is not None:
        for k, v in data.items():
            if isinstance(v, u.Quantity):
                out["mcmc/data/" + k] = np.asarray(v.value)
                meta["data_units"][k] = v.unit.name
            else:
                out["mcmc/data/" + k] = np.asarray(v)
    out["meta"] = np.array(json.dumps(meta))
    (np.savez_compressed if compression else np.savez)(filename, **out)
    return filename


class _Result:
    """what read_run returns: chain / log-prob / blobs with emcee's accessor names"""

    def __init__(self, chain, log_prob, blobs, blob_units, data, labels, run_info, accf):
        self._chain, self._lp, self._blobs = chain, log_prob, blobs
        self.blob_units, self.data, self.labels, self.run_info = blob_units, data, labels, run_info
        self.acceptance_fraction = accf

    def get_chain(self, flat=False, discard=0, thin=1):
        c = self._chain[discard::thin]
        return c.reshape(-1, c.shape[-1]) if flat else c

    def get_log_prob(self, flat=False, discard=0, thin=1):
        c = self._lp[discard::thin]
        return c.reshape(-1) if flat else c

    def get_blobs(self, flat=False, discard=0, thin=1):
        out = []
        for b in self._blobs:
            a = b[discard::thin]
            out.append(a.reshape((-1,) + a.shape[2:]) if flat else a)
        return out

    @property
    def chain(self):
        return np.swapaxes(self._chain, 0, 1)

    @property
    def flatchain(self):
        return self.get_chain(flat=True)

    @property
    def lnprobability(self):
        return self._lp.T


def read_run(filename):
    """Read a run saved by ``save_run``."""
    if not filename.endswith(".npz"):
        filename += ".npz"
    z = np.load(filename, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    blobs, j = [], 0
    while "mcmc/blob%d" % j in z:
        blobs.append(z["mcmc/blob%d" % j])
        j += 1
    data = DataTable()
    for key in z.files:
        if key.startswith("mcmc/data/"):
            name = key[len("mcmc/data/"):]
            un = meta["data_units"].get(name)
            data[name] = u.Quantity(z[key], u.Unit(un)) if un is not None else z[key]
    units = [None if s is None else u.Unit(s) for s in meta["blob_units"]]
    return _Result(z["mcmc/chain"], z["mcmc/log_prob"], blobs, units, data, meta["labels"],
                   meta["run_info"], meta["acceptance_fraction"])


def find_ML(sampler):
    """(max log-probability, its parameter vector): the most probable sample of the chain
    (the first return values of analysis-time ``find_ML`` in the reference, which
    ``save_results_table`` stores as ``MaxLogLikelihood`` / ``ML_pars``; no refit)"""
    lp = np.asarray(sampler.get_log_prob())
    chain = np.asarray(sampler.get_chain())
    idx = np.unravel_index(np.argmax(lp), lp.shape)
    return float(lp[idx]), chain[idx]


def save_results_table(outname, sampler, convert_log=True, last_step=False, include_blobs=True,
                       overwrite=False):
    """Write ``<outname>_results.ecsv``: median and 16th/84th-percentile distances of every
    parameter (plus the de-logged value for ``log10(x)`` / ``log(x)`` labels and every
    scalar blob), with the run information, the most probable parameters and the BIC as
    metadata (analysis.py:165-363 of the reference; ECSV written directly, astropy is not
    available).  Returns the table as a dict of columns + ``meta``."""
    import os

    import yaml
    fname = "{0}_results.ecsv".format(outname)
    if os.path.exists(fname) and not overwrite:
        raise OSError("{0} exists; pass overwrite=True".format(fname))
    labels = list(sampler.labels)
    chain = np.asarray(sampler.get_chain())
    dists = chain[-1] if last_step else chain.reshape(-1, chain.shape[-1])
    quant = [16, 50, 84]
    rows = []

    def add(label, dist):
        lo, med, hi = np.percentile(dist, quant)
        rows.append((label, float(med), float(med - lo), float(hi - med)))

    for p, label in enumerate(labels):
        add(label, dists[:, p])
        if convert_log and ("log10(" in label or "log(" in label):
            nlabel = label.split("(")[-1].split(")")[0]
            add(nlabel, 10 ** dists[:, p] if label.split("(")[0] == "log10" else np.exp(dists[:, p]))
    meta = {"n_samples": int(dists.shape[0])}
    ML, MLp = find_ML(sampler)
    meta["ML_pars"] = [float(x) for x in MLp]
    meta["MaxLogLikelihood"] = ML
    ndata = len(sampler.data["energy"]) if getattr(sampler, "data", None) is not None else 0
    if ndata:
        meta["BIC"] = float(len(MLp) * np.log(ndata) - 2 * ML)
    for k, v in dict(getattr(sampler, "run_info", {})).items():
        meta[k] = v.tolist() if isinstance(v, np.ndarray) else (v.item() if isinstance(
            v, np.generic) else v)
    if include_blobs:
        blobs = sampler.get_blobs() or []
        units = list(getattr(sampler, "blob_units", None) or [None] * len(blobs))
        for idx, b in enumerate(blobs):
            b = np.asarray(b)
            if b.ndim == 2:  # one scalar per walker and step
                add("blob{0}".format(idx), b[-1] if last_step else b.ravel())
                if units[idx] is not None:
                    meta["blob{0}_unit".format(idx)] = units[idx].name
    cols = [("label", "string", "Name of the parameter"),
            ("median", "float64", "Median of the posterior distribution function"),
            ("unc_lo", "float64", "Difference between the median and the 16th percentile of the "
                                  "pdf, ~1sigma lower uncertainty"),
            ("unc_hi", "float64", "Difference between the 84th percentile and the median of the "
                                  "pdf, ~1sigma upper uncertainty")]
    header = {"datatype": [dict(name=n, datatype=t, description=d) for n, t, d in cols],
              "meta": meta}
    with open(fname, "w") as f:
        f.write("# %ECSV 1.0\n# ---\n")
        for line in yaml.safe_dump(header, default_flow_style=None, sort_keys=False).splitlines():
            f.write("# " + line + "\n")
        f.write("label median unc_lo unc_hi\n")
        for label, med, lo, hi in rows:
            f.write('"{0}" {1!r} {2!r} {3!r}\n'.format(label, med, lo, hi))
    return {"label": [r[0] for r in rows], "median": np.array([r[1] for r in rows]),
            "unc_lo": np.array([r[2] for r in rows]), "unc_hi": np.array([r[3] for r in rows]),
            "meta": meta}
