"""Results on disk with the layout of naima's ``save_run`` (analysis.py:366-471 of the
reference: ``mcmc/{chain, log_prob, blobN, data}`` + attributes).  h5py is not installed
on the target image, so the same keys go into a ``.npz``; ``read_run`` rebuilds a
read-only result object with the attributes naima's post-processing reads."""
import json

import numpy as np

from . import units as u
from .datatable import DataTable

__all__ = ["save_run", "read_run"]


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
