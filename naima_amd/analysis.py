"""Results on disk with the layout of naima's ``save_run`` (analysis.py:366-471 of the
reference: group ``mcmc`` with ``chain, log_prob, blobN, data`` + attributes).  With h5py
the file IS the reference's HDF5 file (naima's own ``read_run`` reads it: checked when the
golden schema is generated, tests/golden/gen_golden_run.py); h5py is not installed on the
target image, where the same objects go into a ``.npz`` under the same names.  ``read_run``
rebuilds a read-only result object with the attributes naima's post-processing reads."""
import json

import numpy as np

from . import units as u
from .datatable import DataTable

__all__ = ["save_run", "read_run", "save_results_table", "find_ML"]


def _unit_string(unit):
    """unit -> the string stored with a dataset, in astropy's spelling where the unit's name
    is a plain product / quotient of named units: ``1 / (cm2 s TeV)``, ``erg / (cm2 s)`` --
    factors by decreasing power, then alphabetically (case-insensitive), as
    ``astropy.units.UnitBase.to_string()`` writes them.  Anything else is stored as named;
    the reference reads either back with ``u.Quantity(..., unit=str)``."""
    import re
    if unit is None:
        return ""
    name = unit.name.strip()
    m = re.fullmatch(r"([^/()]*?)\s*(?:/\s*\(?([^/()]*?)\)?)?", name)
    tok = r"[A-Za-z]+\d*"
    if not m:
        return name
    num, den = (m.group(1) or "").strip(), (m.group(2) or "").strip()
    if num == "1":
        num = ""
    parts = []
    for side in (num, den):
        toks = side.split()
        if any(not re.fullmatch(tok, t) for t in toks):
            return name
        split = [re.fullmatch(r"([A-Za-z]+)(\d*)", t).groups() for t in toks]
        split.sort(key=lambda bp: (-int(bp[1] or 1), bp[0].lower()))
        parts.append(" ".join(b + p for b, p in split))
    num, den = parts
    if not den:
        return num
    den = "(%s)" % den if " " in den else den
    return "%s / %s" % (num or "1", den)


def run_layout(sampler):
    """What ``save_run`` writes, container-independent -- exactly the objects of the
    reference's HDF5 file (analysis.py:366-471):

      group ``mcmc``: attrs = run_info + ``acceptance_fraction`` + ``label<i>``
        ``chain``     [nsteps][nwalkers][ndim]
        ``log_prob``  [nsteps][nwalkers]
        ``blob<i>``   blob i of every (step, walker), FLATTENED over steps x walkers
                      (analysis.py:433-438), attr ``unit``
        ``data``      the data table as a structured array (astropy ``write_table_hdf5``)
        ``data.__table_column_meta__``  its YAML column / unit description, one line per row

    Returns (attrs, datasets) with datasets[name] = (array, dataset attrs)."""
    attrs = {}
    for k, v in dict(getattr(sampler, "run_info", {})).items():
        attrs[k] = v
    attrs["acceptance_fraction"] = float(np.mean(sampler.acceptance_fraction))
    for i, label in enumerate(sampler.labels):
        attrs["label{0}".format(i)] = str(label)
    ds = {"chain": (np.asarray(sampler.get_chain(), dtype=float), {}),
          "log_prob": (np.asarray(sampler.get_log_prob(), dtype=float), {})}
    blobs = sampler.get_blobs() or []
    units = list(getattr(sampler, "blob_units", None) or [None] * len(blobs))
    for j, b in enumerate(blobs):
        b = np.asarray(b, dtype=float)
        ds["blob{0}".format(j)] = (b.reshape((-1,) + b.shape[2:]), {"unit": _unit_string(units[j])})
    data = getattr(sampler, "data", None)
    if data is not None:
        fields, lines, ser = [], ["datatype:"], {}
        for k, v in data.items():
            if isinstance(v, u.Quantity):
                arr, un = np.asarray(v.value, dtype=float), _unit_string(v.unit)
            else:
                arr, un = np.asarray(v), None
            if arr.dtype.kind == "i":
                arr = arr.astype(np.int64)
            fields.append((k, arr))
            dt = {"f": "float64", "i": "int64", "b": "bool"}[arr.dtype.kind]
            lines.append("- {{name: {0}, {1}datatype: {2}}}".format(
                k, "unit: {0}, ".format(un) if un is not None else "", dt))
            if un is not None:
                ser[k] = un
        n = len(fields[0][1])
        table = np.zeros(n, dtype=[(k, a.dtype.str.replace("|b1", "?")) for k, a in fields])
        for k, a in fields:
            table[k] = a
        # astropy's serialisation of the Quantity columns (alphabetical, one YAML anchor per
        # distinct unit, in order of first appearance)
        lines += ["meta: !!omap", "- __serialized_columns__:"]
        anchors = {}
        for k in sorted(ser):
            un = ser[k]
            lines += ["    {0}:".format(k), "      __class__: astropy.units.quantity.Quantity"]
            if un in anchors:
                lines.append("      unit: *{0}".format(anchors[un]))
            else:
                anchors[un] = "id{0:03d}".format(len(anchors) + 1)
                lines.append("      unit: &{0} !astropy.units.Unit {{unit: {1}}}".format(anchors[un], un))
            lines.append("      value: !astropy.table.SerializedColumn {{name: {0}}}".format(k))
        ds["data"] = (table, {})
        ds["data.__table_column_meta__"] = (np.array([x.encode() for x in lines]), {})
    return attrs, ds


def save_run(filename, sampler, compression=True, clobber=False):
    """Save the sampler chain, log-probabilities, blobs, data table, parameter labels and
    run info (analysis.py:366-471).  ``*.h5`` / ``*.hdf5``: the reference's HDF5 file,
    object for object (needs h5py; naima's ``read_run`` reads it).  Anything else: the same
    objects under the same names in a ``.npz`` (h5py is not installed on the target image)."""
    import os
    attrs, ds = run_layout(sampler)
    if filename.endswith((".h5", ".hdf5")):
        try:
            import h5py
        except ImportError:
            raise ImportError("writing %s needs h5py; use a .npz file name for the same layout "
                              "without it" % filename)
        if os.path.exists(filename) and not clobber:
            import warnings
            warnings.warn("Not writing file because file exists and clobber is False")
            return None
        with h5py.File(filename, "w") as f:
            g = f.create_group("mcmc")
            for name, (arr, dattrs) in ds.items():
                kw = {"compression": "gzip"} if compression and arr.ndim > 0 and arr.dtype.kind != "S" \
                    else {}
                d = g.create_dataset(name, data=arr, **kw)
                for k, v in dattrs.items():
                    d.attrs[k] = v
            for k, v in attrs.items():
                try:
                    g.attrs[k] = v
                except TypeError:
                    g.attrs[k] = str(v)
        return filename
    if not filename.endswith(".npz"):
        filename += ".npz"
    if os.path.exists(filename) and not clobber:
        raise OSError("{0} exists; pass clobber=True to overwrite".format(filename))
    out = {}
    dattrs = {}
    for name, (arr, da) in ds.items():
        out["mcmc/" + name] = arr
        if da:
            dattrs[name] = da
    meta = {"attrs": {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in attrs.items()},
            "dataset_attrs": dattrs}
    out["mcmc.attrs"] = np.array(json.dumps(meta, default=str))
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


def _table_from(table, meta_lines):
    """structured array + astropy column meta lines -> DataTable (units restored)"""
    units = {}
    for line in [x.decode() if isinstance(x, bytes) else str(x) for x in meta_lines]:
        line = line.strip()
        if line.startswith("- {name:") and "unit:" in line:
            body = line[3:-1]
            name = body.split("name:")[1].split(",")[0].strip()
            units[name] = body.split("unit:")[1].rsplit(", datatype", 1)[0].strip()
    data = DataTable()
    for name in table.dtype.names:
        col = np.asarray(table[name])
        data[name] = u.Quantity(col, u.Unit(units[name])) if name in units else col
    return data


def read_run(filename):
    """Read a run saved by ``save_run`` (either container)."""
    if filename.endswith((".h5", ".hdf5")):
        import h5py
        with h5py.File(filename, "r") as f:
            g = f["mcmc"]
            attrs = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in g.attrs.items()}
            ds = {k: (np.array(g[k]), dict(g[k].attrs)) for k in g.keys()}
    else:
        if not filename.endswith(".npz"):
            filename += ".npz"
        z = np.load(filename, allow_pickle=False)
        meta = json.loads(str(z["mcmc.attrs"]))
        attrs = meta["attrs"]
        ds = {k[len("mcmc/"):]: (z[k], meta["dataset_attrs"].get(k[len("mcmc/"):], {}))
              for k in z.files if k.startswith("mcmc/")}
    chain, log_prob = ds["chain"][0], ds["log_prob"][0]
    nsteps, nwalkers = chain.shape[:2]
    blobs, units, j = [], [], 0
    while "blob%d" % j in ds:
        arr, da = ds["blob%d" % j]
        blobs.append(arr.reshape((nsteps, nwalkers) + arr.shape[1:]))
        un = da.get("unit", "")
        un = un.decode() if isinstance(un, bytes) else un
        units.append(u.Unit(un) if un else None)
        j += 1
    data = _table_from(ds["data"][0], ds["data.__table_column_meta__"][0]) if "data" in ds else None
    labels, i = [], 0
    while "label%d" % i in attrs:
        labels.append(attrs["label%d" % i])
        i += 1
    run_info = {k: v for k, v in attrs.items()
                if not k.startswith("label") and k != "acceptance_fraction"}
    return _Result(chain, log_prob, blobs, units, data, labels, run_info,
                   attrs.get("acceptance_fraction"))


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
