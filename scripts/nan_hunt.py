"""Which proposals of a run from the benchmark's ball make the HOST-driven evaluation return
NaN (the host loop then raises, as emcee does), and what does the oracle say about them?

    python scripts/nan_hunt.py cfg5 256 [steps]
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import naima_amd as na  # noqa: E402
from naima_amd import _lib  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402
from oracle import workloads_np as WN  # noqa: E402
import test_gpu_loops as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 70
_lib.get_context()
model, p0, raw, data, prior = T._problem(na, name, {})
pos = T._bench_ball(name, p0, nw)
d = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=T.BENCH_SEED,
                    naima_style=True, store_blobs=True, device=True)
with np.errstate(all="ignore"):
    st = d.run_mcmc(pos, 2)
    st = d.run_mcmc(st, steps - 2)
ch = d.get_chain()
S, P, Z, L = T._move_stream(T.BENCH_SEED, nw, (2, steps - 2))
props = T._replay_proposals(pos, ch, S, P, Z)
allp = np.concatenate([pos, props])
badc = ~np.isfinite(ch).all(axis=2)
print("non-finite chain entries: %d (step, walker) pairs; first: %s" % (badc.sum(), np.argwhere(badc)[:5].tolist()))
big = np.abs(allp) > 300.0
print("rows of the evaluation set with |coordinate| > 300 or non-finite: %d" % (big | ~np.isfinite(allp)).any(axis=1).sum())
keep = np.isfinite(allp).all(axis=1) & ~big.any(axis=1)
for i in np.flatnonzero(~keep)[:5]:
    print("   dropped row %d (step %d): %s" % (i, (i - len(pos)) // nw if i >= len(pos) else -1, allp[i]))
allp = allp[keep]
with np.errstate(all="ignore"):
    res = na.lnprob(allp.T, data, model, prior)
lnp = np.asarray(res[0], dtype=float)
bad = np.flatnonzero(np.isnan(lnp))
print("%s: %d evaluations, %d NaN log-probabilities on the host path" % (name, len(allp), len(bad)))
flux = np.asarray(res[1].value if hasattr(res[1], "value") else res[1], dtype=float)


def oprior(q):
    return 0.0 if prior is None else float(np.asarray(prior(q)))


for i in bad[:6]:
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        o = WN.lnprob(name, allp[i], raw, prior=oprior)
    print(" pars", allp[i], "\n   host lnp", lnp[i], "flux NaNs", int(np.isnan(flux[i]).sum()), "of", flux.shape[1],
          "\n   oracle lnp", o[0], "oracle flux NaNs", int(np.isnan(np.asarray(o[1])).sum()),
          "\n   host flux[:4]", flux[i][:4], "oracle flux[:4]", np.asarray(o[1])[:4])
