import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem("cfg3", na)
s = EnsembleSampler(512, p0.size, na.lnprob, args=[data, model, prior], seed=20260929, naima_style=True,
                    store_blobs=False, device=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(512, p0.size))
st = s.run_mcmc(pos, 7, store=False)
for rep in range(40):
    st = s.run_mcmc(st, 100, store=False)
    c = np.asarray(st.coords); lp = np.asarray(st.log_prob)
    host = np.asarray(na.lnprob(c.T, data, model, prior)[0])
    bad = ~np.isfinite(c).all(axis=1) | np.isnan(lp) | np.isnan(host) | (np.abs(host - lp) > 1e-6 * np.abs(host) + 1e-6)
    if bad.any():
        print("after", 7 + 100 * (rep + 1), "steps:", int(bad.sum()), "bad walkers")
        for i in np.nonzero(bad)[0][:8]:
            print("  walker", i, "device lnp", lp[i], "host lnp", host[i], "coords", c[i])
        break
else:
    print("no discrepancy in 4000 steps; min lnp", lp.min(), "nan", np.isnan(lp).sum())
