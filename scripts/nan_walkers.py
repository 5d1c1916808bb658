"""Where do the NaN proposals of a long run from the benchmark's ball come from?  Runs the device
loop with nan_policy="reject", then re-evaluates the stretch-move proposals of the last steps on
the host path and prints the parameters of those whose log-probability is NaN.

    python scripts/nan_walkers.py cfg3 512 4000
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
model, p0, raw, data, prior, labels = build_problem(name, na)
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=20260929,
                    naima_style=True, store_blobs=False, device=True, nan_policy="reject")
pos = p0 + 0.1 * p0 * s._rng.normal(size=(nw, p0.size))
with np.errstate(all="ignore"):
    st = s.run_mcmc(pos, steps)
ch = s.get_chain()
lp = s.get_log_prob()
print(name, nw, "walkers,", steps, "steps; NaN proposals rejected:", s.nan_proposals,
      "of", steps * nw, "; acceptance", np.mean(s.acceptance_fraction))
print("final ensemble, per parameter min / median / max:")
for k, lab in enumerate(labels):
    c = ch[-1][:, k]
    print("  %-16s %12.5g %12.5g %12.5g" % (lab, c.min(), np.median(c), c.max()))
print("log-prob of the final ensemble: min %.4g median %.4g max %.4g; -inf: %d" %
      (lp[-1][np.isfinite(lp[-1])].min(), np.median(lp[-1]), lp[-1].max(), np.isinf(lp[-1]).sum()))
# random stretch proposals from the final ensemble, evaluated on the host path
rng = np.random.default_rng(1)
cur = ch[-1]
i, j = rng.integers(nw, size=20000), rng.integers(nw, size=20000)
z = ((2.0 - 1.0) * rng.random(20000) + 1.0) ** 2 / 2.0
prop = cur[j] - (cur[j] - cur[i]) * z[:, None]
with np.errstate(all="ignore"):
    res = na.lnprob(prop.T, data, model, prior)
l = np.asarray(res[0], dtype=float)
bad = np.flatnonzero(np.isnan(l))
print("%d of 20000 random proposals from the final ensemble are NaN on the host path" % len(bad))
for b in bad[:10]:
    print("   ", np.array2string(prop[b], precision=5), " from walker", np.array2string(cur[i[b]], precision=4),
          "lnp", lp[-1][i[b]])
