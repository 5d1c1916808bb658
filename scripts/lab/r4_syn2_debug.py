"""round 4 diagnostics: resident loop, log-domain synchrotron items against the direct form"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem
from naima_amd.sampler import EnsembleSampler
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
model, p0, raw, data, prior, labels = build_problem(name, na)
nd = p0.size
kw = dict(args=[data, model, prior], seed=20260929, naima_style=True, store_blobs=True, nan_policy="reject")
rng = np.random.default_rng(20260929)
pos = p0 + 0.1 * p0 * rng.normal(size=(nw, nd))
out = {}
for mode in ("1", "0"):
    os.environ["NH_RUN_SYN2"] = mode
    d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    with np.errstate(all="ignore"):
        st = d.run_mcmc(pos, 4)
        st = d.run_mcmc(st, 12)
    print(mode, d._dev.resident_info, d._dev.resident_launches)
    out[mode] = (d.get_chain(), d.get_log_prob(), [np.asarray(b, dtype=float) for b in d.get_blobs()])
a, b = out["1"], out["0"]
print("chain equal:", np.array_equal(a[0], b[0]), "max rel", np.nanmax(np.abs(a[0] - b[0]) / np.abs(b[0])))
with np.errstate(all="ignore"):
    rl = np.abs(a[1] - b[1]) / np.abs(b[1])
    rs = np.abs(a[2][0] - b[2][0]) / np.abs(b[2][0])
rl[~np.isfinite(rl)] = 0
rs[~np.isfinite(rs)] = 0
print("logp max rel", rl.max(), "spec max rel", rs.max(), "We max rel", np.nanmax(np.abs(a[2][1] - b[2][1]) / np.abs(b[2][1])))
print("spec rel by energy (max over walkers, steps):")
print(np.array2string(rs.max(axis=(0, 1)), precision=2, max_line_width=200))
idx = np.argwhere(rs > 1e-10)
print(len(idx), "entries > 1e-10")
seen = set()
for s_, w_, k_ in idx[:4000]:
    if (s_, w_) in seen: continue
    seen.add((s_, w_))
    if len(seen) > 12: break
    print("step", s_, "walker", w_, "pars", a[0][s_, w_], "logp", a[1][s_, w_], b[1][s_, w_])
    print("   rel", np.array2string(rs[s_, w_], precision=1, max_line_width=250))
    print("   spec", np.array2string(b[2][0][s_, w_][:40], precision=3, max_line_width=250))
