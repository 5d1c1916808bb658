"""where a timed region of bench.py spends its time outside the resident launch: host time until
run_mcmc returns (launches issued), until the device is idle again, and the same for an empty
launch (the floor of launch + synchronise on this box)

    python scripts/region_overhead.py [workload] [walkers] [steps]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem(name, na)
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=1, naima_style=True,
                    store_blobs=True, device=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, 200, store=False)
for _ in range(8):
    st = s.run_mcmc(st, K)
    s.reset()
issue, total = [], []
for _ in range(200):
    ctx.sync()
    t0 = time.perf_counter()
    st = s.run_mcmc(st, K)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    issue.append(t1 - t0)
    total.append(t2 - t0)
    s.reset()
e = []
for _ in range(200):
    ctx.sync()
    t0 = time.perf_counter()
    ctx.call("nh_memset", s._dev.cursor, 0, 0)
    ctx.sync()
    e.append(time.perf_counter() - t0)
print("%s %d walkers, %d-step regions: run_mcmc returns after %.1f us (median), region %.1f us; "
      "an empty stream operation + synchronise %.1f us" % (name, nw, K, 1e6 * np.median(issue),
                                                         1e6 * np.median(total), 1e6 * np.median(e)))
