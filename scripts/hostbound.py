import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from naima_amd import _lib
from bench import build_problem
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem("cfg3", na)
for nw in (512, 2048, 4096):
    s = EnsembleSampler(nw, 5, na.lnprob, args=[data, model, prior], seed=1, naima_style=True, store_blobs=False, device=True)
    pos = p0 * (1 + 0.005 * s._rng.normal(size=(nw, 5)))
    st = s.run_mcmc(pos, 10, store=False)
    ctx.sync()
    t0 = time.perf_counter()
    st = s.run_mcmc(st, 200, store=False)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print("walkers %5d: host enqueue %.1f us/half-step, total %.1f us/half-step -> %.2f M walker-steps/s" % (nw, (t1 - t0) / 400 * 1e6, (t2 - t0) / 400 * 1e6, nw * 200 / (t2 - t0) / 1e6))
