"""host-side profile of the 20-step timed region (cProfile)"""
import cProfile, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem("cfg3", na)
nw = 512
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=1, naima_style=True,
                    store_blobs=True, device=True, use_graph=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, 200, store=False)
for _ in range(16):
    st = s.run_mcmc(st, 20, store=True); ctx.sync(); s.reset()
pr = cProfile.Profile()
for _ in range(200):
    ctx.sync()
    pr.enable()
    st = s.run_mcmc(st, 20, store=True)
    pr.disable()
    ctx.sync()
    s.reset()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
