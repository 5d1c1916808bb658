"""where a 20-step timed region's time goes on the host: the call that queues the launches
(returns before the GPU is done) and the device sync behind it"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
model, p0, raw, data, prior, labels = build_problem(name, na)
nw = 512
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=1, naima_style=True,
                    store_blobs=True, device=True, use_graph=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, 200, store=False)
ctx.sync()
for _ in range(16):
    st = s.run_mcmc(st, K, store=True); ctx.sync(); s.reset()
a, b = [], []
for _ in range(300):
    ctx.sync()
    t0 = time.perf_counter()
    st = s.run_mcmc(st, K, store=True)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    a.append(t1 - t0); b.append(t2 - t0)
    s.reset()
print(name, K, "steps: queued after %.1f us (median; 10 %% %.1f, 90 %% %.1f), done after %.1f us" % (
    1e6 * np.median(a), 1e6 * np.percentile(a, 10), 1e6 * np.percentile(a, 90), 1e6 * np.median(b)))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    st = s.run_mcmc(st, K, store=True); ctx.sync(); s.reset()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
