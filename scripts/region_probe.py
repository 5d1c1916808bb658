"""where does a timed region of bench.py (barrier + sync, K steps, sync) spend its host time?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model, p0, raw, data, prior, labels = build_problem("cfg3", na)
nw = 512
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=1, naima_style=True,
                    store_blobs=True, device=True, use_graph=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, 200, store=False)
ctx.sync()
stamps = []
orig = ctx.graph_launch
def gl(g):
    stamps.append(time.perf_counter())
    return orig(g)
ctx.graph_launch = gl
calls = {}
orig_call = ctx.call
def call(name, *a):
    t = time.perf_counter()
    r = orig_call(name, *a)
    c = calls.setdefault(name, [0, 0.0])
    c[0] += 1; c[1] += time.perf_counter() - t
    return r
ctx.call = call
rows = []
for rep in range(60):
    ctx.sync()
    del stamps[:]
    t0 = time.perf_counter()
    st = s.run_mcmc(st, K, store=True)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    rows.append((stamps[0] - t0, t1 - t0, t2 - t0, len(stamps)))
    s.reset()
r = np.array(rows[10:])
m = np.median(r, axis=0)
print("K=%d: first graph launch after %.1f us; run_mcmc returns after %.1f us; region %.1f us = %.2f us/step; %d graph launches"
      % (K, m[0] * 1e6, m[1] * 1e6, m[2] * 1e6, m[2] * 1e6 / K, m[3]))
for n, (c, t) in sorted(calls.items(), key=lambda kv: -kv[1][1])[:12]:
    print("   %-28s %5d calls  %8.1f us per region" % (n, c, t * 1e6 / 60))
