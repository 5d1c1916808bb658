"""every library call a timed region (and the reset behind it) makes, in order, with the host
time at which it was made: what sits on the stream around the resident launch"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler

class Proxy:
    def __init__(self, lib):
        self.__dict__["_l"] = lib
        self.__dict__["log"] = None
    def __getattr__(self, k):
        f = getattr(self._l, k)
        if self.log is None or not callable(f):
            return f
        log = self.log
        def w(*a):
            t = time.perf_counter()
            r = f(*a)
            log.append((k, t, time.perf_counter()))
            return r
        return w

ctx = _lib.get_context()
px = Proxy(_lib._lib)
_lib._lib = px
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 512
model, p0, raw, data, prior, labels = build_problem(name, na)
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=1, naima_style=True,
                    store_blobs=True, device=True, use_graph=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, 200, store=False)
ctx.sync()
for _ in range(16):
    st = s.run_mcmc(st, K, store=True); ctx.sync(); s.reset()
for rep in range(2):
    ctx.sync()
    px.__dict__["log"] = []
    t0 = time.perf_counter()
    st = s.run_mcmc(st, K, store=True)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    s.reset()
    t3 = time.perf_counter()
    log, px.__dict__["log"] = px.log, None
    print("region %d: run_mcmc returned at %.1f us, sync at %.1f, reset done at %.1f" % (
        rep, 1e6 * (t1 - t0), 1e6 * (t2 - t0), 1e6 * (t3 - t0)))
    for k, a, b in log:
        print("  %8.1f  +%6.1f  %s" % (1e6 * (a - t0), 1e6 * (b - a), k))
