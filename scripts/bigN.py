"""large ensembles: 16384 and 65536 walkers of cfg3, device loop vs host loop"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from naima_amd import _lib
from bench import build_problem
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem("cfg3", na)
for nw in (16384, 65536):
    kw = dict(args=[data, model, prior], seed=5, naima_style=True, store_blobs=False)
    d = EnsembleSampler(nw, 5, na.lnprob, device=True, **kw)
    pos = p0 * (1 + 0.005 * d._rng.normal(size=(nw, 5)))
    sd = d.run_mcmc(pos, 6)
    ctx.sync(); t0 = time.perf_counter()
    sd = d.run_mcmc(sd, 24)
    ctx.sync(); dt = time.perf_counter() - t0
    print("N=%d: %.2f M walker-steps/s, acceptance %.3f, finite %s" % (
        nw, nw * 24 / dt / 1e6, np.mean(d.acceptance_fraction), np.isfinite(sd.coords).all()))
    if nw == 16384:
        h = EnsembleSampler(nw, 5, na.lnprob, **kw)
        sh = h.run_mcmc(pos, 6)
        sh = h.run_mcmc(sh, 24)
        print("   max |device - host| / |host| =", np.max(np.abs(sd.coords - sh.coords) / np.abs(sh.coords)))
    del d
