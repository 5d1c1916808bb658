"""host enqueue cost vs total time of the device loop (is the loop host- or GPU-bound?)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from naima_amd import _lib
from bench import build_problem
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
for name, walkers in (("cfg1", (32, 512)), ("cfg3", (512, 2048)), ("cfg5", (256,))):
    model, p0, raw, data, prior, labels = build_problem(name, na)
    for nw in walkers:
        for gs in (1, 8):
            s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=1,
                                naima_style=True, store_blobs=False, device=True)
            pos = p0 * (1 + 0.005 * s._rng.normal(size=(nw, p0.size)))
            st = s.run_mcmc(pos, 16, store=False, yield_every=gs)
            ctx.sync()
            t0 = time.perf_counter()
            st = s.run_mcmc(st, 192, store=False, yield_every=gs)
            t1 = time.perf_counter()
            ctx.sync()
            t2 = time.perf_counter()
            print("%s walkers %5d steps/graph %d: host enqueue %.1f us/half-step, total %.1f us/half-step -> %.2f M walker-steps/s" % (
                name, nw, gs, (t1 - t0) / 384 * 1e6, (t2 - t0) / 384 * 1e6, nw * 192 / (t2 - t0) / 1e6))
