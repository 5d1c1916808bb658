"""long run of the device loop: finite chain, stable acceptance, no pool growth"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from naima_amd import _lib
from bench import build_problem
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem("cfg3", na)
s = EnsembleSampler(512, 5, na.lnprob, args=[data, model, prior], seed=5, naima_style=True,
                    store_blobs=False, device=True)
pos = p0 * (1 + 0.005 * s._rng.normal(size=(512, 5)))
st = s.run_mcmc(pos, 200, store=False)
s.reset()
nret0 = len(ctx._retained)
t0 = time.perf_counter()
for rep in range(4):
    st = s.run_mcmc(st, 5000)
    ctx.sync()
dt = time.perf_counter() - t0
chain = s.get_chain()
lp = s.get_log_prob()
print("steps", chain.shape, "time %.2f s -> %.2f M walker-steps/s" % (dt, 512 * 20000 / dt / 1e6))
print("finite:", np.isfinite(chain).all(), np.isfinite(lp).all(), "acceptance %.3f" % np.mean(s.acceptance_fraction))
flat = chain[5000:].reshape(-1, 5)
print("posterior median", np.median(flat, axis=0), "p0", p0)
print("posterior std   ", flat.std(axis=0))
print("retained buffers before/after", nret0, len(ctx._retained), "pool sizes", {k: len(v) for k, v in ctx._pool.items() if v})
