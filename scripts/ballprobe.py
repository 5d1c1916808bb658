"""why is cfg3 slower from naima's 10 % initial ball?  ensemble statistics + ms/step by stage"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem, executed_flop_eq
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
model, p0, raw, data, prior, labels = build_problem(name, na)
for ball in (0.005, 0.1):
    s = EnsembleSampler(512, p0.size, na.lnprob, args=[data, model, prior], seed=20260929,
                        naima_style=True, store_blobs=False, device=True)
    pos = p0 + ball * p0 * s._rng.normal(size=(512, p0.size))
    st = s.run_mcmc(pos, 8, store=False)
    done = 8
    for more in (160, 500, 2000, 4000):
        st = s.run_mcmc(st, more - 8, store=False); ctx.sync()
        t0 = time.perf_counter(); st = s.run_mcmc(st, 8, store=False); ctx.sync()
        t0 = time.perf_counter(); st = s.run_mcmc(st, 200, store=False); ctx.sync()
        dt = time.perf_counter() - t0
        done += more + 200
        c = np.asarray(st.coords); lp = np.asarray(st.log_prob)
        fe = executed_flop_eq(name, raw, c).get("synchrotron", 0) / 50
        print("ball %.3f after %5d steps: %.4f ms/step  lnp med %.1f min %.3g n(-inf) %d | live nodes/walker %.0f" % (
            ball, done, dt / 200 * 1e3, np.median(lp), lp.min(), np.isinf(lp).sum(), fe))
        print("    min", np.array2string(c.min(0), precision=3), "max", np.array2string(c.max(0), precision=3),
              "std/|p0|", np.array2string(c.std(0) / np.abs(p0), precision=4))
