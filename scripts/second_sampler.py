"""Is "the sampler a process builds second is slower"  a property of the
second sampler or of running second?  Builds samplers one after the other on one workload and
times them alternately; prints walker-steps/s of every timed run and each plan's buffers.

    python scripts/second_sampler.py cfg2 256 [steps] [blobs0 blobs1 ...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd import _lib  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
blobs = [bool(int(x)) for x in sys.argv[4:]] or [True, False, True]
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem(name, na)
pos = p0 + 0.1 * p0 * np.random.default_rng(20260929).normal(size=(nw, p0.size))


def timed(s, st, n):
    ctx.sync()
    t0 = time.perf_counter()
    st = s.run_mcmc(st, n, store=True)
    ctx.sync()
    dt = time.perf_counter() - t0
    s.reset()
    return nw * n / dt, st


S, ST = [], []
for i, b in enumerate(blobs):
    s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=20260929,
                        naima_style=True, store_blobs=b, device=True)
    st = s.run_mcmc(pos, 40, store=False)
    for _ in range(3):
        _, st = timed(s, st, steps)
    S.append(s)
    ST.append(st)
    hs = s._dev._plan["hs"] if s._dev and s._dev._plan else None
    print("sampler %d (blobs=%s) built; plan %s" % (i, b, {k: hs[k] for k in hs if k != "plan"} if hs else None),
          flush=True)
    # every sampler built so far, alternately, three rounds
    for rnd in range(3):
        row = []
        for k, sk in enumerate(S):
            v, ST[k] = timed(sk, ST[k], steps)
            row.append("s%d %.3f M" % (k, v / 1e6))
        print("   round %d: %s" % (rnd, "  ".join(row)), flush=True)
# drop the first one, time the rest again
S[0] = None
ST[0] = None
import gc
gc.collect()
for rnd in range(2):
    row = []
    for k, sk in enumerate(S):
        if sk is None:
            continue
        v, ST[k] = timed(sk, ST[k], steps)
        row.append("s%d %.3f M" % (k, v / 1e6))
    print("   after dropping s0, round %d: %s" % (rnd, "  ".join(row)), flush=True)
