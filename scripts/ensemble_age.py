"""Does the loop's speed depend on how far the ensemble has burnt in?  (DESIGN.md, round 2:
"whichever sampler a bench process builds second is slower" -- the second one starts from the
first one's FINAL ensemble.)  Runs one sampler for a long time and prints, per segment:
walker-steps/s, the walkers on the zero-flux plateau, and the executed synchrotron nodes
per walker counted on the host (bench.executed_flop_eq).

    python scripts/ensemble_age.py cfg2 256 [segments] [steps per segment]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na  # noqa: E402
from bench import build_problem, executed_flop_eq  # noqa: E402
from naima_amd import _lib  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nseg = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 400
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem(name, na)
pos = p0 + 0.1 * p0 * np.random.default_rng(20260929).normal(size=(nw, p0.size))
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=20260929,
                    naima_style=True, store_blobs=True, device=True)
st = s.run_mcmc(pos, 40, store=False)
done = 40
for seg in range(nseg):
    ctx.sync()
    t0 = time.perf_counter()
    st = s.run_mcmc(st, steps, store=True)
    ctx.sync()
    dt = time.perf_counter() - t0
    acc = float(np.mean(s.acceptance_fraction))
    s.reset()
    done += steps
    c, l = np.asarray(st.coords), np.asarray(st.log_prob)
    ex = executed_flop_eq(name, raw, c)
    print("after %5d steps: %.3f M walker-steps/s  acceptance %.3f  lnp < -1000: %3d  -inf: %3d  "
          "executed flop-eq per walker %s" % (done, nw * steps / dt / 1e6, acc, int((l < -1000).sum()),
                                               int(np.isinf(l).sum()),
                                               {k: round(v / 1e6, 3) for k, v in ex.items()}), flush=True)
