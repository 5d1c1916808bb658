"""worker of tests/test_gpu_loops.py::test_sharded_loop_three_ranks_again_and_again: the
per-launch sharded loop (launches around an all-gather per half-step) by three processes on the
one GPU of the test box, a fresh sampler -- plan, counters, graphs -- every repetition"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NAIMA_AMD_DEVICE"] = "0"
os.environ["NAIMA_AMD_SHARED"] = "0"
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd.dist import HostComm  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

name, nw, reps, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
comm = HostComm()
model, p0, raw, data, prior, labels = build_problem(name, na)
nd = p0.size
pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((nw, nd)))
kw = dict(args=[data, model, prior], seed=42, naima_style=True, store_blobs=True, device=True,
          nan_policy="reject")
s = EnsembleSampler(nw, nd, na.lnprob, **kw)  # one process, the whole ensemble
s.run_mcmc(pos, nsteps)
want = np.concatenate([s.get_chain().ravel(), s.get_log_prob().ravel()])
del s
bad = []
for rep in range(reps):
    s = EnsembleSampler(nw, nd, na.lnprob, comm=comm, **kw)
    s.run_mcmc(pos, nsteps)
    have = np.concatenate([s.get_chain().ravel(), s.get_log_prob().ravel()])
    assert s._dev.sharded and not s._dev.shared
    for r, row in enumerate(comm.allgather(have.reshape(1, -1))):
        if not np.allclose(row, want, rtol=1e-9, atol=0):
            bad.append((rep, r))
    del s
assert not bad, "repetitions that left the single-process chain (repetition, rank): %s" % bad[:20]
