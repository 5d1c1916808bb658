"""worker of tests/test_gpu_loops.py::test_shared_loop_that_gives_up_is_replayed_by_all_ranks: three
ranks on the one GPU share an ensemble; rank 1's third launch of the resident loop is made to time
out (NH_RUN_FAIL_AT) -- what a GPU taken over by another process would do to one rank."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NAIMA_AMD_DEVICE"] = "0"
if os.environ.get("RANK") == "1":
    os.environ["NH_RUN_FAIL_AT"] = sys.argv[4]
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd.dist import HostComm  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

out, name, nw = sys.argv[1], sys.argv[2], int(sys.argv[3])
flow = sys.argv[5] if len(sys.argv) > 5 else "plain"
comm = HostComm()
model, p0, raw, data, prior, labels = build_problem(name, na)
nd = p0.size
s = EnsembleSampler(nw, nd, na.lnprob, args=[data, model, prior], seed=42, comm=comm,
                    naima_style=True, store_blobs=True, device=True, nan_policy="reject")
pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((nw, nd)))
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    st = s.run_mcmc(pos, 5)
    st = s.run_mcmc(st, 40)            # launches 1 and 2 of the shared loop
    chain1 = s.get_chain()             # (flush: every rank verified, the ensemble kept)
    if flow == "reset":
        # the reference's burn-in -> reset -> run (core.py:483-487, 529-530): the ensemble is kept
        # BEHIND the reset -- iteration 0, acceptance counters 0 -- and the replay starts there
        s.reset()
    st = s.run_mcmc(st, 50)            # launch 3 gives up on rank 1 ...
    st = s.run_mcmc(st, 9, store=False)
    st = s.run_mcmc(st, 12)
    chain = s.get_chain()              # ... found here by every rank: 71 steps made again
dev = s._dev
assert not dev.shared and dev.resident_failed_launches >= 1, (dev.shared, dev.resident_failed_launches)
if comm.rank == 0:
    assert any("goes back" in str(x.message) for x in w), [str(x.message) for x in w]
r = comm.rank
np.save(os.path.join(out, "coords_%d.npy" % r), st.coords)
np.save(os.path.join(out, "logp_%d.npy" % r), st.log_prob)
np.save(os.path.join(out, "chain_%d.npy" % r), chain)
np.save(os.path.join(out, "lnp_%d.npy" % r), s.get_log_prob())
blobs = s.get_blobs()
np.save(os.path.join(out, "blob0_%d.npy" % r), np.asarray(blobs[0]))
np.save(os.path.join(out, "blob1_%d.npy" % r), np.asarray(blobs[1]))
np.save(os.path.join(out, "curblob0_%d.npy" % r), np.asarray(st.blobs[0]))
np.save(os.path.join(out, "acc_%d.npy" % r), s.acceptance_fraction)
