"""worker of tests/test_gpu_loops.py::test_shared_ensemble_two_ranks_one_gpu: the resident loop
over an ensemble shared by two ranks (here: two processes on the one GPU of the test box, each
mapping the other's rings through hipIpc)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NAIMA_AMD_DEVICE"] = "0"  # both ranks share the one GPU of the test box
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd.dist import HostComm  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

out, name, nw = sys.argv[1], sys.argv[2], int(sys.argv[3])
comm = HostComm()  # (RCCL refuses two ranks on one GPU; the shared loop needs the control plane only)
assert "torch" not in sys.modules
model, p0, raw, data, prior, labels = build_problem(name, na)
nd = p0.size
s = EnsembleSampler(nw, nd, na.lnprob, args=[data, model, prior], seed=42, comm=comm,
                    naima_style=True, store_blobs=True, device=True, nan_policy="reject")
pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((nw, nd)))
st = s.run_mcmc(pos, 5)            # warm-up and plan: launches per half-step around the exchange
st = s.run_mcmc(st, 70)            # three launches of the shared loop (32 + 32 + 6 steps)
st = s.run_mcmc(st, 9, store=False)  # ... without a history: blobs merged by stamp
st = s.run_mcmc(st, 4)
dev = s._dev
assert dev.shared and dev.resident_launches >= 5, (dev.shared, getattr(dev, "resident_reason", None))
# somebody looks at every step: launches per half-step around the exchange again (the blobs each
# rank holds are merged first), then the shared loop takes over once more
n0 = dev.resident_launches
for st in s.sample(st, iterations=3):
    assert np.all(np.isfinite(st.coords))
assert dev.resident_launches <= n0 + 1  # (the last step of a block may go as a launch of one step)
st = s.run_mcmc(st, 5)
assert dev.resident_launches >= n0 + 1
r = comm.rank
np.save(os.path.join(out, "coords_%d.npy" % r), st.coords)
np.save(os.path.join(out, "logp_%d.npy" % r), st.log_prob)
b = st.blobs
np.save(os.path.join(out, "curblob0_%d.npy" % r), np.asarray(b[0]))
np.save(os.path.join(out, "curblob1_%d.npy" % r), np.asarray(b[1]))
np.save(os.path.join(out, "chain_%d.npy" % r), s.get_chain())
np.save(os.path.join(out, "lnp_%d.npy" % r), s.get_log_prob())
blobs = s.get_blobs()
np.save(os.path.join(out, "blob0_%d.npy" % r), np.asarray(blobs[0]))
np.save(os.path.join(out, "blob1_%d.npy" % r), np.asarray(blobs[1]))
np.save(os.path.join(out, "acc_%d.npy" % r), s.acceptance_fraction)
if r == 0:
    print("shared loop:", dev.shared_info, dev.resident_info, "launches", dev.resident_launches)
assert s.n_walker_evals < nw * 98 * (1.2 / comm.size)  # each rank evaluated only its shard
