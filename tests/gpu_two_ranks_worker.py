"""worker of tests/test_gpu_parity.py::test_device_loop_two_ranks_one_gpu"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NAIMA_AMD_DEVICE"] = "0"  # both ranks share the one GPU of the test box
# (the per-launch sharded loop is what this worker covers: launches around an all-gather per
# half-step, the fallback of the shared resident loop -- tests/gpu_shared_ranks_worker.py)
os.environ["NAIMA_AMD_SHARED"] = "0"
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd.dist import HostComm  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

out = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
comm = HostComm()  # (RCCL refuses two ranks on one GPU)
assert "torch" not in sys.modules
model, p0, raw, data, prior, labels = build_problem(name, na)
nd = p0.size
s = EnsembleSampler(32, nd, na.lnprob, args=[data, model, prior], seed=42, comm=comm,
                    naima_style=True, store_blobs=True, device=True)
pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((32, nd)))
st = s.run_mcmc(pos, 6)
np.save(os.path.join(out, "coords_%d.npy" % comm.rank), st.coords)
np.save(os.path.join(out, "logp_%d.npy" % comm.rank), st.log_prob)
np.save(os.path.join(out, "chain_%d.npy" % comm.rank), s.get_chain())
blobs = s.get_blobs()
for b, x in enumerate(blobs):
    np.save(os.path.join(out, "blob%d_%d.npy" % (b, comm.rank)), np.asarray(x, dtype=float))
assert s._dev.graph is not None and s._dev.graph2 is not None
assert s.n_walker_evals < 32 * 7  # each rank evaluated only its shard
if name == "cfg4":
    # the Crab model (its SSC seed integral is a launch of its own): two launches of the half-step
    # kernel around it (Context._stage_a), each rank over its block, around the all-gather
    assert s._dev.mega and s._dev._plan.get("staged") and s._dev._plan.get("stage") is not None

# without blobs run_mcmc takes the merged form of the sharded loop (accept + next evaluation
# as one graph between all-gathers)
s2 = EnsembleSampler(32, nd, na.lnprob, args=[data, model, prior], seed=42, comm=comm,
                     naima_style=True, store_blobs=False, device=True)
st2 = s2.run_mcmc(pos, 3)
st2 = s2.run_mcmc(st2, 37 if name != "cfg4" else 9)
assert s2._dev.graph21 is not None and not s2._dev._pending
np.save(os.path.join(out, "chain_noblobs_%d.npy" % comm.rank), s2.get_chain())
