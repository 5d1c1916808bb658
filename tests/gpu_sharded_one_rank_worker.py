"""worker of tests/test_gpu_parity.py::test_sharded_path_with_rccl_on_one_rank: the sharded
device loop (split graphs, RCCL all-gathers between them, likelihood into the send buffer,
accept after the exchange, blobs gathered) with a ONE-rank RCCL communicator"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NAIMA_AMD_FORCE_SHARDED"] = "1"
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd import dist  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

out = sys.argv[1]
comm = dist.from_env("rccl")
assert type(comm).__name__ == "RcclComm", type(comm).__name__
model, p0, raw, data, prior, labels = build_problem("cfg3", na)
s = EnsembleSampler(64, 5, na.lnprob, args=[data, model, prior], seed=42, comm=comm,
                    naima_style=True, store_blobs=True, device=True)
pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((64, 5)))
st = s.run_mcmc(pos, 4)
st = s.run_mcmc(st, 40)
assert s._dev.sharded and s._dev.fused
if os.environ.get("NAIMA_AMD_RCCL_IN_GRAPH", "auto") in ("1", "auto"):  # auto: the probe passes here
    assert s._dev.coll_in_graph and s._dev.step_graph is not None and s._dev.graph2 is None
else:
    assert s._dev.graph is not None and s._dev.graph2 is not None
np.save(os.path.join(out, "chain.npy"), s.get_chain())
np.save(os.path.join(out, "logp.npy"), s.get_log_prob())
blobs = s.get_blobs()
np.save(os.path.join(out, "blob0.npy"), np.asarray(blobs[0]))
np.save(os.path.join(out, "blob1.npy"), np.asarray(blobs[1]))

# without blobs run_mcmc takes the merged form: accept of a half-step + evaluation of the
# next as one graph, the all-gather between two such graphs
s2 = EnsembleSampler(64, 5, na.lnprob, args=[data, model, prior], seed=42, comm=comm,
                     naima_style=True, store_blobs=False, device=True)
st2 = s2.run_mcmc(pos, 4)
st2 = s2.run_mcmc(st2, 70)
if not s2._dev.coll_in_graph:
    assert s2._dev.graph21 is not None and not s2._dev._pending
np.save(os.path.join(out, "chain_noblobs.npy"), s2.get_chain())
np.save(os.path.join(out, "final_noblobs.npy"), st2.coords)
