"""Stress of the resident loop over a shared ensemble: R processes on one GPU walk an ensemble
for many blocks of moves (no history), then every rank's final ensemble is compared with one
process's.  Under torch.distributed.run it is a rank; without, it launches the ranks itself:

    python scripts/shared_stress.py cfg3 256 4 3000      (workload, walkers, ranks, steps)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NAIMA_AMD_DEVICE"] = "0"


def run(name, nw, steps, comm, out=None):
    import naima_amd as na
    from bench import build_problem
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior, labels = build_problem(name, na)
    s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=7, comm=comm,
                        naima_style=True, store_blobs=True, device=True, nan_policy="reject")
    pos = p0 * (1 + 0.01 * np.random.default_rng(3).standard_normal((nw, p0.size)))
    st = s.run_mcmc(pos, 5, store=False)
    done = 5
    while done < steps:  # calls of uneven length: launches of 32, 32, ... and a tail each
        k = min(steps - done, 333)
        st = s.run_mcmc(st, k, store=False)
        done += k
    res = dict(coords=np.array(st.coords), logp=np.array(st.log_prob),
               blob0=np.asarray(st.blobs[0]), acc=s.acceptance_fraction)
    return s, res


if "RANK" in os.environ and len(sys.argv) > 5:
    from naima_amd.dist import HostComm
    name, nw, steps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[4]), sys.argv[5]
    comm = HostComm()
    s, res = run(name, nw, steps, comm)
    assert s._dev.shared, getattr(s._dev, "resident_reason", None)
    np.savez(os.path.join(out, "r%d.npz" % comm.rank), launches=s._dev.resident_launches, **res)
else:
    name, nw, R, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    with tempfile.TemporaryDirectory() as out:
        subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                               "--nproc-per-node=%d" % R, "--master-addr", "127.0.0.1",
                               "--master-port", "29641", os.path.abspath(__file__), name, str(nw),
                               str(R), str(steps), out], cwd=ROOT,
                              env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
        from naima_amd import dist
        s, want = run(name, nw, steps, dist.LocalComm())
        for r in range(R):
            z = np.load(os.path.join(out, "r%d.npz" % r))
            for k, w in want.items():
                fin = np.isfinite(w)
                assert np.array_equal(np.isfinite(z[k]), fin), (k, r)
                np.testing.assert_allclose(z[k][fin], w[fin], rtol=1e-9, atol=1e-300, err_msg="%s rank %d" % (k, r))
        print("%s, %d walkers, %d ranks on one GPU, %d steps (%d shared launches per rank): every "
              "rank's ensemble, log-probabilities, current blobs and acceptance == one process's"
              % (name, nw, R, steps, int(z["launches"])))
