"""Joint synchrotron + inverse-Compton fit of an RX J1713-like spectrum with naima_amd:
the naima workflow (get_sampler -> run_sampler -> save_run) with the ensemble and the step
loop on one MI355X.  The model function is what one would write for naima itself.

    python examples/rxj1713_synic.py [nwalkers] [nburn] [nrun]
    python -m torch.distributed.run --nproc-per-node 8 examples/rxj1713_synic.py 4096
        (one process per GPU: the ranks share the ensemble; reading the chain, the
        blobs and the acceptance is then a collective call -- every rank makes it)

Data: the synthetic X-ray + TeV table of BASELINE workload cfg3 (naima_amd/workloads.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as naima  # noqa: E402
from naima_amd import workloads as W  # noqa: E402
from naima_amd.datatable import make_data  # noqa: E402

u = naima.u


def ElectronSynIC(pars, data):
    ECPL = naima.ExponentialCutoffPowerLaw(10 ** pars[0] / u.eV, 10 * u.TeV, pars[1],
                                           10 ** pars[2] * u.TeV, beta=pars[4])
    IC = naima.InverseCompton(ECPL, seed_photon_fields=["CMB", "FIR", "NIR"], Eemin=100 * u.GeV)
    SYN = naima.Synchrotron(ECPL, B=pars[3] * u.uG)
    model = IC.flux(data, distance=1.0 * u.kpc) + SYN.flux(data, distance=1.0 * u.kpc)
    We = IC.compute_We(Eemin=1 * u.TeV)
    return model, We


def lnprior(pars):
    return (naima.uniform_prior(pars[1], -1, 5) + naima.uniform_prior(pars[3], 0, np.inf)
            + naima.uniform_prior(pars[4], 0.3, 3))


if __name__ == "__main__":
    nwalkers = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nburn = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    nrun = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    p0 = np.array(W.WORKLOADS["cfg3"]["p0"], dtype=float)

    def flux_at_p0(E_eV):
        return ElectronSynIC(p0, {"energy": E_eV * u.eV})[0].to("1/(s cm2 eV)").value

    data = make_data(W.build_data("cfg3", flux_at_p0))
    labels = ["log10(norm)", "index", "log10(cutoff)", "B", "beta"]
    from naima_amd import dist
    comm = dist.from_env()  # LocalComm for one process; RCCL + the control plane for several
    t0 = time.time()
    sampler, pos = naima.run_sampler(data_table=data, p0=p0, labels=labels, model=ElectronSynIC,
                                     prior=lnprior, nwalkers=nwalkers, nburn=nburn, nrun=nrun,
                                     prefit=True, seed=1, verbose=False, comm=comm)
    dt = time.time() - t0
    chain, blobs = sampler.get_chain(), sampler.get_blobs()  # (every rank: collective reads)
    acc = np.mean(sampler.acceptance_fraction)
    if comm.rank != 0:
        sys.exit(0)
    print("chain", chain.shape, "blobs", [np.shape(b) for b in blobs])
    print("%d walkers x (%d + %d) steps in %.2f s on %d GPU(s); acceptance %.2f" % (
        nwalkers, nburn, nrun, dt, comm.size, acc))
    flat = chain[nrun // 2:].reshape(-1, chain.shape[-1])
    for lab, med, lo, hi, t in zip(labels, np.median(flat, 0), *np.percentile(flat, [16, 84], 0), p0):
        print("  %-14s %8.3f  (+%.3f -%.3f)   generated with %.3f" % (lab, med, hi - med, med - lo, t))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rxj1713_synic_run")
    naima.save_run(out, sampler, clobber=True)
    back = naima.read_run(out)
    assert np.array_equal(back.get_chain(), chain)
    print("saved and read back:", out + ".npz")
