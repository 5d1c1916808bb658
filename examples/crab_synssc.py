"""Synchrotron + synchrotron-self-Compton fit of a Crab-Nebula-like spectrum with naima_amd: the
model of naima's examples/CrabNebula_SynSSC.py (its synchrotron spectrum is the seed photon field
of its own inverse-Compton emission, beside CMB / FIR / NIR) made a fit function, run through
naima's workflow (run_sampler -> save_run) with the ensemble and the step loop on one MI355X.  A
half-step of it is two launches of the half-step kernel around the SSC seed integral.

    python examples/crab_synssc.py [nwalkers] [nburn] [nrun]

Data: the synthetic radio-to-TeV table of BASELINE workload cfg4 (naima_amd/workloads.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as naima  # noqa: E402
from naima_amd import workloads as W  # noqa: E402
from naima_amd.datatable import make_data  # noqa: E402

u = naima.u
c_cgs = 29979245800.0 * u.cm / u.s
Rpwn = 2.1 * u.pc
Esy = np.logspace(-7, 9, 100) * u.eV
eopts = {"Eemax": 50 * u.PeV, "Eemin": 0.1 * u.GeV}


def CrabSynSSC(pars, data):
    ECBPL = naima.ExponentialCutoffBrokenPowerLaw(
        amplitude=10 ** pars[0] / u.eV, e_0=1 * u.TeV, e_break=10 ** pars[1] * u.TeV,
        alpha_1=pars[2], alpha_2=pars[3], e_cutoff=10 ** pars[4] * u.TeV, beta=2.0)
    SYN = naima.Synchrotron(ECBPL, B=pars[5] * u.uG, **eopts)
    # photon density of the synchrotron emission inside R = 2.1 pc (examples/CrabNebula_SynSSC.py:27-31)
    Lsy = SYN.flux(Esy, distance=0 * u.cm)
    phn_sy = Lsy / (4 * np.pi * Rpwn ** 2 * c_cgs) * 2.24
    IC = naima.InverseCompton(
        ECBPL, seed_photon_fields=["CMB", ["FIR", 70 * u.K, 0.5 * u.eV / u.cm ** 3],
                                   ["NIR", 5000 * u.K, 1 * u.eV / u.cm ** 3], ["SSC", Esy, phn_sy]],
        **eopts)
    return IC.flux(data, distance=2.0 * u.kpc) + SYN.flux(data, distance=2.0 * u.kpc)


def lnprior(pars):
    # (emcee stops at the first NaN log-probability: bounds on every parameter keep the walkers of
    # naima's 10 % initial ball where the model is defined -- workloads.prior_for("cfg4"))
    U = naima.uniform_prior
    return (U(pars[0], 0.0, 100.0) + U(pars[1], -4, 4) + U(pars[2], -1, 6) + U(pars[3], -1, 6)
            + U(pars[4], -1, 6) + U(pars[5], 0, np.inf))


if __name__ == "__main__":
    nwalkers = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nburn = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    nrun = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    p0 = np.array(W.WORKLOADS["cfg4"]["p0"], dtype=float)

    def flux_at_p0(E_eV):
        return CrabSynSSC(p0, {"energy": E_eV * u.eV}).to("1/(s cm2 eV)").value

    data = make_data(W.build_data("cfg4", flux_at_p0))
    labels = ["log10(norm)", "log10(break)", "index1", "index2", "log10(cutoff)", "B"]
    t0 = time.time()
    sampler, pos = naima.run_sampler(data_table=data, p0=p0, labels=labels, model=CrabSynSSC,
                                     prior=lnprior, nwalkers=nwalkers, nburn=nburn, nrun=nrun,
                                     prefit=False, seed=1, verbose=False)
    dt = time.time() - t0
    chain = sampler.get_chain()
    print("chain", chain.shape)
    print("%d walkers x (%d + %d) steps in %.2f s (%.0f walker-steps/s); acceptance %.2f" % (
        nwalkers, nburn, nrun, dt, nwalkers * (nburn + nrun) / dt, np.mean(sampler.acceptance_fraction)))
    flat = chain[nrun // 2:].reshape(-1, chain.shape[-1])
    for lab, med, lo, hi, t in zip(labels, np.median(flat, 0), *np.percentile(flat, [16, 84], 0), p0):
        print("  %-14s %8.3f  (+%.3f -%.3f)   generated with %.3f" % (lab, med, hi - med, med - lo, t))
    dev = getattr(sampler, "_dev", None)
    if dev is not None and dev._plan is not None:
        print("half-step: %s" % ("two launches of the half-step kernel around the SSC integral"
                                 if dev._plan.get("staged") else "separate kernels"))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "crab_synssc_run")
    naima.save_run(out, sampler, clobber=True)
    back = naima.read_run(out)
    assert np.array_equal(back.get_chain(), chain)
    print("saved and read back:", out + ".npz")
