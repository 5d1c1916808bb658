"""The five BASELINE.json workloads as naima-style model functions.

Every model function is written against a *namespace* ``ns`` that provides
``u`` (units), the particle-distribution classes and the radiative classes, with
naima's spelling.  The same source therefore runs on

  * the reference (``ns`` = astropy.units + naima.models; used only by
    ``tests/golden/gen_golden.py`` in the build container), and
  * this package (``ns`` = ``naima_amd``), for one walker (``pars[ndim]``) or
    for a whole ensemble at once (``pars[ndim, N]``: every ``pars[i]`` is then
    a vector over walkers and every ``.flux()`` is one batched HIP launch).

The reference scripts these follow: examples/RXJ1713_IC_minimal.py:14-25 (cfg1),
examples/RXJ1713_SynIC.py:19-46 (cfg3), examples/CrabNebula_SynSSC.py:13-45
(cfg4).  cfg2 and cfg5 are this build's own fit wrappers.
Synthetic spectra replace the real ECSV tables (same sizes, same units).
"""
import numpy as np

SEED = 20260929


# ----------------------------------------------------------------------------
# model functions
# ----------------------------------------------------------------------------
def cfg1_model(ns):
    u = ns.u

    def ElectronIC(pars, data):
        ECPL = ns.ExponentialCutoffPowerLaw(
            pars[0] / u.eV, 10.0 * u.TeV, pars[1], 10 ** pars[2] * u.TeV)
        IC = ns.InverseCompton(ECPL, seed_photon_fields=["CMB"])
        return IC.flux(data, distance=1.0 * u.kpc)

    return ElectronIC


def cfg2_model(ns):
    u = ns.u

    def ElectronSyn(pars, data):
        ECPL = ns.ExponentialCutoffPowerLaw(
            10 ** pars[0] / u.eV, 10.0 * u.TeV, pars[1], 10 ** pars[2] * u.TeV)
        SYN = ns.Synchrotron(ECPL, B=pars[3] * u.uG, Eemin=1 * u.GeV,
                             Eemax=1 * u.PeV, nEed=50)
        return SYN.flux(data, distance=1.0 * u.kpc)

    return ElectronSyn


def cfg3_model(ns):
    u = ns.u

    def ElectronSynIC(pars, data):
        amplitude = 10 ** pars[0] / u.eV
        alpha = pars[1]
        e_cutoff = (10 ** pars[2]) * u.TeV
        B = pars[3] * u.uG
        beta = pars[4]
        ECPL = ns.ExponentialCutoffPowerLaw(amplitude, 10.0 * u.TeV, alpha, e_cutoff, beta)
        IC = ns.InverseCompton(ECPL, seed_photon_fields=["CMB", "FIR", "NIR"],
                               Eemin=100 * u.GeV)
        SYN = ns.Synchrotron(ECPL, B=B)
        model = IC.flux(data, distance=1.0 * u.kpc) + SYN.flux(data, distance=1.0 * u.kpc)
        return model, IC.compute_We(Eemin=1 * u.TeV)

    return ElectronSynIC


def cfg4_model(ns):
    u = ns.u
    c_cgs = 29979245800.0 * u.cm / u.s
    Rpwn = 2.1 * u.pc
    Esy = np.logspace(-7, 9, 100) * u.eV
    eopts = {"Eemax": 50 * u.PeV, "Eemin": 0.1 * u.GeV}

    def CrabSynSSC(pars, data):
        ECBPL = ns.ExponentialCutoffBrokenPowerLaw(
            amplitude=10 ** pars[0] / u.eV, e_0=1 * u.TeV, e_break=10 ** pars[1] * u.TeV,
            alpha_1=pars[2], alpha_2=pars[3], e_cutoff=10 ** pars[4] * u.TeV, beta=2.0)
        SYN = ns.Synchrotron(ECBPL, B=pars[5] * u.uG, **eopts)
        Lsy = SYN.flux(Esy, distance=0 * u.cm)
        phn_sy = Lsy / (4 * np.pi * Rpwn ** 2 * c_cgs) * 2.24
        IC = ns.InverseCompton(
            ECBPL,
            seed_photon_fields=["CMB",
                                ["FIR", 70 * u.K, 0.5 * u.eV / u.cm ** 3],
                                ["NIR", 5000 * u.K, 1 * u.eV / u.cm ** 3],
                                ["SSC", Esy, phn_sy]],
            **eopts)
        return IC.flux(data, distance=2.0 * u.kpc) + SYN.flux(data, distance=2.0 * u.kpc)

    return CrabSynSSC


def cfg5_model(ns, useLUT=True, nEpd=100):
    u = ns.u
    Epmin = (0.9382720881604903 + 0.27966184 + 1e-4) * u.GeV
    Epmax = Epmin * 10 ** 6.005

    def ProtonPP(pars, data):
        ECBPL = ns.ExponentialCutoffBrokenPowerLaw(
            10 ** pars[0] / u.TeV, 1 * u.TeV, 10 ** pars[1] * u.TeV,
            pars[2], pars[3], 10 ** pars[4] * u.TeV)
        PP = ns.PionDecay(ECBPL, nh=1.0 / u.cm ** 3, useLUT=useLUT, Epmax=Epmax, nEpd=nEpd)
        return PP.flux(data, distance=1.0 * u.kpc), PP.compute_Wp(Epmin=1 * u.TeV)

    return ProtonPP


# ----------------------------------------------------------------------------
# synthetic spectra (plain arrays + unit strings; each namespace wraps them)
# ----------------------------------------------------------------------------
def _noise(n, tag):
    return np.random.default_rng(SEED + tag).standard_normal(n)


def tev_points():
    """28 HESS-like points, 1/(cm2 s TeV), last one an upper limit, cl 0.95."""
    return dict(energy=np.geomspace(0.33, 170.0, 28), energy_unit="TeV",
                flux_unit="1/(cm2 s TeV)", rel_err=0.10, noise=_noise(28, 1),
                ul_last=True, cl=0.95)


def xray_points(every=1):
    """179 Suzaku-XIS-like points (or every 5th), erg/(cm2 s) SED, 2.3 %."""
    e = np.geomspace(0.55, 11.2, 179)[::every]
    return dict(energy=e, energy_unit="keV", flux_unit="erg/(cm2 s)", rel_err=0.023,
                noise=_noise(179, 2)[::every], ul_last=False, cl=0.9)


def crab_points():
    """261 broadband points, SED, 5 %."""
    return dict(energy=np.geomspace(3.4e-7, 75e12, 261), energy_unit="eV",
                flux_unit="erg/(cm2 s)", rel_err=0.05, noise=_noise(261, 4),
                ul_last=False, cl=0.9)


WORKLOADS = {
    "cfg1": dict(model=cfg1_model, p0=(1e30, 3.0, np.log10(30.0)),
                 labels=["norm", "index", "log10(cutoff)"], points=[tev_points],
                 nwalkers=32, blobs=0),
    "cfg2": dict(model=cfg2_model, p0=(33.0, 2.5, np.log10(48.0), 12.0),
                 labels=["log10(norm)", "index", "log10(cutoff)", "B"],
                 points=[xray_points], nwalkers=256, blobs=0),
    "cfg3": dict(model=cfg3_model, p0=(33.0, 2.5, np.log10(48.0), 12.0, 1.0),
                 labels=["log10(norm)", "index", "log10(cutoff)", "B", "beta"],
                 points=[lambda: xray_points(5), tev_points], nwalkers=512, blobs=1),
    "cfg4": dict(model=cfg4_model,
                 p0=(np.log10(3.699e36), np.log10(0.265), 1.5, 3.233, np.log10(1863.0), 125.0),
                 labels=["log10(norm)", "log10(break)", "index1", "index2", "log10(cutoff)", "B"],
                 points=[crab_points], nwalkers=1024, blobs=0),
    "cfg5": dict(model=cfg5_model, p0=(46.5, 0.3, 1.8, 2.6, 2.0),
                 labels=["log10(norm)", "log10(break)", "index1", "index2", "log10(cutoff)"],
                 points=[tev_points], nwalkers=2048, blobs=1),
}


_E_TO_EV = {"eV": 1.0, "keV": 1e3, "MeV": 1e6, "GeV": 1e9, "TeV": 1e12}
_ERG_PER_EV = 1.602176634e-12


def build_data(name, flux_at_p0):
    """Synthetic spectrum for workload ``name`` as plain arrays.

    flux_at_p0(E_eV) -> differential flux 1/(s cm2 eV) of the model at p0.
    Point sets are converted to the representation of the first set,
    concatenated and energy-sorted, as utils.py:38-112 (validate_data_table)
    does for a list of tables.  Returns a dict of arrays + unit strings.
    """
    sets = [f() for f in WORKLOADS[name]["points"]]
    first = sets[0]
    sed = first["flux_unit"] == "erg/(cm2 s)"
    cols = {k: [] for k in ("energy", "flux", "lo", "hi", "ul", "cl")}
    for s in sets:
        e_eV = s["energy"] * _E_TO_EV[s["energy_unit"]]
        f = np.asarray(flux_at_p0(e_eV), dtype=float)
        if sed:
            true = f * e_eV ** 2 * _ERG_PER_EV
        else:  # 1/(cm2 s TeV)
            true = f * 1e12
        obs = true * (1 + s["rel_err"] * s["noise"])
        err = s["rel_err"] * true
        ul = np.zeros(e_eV.size, dtype=bool)
        if s["ul_last"]:
            ul[-1] = True
            obs[-1] = 2.0 * true[-1]
            err[-1] = 0.0
        cols["energy"].append(e_eV / _E_TO_EV[first["energy_unit"]])
        cols["flux"].append(obs)
        cols["lo"].append(err)
        cols["hi"].append(err)
        cols["ul"].append(ul)
        cols["cl"].append(np.full(e_eV.size, s["cl"]))
    cat = {k: np.concatenate(v) for k, v in cols.items()}
    order = np.argsort(cat["energy"], kind="stable")
    return dict(energy=cat["energy"][order], energy_unit=first["energy_unit"],
                flux=cat["flux"][order], flux_error_lo=cat["lo"][order],
                flux_error_hi=cat["hi"][order], ul=cat["ul"][order], cl=cat["cl"][order],
                flux_unit=first["flux_unit"])


def test_vectors(name, n=8, spread=0.03):
    """p0 followed by n-1 seeded perturbations (the parity-test walkers)."""
    p0 = np.asarray(WORKLOADS[name]["p0"], dtype=float)
    rng = np.random.default_rng(SEED + 100 + int(name[-1]))
    out = [p0]
    for _ in range(n - 1):
        out.append(p0 * (1 + spread * rng.standard_normal(p0.size)))
    return np.array(out)


def prior_for(name, ns):
    """Uniform priors in the spirit of examples/RXJ1713_SynIC.py:52-65 (amplitude and B positive,
    index within (-1, 5)), completed with physical bounds on the logarithmic parameters: the
    likelihood has a plateau where the model flux is zero (a cut-off far below the grid), a
    walker of naima's 10 % ball that lands there wanders, and ``10 ** pars[k]`` of a wandered
    coordinate overflows -- a NaN log-probability, where emcee stops the run
    (``ValueError: Probability function returned NaN``), or a zero cut-off energy, where the
    reference's own validator does.  cfg1 keeps the prior of its reference script
    (examples/RXJ1713_IC_minimal.py:28-31: its amplitude is linear)."""
    U = ns.uniform_prior
    if name == "cfg1":
        return lambda pars: U(pars[0], 0.0, np.inf)
    if name == "cfg2":   # log10(norm), index, log10(cutoff / TeV), B / uG
        return lambda pars: (U(pars[0], 0.0, 100.0) + U(pars[1], -1, 5) + U(pars[2], -3, 5)
                             + U(pars[3], 0, np.inf))
    if name == "cfg3":   # ... + beta
        return lambda pars: (U(pars[0], 0.0, 100.0) + U(pars[1], -1, 5) + U(pars[2], -3, 5)
                             + U(pars[3], 0, np.inf) + U(pars[4], 0.1, 5))
    if name == "cfg4":   # log10(norm), log10(break / TeV), index1, index2, log10(cutoff / TeV), B / uG
        return lambda pars: (U(pars[0], 0.0, 100.0) + U(pars[1], -4, 4) + U(pars[2], -1, 6)
                             + U(pars[3], -1, 6) + U(pars[4], -1, 6) + U(pars[5], 0, np.inf))
    if name == "cfg5":   # log10(norm), log10(break / TeV), index1, index2, log10(cutoff / TeV)
        return lambda pars: (U(pars[0], 0.0, 100.0) + U(pars[1], -3, 4) + U(pars[2], -1, 6)
                             + U(pars[3], -1, 6) + U(pars[4], -2, 6))
    return None
