"""The BASELINE workloads evaluated with the NumPy oracle -- TEST INFRASTRUCTURE ONLY.

Mirrors naima_amd/workloads.py one walker at a time, the way the reference is
driven by emcee (core.py:450-457).  Used by tests/ (parity checker) and by the
``cpu_baseline`` leg of bench.py (kind "port").  Never imported by naima_amd/.
"""
import os

import numpy as np

from . import naima_np as O

_E_TO_EV = {"eV": 1.0, "keV": 1e3, "MeV": 1e6, "GeV": 1e9, "TeV": 1e12}
_LUT = {}


def lut_path():
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(here), "naima_amd", "data",
                        "PionDecayKafexhiu14_LUT_NucEnh_Pythia8.npz")


def get_lut():
    if "lut" not in _LUT:
        _LUT["lut"] = O.PionLUT(lut_path())
    return _LUT["lut"]


def data_energy_eV(raw):
    return np.asarray(raw["energy"], dtype=float) * _E_TO_EV[raw["energy_unit"]]


def to_data_repr(flux_eV, raw):
    """differential flux 1/(s cm2 eV) -> the representation of the data
    (core.py:66-71 + utils.py:219-282)."""
    E = data_energy_eV(raw)
    if raw["flux_unit"] == "erg/(cm2 s)":
        return flux_eV * E ** 2 * O.ERG_PER_EV
    if raw["flux_unit"] == "1/(cm2 s TeV)":
        return flux_eV * 1e12
    raise ValueError(raw["flux_unit"])


def _flux(spec, d_kpc):
    return O.to_flux(spec, d_kpc * O.KPC_CM)


def model_cfg1(p, E_eV):
    pd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=p[0], e_0=10e12, alpha=p[1],
                        e_cutoff=10 ** p[2] * 1e12, beta=1.0)
    gam = O.electron_grid(1e9, 1e9 * O.MEC2_EV, 100)
    spec, _ = O.ic_spectrum(E_eV, gam, O.nelec_on(pd, gam), [O.thermal_seed("CMB")])
    return _flux(spec, 1.0), np.nan


def model_cfg2(p, E_eV):
    pd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=10 ** p[0], e_0=10e12,
                        alpha=p[1], e_cutoff=10 ** p[2] * 1e12, beta=1.0)
    gam = O.electron_grid(1e9, 1e15, 50)
    spec = O.synchrotron_spectrum(E_eV, gam, O.nelec_on(pd, gam), p[3] * 1e-6)
    return _flux(spec, 1.0), np.nan


def model_cfg3(p, E_eV):
    pd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=10 ** p[0], e_0=10e12,
                        alpha=p[1], e_cutoff=10 ** p[2] * 1e12, beta=p[4])
    g_ic = O.electron_grid(100e9, 1e9 * O.MEC2_EV, 100)
    g_sy = O.electron_grid(1e9, 1e9 * O.MEC2_EV, 100)
    seeds = [O.thermal_seed(s) for s in ("CMB", "FIR", "NIR")]
    ic, _ = O.ic_spectrum(E_eV, g_ic, O.nelec_on(pd, g_ic), seeds)
    sy = O.synchrotron_spectrum(E_eV, g_sy, O.nelec_on(pd, g_sy), p[3] * 1e-6)
    g_we = O.electron_grid(1e12, 1e9 * O.MEC2_EV, 100)
    We = O.electron_energy_content(pd, g_we)
    return _flux(ic, 1.0) + _flux(sy, 1.0), We


def model_cfg4(p, E_eV):
    pd = O.ParticleDist("ExponentialCutoffBrokenPowerLaw", amplitude=10 ** p[0], e_0=1e12,
                        e_break=10 ** p[1] * 1e12, alpha_1=p[2], alpha_2=p[3],
                        e_cutoff=10 ** p[4] * 1e12, beta=2.0)
    gam = O.electron_grid(0.1e9, 50e15, 100)
    ne = O.nelec_on(pd, gam)
    B = p[5] * 1e-6
    Esy = np.logspace(-7, 9, 100)
    Lsy = O.synchrotron_spectrum(Esy, gam, ne, B)
    R = 2.1 * O.PC_CM
    phn = Lsy / (4 * np.pi * R ** 2 * O.C_CGS) * 2.24  # 1/(eV cm3)
    seeds = [O.thermal_seed("CMB"),
             dict(type="thermal", T=70.0, u=0.5 * O.ERG_PER_EV, theta=None),
             dict(type="thermal", T=5000.0, u=1.0 * O.ERG_PER_EV, theta=None),
             dict(type="array", energy=Esy, density=phn)]
    ic, _ = O.ic_spectrum(E_eV, gam, ne, seeds)
    sy = O.synchrotron_spectrum(E_eV, gam, ne, B)
    return _flux(ic, 2.0) + _flux(sy, 2.0), np.nan


def model_cfg5(p, E_eV, useLUT=True):
    pd = O.ParticleDist("ExponentialCutoffBrokenPowerLaw", amplitude=10 ** p[0] * 1e-12,
                        e_0=1e12, e_break=10 ** p[1] * 1e12, alpha_1=p[2], alpha_2=p[3],
                        e_cutoff=10 ** p[4] * 1e12, beta=1.0)
    Epmin = O.M_P_GEV + O.T_TH_GEV + 1e-4
    Epmax = Epmin * 10 ** 6.005
    Ep = O.proton_grid(Epmin, Epmax, 100)
    J = O.J_on(pd, Ep)
    spec = O.pion_spectrum(E_eV, Ep, J, 1.0, diffsigma=get_lut() if useLUT else None)
    Ep2 = O.log_grid(1e3, Epmax, 100)
    Wp = O.proton_energy_content(pd, Ep2)
    return _flux(spec, 1.0), Wp


MODELS = {"cfg1": model_cfg1, "cfg2": model_cfg2, "cfg3": model_cfg3, "cfg4": model_cfg4,
          "cfg5": model_cfg5}


def lnprob(name, p, raw, prior=None, **kw):
    """(lnprob, flux[1/(s cm2 eV)], blob) for one walker, core.py:97-121."""
    E = data_energy_eV(raw)
    flux, blob = MODELS[name](p, E, **kw)
    lp = 0.0 if prior is None else prior(p)
    if np.isinf(lp):
        return lp, flux, blob
    ll = O.lnprobmodel(to_data_repr(flux, raw), raw)
    return ll + lp, flux, blob


def raw_from_npz(z, prefix="data_"):
    keys = ("energy", "energy_unit", "flux", "flux_error_lo", "flux_error_hi", "ul", "cl",
            "flux_unit")
    raw = {k: z[prefix + k] for k in keys}
    raw["energy_unit"] = str(raw["energy_unit"])
    raw["flux_unit"] = str(raw["flux_unit"])
    return raw
