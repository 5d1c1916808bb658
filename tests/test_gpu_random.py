"""Randomised parity: every particle distribution x every radiative process with random
(physically sensible) parameters, grids and photon energies, the HIP path against the
oracle one walker at a time.  Tolerance 1e-9 (north_star asks for 1e-6): the integrands
are smooth; the LUT mode is held to 1e-7 (the reference's spline rings around zero)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

KINDS = ("PowerLaw", "ExponentialCutoffPowerLaw", "BrokenPowerLaw",
         "ExponentialCutoffBrokenPowerLaw", "LogParabola")


@pytest.fixture(scope="module")
def na():
    import naima_amd
    from naima_amd import _lib
    _lib.get_context()
    return naima_amd


def _draw(rng, kind, N, proton=False):
    """N random parameter sets of one kind: (oracle ParticleDist list, kwargs of arrays)"""
    amp = 10 ** rng.uniform(30, 36, N)
    e0 = 10 ** rng.uniform(11.5, 13.5, N)
    a1 = rng.uniform(1.2, 3.2, N)
    a2 = a1 + rng.uniform(-0.5, 1.5, N)
    eb = 10 ** rng.uniform(11, 13.5, N)
    ec = 10 ** rng.uniform(12.5, 14.8, N)
    be = rng.uniform(0.5, 2.5, N)
    cb = rng.uniform(0.02, 0.3, N)
    if kind == "PowerLaw":
        p = dict(amplitude=amp, e_0=e0, alpha=a1)
    elif kind == "ExponentialCutoffPowerLaw":
        p = dict(amplitude=amp, e_0=e0, alpha=a1, e_cutoff=ec, beta=be)
    elif kind == "BrokenPowerLaw":
        p = dict(amplitude=amp, e_0=e0, e_break=eb, alpha_1=a1, alpha_2=a2)
    elif kind == "ExponentialCutoffBrokenPowerLaw":
        p = dict(amplitude=amp, e_0=e0, e_break=eb, alpha_1=a1, alpha_2=a2, e_cutoff=ec, beta=be)
    else:
        p = dict(amplitude=amp, e_0=e0, alpha=a1, beta=cb)
    return p


def _na_pd(na, kind, p):
    u = na.u
    q = {}
    for k, v in p.items():
        if k == "amplitude":
            q[k] = v / u.eV
        elif k.startswith("e_"):
            q[k] = v * u.eV
        else:
            q[k] = v
    return getattr(na, kind)(**q)


def _o_pd(O, kind, p, i):
    return O.ParticleDist(kind, **{k: float(v[i]) for k, v in p.items()})


@pytest.mark.parametrize("kind", KINDS)
def test_random_models_against_oracle(na, kind):
    from oracle import naima_np as O
    from oracle import workloads_np as WN
    u = na.u
    rng = np.random.default_rng(abs(hash(kind)) % 2 ** 31)
    N = 4
    p = _draw(rng, kind, N)
    pd = _na_pd(na, kind, p)
    d_kpc = 1.7
    # random grids / energies, shared by the batch (per-walker grids: test_gpu_general.py)
    Eemin = 10 ** rng.uniform(8.5, 10.5)
    nEed = int(rng.integers(20, 130))
    Eg = np.sort(10 ** rng.uniform(7, 14.3, 17))
    Ex = np.sort(10 ** rng.uniform(-2, 5.5, 13))
    B = rng.uniform(1, 100, N) * 1e-6
    T = float(rng.uniform(10, 5000))
    uT = float(rng.uniform(0.05, 3.0))
    theta = float(rng.uniform(0.3, 2.8))
    mono_E, mono_u = float(10 ** rng.uniform(-3, 1)), float(rng.uniform(0.1, 2))
    arr_E = np.geomspace(1e-4, 10.0, 23)
    arr_n = 10 ** rng.uniform(-2, 1, 23) / arr_E
    n0 = float(rng.uniform(0.1, 10))

    kw = dict(Eemin=Eemin * u.eV, nEed=nEed)
    syn = na.Synchrotron(pd, B=B * u.G, **kw).flux(Ex * u.eV, d_kpc * u.kpc).value
    seeds_na = ["CMB", ["s1", T * u.K, uT * u.eV / u.cm ** 3],
                ["s2", T * u.K, uT * u.eV / u.cm ** 3, theta * u.rad],
                ["s3", mono_E * u.eV, mono_u * u.eV / u.cm ** 3],
                ["s4", arr_E * u.eV, arr_n / (u.eV * u.cm ** 3)]]
    icm = na.InverseCompton(pd, seed_photon_fields=seeds_na, **kw)
    ic = icm.flux(Eg * u.eV, d_kpc * u.kpc).value
    ic_s2 = icm.flux(Eg * u.eV, d_kpc * u.kpc, seed="s2").value
    br = na.Bremsstrahlung(pd, n0=n0 / u.cm ** 3, Eemin=Eemin * u.eV,
                           nEed=nEed).flux(Eg * u.eV, d_kpc * u.kpc).value
    We = na.Synchrotron(pd, B=B * u.G, **kw).We.to("erg").value
    Epmin_GeV = float(rng.uniform(1.3, 30))
    nEpd = int(rng.integers(30, 120))
    pkw = dict(Epmin=Epmin_GeV * u.GeV, nEpd=nEpd)
    ppa = na.PionDecay(pd, nh=n0 / u.cm ** 3, useLUT=False, **pkw).flux(
        Eg * u.eV, d_kpc * u.kpc).value
    ppl = na.PionDecay(pd, nh=n0 / u.cm ** 3, **pkw).flux(Eg * u.eV, d_kpc * u.kpc).value

    seeds_o = [O.thermal_seed("CMB"),
               dict(type="thermal", T=T, u=uT * O.ERG_PER_EV, theta=None),
               dict(type="thermal", T=T, u=uT * O.ERG_PER_EV, theta=theta),
               dict(type="array", energy=np.array([mono_E]), density=np.array([mono_u])),
               dict(type="array", energy=arr_E, density=arr_n)]
    gam = O.electron_grid(Eemin, 1e9 * O.MEC2_EV, nEed)
    Epmax = 1e4 * 1e3  # the class default 10 PeV in GeV
    Ep = O.proton_grid(Epmin_GeV, Epmax, nEpd)
    dist = d_kpc * O.KPC_CM
    for i in range(N):
        opd = _o_pd(O, kind, p, i)
        ne = O.nelec_on(opd, gam)
        assert_allclose(syn[i], O.to_flux(O.synchrotron_spectrum(Ex, gam, ne, B[i]), dist),
                        rtol=1e-9, atol=1e-300)
        tot, per = O.ic_spectrum(Eg, gam, ne, seeds_o)
        assert_allclose(ic[i], O.to_flux(tot, dist), rtol=1e-9, atol=1e-300)
        assert_allclose(ic_s2[i], O.to_flux(per[2], dist), rtol=1e-9, atol=1e-300)
        assert_allclose(br[i], O.to_flux(O.brems_spectrum(Eg, gam, ne, n0=n0), dist),
                        rtol=1e-9, atol=1e-300)
        assert_allclose(We[i], O.electron_energy_content(opd, gam), rtol=1e-10)
        J = O.J_on(opd, Ep)
        assert_allclose(ppa[i], O.to_flux(O.pion_spectrum(Eg, Ep, J, n0), dist), rtol=1e-9,
                        atol=1e-300)
        ref = O.to_flux(O.pion_spectrum(Eg, Ep, J, n0, diffsigma=WN.get_lut()), dist)
        assert_allclose(ppl[i], ref, rtol=1e-7, atol=1e-9 * np.max(np.abs(ref)))
