"""The reference's own tests/test_models.py, run against naima_amd on the GPU: same
particle distributions, same energies, same known-answer luminosities, same rtol (the
file's default 1e-7), same error behaviour.  ``naima`` -> ``naima_amd``, ``astropy.units``
-> ``naima_amd.units``; nothing else changes."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

na = pytest.importorskip("naima_amd")
from naima_amd import units as u  # noqa: E402
from naima_amd.constants import c, m_e, mec2_unit  # noqa: E402
from naima_amd.models import (BrokenPowerLaw, ExponentialCutoffBrokenPowerLaw,  # noqa: E402
                              ExponentialCutoffPowerLaw, LogParabola, PowerLaw)
from naima_amd.radiative import Bremsstrahlung, InverseCompton, PionDecay, Synchrotron  # noqa: E402
from naima_amd.utils import trapz_loglog  # noqa: E402

e_0 = 20 * u.TeV
e_cutoff = 10 * u.TeV
alpha = 2.0
e_break = 1 * u.TeV
alpha_1 = 1.5
alpha_2 = 2.5

electron_properties = {"Eemin": 100 * u.GeV, "Eemax": 1 * u.PeV}
proton_properties = {"Epmax": 1 * u.PeV}

energy = np.logspace(0, 15, 1000) * u.eV
data = {"energy": energy}
pdist_unit = 1 / mec2_unit


@pytest.fixture
def particle_dists():
    ECPL = ExponentialCutoffPowerLaw(amplitude=1 * pdist_unit, e_0=e_0, alpha=alpha,
                                     e_cutoff=e_cutoff)
    PL = PowerLaw(amplitude=1 * pdist_unit, e_0=e_0, alpha=alpha)
    BPL = BrokenPowerLaw(amplitude=1 * pdist_unit, e_0=e_0, e_break=e_break, alpha_1=alpha_1,
                         alpha_2=alpha_2)
    return ECPL, PL, BPL


def test_synchrotron_lum(particle_dists):
    ECPL, PL, BPL = particle_dists
    lum_ref = [0.00025231296225663107, 0.03316715765695228, 0.00044597089198025806]
    We_ref = [5064124672.902273, 11551172166.866821, 926633861.2898524]
    Wes, lsys = [], []
    for pdist in particle_dists:
        sy = Synchrotron(pdist, **electron_properties)
        Wes.append(sy.We.to("erg").value)
        lsy = trapz_loglog(sy.flux(energy, 0) * energy, energy).to("erg/s")
        assert lsy.unit == u.erg / u.s
        lsys.append(lsy.value)
    assert_allclose(lsys, lum_ref)
    assert_allclose(Wes, We_ref)
    sy = Synchrotron(ECPL, B=1 * u.G, **electron_properties)
    sy.flux(data)
    lsy = trapz_loglog(sy.flux(energy, 0) * energy, energy).to("erg/s")
    assert_allclose(lsy.value, 31374131.90312505)


def test_bolometric_luminosity(particle_dists):
    ECPL, PL, BPL = particle_dists
    sy = Synchrotron(ECPL, B=1 * u.G, **electron_properties)
    sy.flux(energy, distance=0 * u.kpc)
    sy.flux(energy, distance=0)
    sy.sed(energy, distance=0 * u.kpc)
    sy.sed(energy, distance=0)


def test_compute_We(particle_dists):
    ECPL, PL, BPL = particle_dists
    sy = Synchrotron(ECPL, B=1 * u.G, **electron_properties)
    Eemin, Eemax = 10 * u.GeV, 100 * u.TeV
    sy.compute_We()
    sy.compute_We(Eemin=Eemin)
    sy.compute_We(Eemax=Eemax)
    sy.compute_We(Eemin=Eemin, Eemax=Eemax)
    assert sy.We == sy.compute_We(Eemin=sy.Eemin, Eemax=sy.Eemax)
    pp = PionDecay(ECPL)
    Epmin, Epmax = 10 * u.GeV, 100 * u.TeV
    pp.compute_Wp()
    pp.compute_Wp(Epmin=Epmin)
    pp.compute_Wp(Epmax=Epmax)
    pp.compute_Wp(Epmin=Epmin, Epmax=Epmax)


def test_set_We(particle_dists):
    ECPL, PL, BPL = particle_dists
    sy = Synchrotron(ECPL, B=1 * u.G, **electron_properties)
    pp = PionDecay(ECPL)
    W = 1e49 * u.erg
    for Eemin in [1 * u.GeV, 10 * u.GeV, None]:
        for Eemax in [100 * u.TeV, None]:
            sy.set_We(W, Eemin, Eemax)
            assert_allclose(W.value, sy.compute_We(Eemin, Eemax).to("erg").value)
            sy.set_We(W, Eemin, Eemax, amplitude_name="amplitude")
            assert_allclose(W.value, sy.compute_We(Eemin, Eemax).to("erg").value)
            pp.set_Wp(W, Eemin, Eemax)
            assert_allclose(W.value, pp.compute_Wp(Eemin, Eemax).to("erg").value)
            pp.set_Wp(W, Eemin, Eemax, amplitude_name="amplitude")
            assert_allclose(W.value, pp.compute_Wp(Eemin, Eemax).to("erg").value)
    with pytest.raises(AttributeError):
        sy.set_We(W, amplitude_name="norm")
    with pytest.raises(AttributeError):
        pp.set_Wp(W, amplitude_name="norm")


def test_bremsstrahlung_lum(particle_dists):
    ECPL, PL, BPL = particle_dists
    energy2 = np.logspace(8, 14, 100) * u.eV
    brems = Bremsstrahlung(ECPL, n0=1 * u.cm ** -3, Eemin=m_e * c ** 2)
    lbrems = trapz_loglog(brems.flux(energy2, 0) * energy2, energy2).to("erg/s")
    assert_allclose(lbrems.value, 2.3064095039069847e-05)


def test_inverse_compton_lum(particle_dists):
    ECPL, PL, BPL = particle_dists
    lum_ref = [0.0002782201669858555, 0.004821189222961136, 0.00012916582897424096]
    lums = []
    for pdist in particle_dists:
        ic = InverseCompton(pdist, **electron_properties)
        lic = trapz_loglog(ic.flux(energy, 0) * energy, energy).to("erg/s")
        assert lic.unit == u.erg / u.s
        lums.append(lic.value)
    assert_allclose(lums, lum_ref)
    ic = InverseCompton(ECPL, seed_photon_fields=["CMB", "FIR", "NIR"])
    ic.flux(data)
    lic = trapz_loglog(ic.flux(energy, 0) * energy, energy).to("erg/s")
    assert_allclose(lic.value, 0.0005833030059049264)


def test_anisotropic_inverse_compton_lum(particle_dists):
    ECPL, PL, BPL = particle_dists
    angles = [45, 90, 135]
    lum_ref = [48901.363932, 111356.423781, 149800.235776]
    lums = []
    for angle in angles:
        ic = InverseCompton(
            PL, seed_photon_fields=[["Star", 20000 * u.K, 0.1 * u.erg / u.cm ** 3, angle * u.deg]],
            **electron_properties)
        lic = trapz_loglog(ic.flux(energy, 0) * energy, energy).to("erg/s")
        lums.append(lic.value)
    assert_allclose(lums, lum_ref)


def test_monochromatic_inverse_compton():
    """IC monochromatic / tabulated blackbody against the Khangulyan et al. thermal form"""
    PL = PowerLaw(1 / u.eV, 1 * u.TeV, 3)
    Ephbb = np.logspace(-3.5, -1.5, 100) * u.eV
    T = 30 * u.K
    w = 1 * u.eV / u.cm ** 3
    # photon density of a blackbody at T, dn/dE = 8 pi E^2 / (h c)^3 / (exp(E/kT) - 1)
    kT = 8.617333262e-5 * 30  # eV
    hc = 1.239841984e-4  # eV cm
    bbv = 8 * np.pi * Ephbb.value ** 2 / hc ** 3 / np.expm1(Ephbb.value / kT)
    bb = bbv * u.Unit("1/(cm3 eV)")
    Ebbmax = Ephbb[np.argmax(Ephbb.value ** 2 * bbv)]
    ar = 7.565733250280007e-15 * u.Unit("erg/(cm3 K4)")
    bb = bb * (w / (ar * T ** 4)).decompose().value
    eopts = {"Eemax": 10000 * u.GeV, "Eemin": 10 * u.GeV, "nEed": 1000}
    IC_khang = InverseCompton(PL, seed_photon_fields=[["bb", T, w]], **eopts)
    IC_mono = InverseCompton(PL, seed_photon_fields=[["mono", Ebbmax, w]], **eopts)
    IC_bb = InverseCompton(PL, seed_photon_fields=[["bb2", Ephbb, bb]], **eopts)
    IC_bb_ene = InverseCompton(PL, seed_photon_fields=[["bb2", Ephbb, Ephbb ** 2 * bb]], **eopts)
    Eph = np.logspace(-1, 1, 3) * u.GeV
    assert_allclose(IC_khang.sed(Eph).value, IC_mono.sed(Eph).value, rtol=1e-2)
    assert_allclose(IC_khang.sed(Eph).value, IC_bb.sed(Eph).value, rtol=1e-2)
    assert_allclose(IC_khang.sed(Eph).value, IC_bb_ene.sed(Eph).value, rtol=1e-2)


def test_flux_sed(particle_dists):
    ECPL, PL, BPL = particle_dists
    d1, d2 = 2.5 * u.kpc, 10.0 * u.kpc
    ic = InverseCompton(ECPL, seed_photon_fields=["CMB", "FIR", "NIR"], **electron_properties)
    luminosity = trapz_loglog(ic.flux(energy, 0) * energy, energy).to("erg/s").value
    int_flux1 = trapz_loglog(ic.flux(energy, d1) * energy, energy).to("erg/(s cm2)").value
    int_flux2 = trapz_loglog(ic.flux(energy, d2) * energy, energy).to("erg/(s cm2)").value
    assert_allclose(int_flux1 / int_flux2, (d2 / d1).value ** 2.0)
    assert_allclose(int_flux1, luminosity / (4 * np.pi * (d1.to("cm").value) ** 2))
    sed1 = ic.sed(energy, d1).to("erg/(s cm2)").value
    sed0 = (ic.flux(energy, 0) * energy ** 2).to("erg/s").value
    assert_allclose(sed1, sed0 / (4 * np.pi * (d1.to("cm").value) ** 2))


def test_ic_seed_input(particle_dists):
    ECPL, PL, BPL = particle_dists
    InverseCompton(PL, seed_photon_fields="CMB")
    InverseCompton(PL, seed_photon_fields=["CMB", "FIR", "NIR"])
    Eph = (1, 10) * u.eV
    phn = (3, 1) * u.Unit("1/(eV cm3)")
    test_seeds = [
        ["test", 5000 * u.K, 0],
        ["array", Eph, phn],
        ["array-energy", Eph, Eph ** 2 * phn],
        ["mono", Eph[0], phn[0] * Eph[0] ** 2],
        ["mono-array", Eph[:1], phn[:1] * Eph[:1] ** 2],
        ["NIR", 50 * u.K, 1.5 * u.eV / u.cm ** 3],
        ["star", 25000 * u.K, 3 * u.erg / u.cm ** 3, 120 * u.deg],
        ["X-ray", [1, 10] * u.keV, [1, 1e-2] * (1 / (u.eV * u.cm ** 3))],
        ["UV", 50 * u.eV, 15 * u.eV / u.cm ** 3],
    ]
    for seed in test_seeds:
        InverseCompton(PL, seed_photon_fields=["CMB", seed])


def test_ic_seed_fluxes(particle_dists):
    _, PL, _ = particle_dists
    ic = InverseCompton(PL, seed_photon_fields=[
        "CMB", ["test", 5000 * u.K, 0], ["test2", 5000 * u.K, 10 * u.eV / u.cm ** 3],
        ["test3", 5000 * u.K, 10 * u.eV / u.cm ** 3, 90 * u.deg]])
    ene = np.logspace(-3, 0, 5) * u.TeV
    for idx, name in enumerate(["CMB", "test", "test2", "test3"]):
        assert_allclose(ic.sed(ene, seed=name).value, ic.sed(ene, seed=idx).value)
    with pytest.raises(ValueError):
        ic.sed(ene, seed="FIR")
    with pytest.raises(ValueError):
        ic.sed(ene, seed=10)


def test_pion_decay(particle_dists):
    ECPL, PL, BPL = particle_dists
    for pdist in [ECPL, PL, BPL]:
        pdist.amplitude = 1 * (1 / u.TeV)
    lum_ref_LUT = [9.94070311e-13, 2.30256683e-12, 1.57263936e-13]
    lum_ref_noLUT = [9.94144387e-13, 2.30264140e-12, 1.57272216e-13]
    Wp_ref = [5406.36160963, 8727.55086557, 554.13864492]
    en = np.logspace(-3, 3, 60) * u.TeV
    Wps, lpps_LUT, lpps_noLUT = [], [], []
    for pdist in particle_dists:
        pp = PionDecay(pdist, useLUT=True, **proton_properties)
        Wps.append(pp.Wp.to("erg").value)
        lpp = trapz_loglog(pp.flux(en, 0) * en, en).to("erg/s")
        assert lpp.unit == u.erg / u.s
        lpps_LUT.append(lpp.value)
        pp.useLUT = False
        lpp = trapz_loglog(pp.flux(en, 0) * en, en).to("erg/s")
        lpps_noLUT.append(lpp.value)
    assert_allclose(lpps_LUT, lum_ref_LUT)
    assert_allclose(lpps_noLUT, lum_ref_noLUT)
    assert_allclose(Wps, Wp_ref)
    # LUT not packaged for Geant4: falls back to the analytic form with a warning
    with pytest.warns(UserWarning):
        pp = PionDecay(PL, useLUT=True, hiEmodel="Geant4", **proton_properties)
        pp.flux(en, 0)


def test_pion_decay_no_nuc_enh(particle_dists):
    ECPL, PL, BPL = particle_dists
    for pdist in [ECPL, PL, BPL]:
        pdist.amplitude = 1 * (1 / u.TeV)
    en = np.logspace(9, 13, 20) * u.eV
    pp = PionDecay(ECPL, nuclear_enhancement=False, useLUT=False, **proton_properties)
    pp.Wp.to("erg").value
    lpp = trapz_loglog(pp.flux(en, 0) * en, en).to("erg/s")
    assert_allclose(lpp.value, 5.693100769654807e-13)


def test_inputs():
    LP = LogParabola(1.0, e_0, 1.7, 0.2)
    LP(np.logspace(1, 10, 10) * u.TeV)
    LP(10 * u.TeV)
    ECBPL = ExponentialCutoffBrokenPowerLaw(1.0, e_0, e_break, 1.5, 2.5, e_cutoff, 2.0)
    ECBPL(np.logspace(1, 10, 10) * u.TeV)
    with pytest.raises(TypeError):
        LP({"flux": [1, 2, 4]})
