"""Known-answer luminosities through naima_amd's radiative classes on the GPU.

Data-driven: every entry of ``tests/golden/known_answers.json`` (the literals the
reference pins in its tests/test_models.py:77-442, stored here as data) is mapped by
``CASES`` to the class, particle distribution, keyword arguments and energy grid that
produce it; one parametrised test builds the model, integrates ``flux(E, 0) * E`` with
``trapz_loglog`` and compares at the reference's own rtol 1e-7.  The class-API checks
below it (setters, seed grammar, flux/sed identities, errors) are this build's own.
"""
import json
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from naima_amd.utils import trapz_loglog

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))
RTOL = 1e-7  # numpy.testing.assert_allclose default, what the reference's pins are held to


@pytest.fixture(scope="module")
def na():
    import naima_amd
    from naima_amd import _lib
    _lib.get_context()
    return naima_amd


def _dist(na, which, amplitude):
    u = na.u
    e0 = 20 * u.TeV
    if which == "ECPL":
        return na.ExponentialCutoffPowerLaw(amplitude, e0, 2.0, 10 * u.TeV)
    if which == "PL":
        return na.PowerLaw(amplitude, e0, 2.0)
    return na.BrokenPowerLaw(amplitude, e0, 1 * u.TeV, 1.5, 2.5)


GRIDS = {"wide": (0, 15, 1000), "brems": (8, 14, 100), "pp": (9, 15, 60), "nonuc": (9, 13, 20)}
ELECTRONS = dict(Eemin=(100, "GeV"), Eemax=(1, "PeV"))


def _cases():
    """(id, key, index or None, class name, dist, amplitude unit, kwargs, grid, quantity)"""
    out = []
    for i, d in enumerate(("ECPL", "PL", "BPL")):
        out.append(("syn-%s" % d, "syn_lum", i, "Synchrotron", d, "mec2", dict(ELECTRONS), "wide", "lum"))
        out.append(("We-%s" % d, "We", i, "Synchrotron", d, "mec2", dict(ELECTRONS), "wide", "We"))
        out.append(("ic-%s" % d, "ic_lum", i, "InverseCompton", d, "mec2", dict(ELECTRONS), "wide", "lum"))
        out.append(("pp-lut-%s" % d, "pp_lum_LUT", i, "PionDecay", d, "TeV",
                    dict(useLUT=True, Epmax=(1, "PeV")), "pp", "lum"))
        out.append(("pp-analytic-%s" % d, "pp_lum_noLUT", i, "PionDecay", d, "TeV",
                    dict(useLUT=False, Epmax=(1, "PeV")), "pp", "lum"))
        out.append(("Wp-%s" % d, "Wp", i, "PionDecay", d, "TeV", dict(Epmax=(1, "PeV")), "pp", "Wp"))
    for i, ang in enumerate((45, 90, 135)):
        out.append(("ic-star-%d" % ang, "ic_ani_lum", i, "InverseCompton", "PL", "mec2",
                    dict(ELECTRONS, star=ang), "wide", "lum"))
    out.append(("syn-1G", "syn_lum_B1G", None, "Synchrotron", "ECPL", "mec2",
                dict(ELECTRONS, B=(1, "G")), "wide", "lum"))
    out.append(("ic-3seeds", "ic_lum_3seeds", None, "InverseCompton", "ECPL", "mec2",
                dict(seeds=["CMB", "FIR", "NIR"]), "wide", "lum"))
    out.append(("brems", "brems_lum", None, "Bremsstrahlung", "ECPL", "mec2",
                dict(n0_per_cm3=1, Eemin=(1, "mec2")), "brems", "lum"))
    out.append(("pp-no-nuclear-enhancement", "pp_lum_nonuc", None, "PionDecay", "ECPL", "TeV",
                dict(useLUT=False, nuclear_enhancement=False, Epmax=(1, "PeV")), "nonuc", "lum"))
    return out


CASES = _cases()


def _build(na, cls, dist, amp_unit, kwargs):
    from naima_amd.constants import mec2_unit
    u = na.u
    amp = 1 / mec2_unit if amp_unit == "mec2" else 1 / u.TeV
    kw = {}
    for k, v in kwargs.items():
        if k == "star":
            kw["seed_photon_fields"] = [["Star", 20000 * u.K, 0.1 * u.erg / u.cm ** 3, v * u.deg]]
        elif k == "seeds":
            kw["seed_photon_fields"] = v
        elif k == "n0_per_cm3":
            kw["n0"] = v * u.cm ** -3
        elif isinstance(v, tuple):
            kw[k] = v[0] * (mec2_unit if v[1] == "mec2" else u.Unit(v[1]))
        else:
            kw[k] = v
    return getattr(na, cls)(_dist(na, dist, amp), **kw)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_known_answer(na, case):
    _, key, idx, cls, dist, amp_unit, kwargs, grid, what = case
    want = KA[key] if idx is None else KA[key][idx]
    rad = _build(na, cls, dist, amp_unit, kwargs)
    if what in ("We", "Wp"):
        got = getattr(rad, what).to("erg").value
    else:
        lo, hi, n = GRIDS[grid]
        E = np.logspace(lo, hi, n) * na.u.eV
        lum = trapz_loglog(rad.flux(E, 0) * E, E).to("erg/s")
        assert lum.unit == na.u.erg / na.u.s
        got = lum.value
    assert_allclose(got, want, rtol=RTOL)


def test_every_pinned_number_has_a_case():
    covered = {}
    for c in CASES:
        covered.setdefault(c[1], set()).add(c[2])
    for key, val in KA.items():
        if key.startswith("_"):
            continue
        want = set(range(len(val))) if isinstance(val, list) else {None}
        assert covered.get(key) == want, key


# ---------------------------------------------------------------------------------------
# class API around the integrators (radiative.py:88-134, 162-236, 432-545, 712-791)
# ---------------------------------------------------------------------------------------
def test_energy_content_setters_and_limits(na):
    u = na.u
    from naima_amd.constants import mec2_unit
    pd = _dist(na, "ECPL", 1 / mec2_unit)
    sy = na.Synchrotron(pd, B=1 * u.G, Eemin=100 * u.GeV, Eemax=1 * u.PeV)
    pp = na.PionDecay(pd)
    assert sy.We == sy.compute_We(Eemin=sy.Eemin, Eemax=sy.Eemax)
    full = sy.compute_We().to("erg").value
    part = sy.compute_We(Eemin=10 * u.TeV, Eemax=100 * u.TeV).to("erg").value
    assert 0 < part < full
    assert pp.compute_Wp(Epmin=10 * u.GeV, Epmax=100 * u.TeV).to("erg").value > 0
    target = 1e49 * u.erg
    for lo in (1 * u.GeV, None):
        for hi in (100 * u.TeV, None):
            for kw in ({}, {"amplitude_name": "amplitude"}):
                sy.set_We(target, lo, hi, **kw)
                assert_allclose(sy.compute_We(lo, hi).to("erg").value, target.value)
                pp.set_Wp(target, lo, hi, **kw)
                assert_allclose(pp.compute_Wp(lo, hi).to("erg").value, target.value)
    for obj, setter in ((sy, "set_We"), (pp, "set_Wp")):
        with pytest.raises(AttributeError):
            getattr(obj, setter)(target, amplitude_name="norm")


def test_flux_sed_distance_identities(na):
    u = na.u
    from naima_amd.constants import mec2_unit
    E = np.logspace(0, 15, 400) * u.eV
    ic = na.InverseCompton(_dist(na, "ECPL", 1 / mec2_unit),
                           seed_photon_fields=["CMB", "FIR", "NIR"], Eemin=100 * u.GeV,
                           Eemax=1 * u.PeV)
    tz = trapz_loglog
    L = tz(ic.flux(E, 0) * E, E).to("erg/s").value
    for d in (2.5 * u.kpc, 10.0 * u.kpc):
        area = 4 * np.pi * d.to("cm").value ** 2
        assert_allclose(tz(ic.flux(E, d) * E, E).to("erg/(s cm2)").value, L / area)
        assert_allclose(ic.sed(E, d).to("erg/(s cm2)").value,
                        (ic.flux(E, 0) * E ** 2).to("erg/s").value / area)
    for dist0 in (0, 0 * u.kpc):  # a bare 0 and a zero length both mean "luminosity"
        assert ic.flux(E, distance=dist0).unit.physical_type == ic.flux(E, 0).unit.physical_type
        ic.sed(E, distance=dist0)
    ic.flux({"energy": E})  # a table-like with an "energy" column


def test_seed_photon_field_grammar(na):
    u = na.u
    PL = na.PowerLaw(1 / u.eV, 1 * u.TeV, 3)
    na.InverseCompton(PL, seed_photon_fields="CMB")
    Eph = (1, 10) * u.eV
    phn = (3, 1) * u.Unit("1/(eV cm3)")
    accepted = [
        ["zero-density", 5000 * u.K, 0],
        ["density-array", Eph, phn],
        ["energy-density-array", Eph, Eph ** 2 * phn],
        ["mono", Eph[0], phn[0] * Eph[0] ** 2],
        ["mono-1-element", Eph[:1], phn[:1] * Eph[:1] ** 2],
        ["NIR", 50 * u.K, 1.5 * u.eV / u.cm ** 3],
        ["star", 25000 * u.K, 3 * u.erg / u.cm ** 3, 120 * u.deg],
        ["X-ray", [1, 10] * u.keV, [1, 1e-2] * (1 / (u.eV * u.cm ** 3))],
        ["UV", 50 * u.eV, 15 * u.eV / u.cm ** 3],
    ]
    for seed in accepted:
        na.InverseCompton(PL, seed_photon_fields=["CMB", seed])
    ic = na.InverseCompton(PL, seed_photon_fields=[
        "CMB", ["cold", 5000 * u.K, 0], ["warm", 5000 * u.K, 10 * u.eV / u.cm ** 3],
        ["beamed", 5000 * u.K, 10 * u.eV / u.cm ** 3, 90 * u.deg]])
    E = np.logspace(-3, 0, 5) * u.TeV
    for i, name in enumerate(("CMB", "cold", "warm", "beamed")):
        assert_allclose(ic.sed(E, seed=name).value, ic.sed(E, seed=i).value)
    for bad in ("FIR", 10):
        with pytest.raises(ValueError):
            ic.sed(E, seed=bad)


def test_blackbody_as_thermal_mono_and_tabulated_seed(na):
    """one 30 K blackbody given three ways -- Khangulyan's thermal form, a line at its
    peak, a tabulated photon density (plain and as energy density) -- agrees to 1 %"""
    u = na.u
    PL = na.PowerLaw(1 / u.eV, 1 * u.TeV, 3)
    T, w = 30 * u.K, 1 * u.eV / u.cm ** 3
    Eph = np.logspace(-3.5, -1.5, 100) * u.eV
    kT, hc = 8.617333262e-5 * 30, 1.239841984e-4  # eV, eV cm
    dn = 8 * np.pi * Eph.value ** 2 / hc ** 3 / np.expm1(Eph.value / kT)
    peak = Eph[np.argmax(Eph.value ** 2 * dn)]
    a_r = 7.565733250280007e-15 * u.Unit("erg/(cm3 K4)")
    bb = dn * u.Unit("1/(cm3 eV)") * (w / (a_r * T ** 4)).decompose().value
    kw = dict(Eemax=10000 * u.GeV, Eemin=10 * u.GeV, nEed=1000)
    Eg = np.logspace(-1, 1, 3) * u.GeV
    ref = na.InverseCompton(PL, seed_photon_fields=[["bb", T, w]], **kw).sed(Eg).value
    for seed in (["line", peak, w], ["table", Eph, bb], ["table-u", Eph, Eph ** 2 * bb]):
        got = na.InverseCompton(PL, seed_photon_fields=[seed], **kw).sed(Eg).value
        assert_allclose(got, ref, rtol=1e-2)


def test_lut_only_exists_for_pythia8(na):
    u = na.u
    en = np.logspace(-3, 3, 60) * u.TeV
    with pytest.warns(UserWarning):
        pp = na.PionDecay(_dist(na, "PL", 1 / u.TeV), useLUT=True, hiEmodel="Geant4",
                          Epmax=1 * u.PeV)
        pp.flux(en, 0)


def test_particle_distribution_inputs(na):
    u = na.u
    LP = na.LogParabola(1.0, 20 * u.TeV, 1.7, 0.2)
    assert np.asarray(LP(np.logspace(1, 10, 10) * u.TeV)).shape == (10,)
    assert np.ndim(LP(10 * u.TeV)) == 0
    EC = na.ExponentialCutoffBrokenPowerLaw(1.0, 20 * u.TeV, 1 * u.TeV, 1.5, 2.5, 10 * u.TeV, 2.0)
    assert np.asarray(EC(np.logspace(1, 10, 10) * u.TeV)).shape == (10,)
    with pytest.raises(TypeError):
        LP({"flux": [1, 2, 4]})
