"""GPU tests of the parameters that may vary per walker beyond the particle distribution:
target densities and seed energy densities (a per-walker factor on the shared-table
result) and the general path for parameters that shape grids or emission tables
(Eemin/Eemax/nEed, Epmin/..., seed temperatures): every walker as a batch of one.
The scalar evaluation they are compared with is itself held to the reference's golden
vectors in test_gpu_parity.py / test_gpu_models.py."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def na():
    import naima_amd
    from naima_amd import _lib
    _lib.get_context()
    return naima_amd


E_GAMMA = np.geomspace(1e8, 1e14, 23)


def _ecpl(na, amp, ecut):
    u = na.u
    return na.ExponentialCutoffPowerLaw(amp / u.eV, 10 * u.TeV, 2.4, ecut * u.TeV)


def test_per_walker_densities_and_seed_energy_density(na):
    u = na.u
    E = E_GAMMA * u.eV
    amp = np.array([1e33, 2e33, 0.5e33])
    ecut = np.array([30.0, 50.0, 80.0])
    n0 = np.array([0.3, 1.0, 4.0])
    pd = _ecpl(na, amp, ecut)
    br = na.Bremsstrahlung(pd, n0=n0 / u.cm ** 3).flux(E, 1 * u.kpc)
    pp = na.PionDecay(pd, nh=n0 / u.cm ** 3).flux(E, 1 * u.kpc)
    uu = np.array([0.2, 0.5, 1.1])
    ic = na.InverseCompton(pd, seed_photon_fields=[
        "CMB", ["FIR", 30 * u.K, uu * u.eV / u.cm ** 3]])
    fic = ic.flux(E, 1 * u.kpc)
    fic_fir = ic.flux(E, 1 * u.kpc, seed="FIR")
    assert br.shape == pp.shape == fic.shape == (3, E.size)
    for k in range(3):
        pk = _ecpl(na, amp[k], ecut[k])
        assert_allclose(br[k].value, na.Bremsstrahlung(pk, n0=n0[k] / u.cm ** 3).flux(
            E, 1 * u.kpc).value, rtol=1e-13)
        assert_allclose(pp[k].value, na.PionDecay(pk, nh=n0[k] / u.cm ** 3).flux(
            E, 1 * u.kpc).value, rtol=1e-13, atol=1e-300)
        ik = na.InverseCompton(pk, seed_photon_fields=[
            "CMB", ["FIR", 30 * u.K, uu[k] * u.eV / u.cm ** 3]])
        assert_allclose(fic[k].value, ik.flux(E, 1 * u.kpc).value, rtol=1e-13)
        assert_allclose(fic_fir[k].value, ik.flux(E, 1 * u.kpc, seed="FIR").value, rtol=1e-13)


def test_per_walker_density_on_device(na):
    """nh as a device-resident parameter: the factor is applied by nh_lincomb"""
    from naima_amd._lib import get_context
    from naima_amd.darray import DPars
    u = na.u
    ctx = get_context()
    host = np.array([[33.0, 33.3, 32.7, 33.1], [0.3, 1.0, 4.0, 2.0]])
    P = DPars(ctx, ctx.array(host), 2, 4)
    E = E_GAMMA * u.eV
    pd = na.ExponentialCutoffPowerLaw(10 ** P[0] / u.eV, 10 * u.TeV, 2.4, 50 * u.TeV)
    dev = na.PionDecay(pd, nh=P[1] / u.cm ** 3).flux(E, 1 * u.kpc)
    pdh = na.ExponentialCutoffPowerLaw(10 ** host[0] / u.eV, 10 * u.TeV, 2.4, 50 * u.TeV)
    ref = na.PionDecay(pdh, nh=host[1] / u.cm ** 3).flux(E, 1 * u.kpc)
    assert_allclose(np.asarray(dev.value), ref.value, rtol=1e-13, atol=1e-300)
    br = na.Bremsstrahlung(pd, n0=P[1] / u.cm ** 3).flux(E, 1 * u.kpc)
    refb = na.Bremsstrahlung(pdh, n0=host[1] / u.cm ** 3).flux(E, 1 * u.kpc)
    assert_allclose(np.asarray(br.value), refb.value, rtol=1e-13)


def test_general_path_grid_limits_and_seed_temperature(na):
    u = na.u
    E = E_GAMMA * u.eV
    amp = np.array([1e33, 2e33, 0.5e33])
    ecut = np.array([30.0, 50.0, 80.0])
    emin = np.array([1.0, 7.0, 60.0])
    T = np.array([20.0, 30.0, 45.0])
    pd = _ecpl(na, amp, ecut)
    ic = na.InverseCompton(pd, seed_photon_fields=[["FIR", T * u.K, 0.5 * u.eV / u.cm ** 3]],
                           Eemin=emin * u.GeV)
    f = ic.flux(E, 2 * u.kpc)
    sed = ic.sed(E, 2 * u.kpc)
    We = ic.compute_We(Eemin=1 * u.TeV)
    Ex = np.geomspace(1e2, 1e5, 11) * u.eV
    syn = na.Synchrotron(pd, B=np.array([10.0, 20.0, 30.0]) * u.uG, Eemin=emin * u.GeV,
                         nEed=np.array([40, 50, 60]))
    fs = syn.flux(Ex, 2 * u.kpc)
    pp = na.PionDecay(pd, Epmin=np.array([2.0, 5.0, 10.0]) * u.GeV)
    fp = pp.flux(E, 2 * u.kpc)
    Wp = pp.Wp
    assert f.shape == sed.shape == fp.shape == (3, E.size) and fs.shape == (3, Ex.size)
    assert We.shape == Wp.shape == (3,)
    for k in range(3):
        pk = _ecpl(na, amp[k], ecut[k])
        ik = na.InverseCompton(pk, seed_photon_fields=[["FIR", T[k] * u.K, 0.5 * u.eV / u.cm ** 3]],
                               Eemin=emin[k] * u.GeV)
        assert_allclose(f[k].value, ik.flux(E, 2 * u.kpc).value, rtol=1e-14)
        assert_allclose(sed[k].value, ik.sed(E, 2 * u.kpc).value, rtol=1e-14)
        assert_allclose(We[k].value, ik.compute_We(Eemin=1 * u.TeV).value, rtol=1e-14)
        sk = na.Synchrotron(pk, B=[10.0, 20.0, 30.0][k] * u.uG, Eemin=emin[k] * u.GeV,
                            nEed=[40, 50, 60][k])
        assert_allclose(fs[k].value, sk.flux(Ex, 2 * u.kpc).value, rtol=1e-14, atol=1e-300)
        pk2 = na.PionDecay(pk, Epmin=[2.0, 5.0, 10.0][k] * u.GeV)
        assert_allclose(fp[k].value, pk2.flux(E, 2 * u.kpc).value, rtol=1e-14, atol=1e-300)
        assert_allclose(Wp[k].value, pk2.Wp.value, rtol=1e-14)


def test_general_path_refuses_device_values(na):
    from naima_amd._lib import get_context
    from naima_amd.darray import DPars
    u = na.u
    ctx = get_context()
    P = DPars(ctx, ctx.array(np.array([[33.0, 33.2], [1.0, 5.0]])), 2, 2)
    pd = na.ExponentialCutoffPowerLaw(10 ** P[0] / u.eV, 10 * u.TeV, 2.4, 50 * u.TeV)
    with pytest.raises(NotImplementedError):
        na.InverseCompton(pd, Eemin=P[1] * u.GeV).flux(E_GAMMA * u.eV, 1 * u.kpc)
