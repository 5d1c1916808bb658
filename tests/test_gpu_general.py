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
        # (a temperature per walker goes through the general kernel -- the Khangulyan kernel at
        # every node of every walker -- the single walker through its cached table)
        assert_allclose(f[k].value, ik.flux(E, 2 * u.kpc).value, rtol=1e-10)
        assert_allclose(sed[k].value, ik.sed(E, 2 * u.kpc).value, rtol=1e-10)
        assert_allclose(We[k].value, ik.compute_We(Eemin=1 * u.TeV).value, rtol=1e-14)
        sk = na.Synchrotron(pk, B=[10.0, 20.0, 30.0][k] * u.uG, Eemin=emin[k] * u.GeV,
                            nEed=[40, 50, 60][k])
        # (limits AND node density per walker: the general kernel again, against the table path)
        assert_allclose(fs[k].value, sk.flux(Ex, 2 * u.kpc).value, rtol=1e-10, atol=1e-300)
        # (proton limits per walker: nh_general_proton -- the look-up table's spline at every
        # node of every walker's own grid -- against the single walker's table path)
        pk2 = na.PionDecay(pk, Epmin=[2.0, 5.0, 10.0][k] * u.GeV)
        assert_allclose(fp[k].value, pk2.flux(E, 2 * u.kpc).value, rtol=1e-9, atol=1e-300)
        assert_allclose(Wp[k].value, pk2.Wp.value, rtol=1e-10)


def test_general_path_device_values(na):
    """Eemin / Eemax as device-resident per-walker values go through the general kernel (a
    particle grid per walker) and agree with the same values given on the host; so do seed
    temperatures and angles per walker"""
    from naima_amd._lib import get_context
    from naima_amd.darray import DPars
    u = na.u
    ctx = get_context()
    host = np.array([[33.0, 33.2, 32.9], [1.0, 5.0, 40.0], [0.3, 1.0, 2.0]])
    P = DPars(ctx, ctx.array(host), 3, 3)
    E = E_GAMMA * u.eV

    def models(p):
        pd = na.ExponentialCutoffPowerLaw(10 ** p[0] / u.eV, 10 * u.TeV, 2.4, 50 * u.TeV)
        ic = na.InverseCompton(pd, seed_photon_fields=["CMB", ["star", 5000 * u.K, 1 * u.eV / u.cm ** 3,
                                                                 60 * u.deg]],
                               Eemin=p[1] * u.GeV, Eemax=p[2] * u.PeV)
        sy = na.Synchrotron(pd, B=20 * u.uG, Eemin=p[1] * u.GeV, Eemax=p[2] * u.PeV)
        return ic, sy

    icd, syd = models(P)
    ich, syh = models(host)
    Ex = np.geomspace(1e2, 1e5, 9) * u.eV
    assert_allclose(np.asarray(icd.flux(E, 1 * u.kpc).value), ich.flux(E, 1 * u.kpc).value, rtol=1e-13)
    assert_allclose(np.asarray(syd.flux(Ex, 1 * u.kpc).value), syh.flux(Ex, 1 * u.kpc).value,
                    rtol=1e-13, atol=1e-300)
    # a seed temperature / angle per walker (device values, shared grid): the same kernel
    pdd = na.ExponentialCutoffPowerLaw(10 ** P[0] / u.eV, 10 * u.TeV, 2.4, 50 * u.TeV)
    pdh = na.ExponentialCutoffPowerLaw(10 ** host[0] / u.eV, 10 * u.TeV, 2.4, 50 * u.TeV)

    def seeded(pd, p):
        return na.InverseCompton(pd, seed_photon_fields=[
            "CMB", ["FIR", 6 * p[1] * u.K, p[2] * u.eV / u.cm ** 3],
            ["star", 5000 * u.K, 1 * u.eV / u.cm ** 3, 40 * p[2] * u.deg]])

    fd = seeded(pdd, P).flux(E, 1 * u.kpc)
    fh = seeded(pdh, host).flux(E, 1 * u.kpc)
    assert_allclose(np.asarray(fd.value), fh.value, rtol=1e-13)
    for k in range(3):
        pk = na.ExponentialCutoffPowerLaw(10 ** host[0, k] / u.eV, 10 * u.TeV, 2.4, 50 * u.TeV)
        one = seeded(pk, host[:, k]).flux(E, 1 * u.kpc)  # (scalars: the cached tables)
        assert_allclose(fh[k].value, one.value, rtol=1e-10)
    # a non-thermal seed (shared by the walkers) beside a per-walker temperature: the general
    # kernel as well (what = 4), device values == host vectors
    def mixed(pd_, p):
        return na.InverseCompton(pd_, seed_photon_fields=[
            ["FIR", 6 * p[1] * u.K, 0.5 * u.eV / u.cm ** 3],
            ["mono", 1 * u.eV, 1 * u.eV / u.cm ** 3]])

    md, mh = mixed(pdd, P), mixed(pdh, host)
    assert not md._needs_walker_loop() and not mh._needs_walker_loop()
    assert_allclose(np.asarray(md.flux(E, 1 * u.kpc).value), mh.flux(E, 1 * u.kpc).value, rtol=1e-13)
    for k in range(3):
        pk = na.ExponentialCutoffPowerLaw(10 ** host[0, k] / u.eV, 10 * u.TeV, 2.4, 50 * u.TeV)
        one = mixed(pk, host[:, k]).flux(E, 1 * u.kpc)  # (scalars: the cached tables)
        assert_allclose(mh.flux(E, 1 * u.kpc)[k].value, one.value, rtol=1e-10)


def test_general_kernel_against_oracle(na):
    """nh_general_electron (grid, weights, Synchrotron and thermal-IC integrands of EVERY
    walker built in its workgroup) against the oracle on each walker's own grid -- limits and
    node counts differ from walker to walker -- and its overflow report"""
    from naima_amd._lib import NaimaHipError, get_context
    from oracle import naima_np as O
    u = na.u
    rng = np.random.default_rng(5)
    N = 37
    amp = 10 ** rng.normal(33, 0.2, N)
    alpha = rng.uniform(1.8, 2.9, N)
    ecut = rng.uniform(20, 200, N)
    emin = 10 ** rng.uniform(-1, 2.5, N)       # GeV
    emax = 10 ** rng.uniform(4.5, 6.2, N)      # GeV
    B = rng.uniform(3, 80, N)
    pd = na.ExponentialCutoffPowerLaw(amp / u.eV, 10 * u.TeV, alpha, ecut * u.TeV)
    kw = dict(Eemin=emin * u.GeV, Eemax=emax * u.GeV, nEed=37)
    Eg = np.geomspace(1e8, 3e13, 23)
    Ex = np.geomspace(1.0, 1e5, 70)  # (two tiles of 64 energies)
    ic = na.InverseCompton(pd, seed_photon_fields=["CMB", "NIR", ["star", 9000 * u.K, 2 * u.eV / u.cm ** 3,
                                                                   130 * u.deg]], **kw)
    f = ic.flux(Eg * u.eV, 0).value
    per = [ic.flux(Eg * u.eV, 0, seed=j).value for j in range(3)]
    fs = na.Synchrotron(pd, B=B * u.uG, **kw).flux(Ex * u.eV, 0).value
    assert f.shape == (N, Eg.size) and fs.shape == (N, Ex.size)
    nodes = set()
    for i in range(N):
        gam = O.electron_grid(emin[i] * 1e9, emax[i] * 1e9, 37)
        nodes.add(gam.size)
        opd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=amp[i], e_0=1e13, alpha=alpha[i],
                             e_cutoff=ecut[i] * 1e12, beta=1.0)
        ne = O.nelec_on(opd, gam)
        seeds = [O.thermal_seed("CMB"), O.thermal_seed("NIR"),
                 dict(type="thermal", T=9000.0, u=2 * O.ERG_PER_EV, theta=np.deg2rad(130.0))]
        tot, each = O.ic_spectrum(Eg, gam, ne, seeds)
        assert_allclose(f[i], tot, rtol=1e-9, atol=tot.max() * 1e-200)
        for j in range(3):
            assert_allclose(per[j][i], each[j], rtol=1e-9, atol=tot.max() * 1e-200)
        ref = O.synchrotron_spectrum(Ex, gam, ne, B[i] * 1e-6)
        assert_allclose(fs[i], ref, rtol=1e-9, atol=ref.max() * 1e-200)
    assert len(nodes) > 10  # (the walkers really have grids of different lengths)
    ctx = get_context()
    old = ctx.general_nmax
    ctx.general_nmax = 64
    try:
        with pytest.raises(NaimaHipError, match="nodes"):
            na.Synchrotron(pd, B=B * u.uG, **kw).flux(Ex * u.eV, 0)
    finally:
        ctx.general_nmax = old


def test_table_model_particle_distribution(na, golden):
    """TableModel (models.py:425-467) as the particle distribution of every radiative
    class, against the reference (tests/golden/extra.npz, gen_golden_extra.py)"""
    u = na.u
    z = golden("extra")
    tm = na.TableModel(z["tm_energy_eV"] * u.eV, z["tm_values_per_eV"] / u.eV, amplitude=2.5)
    assert_allclose(tm(z["tm_call_e_eV"] * u.eV).to("1/eV").value, z["tm_call"], rtol=1e-13)
    Eg, Ex = z["tm_Egamma_eV"] * u.eV, z["tm_Ex_eV"] * u.eV
    d = 1.5 * u.kpc
    ic = na.InverseCompton(tm, seed_photon_fields=["CMB", "FIR"])
    assert_allclose(ic.flux(Eg, d).to("1/(s cm2 eV)").value, z["tm_ic_flux"], rtol=1e-10)
    assert_allclose(ic.compute_We(Eemin=1 * u.TeV).to("erg").value, z["tm_We_gt_1TeV_erg"],
                    rtol=1e-10)
    syn = na.Synchrotron(tm, B=15 * u.uG)
    assert_allclose(syn.flux(Ex, d).to("1/(s cm2 eV)").value, z["tm_syn_flux"], rtol=1e-10,
                    atol=1e-300)
    br = na.Bremsstrahlung(tm, n0=2 / u.cm ** 3)
    assert_allclose(br.flux(Eg, d).to("1/(s cm2 eV)").value, z["tm_brems_flux"], rtol=1e-10)
    tp = na.TableModel(z["tm_energy_eV"] * u.eV, z["tm_values_per_eV"] / u.eV)
    pp = na.PionDecay(tp, nh=3 / u.cm ** 3, useLUT=False)
    assert_allclose(pp.flux(Eg, d).to("1/(s cm2 eV)").value, z["tm_pp_flux"], rtol=1e-10,
                    atol=1e-300)
    assert_allclose(pp.Wp.to("erg").value, z["tm_Wp_erg"], rtol=1e-10)
    # amplitude per walker: rows scale, shape shared
    tmb = na.TableModel(z["tm_energy_eV"] * u.eV, z["tm_values_per_eV"] / u.eV,
                        amplitude=np.array([2.5, 5.0, 0.25]))
    fb = na.InverseCompton(tmb, seed_photon_fields=["CMB", "FIR"]).flux(Eg, d)
    assert fb.shape == (3, Eg.size)
    assert_allclose(fb.to("1/(s cm2 eV)").value,
                    np.outer([1.0, 2.0, 0.1], z["tm_ic_flux"]), rtol=1e-10)


def test_ebl_absorption_model(na, golden):
    """EblAbsorptionModel.transmission (models.py:470-552) against the reference, and as a
    per-energy factor on a batched flux"""
    u = na.u
    z = golden("extra")
    e = z["ebl_e_eV"] * u.eV
    inside = (z["ebl_e_eV"] >= 1e9) & (z["ebl_e_eV"] <= 1e14)
    for zz in (0.005, 0.5, 1.234, 3.99):
        ebl = na.EblAbsorptionModel(zz)
        assert_allclose(ebl.transmission(e), z["ebl_transmission_z%s" % zz], rtol=1e-12)
        assert_allclose(np.asarray(ebl(e[inside]).value), z["ebl_call_z%s" % zz], rtol=1e-11)
    with pytest.raises(ValueError):
        na.EblAbsorptionModel(0.5, ebl_absorption_model="Franceschini")
    pd = _ecpl(na, np.array([1e33, 2e33]), np.array([30.0, 50.0]))
    ic = na.InverseCompton(pd, seed_photon_fields=["CMB"])
    Eg = np.geomspace(1e10, 5e13, 9) * u.eV
    att = na.EblAbsorptionModel(0.5).transmission(Eg)
    f = ic.flux(Eg, 1 * u.kpc)
    assert_allclose((f * att).value, f.value * att[None, :], rtol=1e-15)


def test_pion_decay_kelner06(na, golden):
    """PionDecayKelner06 (radiative.py:1543-1767): the reference integrates with adaptive
    quad at epsrel = 1e-3, the kernel with a converged fixed rule -- agreement with the
    reference within its tolerance (measured 4e-5), with the converged oracle to 1e-8"""
    from oracle import naima_np as O
    u = na.u
    z = golden("extra")
    E = z["k06_E_eV"]
    pl = na.PowerLaw(4e35 / u.eV, 1 * u.TeV, 2.2)
    ecpl = na.ExponentialCutoffPowerLaw(4e35 / u.eV, 1 * u.TeV, 2.0, 100 * u.TeV)
    opd = {"pl": O.ParticleDist("PowerLaw", amplitude=4e35, e_0=1e12, alpha=2.2),
           "ecpl": O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=4e35, e_0=1e12,
                                  alpha=2.0, e_cutoff=1e14, beta=1.0)}
    for tag, pd in (("pl", pl), ("ecpl", ecpl)):
        k = na.PionDecayKelner06(pd, nh=2 / u.cm ** 3)
        f = k.flux(E * u.eV, 1 * u.kpc).to("1/(s cm2 eV)").value
        assert_allclose(f, z["k06_%s_flux" % tag], rtol=2e-4)
        assert_allclose(k.nhat, float(z["k06_%s_nhat" % tag]), rtol=2e-4)
        conv, nhat = O.k06_spectrum(E, lambda Et, p=opd[tag]: float(p(Et * 1e12)) * 1e12,
                                    nh=2.0, epsrel=1e-11)
        assert_allclose(f, O.to_flux(conv, O.KPC_CM), rtol=1e-8)
        assert_allclose(k.nhat, nhat, rtol=1e-8)
    hi = E >= 1e11
    k = na.PionDecayKelner06(pl, nh=2 / u.cm ** 3)
    assert_allclose(k.flux(E[hi] * u.eV, 1 * u.kpc).to("1/(s cm2 eV)").value,
                    z["k06_pl_flux_hi_only"], rtol=2e-4)
    assert k.nhat == 1.0
    # Wp above threshold: int E J dE from 1.22 GeV (analytic for a power law)
    Wp = k.Wp.to("erg").value
    a = 2.2
    ref_TeV = 4e35 * 1e12 * (1.22e-3 ** (2 - a)) / (a - 2)  # J = 4e47 (E/TeV)^-2.2 per TeV
    assert_allclose(Wp, ref_TeV * 1.602176634, rtol=1e-9)
    # walkers in one launch == one at a time
    amp = np.array([4e35, 1e35, 9e35])
    al = np.array([2.2, 2.0, 2.6])
    kb = na.PionDecayKelner06(na.PowerLaw(amp / u.eV, 1 * u.TeV, al),
                              nh=np.array([2.0, 1.0, 0.5]) / u.cm ** 3)
    fb = kb.flux(E * u.eV, 1 * u.kpc).value
    for j in range(3):
        kj = na.PionDecayKelner06(na.PowerLaw(amp[j] / u.eV, 1 * u.TeV, al[j]),
                                  nh=[2.0, 1.0, 0.5][j] / u.cm ** 3)
        assert_allclose(fb[j], kj.flux(E * u.eV, 1 * u.kpc).value, rtol=1e-13)


def test_prefit_batched_simplex(na, golden):
    """_prefit (core.py:163-217): the batched simplex search on the GPU likelihood ends
    where the sequential algorithm ends on one-walker evaluations of the same likelihood"""
    from bench import build_problem
    from naima_amd.sampler import _prefit
    from oracle.neldermead_np import minimize_sequential
    model, p0, raw, data, prior, labels = build_problem("cfg1", na)
    start = p0 * np.array([1.3, 1.02, 0.97])
    x, is_ml = _prefit(start, data, model, prior)
    seq = minimize_sequential(lambda p: -float(np.asarray(na.lnprob(p, data, model, None)[0])),
                              start, maxfev=500, xtol=1e-1, ftol=1e-3)
    assert_allclose(x, seq["x"], rtol=1e-9)
    assert is_ml == (seq["status"] == 0)
    assert float(np.asarray(na.lnprob(x, data, model, None)[0])) > \
        float(np.asarray(na.lnprob(start, data, model, None)[0]))


def test_device_sampler_with_grid_limits_as_fit_parameters(na):
    """a model whose fit parameters include Eemin (every walker has its own particle grid:
    limits and node count) stays on the device -- no fall-back to the host loop, no warning
    -- and the device-resident loop, the host-driven loop and the oracle agree"""
    import warnings

    from bench import build_problem
    from naima_amd.sampler import EnsembleSampler
    from oracle import naima_np as O
    from oracle import workloads_np as WN
    u = na.u
    _, p0, raw, data, prior, labels = build_problem("cfg1", na)

    def model(pars, data):
        pd = na.ExponentialCutoffPowerLaw(pars[0] / u.eV, 10 * u.TeV, pars[1],
                                          10 ** pars[2] * u.TeV)
        ic = na.InverseCompton(pd, seed_photon_fields=["CMB"], Eemin=pars[3] * u.GeV)
        return ic.flux(data, distance=1 * u.kpc)

    def pri(pars):
        return na.uniform_prior(pars[1], -1, 5) + na.uniform_prior(pars[3], 0.1, 100)

    start = np.append(p0, 3.0)
    kw = dict(args=[data, model, pri], seed=2, naima_style=True, store_blobs=False)
    pos = start * (1 + 0.01 * np.random.default_rng(0).standard_normal((12, 4)))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        d = EnsembleSampler(12, 4, na.lnprob, device=True, **kw)
        sd = d.run_mcmc(pos, 3)
        sd = d.run_mcmc(sd, 9)
    assert d.device is True and d._dev is not None and d._dev.graph is not None
    h = EnsembleSampler(12, 4, na.lnprob, **kw)
    sh = h.run_mcmc(pos, 12)
    assert_allclose(sd.coords, sh.coords, rtol=1e-9)
    assert_allclose(sd.log_prob, sh.log_prob, rtol=1e-7)
    # ... and each walker's likelihood is the oracle's on ITS grid
    E = WN.data_energy_eV(raw)
    for i in (0, 5, 11):
        c = np.asarray(sd.coords)[i]
        gam = O.electron_grid(c[3] * 1e9, 1e9 * O.MEC2_EV, 100)
        opd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=c[0], e_0=10e12, alpha=c[1],
                             e_cutoff=10 ** c[2] * 1e12, beta=1.0)
        spec, _ = O.ic_spectrum(E, gam, O.nelec_on(opd, gam), [O.thermal_seed("CMB")])
        ll = O.lnprobmodel(WN.to_data_repr(O.to_flux(spec, O.KPC_CM), raw), raw)
        assert_allclose(np.asarray(sd.log_prob)[i], ll + float(np.asarray(pri(c))), rtol=1e-7)


def test_device_sampler_with_seed_temperature_as_fit_parameter(na):
    """the temperature of a thermal seed field as a fit parameter (the reference takes it as
    per-call state, radiative.py:430-545): no table can be shared between walkers, the model
    still runs in the device-resident loop and agrees with the host loop and the oracle"""
    import warnings

    from bench import build_problem
    from naima_amd.sampler import EnsembleSampler
    from oracle import naima_np as O
    from oracle import workloads_np as WN
    u = na.u
    _, p0, raw, data, prior, labels = build_problem("cfg1", na)

    def model(pars, data):
        pd = na.ExponentialCutoffPowerLaw(pars[0] / u.eV, 10 * u.TeV, pars[1],
                                          10 ** pars[2] * u.TeV)
        ic = na.InverseCompton(pd, seed_photon_fields=[
            "CMB", ["FIR", pars[3] * u.K, 0.4 * u.eV / u.cm ** 3]])
        return ic.flux(data, distance=1 * u.kpc)

    def pri(pars):
        return na.uniform_prior(pars[1], -1, 5) + na.uniform_prior(pars[3], 5, 200)

    start = np.append(p0, 35.0)
    kw = dict(args=[data, model, pri], seed=4, naima_style=True, store_blobs=False)
    pos = start * (1 + 0.01 * np.random.default_rng(1).standard_normal((12, 4)))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        d = EnsembleSampler(12, 4, na.lnprob, device=True, **kw)
        sd = d.run_mcmc(pos, 3)
        sd = d.run_mcmc(sd, 9)
    assert d.device is True and d._dev is not None and d._dev.graph is not None
    h = EnsembleSampler(12, 4, na.lnprob, **kw)
    sh = h.run_mcmc(pos, 12)
    assert_allclose(sd.coords, sh.coords, rtol=1e-9)
    assert_allclose(sd.log_prob, sh.log_prob, rtol=1e-7)
    E = WN.data_energy_eV(raw)
    gam = O.electron_grid(1e9, 1e9 * O.MEC2_EV, 100)
    for i in (0, 5, 11):
        c = np.asarray(sd.coords)[i]
        opd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=c[0], e_0=10e12, alpha=c[1],
                             e_cutoff=10 ** c[2] * 1e12, beta=1.0)
        seeds = [O.thermal_seed("CMB"), dict(type="thermal", T=c[3], u=0.4 * O.ERG_PER_EV, theta=None)]
        spec, _ = O.ic_spectrum(E, gam, O.nelec_on(opd, gam), seeds)
        ll = O.lnprobmodel(WN.to_data_repr(O.to_flux(spec, O.KPC_CM), raw), raw)
        assert_allclose(np.asarray(sd.log_prob)[i], ll + float(np.asarray(pri(c))), rtol=1e-7)


def test_device_sampler_falls_back_for_models_outside_naima_amd(na):
    """models the reference accepts but that are not built from naima_amd's radiative
    classes -- plain numpy arithmetic on the parameters, a functional model returning a
    host array -- cannot run on device-resident parameters; get_sampler's default
    device=True warns and samples them with the host loop instead of crashing"""
    from bench import build_problem
    from naima_amd.sampler import EnsembleSampler
    u = na.u
    _, p0, raw, data, prior, labels = build_problem("cfg1", na)
    E = data["energy"].to("TeV").value

    def numpy_model(pars, data):  # a power law written with numpy
        amp = np.exp(np.asarray(pars[0]))
        idx = np.asarray(pars[1])
        f = amp[..., None] * E ** (-idx[..., None]) if amp.ndim else amp * E ** (-idx)
        return f * u.Unit("1/(cm2 s TeV)")

    def functional_model(pars, data):  # the particle-distribution class used as a function
        pl = na.PowerLaw(10 ** pars[0] / u.Unit("cm2 s TeV"), 1 * u.TeV, pars[1])
        return pl(data)

    for model, start in ((numpy_model, np.array([-25.0, 2.3])),
                         (functional_model, np.array([-11.0, 2.3]))):
        kw = dict(args=[data, model, None], seed=2, naima_style=True)
        pos = start * (1 + 0.01 * np.random.default_rng(0).standard_normal((8, 2)))
        with pytest.warns(UserWarning, match="host-driven loop"):
            d = EnsembleSampler(8, 2, na.lnprob, device=True, **kw)
            sd = d.run_mcmc(pos, 3)
        assert d.device is False
        h = EnsembleSampler(8, 2, na.lnprob, **kw)
        sh = h.run_mcmc(pos, 3)
        assert_allclose(sd.coords, sh.coords, rtol=1e-12)
        assert d.get_blobs()[0].shape == (3, 8, len(E))


def test_trapz_loglog_intervals(na):
    """utils.trapz_loglog(..., intervals=True) (utils.py:350-351): the per-segment terms,
    along any axis, against the oracle; their sum is the integral"""
    from naima_amd.utils import trapz_loglog
    from oracle import naima_np as O
    rng = np.random.default_rng(6)
    x = np.geomspace(1.0, 1e4, 37)
    y = 10 ** rng.normal(size=(5, 37)) * x ** -1.7
    y[2, 10] = 0.0  # a zero node: both neighbouring terms vanish
    got = trapz_loglog(y, x, axis=-1, intervals=True)
    ref = O.trapz_loglog(y, x, axis=-1, intervals=True)
    assert got.shape == (5, 36)
    assert_allclose(got, ref, rtol=1e-12)
    assert_allclose(got.sum(axis=-1), trapz_loglog(y, x), rtol=1e-13)
    got0 = trapz_loglog(y.T, x, axis=0, intervals=True)
    assert_allclose(got0, ref.T, rtol=1e-12)


def test_run_sampler_surface_on_the_device_loop(na, tmp_path):
    """naima's workflow get_sampler -> run_sampler -> save_run/read_run (core.py:220-538,
    analysis.py:366-471) with the ensemble on the GPU: prefit, burn-in, run, the attributes
    naima's analysis reads, the host loop as cross-check"""
    from bench import build_problem
    model, p0, raw, data, prior, labels = build_problem("cfg3", na)
    kw = dict(data_table=data, p0=p0, labels=labels, model=model, prior=prior, nwalkers=32,
              nburn=6, nrun=20, prefit=True, seed=4, verbose=False)
    s, pos = na.run_sampler(**kw)
    assert s.device and s._dev is not None and s._dev.fused
    assert s.get_chain().shape == (20, 32, 5) and s.get_log_prob().shape == (20, 32)
    blobs = s.get_blobs()
    assert np.shape(blobs[0]) == (20, 32, 64) and np.shape(blobs[1]) == (20, 32)
    assert s.labels == list(labels) and s.run_info["n_walkers"] == 32
    assert s.run_info["n_burn"] == 6 and s.run_info["n_run"] == 20
    assert 0.1 < np.mean(s.acceptance_fraction) < 0.9
    h, _ = na.run_sampler(device=False, **kw)
    assert_allclose(s.get_chain(), h.get_chain(), rtol=1e-8)
    assert_allclose(np.asarray(blobs[1]), np.asarray(h.get_blobs()[1]), rtol=1e-8)
    na.save_run(str(tmp_path / "run"), s)
    back = na.read_run(str(tmp_path / "run"))
    assert np.array_equal(back.get_chain(), s.get_chain())
    assert back.labels == s.labels


def test_estimate_B(na):
    """utils.estimate_B (utils.py:484-542): sqrt(8 pi u_ph L_x / L_gamma) from two data
    tables, with the oracle's trapz_loglog as the reference"""
    from naima_amd.datatable import make_data
    from naima_amd.utils import estimate_B
    from oracle import naima_np as O
    ex = np.geomspace(0.5, 10, 25)          # keV
    fx = 3e-2 * ex ** -2.2                  # 1/(cm2 s keV)
    eg = np.geomspace(0.3, 80, 18)          # TeV
    fg = 2e-11 * eg ** -2.4 * np.exp(-eg / 30)   # 1/(cm2 s TeV)

    def table(e, eu, f, fu):
        return make_data(dict(energy=e, energy_unit=eu, flux=f, flux_error_lo=0.1 * f,
                              flux_error_hi=0.1 * f, ul=np.zeros(e.size, dtype=bool),
                              cl=np.full(e.size, 0.9), flux_unit=fu))
    B = estimate_B(table(ex, "keV", fx, "1/(cm2 s keV)"), table(eg, "TeV", fg, "1/(cm2 s TeV)"))
    kev, tev = 1.602176634e-9, 1.602176634
    Lx = O.trapz_loglog(fx / kev * ex * kev, ex * kev)
    Lg = O.trapz_loglog(fg / tev * eg * tev, eg * tev)
    ref = np.sqrt(Lx / Lg * 8 * np.pi * 0.261 * 1.602176634e-12) * 1e6
    assert_allclose(B.to("uG").value, ref, rtol=1e-12)
    assert 1.0 < B.to("uG").value < 1e4


def test_general_kernel_on_an_int_boundary_and_per_walker_density(na):
    """The node count of a walker's grid is int(nEed * (log10 gmax - log10 gmin))
    (radiative.py:152-154): limits that put nEed * decades exactly ON an integer, and one ulp to
    either side of it, must give the count numpy gives -- the kernel forms the limits with the
    reference's own expression ((value / mec2[erg]) * unit) -- and every evaluation that sat within
    rounding of the boundary is reported (Context.check_general warns), never silently different.
    Also: nEed per walker, and We / compute_We with per-walker limits (radiative.py:162-195),
    against the oracle on each walker's own grid."""
    import warnings
    from naima_amd import constants as K
    from naima_amd._lib import get_context
    from oracle import naima_np as O
    u = na.u
    ctx = get_context()
    ctx.check_general()
    # gmin = 100, gmax = 1e5 (three decades exactly), +- a few ulps on gmin
    base = 100.0 * K.MEC2_ERG
    emin_erg = np.array([np.nextafter(base, 0.0), base, np.nextafter(base, 1.0),
                         base * (1 - 3e-16), base * (1 + 3e-16), 95.0 * K.MEC2_ERG])
    emax_erg = np.full(emin_erg.size, 1e5 * K.MEC2_ERG)
    N = emin_erg.size
    amp = np.full(N, 1e33)
    pd = na.ExponentialCutoffPowerLaw(amp / u.eV, 10 * u.TeV, np.full(N, 2.3), np.full(N, 40.0) * u.TeV)
    Ex = np.geomspace(1.0, 1e5, 23)
    want_n = []
    for a, b in zip(emin_erg, emax_erg):
        l0, l1 = np.log10((a / K.MEC2_ERG) * 1.0), np.log10((b / K.MEC2_ERG) * 1.0)
        want_n.append(max(10, int(100 * (l1 - l0))))
    assert len(set(want_n[:5])) >= 1 and 300 in want_n
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        syn = na.Synchrotron(pd, B=20 * u.uG, Eemin=emin_erg * u.erg, Eemax=emax_erg * u.erg, nEed=100)
        fs = syn.flux(Ex * u.eV, 0).value
        We = syn.We.to("erg").value
    assert any("int()" in str(w.message) for w in rec), "boundary evaluations were not reported"
    for i in range(N):
        l0 = np.log10((emin_erg[i] / K.MEC2_ERG) * 1.0)
        l1 = np.log10((emax_erg[i] / K.MEC2_ERG) * 1.0)
        gam = np.logspace(l0, l1, want_n[i])
        opd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=1e33, e_0=1e13, alpha=2.3,
                             e_cutoff=40e12, beta=1.0)
        ne = O.nelec_on(opd, gam)
        ref = O.synchrotron_spectrum(Ex, gam, ne, 20e-6)
        assert_allclose(fs[i], ref, rtol=1e-9, atol=ref.max() * 1e-200,
                        err_msg="walker %d: %d nodes expected" % (i, want_n[i]))
        assert_allclose(We[i], O.electron_energy_content(opd, gam), rtol=1e-10)
    # nEed per walker + compute_We between per-walker limits
    rng = np.random.default_rng(3)
    M = 9
    ned = rng.integers(20, 120, M).astype(float)
    emin = 10 ** rng.uniform(-1, 2, M)
    pd2 = na.ExponentialCutoffPowerLaw(np.full(M, 1e33) / u.eV, 10 * u.TeV, np.full(M, 2.1),
                                       np.full(M, 80.0) * u.TeV)
    ic = na.InverseCompton(pd2, seed_photon_fields=["CMB"], Eemin=emin * u.GeV, Eemax=510 * u.TeV,
                           nEed=ned)
    Eg = np.geomspace(1e9, 3e13, 11)
    f = ic.flux(Eg * u.eV, 0).value
    Wlo = 10 ** rng.uniform(2.5, 3.5, M)
    W = ic.compute_We(Eemin=Wlo * u.GeV).to("erg").value
    for i in range(M):
        gam = O.electron_grid(emin[i] * 1e9, 510e12, ned[i])
        opd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=1e33, e_0=1e13, alpha=2.1,
                             e_cutoff=80e12, beta=1.0)
        tot, _ = O.ic_spectrum(Eg, gam, O.nelec_on(opd, gam), [O.thermal_seed("CMB")])
        assert_allclose(f[i], tot, rtol=1e-9, atol=tot.max() * 1e-200)
        g2 = O.electron_grid(Wlo[i] * 1e9, 510e12, ned[i])
        assert_allclose(W[i], O.electron_energy_content(opd, g2), rtol=1e-9)


def test_general_proton_kernel_against_oracle(na):
    """nh_general_proton: Epmin, Epmax AND nEpd per walker (radiative.py:1002-1055, 1495-1536) --
    the analytic Kafexhiu+14 cross section and the look-up table's spline evaluated on every
    walker's own proton grid -- and Wp / compute_Wp between per-walker limits, against the oracle
    on each walker's grid"""
    from oracle import naima_np as O
    from oracle import workloads_np as WN
    u = na.u
    rng = np.random.default_rng(11)
    N = 13
    amp = 10 ** rng.normal(46, 0.3, N)
    a1, a2 = rng.uniform(1.6, 2.2, N), rng.uniform(2.3, 3.0, N)
    eb, ec = rng.uniform(0.5, 5.0, N), rng.uniform(50, 500, N)
    epmin = rng.uniform(1.3, 30.0, N)            # GeV
    epmax = 10 ** rng.uniform(5.0, 7.0, N)       # GeV
    ned = rng.integers(30, 120, N).astype(float)
    pd = na.ExponentialCutoffBrokenPowerLaw(amp / u.TeV, 1 * u.TeV, eb * u.TeV, a1, a2, ec * u.TeV)
    Eg = np.geomspace(3e8, 1e14, 19)
    kw = dict(Epmin=epmin * u.GeV, Epmax=epmax * u.GeV, nEpd=ned)
    f_an = na.PionDecay(pd, nh=2 / u.cm ** 3, useLUT=False, **kw).flux(Eg * u.eV, 0).value
    pl = na.PionDecay(pd, nh=2 / u.cm ** 3, **kw)
    f_lut = pl.flux(Eg * u.eV, 0).value
    Wp = pl.Wp.to("erg").value
    wlo = rng.uniform(100.0, 3000.0, N)
    Wp2 = pl.compute_Wp(Epmin=wlo * u.GeV).to("erg").value
    assert f_an.shape == f_lut.shape == (N, Eg.size) and Wp.shape == Wp2.shape == (N,)
    lut = WN.get_lut()
    for i in range(N):
        Ep = O.proton_grid(epmin[i], epmax[i], ned[i])
        opd = O.ParticleDist("ExponentialCutoffBrokenPowerLaw", amplitude=amp[i] * 1e-12, e_0=1e12,
                             e_break=eb[i] * 1e12, alpha_1=a1[i], alpha_2=a2[i],
                             e_cutoff=ec[i] * 1e12, beta=1.0)
        J = O.J_on(opd, Ep)
        ref = O.pion_spectrum(Eg, Ep, J, 2.0)
        assert_allclose(f_an[i], ref, rtol=1e-9, atol=np.abs(ref).max() * 1e-200)
        refl = O.pion_spectrum(Eg, Ep, J, 2.0, diffsigma=lut)
        assert_allclose(f_lut[i], refl, rtol=1e-7, atol=np.abs(refl).max() * 1e-12)
        assert_allclose(Wp[i], O.proton_energy_content(opd, Ep), rtol=1e-9)
        l0, l1 = np.log10(wlo[i]), np.log10(epmax[i])
        Ep2 = np.logspace(l0, l1, max(10, int(ned[i] * (l1 - l0))))
        assert_allclose(Wp2[i], O.proton_energy_content(opd, Ep2), rtol=1e-9)


def test_general_bremsstrahlung_against_oracle(na):
    """Bremsstrahlung with Eemin / Eemax (and the target density) per walker: the general kernel
    builds every walker's own grid (radiative.py:147-154) and evaluates the Baring+99 cross
    sections (radiative.py:838-928) at its nodes -- e-e and e-ion emissivities, negative lobes of
    the fits and the 2 MeV switch included -- against the oracle on that grid; host vectors and
    device-resident parameters agree; nothing goes one walker at a time any more"""
    from naima_amd._lib import get_context
    from naima_amd.darray import DPars
    from oracle import naima_np as O
    u = na.u
    rng = np.random.default_rng(9)
    N = 29
    amp = 10 ** rng.normal(36, 0.2, N)
    alpha = rng.uniform(1.8, 2.9, N)
    ecut = rng.uniform(5, 200, N)
    emin = 10 ** rng.uniform(-3.2, -0.5, N)     # GeV: grids that start below and above 2 MeV
    emax = 10 ** rng.uniform(3.0, 5.5, N)       # GeV
    n0 = rng.uniform(0.5, 30, N)
    Eg = np.geomspace(1e4, 1e13, 75)            # (10 keV ... 10 TeV; two tiles of 64 energies)
    pd = na.ExponentialCutoffPowerLaw(amp / u.eV, 1 * u.TeV, alpha, ecut * u.TeV)
    br = na.Bremsstrahlung(pd, n0=n0 / u.cm ** 3, Eemin=emin * u.GeV, Eemax=emax * u.GeV, nEed=41)
    assert not br._needs_walker_loop()
    f = br.flux(Eg * u.eV, 0).value
    assert f.shape == (N, Eg.size)
    nodes = set()
    for i in range(N):
        gam = O.electron_grid(emin[i] * 1e9, emax[i] * 1e9, 41)
        nodes.add(gam.size)
        opd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=amp[i], e_0=1e12, alpha=alpha[i],
                             e_cutoff=ecut[i] * 1e12, beta=1.0)
        ref = O.brems_spectrum(Eg, gam, O.nelec_on(opd, gam), n0=n0[i])
        assert_allclose(f[i], ref, rtol=1e-9, atol=np.abs(ref).max() * 1e-13)
    assert len(nodes) > 8
    # the same with every parameter resident on the device (as the device loop hands them over)
    ctx = get_context()
    host = np.stack([np.log10(amp), alpha, np.log10(ecut), np.log10(emin), np.log10(n0)])
    pars = DPars(ctx, ctx.array(host), 5, N)
    pdd = na.ExponentialCutoffPowerLaw(10 ** pars[0] / u.eV, 1 * u.TeV, pars[1],
                                       10 ** pars[2] * u.TeV)
    brd = na.Bremsstrahlung(pdd, n0=10 ** pars[4] / u.cm ** 3, Eemin=10 ** pars[3] * u.GeV,
                            Eemax=1e4 * u.GeV, nEed=41)
    fd = np.asarray(brd.flux(Eg * u.eV, 0).value).reshape(N, Eg.size)
    fh = na.Bremsstrahlung(pd, n0=n0 / u.cm ** 3, Eemin=emin * u.GeV, Eemax=1e4 * u.GeV,
                           nEed=41).flux(Eg * u.eV, 0).value
    assert_allclose(fd, fh, rtol=1e-10, atol=np.abs(fh).max() * 1e-13)


def test_general_inverse_compton_on_shared_non_thermal_seeds(na):
    """InverseCompton with Eemin / Eemax per walker AND monochromatic / tabulated seed fields
    (the same for every walker) beside thermal ones: every seed through the general kernel on
    each walker's own grid -- the inner trapz_loglog over a tabulated seed's energies
    (radiative.py:609-655) at every (node, photon energy) -- against the oracle, total and per
    seed; nothing goes one walker at a time"""
    from oracle import naima_np as O
    u = na.u
    rng = np.random.default_rng(21)
    N = 19
    amp = 10 ** rng.normal(34, 0.2, N)
    alpha = rng.uniform(1.9, 2.8, N)
    ecut = rng.uniform(10, 150, N)
    emin = 10 ** rng.uniform(-1, 2, N)       # GeV
    emax = 10 ** rng.uniform(4.2, 5.8, N)    # GeV
    mono_E, mono_u = 0.7, 1.3                # eV, eV/cm3
    arr_E = np.geomspace(1e-3, 30.0, 23)     # eV
    arr_n = 5e2 * arr_E ** -1.4 * np.exp(-arr_E / 8.0)  # 1/(eV cm3)
    pd = na.ExponentialCutoffPowerLaw(amp / u.eV, 10 * u.TeV, alpha, ecut * u.TeV)
    seeds_na = ["CMB", ["mono", mono_E * u.eV, mono_u * u.eV / u.cm ** 3],
                ["star", 6000 * u.K, 3 * u.eV / u.cm ** 3, 100 * u.deg],
                ["tab", arr_E * u.eV, arr_n / (u.eV * u.cm ** 3)]]
    ic = na.InverseCompton(pd, seed_photon_fields=seeds_na, Eemin=emin * u.GeV, Eemax=emax * u.GeV,
                           nEed=33)
    assert not ic._needs_walker_loop()
    Eg = np.geomspace(1e7, 1e14, 31)
    f = ic.flux(Eg * u.eV, 0).value
    per = [ic.flux(Eg * u.eV, 0, seed=n).value for n in ("CMB", "mono", "star", "tab")]
    seeds_o = [O.thermal_seed("CMB"),
               dict(type="array", energy=np.array([mono_E]), density=np.array([mono_u])),
               dict(type="thermal", T=6000.0, u=3 * O.ERG_PER_EV, theta=np.deg2rad(100.0)),
               dict(type="array", energy=arr_E, density=arr_n)]
    for i in range(N):
        gam = O.electron_grid(emin[i] * 1e9, emax[i] * 1e9, 33)
        opd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=amp[i], e_0=1e13, alpha=alpha[i],
                             e_cutoff=ecut[i] * 1e12, beta=1.0)
        tot, each = O.ic_spectrum(Eg, gam, O.nelec_on(opd, gam), seeds_o)
        assert_allclose(f[i], tot, rtol=1e-9, atol=tot.max() * 1e-200)
        for j in range(4):
            assert_allclose(per[j][i], each[j], rtol=1e-9, atol=tot.max() * 1e-200)


def test_general_inverse_compton_with_a_seed_density_per_walker(na):
    """InverseCompton with Eemin / Eemax per walker AND a tabulated seed whose photon density is
    given per walker -- the synchrotron-self-Compton seed of examples/CrabNebula_SynSSC.py:29-45,
    each walker's own synchrotron photons -- beside a thermal and a shared tabulated seed
    (InverseCompton takes any keyword per call, radiative.py:430; inner integral :609-655): the
    whole batch in launches of the general kernel (nh_general_electron_seed_rows), nothing one
    walker at a time; against the oracle on each walker's grid and density, total and per seed.
    The same through the model function of a device loop: Synchrotron.flux at Esy as a device
    expression feeds the seed (host loop == device loop)."""
    from oracle import naima_np as O
    u = na.u
    rng = np.random.default_rng(23)
    N = 17
    amp = 10 ** rng.normal(34, 0.2, N)
    alpha = rng.uniform(1.9, 2.8, N)
    ecut = rng.uniform(10, 150, N)
    emin = 10 ** rng.uniform(-1, 2, N)       # GeV
    emax = 10 ** rng.uniform(4.2, 5.8, N)    # GeV
    arr_E = np.geomspace(1e-3, 30.0, 23)     # eV
    arr_n = 5e2 * arr_E ** -1.4 * np.exp(-arr_E / 8.0)  # 1/(eV cm3)
    ssc_E = np.geomspace(1e-6, 1e4, 40)      # eV
    ssc_n = (10 ** rng.uniform(-2, 2, N))[:, None] * ssc_E[None, :] ** -rng.uniform(1.2, 1.9, N)[:, None] \
        * np.exp(-ssc_E[None, :] / 10 ** rng.uniform(1, 3.5, N)[:, None])
    ssc_n[3, :5] = 0.0                       # (zero-density seed nodes: utils.py:347-348)
    pd = na.ExponentialCutoffPowerLaw(amp / u.eV, 10 * u.TeV, alpha, ecut * u.TeV)
    seeds_na = ["CMB", ["tab", arr_E * u.eV, arr_n / (u.eV * u.cm ** 3)],
                ["SSC", ssc_E * u.eV, ssc_n / (u.eV * u.cm ** 3)]]
    ic = na.InverseCompton(pd, seed_photon_fields=seeds_na, Eemin=emin * u.GeV, Eemax=emax * u.GeV,
                           nEed=33)
    assert not ic._needs_walker_loop()
    Eg = np.geomspace(1e7, 1e14, 29)
    f = ic.flux(Eg * u.eV, 0).value
    per = [ic.flux(Eg * u.eV, 0, seed=n).value for n in ("CMB", "tab", "SSC")]
    for i in range(N):
        seeds_o = [O.thermal_seed("CMB"), dict(type="array", energy=arr_E, density=arr_n),
                   dict(type="array", energy=ssc_E, density=ssc_n[i])]
        gam = O.electron_grid(emin[i] * 1e9, emax[i] * 1e9, 33)
        opd = O.ParticleDist("ExponentialCutoffPowerLaw", amplitude=amp[i], e_0=1e13, alpha=alpha[i],
                             e_cutoff=ecut[i] * 1e12, beta=1.0)
        tot, each = O.ic_spectrum(Eg, gam, O.nelec_on(opd, gam), seeds_o)
        assert_allclose(f[i], tot, rtol=1e-9, atol=tot.max() * 1e-200)
        for j in range(3):
            assert_allclose(per[j][i], each[j], rtol=1e-9, atol=tot.max() * 1e-200)
    # the walker loop it replaces gives the same numbers (the table kernels, one walker at a time)
    one = ic._loop_walkers("flux", Eg * u.eV, distance=0).value
    assert_allclose(f, one, rtol=1e-9, atol=f.max() * 1e-200)
    # ... and as a device loop's model function sees it: parameters resident in HBM, the seed a
    # device expression of the walker's own synchrotron luminosity, Eemin a fit parameter
    from naima_amd._lib import get_context
    from naima_amd.darray import DPars
    ctx = get_context()
    host = np.array([np.log10(amp[:6]), emin[:6], rng.uniform(5, 50, 6)])
    Esy = np.geomspace(1e-5, 1e5, 30) * u.eV

    def crab_like(p):
        pd6 = na.ExponentialCutoffPowerLaw(10 ** p[0] / u.eV, 10 * u.TeV, 2.3, 40 * u.TeV)
        syn = na.Synchrotron(pd6, B=p[2] * u.uG, Eemin=p[1] * u.GeV)
        phn = syn.flux(Esy, distance=0 * u.cm) / (4 * np.pi * (2.1 * u.pc) ** 2 * (29979245800.0 * u.cm / u.s)) * 2.24
        icm = na.InverseCompton(pd6, seed_photon_fields=["CMB", ["SSC", Esy, phn]], Eemin=p[1] * u.GeV)
        assert not icm._needs_walker_loop()
        return icm.flux(Eg * u.eV, 2 * u.kpc)

    fd = crab_like(DPars(ctx, ctx.array(host), 3, 6))
    fh = crab_like(host)
    assert_allclose(np.asarray(fd.value), fh.value, rtol=1e-12, atol=fh.value.max() * 1e-200)



@pytest.mark.gpu
@pytest.mark.parametrize("script,args", [("rxj1713_synic.py", ["64", "6", "12"]),
                                         ("crab_synssc.py", ["32", "3", "6"])])
def test_examples_run(tmp_path, script, args):
    """examples/: naima's workflow (run_sampler with burn-in, save_run, read_run) on the device
    loop, at toy sizes -- the RXJ1713 Syn+IC fit and the Crab Syn+SSC fit"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", script)] + args, cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "saved and read back" in out.stdout
