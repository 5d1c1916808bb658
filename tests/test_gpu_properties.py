"""Size-independent properties at BASELINE.json's full sizes (cfg3: 512 walkers, cfg5: 2048
walkers), where a walker-by-walker comparison with the oracle would take minutes:
linearity in the amplitude, independence of a walker from its batch (permutation, split),
additivity over seed photon fields, distance and E^2 scaling, a zero-residual likelihood,
plus an oracle spot check on a few walkers of the full batch."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def na():
    import naima_amd
    from naima_amd import _lib
    _lib.get_context()
    return naima_amd


def _problem(na, name, nwalkers, seed=5):
    from bench import build_problem
    model, p0, raw, data, prior, labels = build_problem(name, na)
    rng = np.random.default_rng(seed)
    pars = p0 * (1 + 0.02 * rng.standard_normal((nwalkers, p0.size)))
    return model, p0, raw, data, prior, pars


def test_cfg3_full_batch_properties(na):
    from oracle import workloads_np as WN
    model, p0, raw, data, prior, pars = _problem(na, "cfg3", 512)
    res = na.lnprob(pars.T, data, model, prior)
    lp, flux, We = np.asarray(res[0]), res[1].to("1/(s cm2 eV)").value, res[2].to("erg").value
    assert flux.shape == (512, 64) and lp.shape == We.shape == (512,)
    assert np.all(np.isfinite(flux)) and np.all(flux >= 0)
    # a walker does not depend on its batch: permuted, and split into two batches
    perm = np.random.default_rng(1).permutation(512)
    rp = na.lnprob(pars[perm].T, data, model, prior)
    assert_array_equal(rp[1].to("1/(s cm2 eV)").value, flux[perm])
    assert_array_equal(np.asarray(rp[0]), lp[perm])
    ra = na.lnprob(pars[:256].T, data, model, prior)
    rb = na.lnprob(pars[256:].T, data, model, prior)
    assert_allclose(np.concatenate([ra[1].value, rb[1].value]), res[1].value, rtol=1e-14)
    assert_allclose(np.concatenate([np.asarray(ra[0]), np.asarray(rb[0])]), lp, rtol=1e-12)
    # linearity in the amplitude: p[0] is log10 of it (10**(x + log10 2) is 2 * 10**x to an ulp or two)
    p2 = pars.copy()
    p2[:, 0] += np.log10(2.0)
    r2 = na.lnprob(p2.T, data, model, prior)
    assert_allclose(r2[1].to("1/(s cm2 eV)").value, 2 * flux, rtol=1e-13)
    assert_allclose(r2[2].to("erg").value, 2 * We, rtol=1e-13)
    # oracle spot check inside the full batch
    for i in (0, 101, 255, 256, 400, 511):
        olp, oflux, oWe = WN.lnprob("cfg3", pars[i], raw, prior=None)
        assert_allclose(flux[i], oflux, rtol=1e-10, atol=1e-300)
        assert_allclose(We[i], oWe, rtol=1e-10)


def test_cfg3_component_identities(na):
    """IC total = sum over seeds; flux * d^2 is distance independent; sed = flux E^2;
    synchrotron scales as B^(1+p)/... only through Ec: checked as B -> flux(E B'/B)"""
    u = na.u
    rng = np.random.default_rng(2)
    N = 512
    amp = 10 ** (33 + 0.3 * rng.standard_normal(N))
    pd = na.ExponentialCutoffPowerLaw(amp / u.eV, 10 * u.TeV, 2.0 + 0.5 * rng.random(N),
                                      10 ** (1.5 + 0.3 * rng.standard_normal(N)) * u.TeV,
                                      0.8 + 0.4 * rng.random(N))
    E = np.geomspace(1e9, 1e14, 64) * u.eV
    ic = na.InverseCompton(pd, seed_photon_fields=["CMB", "FIR", "NIR"], Eemin=100 * u.GeV)
    tot = ic.flux(E, 1 * u.kpc).value
    parts = sum(ic.flux(E, 1 * u.kpc, seed=s).value for s in ("CMB", "FIR", "NIR"))
    assert_allclose(parts, tot, rtol=1e-14)
    assert_allclose(ic.flux(E, 3 * u.kpc).value * 9, tot, rtol=1e-14)
    lum = ic.flux(E, 0).to("1/(s eV)").value
    assert_allclose(lum / (4 * np.pi * 3.0856775814913673e21 ** 2), tot, rtol=1e-14)
    sed = ic.sed(E, 1 * u.kpc).to("erg/(cm2 s)").value
    assert_allclose(sed, tot * E.value ** 2 * 1.602176634e-12, rtol=1e-14)
    # synchrotron: x = E/(B gamma^2 const) and the prefactor is B/E, so
    # flux(E; 2B) = flux(E/2; B)  (same x, same B/E)
    B = (5 + 20 * rng.random(N)) * u.uG
    Ex = np.geomspace(1e2, 1e5, 24) * u.eV
    f1 = na.Synchrotron(pd, B=B).flux(Ex, 1 * u.kpc).value
    f2 = na.Synchrotron(pd, B=2 * B).flux(2 * Ex, 1 * u.kpc).value
    assert_allclose(f2, f1, rtol=1e-12, atol=1e-300)


def test_cfg5_full_batch_properties(na):
    model, p0, raw, data, prior, pars = _problem(na, "cfg5", 2048)
    res = na.lnprob(pars.T, data, model, prior)
    flux = res[1].to("1/(s cm2 eV)").value
    assert flux.shape == (2048, 28)
    perm = np.random.default_rng(3).permutation(2048)
    rp = na.lnprob(pars[perm].T, data, model, prior)
    assert_array_equal(rp[1].to("1/(s cm2 eV)").value, flux[perm])
    p2 = pars.copy()
    p2[:, 0] += np.log10(4.0)
    r2 = na.lnprob(p2.T, data, model, prior)
    assert_allclose(r2[1].to("1/(s cm2 eV)").value, 4 * flux, rtol=1e-13)
    assert_allclose(r2[2].to("erg").value, 4 * res[2].to("erg").value, rtol=1e-13)


def test_zero_residual_likelihood(na):
    """data built from the model itself: every residual vanishes and the upper limit is
    respected, so lnL = 0 for all 512 walkers"""
    from bench import build_problem
    from naima_amd.datatable import make_data
    from naima_amd import workloads as W
    model, p0, raw, data, prior, labels = build_problem("cfg3", na)
    f0 = model(p0, data)[0]
    conv = f0.to("1/(s cm2 eV)").value
    raw2 = dict(raw)
    E_eV = (np.asarray(raw["energy"]) * na.u.Unit(str(raw["energy_unit"]))).to("eV").value
    # data flux in the table's own representation
    sed_like = "erg" in str(raw["flux_unit"])
    vals = conv * E_eV ** 2 * 1.602176634e-12 if sed_like else conv
    exact = (vals * na.u.Unit("erg/(cm2 s)" if sed_like else "1/(s cm2 eV)")).to(
        str(raw["flux_unit"])).value
    ul = np.asarray(raw["ul"], dtype=bool)  # the upper limit (2x the model) is not violated
    raw2["flux"] = np.where(ul, np.asarray(raw["flux"], dtype=float), exact)
    d2 = make_data(raw2)
    pars = np.tile(p0, (512, 1))
    lp = np.asarray(na.lnprob(pars.T, d2, model, None)[0])
    assert_allclose(lp, 0.0, atol=1e-12)


def test_full_size_device_loop_equals_host_loop(na):
    """512 walkers, the sampler of bench.py: device-resident loop (fused launches, graphs,
    chain kept on the device) against the host-driven loop on the same move stream.  240
    steps in ONE run_mcmc: eight blocks of the move generator's pinned ring (depth 4) go by
    while the GPU lags the host by whole blocks -- an upload whose source block had been
    handed back to the generator too early would tear the stream"""
    from bench import build_problem
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior, labels = build_problem("cfg3", na)
    kw = dict(args=[data, model, prior], seed=20260929, naima_style=True, store_blobs=False)
    h = EnsembleSampler(512, 5, na.lnprob, **kw)
    d = EnsembleSampler(512, 5, na.lnprob, device=True, **kw)
    pos = p0 * (1 + 0.005 * np.random.default_rng(0).standard_normal((512, 5)))
    sh, sd = h.run_mcmc(pos, 3), d.run_mcmc(pos, 3)
    sd = d.run_mcmc(sd, 240)  # (the device loop first: nothing slows the host down)
    sh = h.run_mcmc(sh, 240)
    assert_allclose(sd.coords, sh.coords, rtol=1e-8)
    assert_allclose(d.get_chain(), h.get_chain(), rtol=1e-8)
    assert_allclose(d.get_log_prob(), h.get_log_prob(), rtol=1e-6)
    assert_allclose(d.acceptance_fraction, h.acceptance_fraction)
    assert 0.2 < np.mean(d.acceptance_fraction) < 0.8
