"""Loop-level parity on EVERY BASELINE configuration (core.py:97-121, 128, 450-457).

The device-resident step loop (k_step_front -> table reductions / synchrotron -> likelihood
+ accept, captured in hipGraphs, eight steps per launch) takes a different launch sequence
per workload: single-row reductions (cfg1), three energy tiles and a separate likelihood
launch (cfg2), the fused three-launch half-step (cfg3), the SSC seed inside the graph
(cfg4), the signed LUT reduction with the likelihood as its epilogue (cfg5, and cfg5 with
the analytic cross-section).  For each of them, at the per-GPU walker count bench.py
uses, with blobs kept and not kept:

  * device loop == host-driven loop on the same move stream, chain / log-prob / blobs /
    acceptance, across a random-block boundary (32 steps) and through the multi-step graph;
  * device loop == the NumPy oracle driving ``stretch_move_reference`` one walker at a
    time (small ensembles: the oracle needs ~1 s per cfg4 evaluation).
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

# (id, workload, model kwargs, walkers per GPU in bench.py)
CONFIGS = [("cfg1", "cfg1", {}, 32), ("cfg2", "cfg2", {}, 256), ("cfg3", "cfg3", {}, 512),
           ("cfg4", "cfg4", {}, 256), ("cfg5-lut", "cfg5", {}, 256),
           ("cfg5-analytic", "cfg5", {"useLUT": False}, 256)]


@pytest.fixture(scope="module")
def na():
    import naima_amd
    from naima_amd import _lib
    _lib.get_context()
    return naima_amd


def _problem(na, name, mkw):
    from bench import build_problem
    from naima_amd import workloads as W
    model, p0, raw, data, prior, labels = build_problem(name, na)
    if mkw:  # same data, other evaluation mode of the model (cfg5: analytic cross-section)
        model = W.WORKLOADS[name]["model"](na, **mkw)
    return model, p0, raw, data, prior


@pytest.mark.parametrize("store_blobs", [False, True], ids=["noblobs", "blobs"])
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_device_loop_equals_host_loop(na, cfg, store_blobs):
    from naima_amd.sampler import EnsembleSampler
    _, name, mkw, nw = cfg
    model, p0, raw, data, prior = _problem(na, name, mkw)
    nd = p0.size
    kw = dict(args=[data, model, prior], seed=31, naima_style=True, store_blobs=store_blobs)
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    pos = p0 * (1 + 0.003 * np.random.default_rng(5).standard_normal((nw, nd)))
    first, more = (2, 8) if name == "cfg4" else (3, 37)  # 40 steps cross a 32-step block
    sh, sd = h.run_mcmc(pos, first), d.run_mcmc(pos, first)
    sh, sd = h.run_mcmc(sh, more), d.run_mcmc(sd, more)
    assert d._dev is not None and d.device  # no silent fall-back to the host loop
    assert d._dev.graph is not None or d._dev.step_graph is not None
    assert_allclose(sd.coords, sh.coords, rtol=1e-8)
    assert_allclose(sd.log_prob, sh.log_prob, rtol=1e-6)
    assert_allclose(d.get_chain(), h.get_chain(), rtol=1e-8)
    assert_allclose(d.get_log_prob(), h.get_log_prob(), rtol=1e-6)
    assert_allclose(d.acceptance_fraction, h.acceptance_fraction)
    assert 0.05 < np.mean(d.acceptance_fraction) < 0.95
    bh, bd = h.get_blobs(), d.get_blobs()
    if store_blobs:
        assert bd is not None and len(bd) == len(bh)
        for x, y in zip(bd, bh):
            x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
            assert x.shape == y.shape and x.shape[:2] == (first + more, nw)
            assert_allclose(x, y, rtol=1e-8, atol=1e-300, equal_nan=True)
    else:
        assert bd is None


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_device_loop_equals_oracle_driven_sampler(na, cfg):
    """>= 4 ensemble steps of the device loop against oracle.stretch_move_reference fed with
    the same move stream and the oracle's lnprob (NumPy, one walker at a time)"""
    from naima_amd._lib import Moves
    from naima_amd.sampler import EnsembleSampler
    from oracle import naima_np as O
    from oracle import workloads_np as WN
    _, name, mkw, _ = cfg
    model, p0, raw, data, prior = _problem(na, name, mkw)
    nd = p0.size
    nw, nsteps = 2 * nd + 2, 4
    s = EnsembleSampler(nw, nd, na.lnprob, args=[data, model, prior], seed=17,
                        naima_style=True, store_blobs=False, device=True)
    start = p0 * (1 + 0.003 * np.random.default_rng(9).standard_normal((nw, nd)))
    st = s.run_mcmc(start, nsteps)
    assert s._dev is not None and s.device

    def oprior(q):
        return 0.0 if prior is None else float(np.asarray(prior(q)))

    def lnp(x):
        return np.array([WN.lnprob(name, p, raw, prior=oprior, **mkw)[0]
                         for p in np.atleast_2d(x)])

    m = Moves(17, nw, 2.0, ksteps=32, depth=4)
    addr, got = m.take(nsteps)
    S, P, Z, L = m.view(addr, got)
    c, l = start.copy(), lnp(start)
    for k in range(nsteps):
        c, l, _ = O.stretch_move_reference(c, l, lnp, S[k], P[k], Z[k], L[k])
    assert_allclose(st.coords, c, rtol=1e-8)
    assert_allclose(st.log_prob, l, rtol=1e-6)
    assert_allclose(s.get_chain()[-1], c, rtol=1e-8)


def test_one_launch_half_step_with_more_grid_nodes_than_register_units(na):
    """a particle grid of 1800 nodes: the waves hold two units of 64 nodes each in registers,
    the rest of the grid is evaluated from memory -- the one-launch half-step must still be
    the path taken, and agree with the host-driven loop"""
    from bench import build_problem
    from naima_amd.sampler import EnsembleSampler
    u = na.u
    _, p0, raw, data, prior = _problem(na, "cfg2", {})

    def model(pars, data):
        pd = na.ExponentialCutoffPowerLaw(10 ** pars[0] / u.eV, 10 * u.TeV, pars[1],
                                          10 ** pars[2] * u.TeV)
        syn = na.Synchrotron(pd, B=pars[3] * u.uG, Eemin=1 * u.GeV, Eemax=1 * u.PeV, nEed=300)
        return syn.sed(data, distance=1 * u.kpc)

    nw, nd = 24, p0.size
    kw = dict(args=[data, model, prior], seed=7, naima_style=True, store_blobs=False)
    pos = p0 * (1 + 0.003 * np.random.default_rng(2).standard_normal((nw, nd)))
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    sh, sd = h.run_mcmc(pos, 3), d.run_mcmc(pos, 3)
    sh, sd = h.run_mcmc(sh, 9), d.run_mcmc(sd, 9)
    dev = d._dev
    assert dev is not None and dev.mega and dev._plan["hs"] is not None  # ONE launch per half-step
    assert dev._plan["hs"]["threads"] == 1024
    assert_allclose(sd.coords, sh.coords, rtol=1e-8)
    assert_allclose(sd.log_prob, sh.log_prob, rtol=1e-6)
    assert_allclose(d.get_chain(), h.get_chain(), rtol=1e-8)


@pytest.mark.parametrize("cfg,nw,steps", [("cfg3", 256, 150), ("cfg2", 256, 150), ("cfg3", 24, 60)],
                         ids=["cfg3-256", "cfg2-256", "cfg3-24"])
def test_split_launch_equals_unsplit_launch(na, monkeypatch, cfg, nw, steps):
    """A launch of fewer walkers than the chip has compute units gives K workgroups to every
    walker; their partial spectra meet through device memory inside the launch (write-through
    stores, an arrival ticket, the last arriver sums in index order).  Tens of thousands of
    such hand-offs, blobs kept, against the same loop with one workgroup per walker
    (NH_HS_SPLIT=1) and against the host-driven loop: a stale or torn partial would show as a
    wrong spectrum (the blobs), a wrong log-probability and, from there on, another chain."""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, cfg, {})
    nd = p0.size
    kw = dict(args=[data, model, prior], seed=23, naima_style=True, store_blobs=True)
    pos = p0 * (1 + 0.01 * np.random.default_rng(8).standard_normal((nw, nd)))
    runs = {}
    for k in ("1", "8"):
        monkeypatch.setenv("NH_HS_SPLIT", k)
        d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
        st = d.run_mcmc(pos, 2)
        st = d.run_mcmc(st, steps - 2)
        hs = d._dev._plan["hs"]
        assert hs is not None and d._dev.mega
        runs[k] = (hs["split"], st, d.get_chain(), d.get_log_prob(), d.get_blobs(),
                   d.acceptance_fraction)
    assert runs["1"][0] == 1
    assert runs["8"][0] == (2 if nw == 256 else 8)  # (nw/2 walkers per launch, 256 CUs)
    (_, s1, c1, l1, b1, a1), (_, s8, c8, l8, b8, a8) = runs["1"], runs["8"]
    assert_allclose(c8, c1, rtol=1e-8)
    assert_allclose(l8, l1, rtol=1e-6)
    assert_allclose(a8, a1)
    for x, y in zip(b8, b1):
        assert_allclose(np.asarray(x, dtype=float), np.asarray(y, dtype=float), rtol=1e-8,
                        atol=1e-300, equal_nan=True)
    # twice the same split run: the sum of the partials does not depend on who arrived last
    monkeypatch.setenv("NH_HS_SPLIT", "8")
    d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    st = d.run_mcmc(pos, 2)
    st = d.run_mcmc(st, steps - 2)
    assert np.array_equal(d.get_chain(), c8) and np.array_equal(d.get_log_prob(), l8)
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    sh = h.run_mcmc(pos, 2)
    sh = h.run_mcmc(sh, 38)
    assert_allclose(c8[:40], h.get_chain(), rtol=1e-8)


def test_split_launch_of_the_table_only_instance(na, monkeypatch):
    """A model without a synchrotron component runs k_half_step<false>; with 200 photon
    energies its table items are most of a launch, so a small ensemble gives every walker
    several workgroups there too: device loop (split) == device loop (unsplit) == host loop,
    for the pi0 look-up table (signed segments, unpacked 64-column tiles)"""
    from naima_amd import workloads as W
    from naima_amd.datatable import make_data
    from naima_amd.sampler import EnsembleSampler
    u = na.u
    model = W.WORKLOADS["cfg5"]["model"](na)
    p0 = np.asarray(W.WORKLOADS["cfg5"]["p0"], dtype=float)
    E = np.geomspace(0.05, 300.0, 200)

    def flux_at(E_TeV):
        out = model(p0, {"energy": E_TeV * u.TeV})[0]
        return out.to("1/(s cm2 TeV)").value

    true = flux_at(E)
    rng = np.random.default_rng(12)
    raw = dict(energy=E, energy_unit="TeV", flux=true * (1 + 0.1 * rng.standard_normal(E.size)),
               flux_error_lo=0.1 * true, flux_error_hi=0.1 * true,
               ul=np.zeros(E.size, dtype=bool), cl=np.full(E.size, 0.9), flux_unit="1/(cm2 s TeV)")
    data = make_data(raw)
    nw, nd, steps = 24, p0.size, 40
    kw = dict(args=[data, model, None], seed=3, naima_style=True, store_blobs=True)
    pos = p0 * (1 + 0.003 * np.random.default_rng(6).standard_normal((nw, nd)))
    runs = {}
    for k in ("1", "8"):
        monkeypatch.setenv("NH_HS_SPLIT", k)
        d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
        st = d.run_mcmc(pos, 2)
        st = d.run_mcmc(st, steps - 2)
        hs = d._dev._plan["hs"]
        assert hs is not None and d._dev.mega
        runs[k] = (hs["split"], d.get_chain(), d.get_log_prob(), d.get_blobs())
    assert runs["1"][0] == 1 and runs["8"][0] > 1
    assert_allclose(runs["8"][1], runs["1"][1], rtol=1e-8)
    assert_allclose(runs["8"][2], runs["1"][2], rtol=1e-6)
    for x, y in zip(runs["8"][3], runs["1"][3]):
        assert_allclose(np.asarray(x, dtype=float), np.asarray(y, dtype=float), rtol=1e-8,
                        atol=1e-300)
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    sh = h.run_mcmc(pos, 2)
    sh = h.run_mcmc(sh, steps - 2)
    assert_allclose(runs["8"][1], h.get_chain(), rtol=1e-8)
    assert_allclose(runs["8"][2], h.get_log_prob(), rtol=1e-6)
