"""Loop-level parity on EVERY BASELINE configuration (core.py:97-121, 128, 450-457).

The device-resident step loop runs a half-step as ONE launch (k_half_step: proposal ->
parameter packs -> particle weights -> table reductions + synchrotron items -> likelihood ->
accept, sixteen launches per hipGraph) wherever the model's launch sequence can be absorbed:
cfg1 / cfg5 take the table-only instance, cfg2 / cfg3 the instance with synchrotron items
(K workgroups per walker when a launch has fewer walkers than the chip has CUs); cfg4 keeps
the separate kernels with the SSC seed integral inside the graph.  For each of them, at the
per-GPU walker count bench.py uses, with blobs kept and not kept:

  * device loop == host-driven loop on the same move stream, chain / log-prob / blobs /
    acceptance, across a random-block boundary (32 steps) and through the multi-step graph;
  * device loop == the NumPy oracle driving ``stretch_move_reference`` one walker at a
    time (small ensembles: the oracle needs ~1 s per cfg4 evaluation);
  * the same two comparisons UNDER THE BENCHMARK'S OWN CONDITIONS: naima's 10 % initial ball
    (core.py:477-481) with bench.py's seed, plus hand-placed walkers outside every uniform
    prior, with a negative magnetic field, with a cut-off energy so low that every particle
    weight underflows, and with a log-probability of -inf -- the walkers for which the kernel
    takes its exact short-cuts (a proposal the prior forbids is not integrated; a grid of zero
    weights contributes exact zeros).  The tests count such proposals on the host and fail
    if none occurred;
  * the resident loop (one launch per block of moves, walkers handed over by tagged records;
    its own copies of the emission tables with sorted columns and skipped zero rows) == the
    per-launch kernel, bit for bit, up to ensembles of more walkers than CUs;
  * the resident loop over an ensemble SHARED by two, three and four ranks (one process each,
    all on the one GPU of the box, rings mapped through hipIpc) == one process: ensemble, chain,
    log-probabilities, blob history, merged current blobs, acceptance.
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
    assert d._dev.graph is not None or d._dev.step_graph is not None or d._dev.resident_launches > 0
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


def test_cfg4_at_its_baseline_ensemble_size(na):
    """BASELINE's CrabNebula Syn+SSC configuration at its full ensemble, 1024 walkers (512
    proposals per half-step: the SSC seed kernel's walker groups, the table reductions and the
    likelihood at four times the size the other loop tests use): device loop == host-driven
    loop, chain, log-probability and acceptance.  From a 2 % ball: in naima's 10 % ball some
    proposal has a negative magnetic field within a few steps -- the prior forbids it, but the
    reference evaluates the model before it looks at the prior (core.py:103-119), the model
    function's own validation refuses the NaN photon density ("SSC-density value is NaN or Inf",
    extern/validator.py) and the host-driven run ends there, as the reference's would."""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, "cfg4", {})
    nw, nd = 1024, p0.size
    kw = dict(args=[data, model, prior], seed=BENCH_SEED, naima_style=True, store_blobs=False)
    pos = p0 + 0.02 * p0 * np.random.default_rng(BENCH_SEED).normal(size=(nw, nd))
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    with np.errstate(all="ignore"):
        sh, sd = h.run_mcmc(pos, 2), d.run_mcmc(pos, 2)
        sh, sd = h.run_mcmc(sh, 6), d.run_mcmc(sd, 6)
    assert d._dev is not None and d.device
    # (the launch that integrates both synchrotron spectra -- 361 energies over 869 nodes: the
    # log-domain form with its table in LDS, not the direct form it falls back to when LDS is short)
    import ctypes as C
    from naima_amd import _lib
    form = C.c_int(0)
    _lib._chk(_lib._lib.nh_half_step_syn_form(d._dev._plan["stage"]["plan"], C.byref(form)))
    assert form.value == 2, form.value
    assert_allclose(d.get_chain(), h.get_chain(), rtol=1e-8)
    lh, ld = h.get_log_prob(), d.get_log_prob()
    assert np.array_equal(np.isinf(ld), np.isinf(lh))
    fin = np.isfinite(lh)
    assert_allclose(ld[fin], lh[fin], rtol=1e-6)
    assert_allclose(d.acceptance_fraction, h.acceptance_fraction)
    assert d.nan_proposals == 0 and h.nan_proposals == 0


@pytest.mark.parametrize("kind", ["three-seeds", "ic+bremsstrahlung"])
def test_register_resident_table_items_of_a_three_seed_model(na, monkeypatch, kind):
    """(three-seeds) InverseCompton on CMB + FIR + NIR at cfg1's 28 TeV energies over a 700-node grid: ONE table
    of 84 columns, two column tiles with lane = column (the form of hs_rt_item that cfg1's own
    narrow table does not take), sixteen proposals per half-step, eight workgroups per walker:
    each keeps THEIR items' rows in registers -- the rotation of the items over a walker's
    workgroups is the same at load time and in every slice.  The resident loop with the rows in
    registers == with the rows streamed == one launch per half-step == the host-driven loop.
    (ic+bremsstrahlung) InverseCompton(CMB) + Bremsstrahlung: TWO tables over two particle grids
    (570 and 1340 nodes; the bremsstrahlung one of 56 columns, electron-electron and
    electron-proton side by side), their items spread over the same waves."""
    from naima_amd.sampler import EnsembleSampler
    u = na.u
    _, p0, raw, data, prior = _problem(na, "cfg1", {})

    def ElectronIC(pars, data):
        ECPL = na.ExponentialCutoffPowerLaw(pars[0] / u.eV, 10.0 * u.TeV, pars[1], 10 ** pars[2] * u.TeV)
        if kind == "three-seeds":
            IC = na.InverseCompton(ECPL, seed_photon_fields=["CMB", "FIR", "NIR"], Eemin=0.05 * u.GeV)
            return IC.flux(data, distance=1.0 * u.kpc)
        IC = na.InverseCompton(ECPL, seed_photon_fields=["CMB"])
        BR = na.Bremsstrahlung(ECPL, n0=1.0 / u.cm ** 3, nEed=200)
        return IC.flux(data, distance=1.0 * u.kpc) + BR.flux(data, distance=1.0 * u.kpc)

    nw, nd = 32, p0.size
    kw = dict(args=[data, ElectronIC, prior], seed=11, naima_style=True, store_blobs=True)
    pos = p0 * (1 + 0.01 * np.random.default_rng(2).standard_normal((nw, nd)))
    runs = {}
    for mode in ("registers", "streamed", "per-launch", "host"):
        monkeypatch.setenv("NH_RUN_RT", "0" if mode == "streamed" else "1")
        monkeypatch.setenv("NAIMA_AMD_RESIDENT", "0" if mode == "per-launch" else "1")
        d = EnsembleSampler(nw, nd, na.lnprob, device=mode != "host", **kw)
        st = d.run_mcmc(pos, 3)
        st = d.run_mcmc(st, 45)
        if mode in ("registers", "streamed"):
            info = d._dev.resident_info
            assert d._dev.resident_launches > 0 and info["tables_in_registers"] == (mode == "registers"), info
            assert info["grid"] == nw // 2 and d._dev._plan["hs"]["split"] > 1, (info, d._dev._plan["hs"]["split"])
        runs[mode] = (d.get_chain(), d.get_log_prob(), np.asarray(d.get_blobs()[0], dtype=float))
    a = runs["streamed"]
    for mode in ("registers", "per-launch"):
        assert np.array_equal(runs[mode][0], a[0]), mode  # (the same accept decisions)
        assert_allclose(runs[mode][1], a[1], rtol=1e-11)
        assert_allclose(runs[mode][2], a[2], rtol=1e-11, atol=1e-300)
    assert_allclose(runs["host"][0], a[0], rtol=1e-8)
    assert_allclose(runs["host"][1], a[1], rtol=1e-6)


@pytest.mark.parametrize("shape", ["syn+ic", "ic"])
@pytest.mark.parametrize("kind", ["PowerLaw", "ExponentialCutoffPowerLaw", "BrokenPowerLaw",
                                  "ExponentialCutoffBrokenPowerLaw", "LogParabola"])
def test_resident_loop_every_particle_distribution(na, monkeypatch, kind, shape):
    """The resident loop forms a slice's particle weights in a function specialised by the KIND of
    distribution (hsr_weights<KIND, ...>, round 5); the BASELINE workloads all use the cut-off
    power law.  Every kind of models.py:49-407, a break inside the grid for the broken ones, in the
    synchrotron + IC instance (log-domain items: ln w of one grid, w of the others, We as a blob)
    and in the table-only one (rows in registers): resident loop == one launch per half-step ==
    the host-driven loop (the class kernels of nh_core.hip: other code)."""
    from naima_amd.sampler import EnsembleSampler
    u = na.u
    _, _, raw, data, _ = _problem(na, "cfg3" if shape == "syn+ic" else "cfg1", {})
    # pars: log10 amplitude, alpha, log10(e_cutoff or e_break / TeV), B / uG, the second index
    p0 = np.array([33.2 if shape == "syn+ic" else 33.0, 2.1, 1.2, 12.0, 2.9])

    def pdist(pars):
        amp, e0 = 10 ** pars[0] / u.eV, 10.0 * u.TeV
        if kind == "PowerLaw":
            return na.PowerLaw(amp, e0, pars[1] + 0.6)
        if kind == "ExponentialCutoffPowerLaw":
            return na.ExponentialCutoffPowerLaw(amp, e0, pars[1], 10 ** pars[2] * u.TeV)
        if kind == "BrokenPowerLaw":
            return na.BrokenPowerLaw(amp, e0, 10 ** pars[2] * u.TeV, pars[1], pars[4])
        if kind == "ExponentialCutoffBrokenPowerLaw":
            return na.ExponentialCutoffBrokenPowerLaw(amp, e0, 10 ** (pars[2] - 1.0) * u.TeV, pars[1], pars[4],
                                                      10 ** (pars[2] + 0.8) * u.TeV)
        return na.LogParabola(amp, e0, pars[1] + 0.3, 0.1 * pars[4])

    def model(pars, data):
        pd = pdist(pars)
        if shape == "ic":
            IC = na.InverseCompton(pd, seed_photon_fields=["CMB"])
            return IC.flux(data, distance=1.0 * u.kpc)
        IC = na.InverseCompton(pd, seed_photon_fields=["CMB", "FIR"], Eemin=100 * u.GeV)
        SYN = na.Synchrotron(pd, B=pars[3] * u.uG)
        return (IC.flux(data, distance=1.0 * u.kpc) + SYN.flux(data, distance=1.0 * u.kpc),
                IC.compute_We(Eemin=1 * u.TeV))

    def prior(pars):
        return (na.uniform_prior(pars[1], 0.5, 4.0) + na.uniform_prior(pars[2], -1.0, 3.0) +
                na.uniform_prior(pars[3], 0.1, 100.0) + na.uniform_prior(pars[4], 2.0, 5.0))

    nw, nd = 48, p0.size
    kw = dict(args=[data, model, prior], seed=5, naima_style=True, store_blobs=True)
    pos = p0 * (1 + 0.005 * np.random.default_rng(3).standard_normal((nw, nd)))
    runs = {}
    for mode in ("resident", "per-launch", "host"):
        monkeypatch.setenv("NAIMA_AMD_RESIDENT", "0" if mode == "per-launch" else "1")
        d = EnsembleSampler(nw, nd, na.lnprob, device=mode != "host", **kw)
        st = d.run_mcmc(pos, 3)
        st = d.run_mcmc(st, 37)
        if mode == "resident":
            assert d._dev is not None and d._dev.resident_launches > 0, d._dev.resident_reason
            info = d._dev.resident_info
            assert info["syn_log_domain"] == (shape == "syn+ic"), info
        if mode == "per-launch":
            assert d._dev.resident_launches == 0 and d._dev.mega
        runs[mode] = (d.get_chain(), d.get_log_prob(),
                      [np.asarray(b, dtype=float) for b in d.get_blobs()], d.acceptance_fraction)
    r = runs["resident"]
    assert np.isfinite(r[1]).all() and 0.05 < r[3].mean() < 0.95, r[3].mean()
    assert np.array_equal(runs["per-launch"][0], r[0])  # (the same accept decisions)
    assert_allclose(runs["per-launch"][1], r[1], rtol=1e-10)
    for x, y in zip(runs["per-launch"][2], r[2]):
        assert_allclose(x, y, rtol=1e-10, atol=1e-300)
    assert_allclose(runs["host"][0], r[0], rtol=1e-8)
    assert_allclose(runs["host"][1], r[1], rtol=1e-6)
    for x, y in zip(runs["host"][2], r[2]):
        assert_allclose(x, y, rtol=1e-8, atol=1e-300)


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
    # (2 ndim + 2 walkers: cfg4 runs at 14 -- the NumPy oracle materialises the (100 x 261 x 869)
    # seed tensor of radiative.py:609-655 and costs ~1 s per evaluation, 4 steps of 14 walkers are
    # 70 of them; cfg4 at its BASELINE size, 1024 walkers, is held to the host-driven loop instead:
    # test_cfg4_at_its_baseline_ensemble_size)
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


# ---------------------------------------------------------------------------------------------
# the benchmark's own conditions (bench.py: naima's 10 % ball, seed 20260929) + degenerate walkers
# ---------------------------------------------------------------------------------------------
BENCH_SEED = 20260929

# hand-placed start positions, as factors on p0 (None: keep the ball's value)
#   priors (workloads.prior_for): 0 <= log10 norm <= 100, -1 <= index <= 5, -3 <= log10 cut-off <= 5,
#   B >= 0, 0.1 <= beta <= 5 (cfg3); cfg5: -3 <= log10 break <= 4, indices within (-1, 6) ...
HAND = {
    "cfg3": [("norm below its prior", 0, -0.03), ("index above its prior", 1, 2.4),
             ("index below its prior", 1, -0.8), ("negative B (below its prior)", 3, -0.4),
             ("beta below its prior", 4, 0.05), ("beta above its prior", 4, 6.0),
             ("cut-off 10 keV: every weight underflows (below the bench prior)", 2, -8.0 / np.log10(48.0))],
    "cfg2": [("norm below its prior", 0, -0.03), ("negative B (below its prior)", 3, -0.4),
             ("cut-off below its prior", 2, -8.0 / np.log10(48.0)), ("index above its prior", 1, 2.4)],
    "cfg5": [("first index above its prior", 2, 4.0), ("break below its prior", 1, -20.0),
             ("cut-off above its prior", 4, 3.5)],
}
# (The cut-off sits five decades below the particle grids -- exp(-(E/E_c)^beta) underflows at
# every node -- but not absurdly far: a stretch move that uses such a walker as the partner
# lands at 10^(+11) TeV, still a number.)


def _loose_prior(na, name):
    """round 3's priors (cfg3: examples/RXJ1713_SynIC.py:52-65 + beta; cfg2: the amplitude only):
    without a bound on log10(cut-off) a walker can sit on the zero-flux plateau -- every particle
    weight an exact zero, the kernel's skipped-items short-cut -- and, for cfg2, propose a
    negative magnetic field (a NaN log-probability).  The benchmark's priors forbid both."""
    U = na.uniform_prior
    if name == "cfg3":
        return lambda pars: (U(pars[0], 0.0, np.inf) + U(pars[1], -1, 5) + U(pars[3], 0, np.inf)
                             + U(pars[4], 0.1, 5))
    return lambda pars: U(pars[0], 0.0, np.inf)


def _bench_ball(name, p0, nw):
    """bench.py's initial ensemble (p0 + 0.1 p0 N(0,1) from the sampler's own generator,
    core.py:477-481) with the hand-placed walkers written over its first rows"""
    rng = np.random.default_rng(BENCH_SEED)
    pos = p0 + 0.1 * p0 * rng.normal(size=(nw, p0.size))
    for i, (_, col, fac) in enumerate(HAND[name]):
        pos[i] = p0
        pos[i, col] = p0[col] * fac
    return pos


def _move_stream(seed, nw, calls):
    """the moves the sampler consumes in ``calls`` consecutive run_mcmc calls (it takes them
    from the generator in blocks of at most 32 steps): S, P, Z, L as [steps, 2, ns] copies"""
    from naima_amd._lib import Moves
    m = Moves(seed, nw, 2.0, ksteps=32, depth=4)
    out = [[], [], [], []]
    for n in calls:
        while n > 0:
            addr, got = m.take(min(32, n))
            for o, v in zip(out, m.view(addr, got)):
                o.append(np.array(v))
            n -= got
    m.close()
    return [np.concatenate(o) for o in out]


def _replay_proposals(start, chain, S, P, Z):
    """every proposal q = c - (c - s) z the run made, rebuilt from the positions it kept:
    the first half of step k moves S[k, 0] against partners that are where step k - 1 left
    them; the second half moves S[k, 1] (still there) against the first half's walkers, which
    are already where step k leaves them"""
    props, prev = [], start
    for k in range(len(chain)):
        cur = chain[k]
        c0, s0 = prev[P[k, 0]], prev[S[k, 0]]
        c1, s1 = cur[P[k, 1]], prev[S[k, 1]]
        props += [c0 - (c0 - s0) * Z[k, 0][:, None], c1 - (c1 - s1) * Z[k, 1][:, None]]
        prev = cur
    return np.concatenate(props)


def _count_shortcuts(na, model, prior, data, props):
    """on the host: how many proposals did the prior forbid (the kernel's HI_DEAD: no
    integral evaluated), how many allowed ones have a spectrum of exact zeros (a grid on which
    every particle weight is zero: its work items are skipped)"""
    dead = np.zeros(len(props), dtype=bool)
    if prior is not None:
        dead = np.isinf(np.asarray(prior(props.T), dtype=float))
    zero = np.zeros(len(props), dtype=bool)
    for lo in range(0, len(props), 4096):
        out = model(props[lo:lo + 4096].T, data)
        flux = np.asarray((out[0] if isinstance(out, tuple) else out).value, dtype=float)
        zero[lo:lo + 4096] = np.all(flux == 0.0, axis=1)
    return int(dead.sum()), int((zero & ~dead).sum())


@pytest.mark.parametrize("name,nw,loose", [("cfg3", 512, False), ("cfg3", 512, True), ("cfg2", 256, False),
                                           ("cfg5", 256, False)],
                         ids=["cfg3-512", "cfg3-512-round-3-prior", "cfg2-256", "cfg5-256"])
def test_device_loop_equals_host_loop_from_the_benchmarks_ball(na, name, nw, loose):
    """>= 64 steps with the blobs kept, from bench.py's initial ensemble and the hand-placed
    degenerate walkers: chain, log-probability, blobs and acceptance of the device loop are
    the host-driven loop's (which evaluates every proposal in full with the separate,
    golden-pinned kernels and discards what the prior forbids, as core.py:103-119 does)"""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, name, {})
    if loose:
        prior = _loose_prior(na, name)
    nd = p0.size
    # (under round 3's priors walkers the ball throws far off wander on, and within ~30 steps one
    # of them proposes parameters for which the reference's own arithmetic gives a NaN
    # log-probability -- the oracle agrees -- where emcee raises ValueError and the run is over,
    # here as there (test_nan_log_probability_is_emcees_error).  To compare all 70 steps both
    # loops run with nan_policy="reject": such a proposal is never accepted, and counted.)
    kw = dict(args=[data, model, prior], seed=BENCH_SEED, naima_style=True, store_blobs=True,
              nan_policy="reject")
    pos = _bench_ball(name, p0, nw)
    calls = (2, 68)  # 70 steps: three blocks of moves, several multi-step graphs
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    died = None
    with np.errstate(all="ignore"):
        sd = d.run_mcmc(pos, calls[0])
        sd = d.run_mcmc(sd, calls[1])
        try:
            sh = h.run_mcmc(pos, calls[0])
            sh = h.run_mcmc(sh, calls[1])
        except ValueError as e:
            # cfg5 has no prior at all: a far-off walker of the ball walks on until 10**x
            # overflows, and the model function's own parameter validation ("e_cutoff value is
            # NaN or Inf", the reference's extern/validator.py) ends the host-driven run as it
            # ends the reference's.  The device loop validates nothing per step and carries on;
            # the steps both completed are compared.
            died = str(e)
    dev = d._dev
    assert dev is not None and d.device and dev.mega and dev._plan["hs"] is not None
    assert dev.graph is not None or dev.step_graph is not None or dev.resident_launches > 0
    ch, cd = h.get_chain(), d.get_chain()
    lh, ld = h.get_log_prob(), d.get_log_prob()
    nok = len(ch)
    assert cd.shape == (sum(calls), nw, nd) and ch.shape == (nok, nw, nd)
    if died is None:
        assert nok == sum(calls)
    else:
        print("%s: the host-driven loop stopped after %d steps (%s)" % (name, nok, died))
        assert name != "cfg3" and nok >= 8
    cd, ld = cd[:nok], ld[:nok]
    assert_allclose(cd, ch, rtol=1e-8)
    assert np.array_equal(np.isinf(ld), np.isinf(lh))
    fin = np.isfinite(lh)
    assert_allclose(ld[fin], lh[fin], rtol=1e-6)
    if died is None:
        assert_allclose(sd.coords, sh.coords, rtol=1e-8)
        assert_allclose(d.acceptance_fraction, h.acceptance_fraction)
    for x, y in zip(d.get_blobs(), h.get_blobs()):
        x, y = np.asarray(x, dtype=float)[:nok], np.asarray(y, dtype=float)
        assert x.shape == y.shape and x.shape[:2] == (nok, nw)
        assert_allclose(x, y, rtol=1e-8, atol=1e-300, equal_nan=True)
    # ... and the run did go through the kernel's short-cuts
    S, P, Z, L = _move_stream(BENCH_SEED, nw, calls)
    props = _replay_proposals(pos, ch, S, P, Z)
    assert props.shape == (nok * nw, nd)
    with np.errstate(all="ignore"):
        ndead, nzero = _count_shortcuts(na, model, prior, data, props)
    d.get_chain()  # (flush: the device loop's NaN count reaches the host)
    print("%s: %d proposals, %d forbidden by the prior, %d with a grid of zero weights; "
          "%d walkers end at lnp < -1000, %d at -inf; NaN proposals rejected: host %d, device %d"
          % (name, len(props), ndead, nzero, int((lh[-1] < -1000).sum()),
             int(np.isinf(lh[-1]).sum()), h.nan_proposals, d.nan_proposals))
    if died is None:
        assert d.nan_proposals == h.nan_proposals
    assert ndead > 0, "no proposal was forbidden by the prior: HI_DEAD not exercised"
    # (the kernels' own count of them: the run's first two half-steps go piece by piece, uncounted)
    assert 0.9 * ndead <= d.prior_forbidden_proposals <= ndead
    if loose:
        assert nzero > 0, "no proposal had a grid of zero weights: that short-cut not exercised"
    else:
        # the benchmark's priors keep every walker off the plateau and out of the NaN region
        assert d.nan_proposals == 0 and died is None


@pytest.mark.parametrize("name,nw,mkw", [("cfg3", 512, {}), ("cfg1", 32, {}), ("cfg5", 256, {}),
                                         ("cfg2", 256, {})],
                         ids=["cfg3-512", "cfg1-32", "cfg5-256", "cfg2-256-two-per-walker"])
def test_device_loop_equals_oracle_driven_sampler_at_the_benchmarks_size(na, name, nw, mkw):
    """The kernels that produce the bench numbers, DIRECTLY against the oracle: every BASELINE
    workload the resident loop runs, at bench.py's walker count -- cfg3 at 512 walkers is 256
    per launch, the one-workgroup-per-walker instance of the headline, on its own sorted copies
    of the inverse-Compton tables and with the synchrotron integrand in the log domain -- from
    bench.py's ball plus the degenerate walkers, against oracle.stretch_move_reference with the
    NumPy oracle's lnprob.  The second call runs as a launch of the resident loop (asserted)."""
    import warnings
    from naima_amd.sampler import EnsembleSampler
    from oracle import naima_np as O
    from oracle import workloads_np as WN
    nsteps = 2
    model, p0, raw, data, prior = _problem(na, name, mkw)
    nd = p0.size
    s = EnsembleSampler(nw, nd, na.lnprob, args=[data, model, prior], seed=BENCH_SEED,
                        naima_style=True, store_blobs=True, device=True)
    start = _bench_ball(name, p0, nw) if name in HAND else \
        p0 + 0.1 * p0 * np.random.default_rng(BENCH_SEED).normal(size=(nw, nd))
    with np.errstate(all="ignore"):
        st = s.run_mcmc(start, nsteps)
        st = s.run_mcmc(st, nsteps)  # (the first call's half-steps settle and record the plan)
    dev = s._dev
    assert dev is not None and dev.mega
    assert dev.resident_launches > 0, getattr(dev, "resident_reason", "")
    if name == "cfg3":
        assert dev._plan["hs"]["split"] == 1 and dev._plan["hs"].get("sorted")
        assert dev.resident_info["syn_log_domain"]
    if name == "cfg1":
        assert dev._plan["hs"].get("sorted")
    if name == "cfg2":
        assert dev._plan["hs"]["split"] == 2 and dev.resident_info["syn_log_domain"]

    def oprior(q):
        return float(np.asarray(prior(q)))

    def lnp(x):
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            return np.array([WN.lnprob(name, p, raw, prior=oprior, **mkw)[0] for p in np.atleast_2d(x)])

    S, P, Z, L = _move_stream(BENCH_SEED, nw, (nsteps, nsteps))
    c, l = start.copy(), lnp(start)
    if name in HAND:
        assert np.isinf(l).sum() >= len(HAND[name]) - 1 and np.isfinite(l).sum() >= nw - 8
    chain = []
    for k in range(2 * nsteps):
        c, l, _ = O.stretch_move_reference(c, l, lnp, S[k], P[k], Z[k], L[k])
        chain.append(c.copy())
    assert_allclose(s.get_chain(), np.array(chain), rtol=1e-8)
    got = s.get_log_prob()[-1]
    assert np.array_equal(np.isinf(got), np.isinf(l))
    fin = np.isfinite(l)
    assert_allclose(got[fin], l[fin], rtol=1e-6)
    assert_allclose(st.coords, c, rtol=1e-8)
    # ... and what a user READS of the run -- the blobs the resident kernel kept: every walker's
    # model spectrum (and We / Wp) at the position it holds after the last step -- directly against
    # the oracle's spectra there, not through the log-probability
    blobs, units = s.get_blobs(), s.blob_units
    assert blobs is not None and np.shape(blobs[0])[:2] == (2 * nsteps, nw)
    flux_dev = np.asarray(blobs[0][-1], dtype=float)
    if units and units[0] is not None:
        flux_dev = na.u.Quantity(flux_dev, units[0]).to("1/(s cm2 eV)").value
    checked = 0
    for i in np.flatnonzero(fin)[::max(1, int(fin.sum()) // 64)]:
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            _, f_or, b_or = WN.lnprob(name, c[i], raw, prior=oprior, **mkw)
        assert_allclose(flux_dev[i], f_or, rtol=1e-9 if name != "cfg5" else 1e-7, atol=1e-300,
                        err_msg="model spectrum of walker %d" % i)
        if len(blobs) > 1 and np.isfinite(np.asarray(b_or, dtype=float)).all():
            b_dev = np.asarray(blobs[1][-1], dtype=float)[i]
            if units[1] is not None:
                b_dev = na.u.Quantity(b_dev, units[1]).to("erg").value
            assert_allclose(b_dev, b_or, rtol=1e-9 if name != "cfg5" else 1e-7)
        checked += 1
    assert checked >= min(32, int(fin.sum()))


def test_rejected_one_launch_plan_falls_back_to_the_three_launch_loop(na, monkeypatch):
    """the device loop's own estimate says "one launch", nh_half_step_create says no (test
    hook NH_HS_TEST_REJECT): the sampler warns once, keeps the three-launch fused half-step
    and still agrees with the host-driven loop"""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, "cfg3", {})
    nw, nd = 24, p0.size
    kw = dict(args=[data, model, prior], seed=13, naima_style=True, store_blobs=True)
    pos = p0 * (1 + 0.003 * np.random.default_rng(4).standard_normal((nw, nd)))
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    sh = h.run_mcmc(pos, 3)
    sh = h.run_mcmc(sh, 11)
    monkeypatch.setenv("NH_HS_TEST_REJECT", "1")
    d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    with pytest.warns(UserWarning, match="turned the one-launch plan down"):
        sd = d.run_mcmc(pos, 3)
    sd = d.run_mcmc(sd, 11)
    assert d._dev is not None and d._dev.fused and not d._dev.mega
    assert_allclose(d.get_chain(), h.get_chain(), rtol=1e-8)
    assert_allclose(d.get_log_prob(), h.get_log_prob(), rtol=1e-6)
    for x, y in zip(d.get_blobs(), h.get_blobs()):
        assert_allclose(np.asarray(x, dtype=float), np.asarray(y, dtype=float), rtol=1e-8,
                        atol=1e-300, equal_nan=True)


@pytest.mark.parametrize("name,nw,mkw", [("cfg3", 512, {}), ("cfg5", 256, {}), ("cfg1", 32, {}),
                                         ("cfg5", 256, {"useLUT": False}), ("cfg2", 256, {}),
                                         ("cfg3", 256, {}), ("cfg3", 48, {}), ("cfg3", 1280, {}),
                                         ("cfg5", 256, {"nEpd": 50}), ("cfg5", 500, {"nEpd": 40, "useLUT": False})],
                         ids=["cfg3-512", "cfg5-256", "cfg1-32", "cfg5-analytic-256", "cfg2-256-two-per-walker",
                              "cfg3-256-two-per-walker", "cfg3-48-eight-per-walker",
                              "cfg3-1280-three-walkers-per-workgroup", "cfg5-lut-300-nodes-rows-in-registers",
                              "cfg5-analytic-240-nodes-rows-in-registers"])
def test_resident_loop_equals_per_launch_loop(na, monkeypatch, name, nw, mkw):
    """nh_half_step_run -- a whole block of moves in ONE launch, walkers handed from half-step
    to half-step through per-walker records (write-through granules with tags) instead of
    through the kernel boundary -- against the same kernel arithmetic launched once per
    half-step (NAIMA_AMD_RESIDENT=0), from the benchmark's ball (uneven load: degenerate
    walkers finish early, so consumers do wait for records): chain, log-probability, blobs
    and acceptance of 100 steps (four blocks of moves, a 4-step tail)"""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, name, mkw)
    nd = p0.size
    kw = dict(args=[data, model, prior], seed=BENCH_SEED, naima_style=True, store_blobs=True,
              nan_policy="reject")  # (cfg5's ball: far-off walkers propose NaN log-probabilities)
    rng = np.random.default_rng(BENCH_SEED)
    pos = p0 + 0.1 * p0 * rng.normal(size=(nw, nd))
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NAIMA_AMD_RESIDENT", mode)
        d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
        with np.errstate(all="ignore"):
            st = d.run_mcmc(pos, 4)
            st = d.run_mcmc(st, 96)
            st2 = d.run_mcmc(st, 7, store=False)  # (no history: blobs go to the current array)
        assert d._dev.mega and d._dev._plan["hs"] is not None
        assert (d._dev.resident_launches > 0) == (mode == "1"), getattr(d._dev, "resident_reason", "")
        if mode == "1" and (name == "cfg1" or "nEpd" in mkw):
            # (a table-only model whose items' rows fit a lane's registers: workgroups of 512
            # threads, the rows loaded once per launch -- hs_rt_item; signed tables (the LUT's) and
            # non-negative ones)
            assert d._dev.resident_info["tables_in_registers"] and d._dev.resident_info["threads"] <= 512, \
                d._dev.resident_info
        if mode == "1" and name == "cfg5" and not mkw:
            # (600 nodes, no zero rows, 128 proposals per half-step on 256 CUs: two workgroups per
            # walker that halve the grid's ROWS -- each forms its half of the weights, reduces its half
            # of the table from registers, and the second hands its partial spectrum and its part of Wp
            # to the first)
            info = d._dev.resident_info
            assert info["rows_split"] and info["workgroups_per_walker"] == 2 and info["tables_in_registers"], info
        if mode == "1" and name == "cfg3":
            # (more walkers of a half-step than resident workgroups: two walkers of a workgroup in
            # flight -- the next one's records, proposal and packs made during this one's work items)
            assert d._dev.resident_info["two_walkers_in_flight"] == (nw // 2 > d._dev.resident_info["grid"])
        if mode == "1" and name in ("cfg3", "cfg1"):
            # (the resident loop walks its own copies of the inverse-Compton tables, columns sorted
            # by their first non-zero row, rows below a tile's first one skipped: same spectra)
            assert d._dev._plan["hs"].get("sorted"), "sorted tables not installed"
        runs[mode] = (d.get_chain(), d.get_log_prob(), d.get_blobs(), d.acceptance_fraction,
                      np.array(st2.coords), np.array(st2.log_prob),
                      [np.array(b) for b in (st2.blobs or [])])
    a, b = runs["0"], runs["1"]
    assert a[0].shape == (100, nw, nd)
    assert_allclose(b[0], a[0], rtol=1e-10)
    assert np.array_equal(np.isinf(b[1]), np.isinf(a[1]))
    fin = np.isfinite(a[1])
    assert_allclose(b[1][fin], a[1][fin], rtol=1e-9)
    for x, y in zip(b[2], a[2]):
        assert_allclose(np.asarray(x, dtype=float), np.asarray(y, dtype=float), rtol=1e-10,
                        atol=1e-300, equal_nan=True)
    assert_allclose(b[3], a[3])
    assert_allclose(b[4], a[4], rtol=1e-10)
    fin = np.isfinite(a[5])
    assert_allclose(b[5][fin], a[5][fin], rtol=1e-9)
    for x, y in zip(b[6], a[6]):
        assert_allclose(np.asarray(x, dtype=float), np.asarray(y, dtype=float), rtol=1e-10,
                        atol=1e-300, equal_nan=True)
    print("%s: resident == per-launch; bit-identical chain: %s" % (name, np.array_equal(a[0], b[0])))


@pytest.mark.parametrize("mkw", [{"useLUT": False}, {}], ids=["analytic-zero-rows", "lut"])
def test_rows_split_with_weights_that_are_not_finite(na, monkeypatch, mkw):
    """cfg5 at 256 walkers -- two workgroups per walker that split the grid's ROWS -- with walkers whose
    amplitude overflows double precision at the grid's low energies and not at its high ones (a prior
    wide enough to let them be evaluated): 0 x inf = NaN for the table's zero rows, as in the
    reference (utils.py:336-348), so the items walk EVERY row -- each workgroup within the nodes it
    formed (the unbalanced chunk ranges reached, for the second workgroup, below them).
    The resident loop against the per-launch kernel (one workgroup per walker or interleaved items:
    every node formed by whoever walks it): the same proposals are NaN, the same are accepted, the
    chains agree"""
    from naima_amd.sampler import EnsembleSampler
    name, nw = "cfg5", 256
    model, p0, raw, data, _ = _problem(na, name, mkw)
    nd = p0.size
    U = na.uniform_prior
    prior = lambda pars: (U(pars[0], 0.0, 1000.0) + U(pars[1], -3, 4) + U(pars[2], -1, 6)  # noqa: E731
                          + U(pars[3], -1, 6) + U(pars[4], -2, 6))
    rng = np.random.default_rng(77)
    pos = p0 + 0.05 * p0 * rng.normal(size=(nw, nd))
    # a quarter of the ensemble far up in log10(amplitude): finite weights (a log-probability of -inf
    # or hugely negative, not NaN), but a stretch towards them overflows at the low-energy rows first
    far = rng.choice(nw, nw // 4, replace=False)
    pos[far, 0] = rng.uniform(262.0, 296.0, size=far.size)
    kw = dict(args=[data, model, prior], seed=BENCH_SEED, naima_style=True, store_blobs=False,
              nan_policy="reject")
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NAIMA_AMD_RESIDENT", mode)
        d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
        with np.errstate(all="ignore"):
            st = d.run_mcmc(pos, 4)
            st = d.run_mcmc(st, 60)
        if mode == "1":
            info = d._dev.resident_info
            assert d._dev.resident_launches > 0 and info["rows_split"] and info["workgroups_per_walker"] == 2, info
        runs[mode] = (d.get_chain(), d.get_log_prob(), d.acceptance_fraction, int(d.nan_proposals))
    a, b = runs["0"], runs["1"]
    assert a[3] > 0, "no proposal of this ensemble had a NaN log-probability: the test tests nothing"
    assert a[3] == b[3], (a[3], b[3])
    assert np.array_equal(np.isnan(a[1]), np.isnan(b[1])) and np.array_equal(np.isinf(a[1]), np.isinf(b[1]))
    fin = np.isfinite(a[1])
    assert_allclose(b[1][fin], a[1][fin], rtol=1e-9)
    assert_allclose(b[0], a[0], rtol=1e-10)
    assert_allclose(b[2], a[2])


@pytest.mark.parametrize("name", ["cfg3", "cfg2"], ids=["cfg3-syn+tables", "cfg2-syn-only"])
@pytest.mark.parametrize("syn2", ["1", "0"], ids=["log-domain", "direct-form"])
def test_two_walkers_in_flight_changes_no_bit(na, monkeypatch, syn2, name):
    """cfg3 (synchrotron + three inverse-Compton tables) and cfg2 (synchrotron alone: no table items)
    with 1280 walkers on 256 CUs: a workgroup takes two or three walkers of every half-step.
    With two of them in flight (the DEEP instance: the next walker's phase A made ahead during this
    one's work items, barrier 1 skipped, the tile waves gated on the previous likelihood having read
    its columns) the arithmetic and its order are the ones of the strictly serial turns
    (NH_RUN_PIPELINE=0): chain, log-probabilities, blobs and acceptance bit for bit -- for the
    synchrotron items in the log domain and in the direct form, through a block boundary and a
    tail, with and without a history"""
    from naima_amd.sampler import EnsembleSampler
    nw = 1280
    model, p0, raw, data, prior = _problem(na, name, {})
    nd = p0.size
    kw = dict(args=[data, model, prior], seed=BENCH_SEED, naima_style=True, store_blobs=True,
              nan_policy="reject")
    pos = p0 + 0.1 * p0 * np.random.default_rng(BENCH_SEED).normal(size=(nw, nd))
    monkeypatch.setenv("NH_RUN_SYN2", syn2)
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NH_RUN_PIPELINE", mode)
        d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
        with np.errstate(all="ignore"):
            st = d.run_mcmc(pos, 3)
            st = d.run_mcmc(st, 45)
            st2 = d.run_mcmc(st, 5, store=False)
        info = d._dev.resident_info
        assert d._dev.resident_launches > 0 and info["two_walkers_in_flight"] == (mode == "1"), info
        assert info["syn_log_domain"] == (syn2 == "1")
        runs[mode] = (d.get_chain(), d.get_log_prob(), d.get_blobs(), d.acceptance_fraction,
                      np.array(st2.coords), np.array(st2.log_prob),
                      [np.array(b) for b in (st2.blobs or [])])
    a, b = runs["0"], runs["1"]
    assert a[0].shape == (48, nw, nd)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True)
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(np.asarray(x, dtype=float), np.asarray(y, dtype=float), equal_nan=True)
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert np.array_equal(a[5], b[5], equal_nan=True)
    for x, y in zip(a[6], b[6]):
        assert np.array_equal(np.asarray(x, dtype=float), np.asarray(y, dtype=float), equal_nan=True)


@pytest.mark.parametrize("name,nw", [("cfg3", 512), ("cfg2", 256), ("cfg3", 96)],
                         ids=["cfg3-512", "cfg2-256-two-per-walker", "cfg3-96-four-per-walker"])
def test_resident_synchrotron_items_in_the_log_domain(na, monkeypatch, name, nw):
    """The resident loop evaluates Synchrotron._spectrum's integrand (radiative.py:300-340) as one
    exponent per node, ln Gtilde from a table on the particle grid's comb (csrc/nh_syn2.h), not
    with the arithmetic of the golden-pinned k_synchrotron.  Direct check: the model spectra
    (blobs) of every walker and step of a resident run from the benchmark's ball against the
    host-driven loop -- whose spectra are k_synchrotron's, held to the reference's at 1e-10 by
    test_gpu_parity -- at 2e-11, and against the same resident loop with the direct form
    (NH_RUN_SYN2=0)."""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, name, {})
    nd = p0.size
    kw = dict(args=[data, model, prior], seed=BENCH_SEED, naima_style=True, store_blobs=True,
              nan_policy="reject")
    rng = np.random.default_rng(BENCH_SEED)
    pos = p0 + 0.1 * p0 * rng.normal(size=(nw, nd))
    steps = 12
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NH_RUN_SYN2", mode)
        d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
        with np.errstate(all="ignore"):
            st = d.run_mcmc(pos, 4)
            st = d.run_mcmc(st, steps)
        assert d._dev.resident_launches > 0, getattr(d._dev, "resident_reason", "")
        info = d._dev.resident_info
        assert info["syn_log_domain"] == (mode == "1"), info
        if mode == "1":
            assert info["syn_nodes_per_piece"] in (2, 4) and 200 < info["syn_pieces"] < 480, info
        out[mode] = (d.get_chain(), d.get_log_prob(), [np.asarray(b, dtype=float) for b in d.get_blobs()])
    monkeypatch.delenv("NH_RUN_SYN2")
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    with np.errstate(all="ignore"):
        sh = h.run_mcmc(pos, 4)
        sh = h.run_mcmc(sh, steps)
    ref = (h.get_chain(), h.get_log_prob(), [np.asarray(b, dtype=float) for b in h.get_blobs()])
    for other in (out["0"], ref):
        assert_allclose(out["1"][0], other[0], rtol=1e-10)
        fin = np.isfinite(other[1])
        assert np.array_equal(fin, np.isfinite(out["1"][1]))
        assert_allclose(out["1"][1][fin], other[1][fin], rtol=1e-9)
        for x, y in zip(out["1"][2], other[2]):
            assert x.shape == y.shape
            assert_allclose(x, y, rtol=2e-11, atol=1e-300, equal_nan=True)
    spec = out["1"][2][0]
    assert spec.shape[:2] == (4 + steps, nw) and np.isfinite(spec).mean() > 0.9


def test_table_only_model_with_more_walkers_than_compute_units(na):
    """cfg5 at BASELINE's 2048 walkers on one GPU: 1024 walkers per half-step, four per compute
    unit -- the plan picks 256-thread workgroups (several walkers share a CU, one's prologue
    beside another's items; the resident loop declines: its workgroups would each take several
    walkers per slice): device loop == host-driven loop"""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, "cfg5", {})
    nw, nd = 2048, p0.size
    kw = dict(args=[data, model, prior], seed=41, naima_style=True, store_blobs=True)
    pos = p0 * (1 + 0.01 * np.random.default_rng(3).standard_normal((nw, nd)))
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    sh, sd = h.run_mcmc(pos, 3), d.run_mcmc(pos, 3)
    sh, sd = h.run_mcmc(sh, 37), d.run_mcmc(sd, 37)
    dev = d._dev
    assert dev.mega and dev._plan["hs"] is not None and dev._plan["hs"]["threads"] == 256
    assert_allclose(d.get_chain(), h.get_chain(), rtol=1e-8)
    assert_allclose(d.get_log_prob(), h.get_log_prob(), rtol=1e-6)
    assert_allclose(d.acceptance_fraction, h.acceptance_fraction)
    for x, y in zip(d.get_blobs(), h.get_blobs()):
        assert_allclose(np.asarray(x, dtype=float), np.asarray(y, dtype=float), rtol=1e-8,
                        atol=1e-300)


@pytest.mark.parametrize("mega", ["1", "0", "per-launch"],
                         ids=["one-launch", "separate-kernels", "one-launch-per-half-step"])
def test_nan_log_probability_is_emcees_error(na, monkeypatch, mega):
    """a proposal whose log-probability is NaN ends an emcee run with ValueError("Probability
    function returned NaN") (EnsembleSampler.compute_log_prob; the reference just lets it
    through, core.py:128): the host-driven loop raises on the spot, the device loop -- whose
    launches cannot -- when the run's results next reach the host.  cfg2 from the benchmark's
    ball under round 3's prior (the amplitude only) meets its first such proposal -- a negative
    magnetic field -- within ~30 steps."""
    from naima_amd.sampler import EnsembleSampler
    if mega == "per-launch":
        # (k_half_step<true, true> instead of the resident loop: its log-domain synchrotron items
        # take ln B -- the NaN is made by its own test of the field, as in k_half_step_run)
        monkeypatch.setenv("NAIMA_AMD_RESIDENT", "0")
        mega = "1"
    monkeypatch.setenv("NAIMA_AMD_MEGA", mega)  # ("0": the accept rides in k_lnprobmodel / k_synchrotron)
    model, p0, raw, data, prior = _problem(na, "cfg2", {})
    prior = _loose_prior(na, "cfg2")  # (the benchmark's prior forbids B < 0)
    nw, nd = 256, p0.size
    rng = np.random.default_rng(BENCH_SEED)
    pos = p0 + 0.1 * p0 * rng.normal(size=(nw, nd))
    pos[0] = p0
    pos[0, 0] = -0.03 * p0[0]
    kw = dict(args=[data, model, prior], seed=BENCH_SEED, naima_style=True, store_blobs=False)
    h = EnsembleSampler(nw, nd, na.lnprob, **kw)
    with pytest.raises(ValueError, match="returned NaN"), np.errstate(all="ignore"):
        h.run_mcmc(pos, 70)
    d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    with pytest.raises(ValueError, match="returned NaN"), np.errstate(all="ignore"):
        st = d.run_mcmc(pos, 2)
        st = d.run_mcmc(st, 68)
        st.coords
    assert d._dev.mega == (mega == "1")
    # ... and counted instead, with nan_policy="reject": the same number in either loop
    kw["nan_policy"] = "reject"
    h2 = EnsembleSampler(nw, nd, na.lnprob, **kw)
    d2 = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
    with np.errstate(all="ignore"):
        h2.run_mcmc(pos, 40)
        st = d2.run_mcmc(pos, 2)
        st = d2.run_mcmc(st, 38)
        st.coords
    assert h2.nan_proposals > 0 and d2.nan_proposals == h2.nan_proposals


def test_resident_loop_gives_up_instead_of_hanging(na, monkeypatch):
    """every wait of the resident loop is bounded: with a poll limit of ONE (NH_RUN_SPIN_LIMIT=1)
    the first walker whose record is not there yet makes its workgroup latch an error and leave,
    every other workgroup follows, the launch ends -- the device is not left spinning -- and its
    epilogue leaves the ensemble, the counters and the move stream as they were.  The loop's
    first launches are checked as they are made: the sampler warns once, replays the block of
    moves with the per-launch kernel and stays with it -- the chain is the per-launch loop's."""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, "cfg3", {})
    nw, nd = 512, p0.size
    pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((nw, nd)))
    kw = dict(args=[data, model, prior], seed=5, naima_style=True, store_blobs=True, device=True)
    monkeypatch.setenv("NAIMA_AMD_RESIDENT", "0")
    ref = EnsembleSampler(nw, nd, na.lnprob, **kw)
    st = ref.run_mcmc(pos, 4)
    st = ref.run_mcmc(st, 40)
    assert ref._dev.resident_launches == 0
    monkeypatch.delenv("NAIMA_AMD_RESIDENT")
    monkeypatch.setenv("NH_RUN_SPIN_LIMIT", "1")
    d = EnsembleSampler(nw, nd, na.lnprob, **kw)
    st = d.run_mcmc(pos, 4)
    with pytest.warns(UserWarning, match="gave up waiting"):
        st = d.run_mcmc(st, 40)
    assert d._dev.resident_failed_launches == 1 and d._dev.resident_launches == 0
    assert np.array_equal(d.get_chain(), ref.get_chain())
    assert np.array_equal(d.get_log_prob(), ref.get_log_prob())
    for x, y in zip(d.get_blobs(), ref.get_blobs()):
        assert np.array_equal(np.asarray(x, dtype=float), np.asarray(y, dtype=float), equal_nan=True)
    assert_allclose(d.acceptance_fraction, ref.acceptance_fraction)
    assert d.nan_proposals == ref.nan_proposals
    assert d.prior_forbidden_proposals == ref.prior_forbidden_proposals
    st = d.run_mcmc(st, 10, store=False)  # (and it carries on)
    # the context is still usable: a fresh sampler with the default limit runs resident
    monkeypatch.delenv("NH_RUN_SPIN_LIMIT")
    d2 = EnsembleSampler(nw, nd, na.lnprob, args=[data, model, prior], seed=5, naima_style=True,
                         store_blobs=False, device=True)
    st = d2.run_mcmc(pos, 4)
    st = d2.run_mcmc(st, 40)
    assert np.all(np.isfinite(st.coords)) and d2._dev.resident_launches > 0
    assert np.array_equal(d2.get_chain(), ref.get_chain())


@pytest.mark.parametrize("fail_at,blobs", [(4, True), (7, False), (9, True), (10, True), (11, True)],
                         ids=["fourth-launch", "seventh-no-blobs", "ninth-in-a-call-without-history",
                              "tenth-the-last-of-a-call-without-history", "a-call's-last-launch"])
def test_a_later_launch_that_gives_up_is_replayed(na, monkeypatch, fail_at, blobs):
    """a launch of the resident loop that gives up LATER in a run (NH_RUN_FAIL_AT: that launch's
    first wait times out; what another process taking the GPU's CUs would do): the sampler learns
    of it one launch later from the page-locked report its epilogue wrote (nothing synchronises
    per launch), the launch queued behind it found the same status and changed nothing, the books
    go back two blocks of moves, the move stream is made again up to that step and the per-launch
    kernel carries the run on -- chain, log-probabilities, blobs, acceptance and counters are the
    per-launch loop's, bit for bit.  Launches 8-10 are the call WITHOUT a history, whose launches
    write accepted blobs straight into the current-blob arrays: the launch queued behind one that
    gave up runs to its end as a void launch and writes blobs of proposals that are then
    discarded -- the current blobs go back to the copy taken ahead of the first void launch, or
    the blob rows of the next call's rejected steps would belong to positions never held."""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, "cfg3", {})
    nw, nd = 512, p0.size
    pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((nw, nd)))
    kw = dict(args=[data, model, prior], seed=5, naima_style=True, store_blobs=blobs, device=True)

    def run(d):
        st = d.run_mcmc(pos, 4)
        st = d.run_mcmc(st, 200)   # (seven launches of 32 steps at most)
        st = d.run_mcmc(st, 70, store=False)
        st = d.run_mcmc(st, 50)    # (launch 11 is the last of the third call when nothing failed)
        return st

    monkeypatch.setenv("NAIMA_AMD_RESIDENT", "0")
    ref = EnsembleSampler(nw, nd, na.lnprob, **kw)
    sr = run(ref)
    monkeypatch.delenv("NAIMA_AMD_RESIDENT")
    monkeypatch.setenv("NH_RUN_FAIL_AT", str(fail_at))
    d = EnsembleSampler(nw, nd, na.lnprob, **kw)
    with pytest.warns(UserWarning, match="gave up waiting"):
        sd = run(d)
    assert d._dev.resident_failed_launches >= 1 and d._dev.resident_launches == fail_at - 1, \
        (d._dev.resident_failed_launches, d._dev.resident_launches)
    # (the same accept decisions, hence the same positions to the last bit; the launches that ran
    # resident before the one that gave up cut the tables' rows into work items differently: their
    # log-probabilities and spectra agree to rounding, see test_resident_loop_equals_per_launch_loop)
    assert np.array_equal(d.get_chain(), ref.get_chain())
    assert_allclose(d.get_log_prob(), ref.get_log_prob(), rtol=1e-11)
    assert np.array_equal(np.asarray(sd.coords), np.asarray(sr.coords))
    if blobs:
        for x, y in zip(sd.blobs, sr.blobs):  # (the current blobs: where every walker IS)
            assert_allclose(np.asarray(x, dtype=float), np.asarray(y, dtype=float), rtol=1e-10, atol=1e-300)
        for x, y in zip(d.get_blobs(), ref.get_blobs()):
            assert_allclose(np.asarray(x, dtype=float), np.asarray(y, dtype=float), rtol=1e-10, atol=1e-300,
                            equal_nan=True)
    assert_allclose(d.acceptance_fraction, ref.acceptance_fraction)
    assert d.nan_proposals == ref.nan_proposals
    assert d.prior_forbidden_proposals == ref.prior_forbidden_proposals
    assert d.iteration == ref.iteration and d.steps_total == ref.steps_total


def test_resident_loop_through_many_blocks_of_moves(na, monkeypatch):
    """one run_mcmc call of 330 steps: eleven launches queued back to back, the device block of
    moves (a ring of 128 steps, refilled on a copy stream while launches run) wraps twice --
    against the per-launch loop, which uploads every block in stream order.  A refill that
    overtook a launch still reading the old bytes would change the chain from there on."""
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, "cfg5", {})
    nw, nd = 256, p0.size
    pos = p0 * (1 + 0.01 * np.random.default_rng(2).standard_normal((nw, nd)))
    kw = dict(args=[data, model, prior], seed=77, naima_style=True, store_blobs=False,
              nan_policy="reject")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NAIMA_AMD_RESIDENT", mode)
        d = EnsembleSampler(nw, nd, na.lnprob, device=True, **kw)
        st = d.run_mcmc(pos, 4)
        st = d.run_mcmc(st, 330)
        st = d.run_mcmc(st, 50)
        out[mode] = (d.get_chain(), d.get_log_prob(), d._dev.resident_launches)
    assert out["0"][2] == 0 and out["1"][2] >= 13
    # (the same accept decisions, hence the same positions to the last bit; the log-probabilities
    # to rounding -- the resident loop cuts a table's rows into its work items differently, see
    # hs_run.rebalance)
    assert np.array_equal(out["1"][0], out["0"][0])
    assert_allclose(out["1"][1], out["0"][1], rtol=1e-12)


@pytest.mark.parametrize("name,nw,nranks", [("cfg1", 32, 3), ("cfg3", 40, 4)], ids=["cfg1-3", "cfg3-4"])
def test_sharded_loop_three_ranks_again_and_again(na, name, nw, nranks):
    """three (four) processes on the one GPU, a fresh sampler a hundred times over: every
    repetition's chain is the single process's.  (What this caught: a plan's counters zeroed by
    the NULL stream's hipMemset, which orders nothing against the context's non-blocking streams
    and for device memory returns before the fill -- with three processes on the GPU the fill
    landed some launches later about once in thirty samplers and reset the slice counter in the
    middle of a block of moves.  nh_fill_now.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + (os.getpid() % 1000)
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nranks,
         "--master-addr", "127.0.0.1", "--master-port", str(port),
         os.path.join(root, "tests", "gpu_repeat_ranks_worker.py"), name, str(nw), "100", "5"],
        cwd=root, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))


@pytest.mark.parametrize("name,nw,nranks", [("cfg3", 32, 2), ("cfg5", 64, 2), ("cfg2", 48, 2),
                                            ("cfg3", 30, 2), ("cfg3", 40, 4), ("cfg1", 32, 3)],
                         ids=["cfg3-32", "cfg5-64", "cfg2-48", "cfg3-30-uneven-blocks",
                              "cfg3-40-four-ranks", "cfg1-32-three-ranks-uneven"])
def test_shared_ensemble_two_ranks_one_gpu(na, tmp_path, name, nw, nranks):
    """the resident loop over an ensemble SHARED by two (three, four) ranks
    (nh_half_step_run_create_shared): one process per rank, here all on the one GPU of the box,
    each mapping the others' rings through hipIpc; a mover stores its walker's record into every
    ring, nobody launches or gathers per half-step.  Every rank ends with the same ensemble, chain, log-probabilities, blobs (history
    rows gathered from whoever moved the walker, rows of rejected moves filled from the row
    before; current blobs merged by stamp after a call without history) and acceptance counts
    as one process on one GPU -- bit for bit: the walkers' arithmetic is the same code."""
    import os
    import subprocess
    import sys
    from naima_amd.sampler import EnsembleSampler
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29800 + (os.getpid() % 1000)
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node=%d" % nranks, "--master-addr", "127.0.0.1", "--master-port", str(port),
         os.path.join(root, "tests", "gpu_shared_ranks_worker.py"), str(tmp_path), name, str(nw)],
        cwd=root, timeout=600,
        env=dict(os.environ, MASTER_ADDR="127.0.0.1", NH_RUN_SPIN_LIMIT=str(1 << 24)))
    model, p0, raw, data, prior = _problem(na, name, {})
    nd = p0.size
    s = EnsembleSampler(nw, nd, na.lnprob, args=[data, model, prior], seed=42, naima_style=True,
                        store_blobs=True, device=True, nan_policy="reject")
    pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((nw, nd)))
    st = s.run_mcmc(pos, 5)
    st = s.run_mcmc(st, 70)
    st = s.run_mcmc(st, 9, store=False)
    st = s.run_mcmc(st, 4)
    for st in s.sample(st, iterations=3):
        pass
    st = s.run_mcmc(st, 5)
    want = dict(coords=st.coords, logp=st.log_prob, curblob0=np.asarray(st.blobs[0]),
                curblob1=np.asarray(st.blobs[1]), chain=s.get_chain(), lnp=s.get_log_prob(),
                blob0=np.asarray(s.get_blobs()[0]), blob1=np.asarray(s.get_blobs()[1]),
                acc=s.acceptance_fraction)
    assert want["chain"].shape[0] == 87
    # (first: every rank holds what rank 0 holds -- a failure here is the shared loop's, one below
    # is a difference between the shared loop and one process)
    for r in range(1, nranks):
        for key in want:
            assert np.array_equal(np.load(tmp_path / ("%s_%d.npy" % (key, r))),
                                  np.load(tmp_path / ("%s_0.npy" % key)), equal_nan=True), (key, r)
    for r in range(nranks):
        for key, w in want.items():
            have = np.load(tmp_path / ("%s_%d.npy" % (key, r)))
            assert have.shape == w.shape, (key, r)
            assert_allclose(have, w, rtol=1e-10, atol=1e-300, err_msg="%s of rank %d" % (key, r))


@pytest.mark.parametrize("fail_at,flow", [(3, "plain"), (4, "plain"), (3, "reset")],
                         ids=["third-launch", "fourth-launch", "third-launch-after-reset"])
def test_shared_loop_that_gives_up_is_replayed_by_all_ranks(na, tmp_path, fail_at, flow):
    """three ranks on the one GPU share cfg3's ensemble; one launch of ONE rank's resident loop is
    made to time out (NH_RUN_FAIL_AT: what another process taking the GPU's CUs would do).  The
    other ranks' launches starve of its records and give up too; nobody raises: at the next point
    where every rank is (a flush) the status is reduced over the ranks, all of them go back to the
    ensemble they kept at the previous such point, make the move stream again and repeat the steps
    with one launch and one all-gather per half-step -- the final chain, log-probabilities, blobs
    and acceptance equal one process's (the reference's Pool carries on likewise,
    core.py:523-536).  `reset`: the reference's burn-in -> reset -> run flow (core.py:483-487,
    529-530) with the failure behind the reset -- the kept ensemble is the one BEHIND the reset
    (iteration 0, acceptance counters 0), so the acceptance fraction of the replayed run is the
    run's, not the burn-in's plus the run's."""
    import os
    import subprocess
    import sys
    from naima_amd.sampler import EnsembleSampler
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29900 + (os.getpid() % 1000)
    nw, name = 30, "cfg3"
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
         "--master-addr", "127.0.0.1", "--master-port", str(port),
         os.path.join(root, "tests", "gpu_shared_fail_worker.py"), str(tmp_path), name, str(nw), str(fail_at),
         flow],
        cwd=root, timeout=900,
        env=dict(os.environ, MASTER_ADDR="127.0.0.1", NH_RUN_SPIN_LIMIT=str(1 << 22)))
    model, p0, raw, data, prior = _problem(na, name, {})
    nd = p0.size
    s = EnsembleSampler(nw, nd, na.lnprob, args=[data, model, prior], seed=42, naima_style=True,
                        store_blobs=True, device=True, nan_policy="reject")
    pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((nw, nd)))
    st = s.run_mcmc(pos, 5)
    st = s.run_mcmc(st, 40)
    if flow == "reset":
        s.reset()
    st = s.run_mcmc(st, 50)
    st = s.run_mcmc(st, 9, store=False)
    st = s.run_mcmc(st, 12)
    want = dict(coords=st.coords, logp=st.log_prob, chain=s.get_chain(), lnp=s.get_log_prob(),
                blob0=np.asarray(s.get_blobs()[0]), blob1=np.asarray(s.get_blobs()[1]),
                curblob0=np.asarray(st.blobs[0]), acc=s.acceptance_fraction)
    assert want["chain"].shape[0] == (62 if flow == "reset" else 107)
    for r in range(3):
        for key, w in want.items():
            have = np.load(tmp_path / ("%s_%d.npy" % (key, r)))
            assert have.shape == w.shape, (key, r)
            # (the steps made again ran one launch per half-step: the resident loop's arithmetic to
            # rounding, test_resident_loop_equals_per_launch_loop)
            assert_allclose(have, w, rtol=1e-9 if key in ("logp", "lnp") else 1e-10, atol=1e-300,
                            err_msg="%s of rank %d" % (key, r))


def test_sorted_table_trailers_are_checked_at_the_abi(na):
    """nh_half_step_run_tables reads the trailers of the table copies back and refuses a first
    row out of range or a column order that is not a permutation (they decide where the kernel
    writes in LDS); the loop keeps the tables it had"""
    import ctypes as C
    from naima_amd import _lib
    from naima_amd.sampler import EnsembleSampler
    model, p0, raw, data, prior = _problem(na, "cfg1", {})
    nd = p0.size
    d = EnsembleSampler(32, nd, na.lnprob, args=[data, model, prior], seed=5, naima_style=True,
                        store_blobs=True, device=True)
    pos = p0 * (1 + 0.003 * np.random.default_rng(1).standard_normal((32, nd)))
    st = d.run_mcmc(pos, 8)
    st = d.run_mcmc(st, 8)
    dev = d._dev
    hs = dev._plan["hs"]
    assert dev.resident_launches > 0 and hs.get("sorted")
    ctx = dev.ctx
    (Kt, dKt, nG, nK, lx, nonneg), good = hs["tabs"][0], hs["sorted"][0][0]
    host = good.get()
    trail = host[2 * nG * nK:].view(np.int32).copy()
    for what, (idx, val) in {"permutation": (8 + 3, nK), "first row": (0, nG + 1)}.items():
        bad_t = trail.copy()
        bad_t[idx] = val
        bad = host.copy()
        bad[2 * nG * nK:] = bad_t.view(np.float64)
        buf = ctx.array(bad)
        ptrs = (C.c_void_p * 4)(buf.ptr, None, None, None)
        rc = _lib._lib.nh_half_step_run_tables(ctx.h, hs["plan"], dev._run, ptrs, 4)
        assert rc != 0 and what in _lib._lib.nh_last_error().decode(), what
    ref = EnsembleSampler(32, nd, na.lnprob, args=[data, model, prior], seed=5, naima_style=True,
                          store_blobs=True, device=True)
    st2 = ref.run_mcmc(ref.run_mcmc(pos, 8), 8)
    st, st2 = d.run_mcmc(st, 40), ref.run_mcmc(st2, 40)
    assert np.array_equal(st.coords, st2.coords)
