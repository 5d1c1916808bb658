"""The ensemble driver on CPU: move statistics, shapes, sharding over ranks with
gloo (world_size 2), and the device-side propose/accept formulas' host twins.
The log-probability here is either analytic or the NumPy oracle (tests may use
the oracle as a stand-in evaluator; the product never does)."""
import os
import subprocess
import sys

import numpy as np
import pytest
from numpy.testing import assert_allclose

from naima_amd.dist import LocalComm, shard_bounds, shard_counts
from naima_amd.sampler import EnsembleSampler, State

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gauss(x):
    return -0.5 * np.sum((x - 1.5) ** 2 / 0.25, axis=1), np.sum(x, axis=1), x[:, :2] * 2.0


def test_shard_bounds():
    for n in (0, 1, 7, 128, 255):
        for size in (1, 2, 3, 8):
            cnt = shard_counts(n, size)
            assert sum(cnt) == n and max(cnt) - min(cnt) <= 1
            assert shard_bounds(n, size - 1, size)[1] == n


def test_stretch_move_statistics_and_surface():
    s = EnsembleSampler(64, 3, gauss, seed=1)
    p0 = np.random.default_rng(0).normal(size=(64, 3))
    st = s.run_mcmc(p0, 700)
    c = s.get_chain(discard=200, flat=True)
    assert_allclose(c.mean(0), 1.5, atol=0.06)
    assert_allclose(c.std(0), 0.5, atol=0.05)
    assert 0.3 < s.acceptance_fraction.mean() < 0.9
    assert s.get_chain().shape == (700, 64, 3) and s.chain.shape == (64, 700, 3)
    assert s.get_log_prob().shape == (700, 64) and s.lnprobability.shape == (64, 700)
    b = s.get_blobs()
    assert b[0].shape == (700, 64) and b[1].shape == (700, 64, 2)
    # blobs follow the accepted walkers
    assert_allclose(b[0][-1], st.coords.sum(axis=1))
    assert_allclose(b[1][-1], st.coords[:, :2] * 2)
    assert_allclose(st.log_prob, gauss(st.coords)[0])
    s.reset()
    assert s.iteration == 0 and s.get_chain().shape[0] == 0
    with pytest.raises(ValueError):
        EnsembleSampler(5, 3, gauss)
    with pytest.raises(ValueError):
        EnsembleSampler(8, 3, lambda x: np.full(len(x), np.nan)).run_mcmc(p0[:8], 1)


def test_same_seed_same_chain_and_restart():
    a = EnsembleSampler(32, 3, gauss, seed=5)
    b = EnsembleSampler(32, 3, gauss, seed=5)
    p0 = np.random.default_rng(2).normal(size=(32, 3))
    sa = a.run_mcmc(p0, 20)
    sb = b.run_mcmc(p0, 10)
    sb = b.run_mcmc(sb, 10)  # continuing from a State does not re-evaluate
    assert_allclose(sa.coords, sb.coords)


def test_reference_move_restatement_agrees():
    """sampler.sample == oracle.stretch_move_reference fed with the same move stream;
    the stream is independent of how many steps are taken at once"""
    from naima_amd._lib import Moves
    from oracle import naima_np as O
    p0 = np.random.default_rng(3).normal(size=(16, 2))
    lp = lambda x: -0.5 * np.sum(x ** 2, axis=1)  # noqa: E731
    s = EnsembleSampler(16, 2, lp, seed=9, store_blobs=False)
    st = s.run_mcmc(p0, 5)
    m = Moves(9, 16, 2.0, ksteps=32, depth=4)
    c, l = p0.copy(), lp(p0)
    addr, got = m.take(5)
    S, P, Z, L = m.view(addr, got)
    for k in range(5):
        c, l, _ = O.stretch_move_reference(c, l, lp, S[k], P[k], Z[k], L[k])
    assert_allclose(st.coords, c)
    assert_allclose(st.log_prob, l)
    m2 = Moves(9, 16, 2.0, ksteps=3, depth=3)  # other blocking, one step at a time
    for k in range(5):
        a2, g2 = m2.take(1)
        S2, P2, Z2, L2 = m2.view(a2, g2)
        assert g2 == 1 and (S2[0] == S[k]).all() and (P2[0] == P[k]).all()
        assert_allclose(Z2[0], Z[k]) and assert_allclose(L2[0], L[k])
    # every step: a permutation split in halves, partners from the other half
    for k in range(5):
        assert sorted(np.concatenate([S[k, 0], S[k, 1]])) == list(range(16))
        assert np.isin(P[k, 0], S[k, 1]).all() and np.isin(P[k, 1], S[k, 0]).all()
    assert (Z >= 0.5).all() and (Z <= 2.0).all() and (L <= 0).all()


def test_move_ring_keeps_the_previous_block_intact():
    """nh_moves_take's lifetime contract (the device loop ships a block with an asynchronous
    copy and takes the next one while that copy may still be queued): the memory of take
    j-1 is untouched while take j is current -- the producer gets block j-1 back only when
    the consumer moves on to j+1 -- whatever the blocking and however eager the producer"""
    import time

    from naima_amd._lib import Moves
    for ks, depth, takes in ((4, 3, (4,) * 12), (4, 4, (4,) * 12), (5, 3, (2, 3, 5, 1, 4, 5, 5))):
        ref = Moves(77, 64, 2.0, ksteps=64, depth=3)
        a, g = ref.take(sum(takes))
        want = np.concatenate([v.reshape(g, -1) for v in ref.view(a, g)], axis=1).copy()
        m = Moves(77, 64, 2.0, ksteps=ks, depth=depth)
        prev, done = None, 0
        for t in takes:
            addr, got = m.take(t)
            assert got == t  # (the takes above never straddle a block)
            time.sleep(0.01)  # let the producer run as far ahead as the ring allows
            cur = np.concatenate([v.reshape(got, -1) for v in m.view(addr, got)], axis=1)
            assert_allclose(cur, want[done:done + got])
            if prev is not None:  # the previous take's memory, read again NOW
                pa, pg, pd = prev
                old = np.concatenate([v.reshape(pg, -1) for v in m.view(pa, pg)], axis=1)
                assert_allclose(old, want[pd:pd + pg])
            prev, done = (addr, got, done), done + got


def test_naima_style_with_oracle_model(golden):
    """naima's (pars, data) -> (flux, blob) contract through the sampler, with the
    oracle standing in for the GPU evaluator (cfg1, 16 walkers, 3 steps)"""
    from oracle import workloads_np as WN
    z = golden("cfg1")
    raw = WN.raw_from_npz(z)

    def lnprob_batch(parsT, raw_):
        out = [WN.lnprob("cfg1", p, raw_) for p in np.asarray(parsT).T]
        return (np.array([o[0] for o in out]), np.array([o[1] for o in out]))

    p0 = z["pars"][0]
    s = EnsembleSampler(16, 3, lnprob_batch, args=[raw], seed=3, naima_style=True)
    pos = p0 * (1 + 0.01 * s._rng.normal(size=(16, 3)))
    st = s.run_mcmc(pos, 3)
    assert s.get_blobs()[0].shape == (3, 16, 28)
    assert np.all(np.isfinite(st.log_prob))


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from naima_amd.sampler import EnsembleSampler
if %(backend)r == "gloo":
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    from gloo_comm import GlooComm
    comm = GlooComm()
else:
    from naima_amd import dist
    comm = dist.from_env("host")
    assert type(comm).__name__ == "HostComm" and "torch" not in sys.modules
calls = []
def lp(x):
    calls.append(len(x))
    return -0.5 * np.sum((x - 1.5) ** 2 / 0.25, axis=1), np.sum(x, axis=1)
s = EnsembleSampler(30, 3, lp, seed=11, comm=comm)
p0 = np.random.default_rng(0).normal(size=(30, 3))
st = s.run_mcmc(p0, 25)
np.save(os.path.join(%(out)r, "coords_%%d.npy" %% comm.rank), st.coords)
np.save(os.path.join(%(out)r, "logp_%%d.npy" %% comm.rank), st.log_prob)
np.save(os.path.join(%(out)r, "calls_%%d.npy" %% comm.rank), np.array(calls))
# the blob of this log-probability is a function of the position: every rank holds the
# blob that belongs to every walker's CURRENT position, whoever evaluated it
blob = s.get_blobs()[0]
assert blob.shape == (25, 30)
np.testing.assert_allclose(blob, s.get_chain().sum(axis=2), rtol=1e-13)
g = comm.allgather(np.full((2, 3), float(comm.rank)))
assert g.shape == (4, 3) and g[0, 0] == 0 and g[3, 0] == 1
assert comm.max(comm.rank) == 1.0
comm.barrier()
'''


@pytest.mark.parametrize("backend", ["gloo", "host"])
def test_two_ranks_gloo_match_single_process(tmp_path, backend):
    """world_size 2 -- over torch.distributed/gloo (test-only helper tests/gloo_comm.py) and
    over the product's own torch-free control plane (dist.HostComm): each rank evaluates
    its shard, one all-gather per half-step, and the ensemble is identical on both ranks
    and to a 1-rank run"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path), "backend": backend})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() % 2000)
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
        env=env, cwd=ROOT, timeout=300)
    c0, c1 = np.load(tmp_path / "coords_0.npy"), np.load(tmp_path / "coords_1.npy")
    assert_allclose(c0, c1)
    assert_allclose(np.load(tmp_path / "logp_0.npy"), np.load(tmp_path / "logp_1.npy"))

    def lp(x):
        return -0.5 * np.sum((x - 1.5) ** 2 / 0.25, axis=1), np.sum(x, axis=1)
    s = EnsembleSampler(30, 3, lp, seed=11)
    st = s.run_mcmc(np.random.default_rng(0).normal(size=(30, 3)), 25)
    assert_allclose(st.coords, c0)
    calls0, calls1 = np.load(tmp_path / "calls_0.npy"), np.load(tmp_path / "calls_1.npy")
    # 15 proposals per half-step split 8 + 7 (initial evaluation: 30 split 15 + 15)
    assert calls0[0] == 15 and calls1[0] == 15
    assert set(calls0[1:]) == {8} and set(calls1[1:]) == {7}


def test_save_read_run_roundtrip(tmp_path):
    """naima's save_run layout (mcmc/chain, log_prob, blobN, data + attributes) in .npz"""
    import naima_amd as na
    from naima_amd import units as u
    from naima_amd.datatable import make_data
    s = EnsembleSampler(16, 3, gauss, seed=4)
    s.run_mcmc(np.random.default_rng(0).normal(size=(16, 3)), 7)
    s.labels = ["norm", "index", "cutoff"]
    s.run_info = {"n_walkers": 16, "n_burn": 0, "n_run": 7}
    n = 5
    s.data = make_data(dict(energy=np.geomspace(1, 10, n), energy_unit="TeV",
                            flux=np.ones(n), flux_error_lo=0.1 * np.ones(n),
                            flux_error_hi=0.1 * np.ones(n), ul=np.zeros(n, bool), cl=0.9,
                            flux_unit="1/(cm2 s TeV)"))
    fn = na.save_run(str(tmp_path / "run"), s)
    r = na.read_run(fn)
    assert_allclose(r.get_chain(), s.get_chain())
    assert_allclose(r.get_log_prob(), s.get_log_prob())
    assert_allclose(r.get_blobs()[1], s.get_blobs()[1])
    assert r.labels == s.labels and r.run_info["n_run"] == 7
    assert r.data["energy"].unit == u.TeV and r.data["flux"].unit.physical_type == "differential flux"
    assert r.chain.shape == (16, 7, 3) and r.flatchain.shape == (112, 3)
    with pytest.raises(OSError):
        na.save_run(str(tmp_path / "run"), s)


def test_save_results_table(tmp_path):
    """analysis.save_results_table: medians / 1-sigma distances per parameter, de-logged
    labels, scalar blobs, run info + ML + BIC as ECSV metadata (analysis.py:165-363)"""
    import yaml
    from naima_amd.analysis import read_run, save_results_table, save_run
    from naima_amd.datatable import read_ecsv
    rng = np.random.default_rng(0)

    def lp(x):
        return -0.5 * np.sum((x - np.array([1.0, 2.0])) ** 2, axis=1), x[:, 0] ** 2

    s = EnsembleSampler(20, 2, lp, seed=3)
    s.run_mcmc(np.array([1.0, 2.0]) + 0.1 * rng.normal(size=(20, 2)), 60)
    s.labels = ["log10(norm)", "index"]
    s.run_info = {"n_walkers": 20, "n_burn": 0, "n_run": 60, "p0": [1.0, 2.0]}
    s.data = {"energy": np.arange(7.0)}
    t = save_results_table(str(tmp_path / "fit"), s)
    assert t["label"] == ["log10(norm)", "norm", "index", "blob0"]
    flat = s.get_chain(flat=True)
    assert_allclose(t["median"][0], np.median(flat[:, 0]))
    assert_allclose(t["median"][1], np.median(10 ** flat[:, 0]))
    assert_allclose(t["unc_hi"][2], np.percentile(flat[:, 1], 84) - np.median(flat[:, 1]))
    assert_allclose(t["median"][3], np.median(np.asarray(s.get_blobs()[0])))
    assert t["meta"]["n_samples"] == 1200 and t["meta"]["n_run"] == 60
    assert_allclose(t["meta"]["BIC"], 2 * np.log(7) - 2 * t["meta"]["MaxLogLikelihood"])
    assert_allclose(t["meta"]["MaxLogLikelihood"], np.max(s.get_log_prob()))
    text = open(tmp_path / "fit_results.ecsv").read().splitlines()
    assert text[0] == "# %ECSV 1.0"
    hdr = yaml.safe_load("\n".join(l[2:] for l in text[2:] if l.startswith("# ")))
    assert [c["name"] for c in hdr["datatype"]] == ["label", "median", "unc_lo", "unc_hi"]
    assert hdr["meta"]["ML_pars"] == t["meta"]["ML_pars"]
    with pytest.raises(OSError):
        save_results_table(str(tmp_path / "fit"), s)
    t2 = save_results_table(str(tmp_path / "fit"), s, last_step=True, overwrite=True)
    assert t2["meta"]["n_samples"] == 20


def test_save_run_layout_is_the_reference_hdf5_layout(tmp_path):
    """results on disk pinned by the reference: save_run_schema.json holds every object the
    reference's save_run (analysis.py:366-471) wrote for the run in run_inputs.npz -- names,
    shapes, dtypes, attributes, astropy's column-meta lines; ``run_layout`` (what this
    build's save_run writes, to HDF5 where h5py exists, to .npz elsewhere) must produce the
    same objects.  The schema also records that the reference's read_run read a file written
    by this build (checked when the schema was generated: gen_golden_run.py)."""
    import json

    import naima_amd as na
    from naima_amd import analysis as A
    from naima_amd import datatable as DT
    gold = os.path.join(ROOT, "tests", "golden")
    schema = json.load(open(os.path.join(gold, "save_run_schema.json")))
    assert schema["reference_read_run_reads_naima_amd_file"] is True
    z = np.load(os.path.join(gold, "run_inputs.npz"))

    class Run:
        pass

    m = Run()
    m.get_chain, m.get_log_prob = (lambda **k: z["chain"]), (lambda **k: z["log_prob"])
    m.get_blobs = lambda **k: [z["blob0"], z["blob1"]]
    m.blob_units = [na.u.Unit(str(s)) for s in z["blob_units"]]
    m.data = DT.validate_data_table(DT.read(os.path.join(gold, "data", "CrabNebula_HESS_ipac.dat")))
    m.labels = [str(s) for s in z["labels"]]
    m.run_info = json.loads(str(z["run_info"]))
    m.acceptance_fraction = z["acceptance"]
    attrs, ds = A.run_layout(m)
    assert set(ds) == set(schema["objects"])
    for name, want in schema["objects"].items():
        arr, dattrs = ds[name]
        assert list(arr.shape) == want["shape"], name
        if isinstance(want["dtype"], list):  # the data table: field names, order and types
            have = [[n, str(arr.dtype[n])] for n in arr.dtype.names]
            assert have == want["dtype"], (name, have)
        elif name.endswith("__table_column_meta__"):
            assert [x.decode() for x in arr] == want["lines"]
        else:
            assert str(arr.dtype) == want["dtype"], name
        assert dattrs == want["attrs"], (name, dattrs, want["attrs"])
    assert set(attrs) == set(schema["group_attrs"])
    for k, v in schema["group_attrs"].items():
        assert np.all(np.asarray(attrs[k]) == np.asarray(v)), k
    # ... and the .npz container round-trips the same objects
    f = A.save_run(str(tmp_path / "run"), m)
    r = A.read_run(f)
    assert_allclose(r.get_chain(), z["chain"])
    assert_allclose(r.get_blobs()[0], z["blob0"])
    assert r.blob_units[1] == na.u.erg and r.labels == m.labels
    assert_allclose(r.data["flux"].value, m.data["flux"].value)
    assert r.data["flux"].unit == m.data["flux"].unit and r.run_info["n_run"] == 3


def test_shared_ensemble_history_merge_on_two_ranks():
    """DeviceLoop._merge_shared_block (host arithmetic only; the launches that produce its input
    are covered on the GPU by test_gpu_loops.py::test_shared_ensemble_two_ranks_one_gpu): rows a
    shared ensemble's launches wrote hold, on each rank, only the walkers that rank moved (flags
    -1 | 0 | 1); rows of launches per half-step are whole everywhere.  Two ranks (threads, an
    in-memory control plane) each end with the full chain, blob rows of rejected moves filled
    from the row before (the blobs of where the walker IS, as emcee keeps them)."""
    import threading
    import types

    from naima_amd.device_sampler import DeviceLoop

    n, N, ndim, ms = 7, 10, 3, (4, 1)
    rng = np.random.default_rng(5)
    shared_rows = [(2, 5), (5, 7)]
    owner = rng.integers(0, 2, size=(n, N))
    accepted = rng.random((n, N)) < 0.5
    truth_c, truth_l = rng.normal(size=(n, N, ndim)), rng.normal(size=(n, N))
    cur0 = [rng.normal(size=(N, m)) for m in ms]
    prop = [rng.normal(size=(n, N, m)) for m in ms]  # the blobs of every step's proposal
    truth_b = []
    for b, c0 in zip(prop, cur0):
        full = np.empty_like(b)
        for t in range(n):
            prev = full[t - 1] if t else c0
            full[t] = np.where(accepted[t][:, None], b[t], prev)
        truth_b.append(full)
    shared = np.zeros(n, dtype=bool)
    for a, b in shared_rows:
        shared[a:b] = True

    class Group:
        def __init__(self):
            self.bar, self.slots = threading.Barrier(2), [None, None]

        def allgather_bytes(self, rank, payload):
            self.slots[rank] = bytes(payload)
            self.bar.wait()
            out = list(self.slots)
            self.bar.wait()
            return out

    group, results, errors = Group(), {}, []

    def run(rank):
        try:
            mine = shared[:, None] & (owner == rank)
            whole = ~shared[:, None] & np.ones((n, N), dtype=bool)
            c = np.where((mine | whole)[:, :, None], truth_c, np.nan)
            l = np.where(mine | whole, truth_l, np.nan)
            per = [np.where(whole[:, :, None], tb, np.where((mine & accepted)[:, :, None], b, np.nan))
                   for tb, b in zip(truth_b, prop)]
            own = np.where(mine, accepted.astype(np.int32), -1).astype(np.int32)
            own[~shared] = rng.integers(-1, 2, size=(int((~shared).sum()), N))  # (never written: anything)
            g = types.SimpleNamespace(allgather_bytes=lambda p, r=rank: group.allgather_bytes(r, p))
            fake = types.SimpleNamespace(
                s=types.SimpleNamespace(comm=types.SimpleNamespace(group=g, rank=rank, size=2)),
                N=N, ndim=ndim)
            block = dict(own=types.SimpleNamespace(get=lambda: own.copy()), shared_rows=shared_rows,
                         cur0=[x.copy() for x in cur0])
            DeviceLoop._merge_shared_block(fake, block, n, c, l, per)
            results[rank] = (c, l, per)
        except Exception as e:  # pragma: no cover
            errors.append(e)
            group.bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert not errors, errors
    for rank in (0, 1):
        c, l, per = results[rank]
        assert np.array_equal(c, truth_c) and np.array_equal(l, truth_l)
        for have, want in zip(per, truth_b):
            assert np.array_equal(have, want)


def test_control_plane_client_does_not_take_itself_for_the_hub():
    """A rank that looks for the hub before the hub listens can be handed the very port it
    connects to as its own source port on loopback (TCP simultaneous open): the socket is then
    connected to ITSELF and echoes the hello.  Round 6 found one run in ~10 of two self-started
    ranks dying of it ("control-plane message of 17592186044417 bytes": the rank's own four bytes
    + half of its own header).  The handshake must refuse such a socket -- by the addresses, and
    because the hub's answer is not a prefix of the hello."""
    import socket
    import struct

    from naima_amd import dist
    g = dist.SocketGroup(1, 1)  # (size 1: no connection is made; the handshake is driven by hand)
    g.size = 2
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    try:
        s.connect(s.getsockname())  # the self-connection, made deterministically
    except OSError:
        s.close()
        pytest.skip("this kernel refuses a TCP self-connection")
    try:
        assert s.getsockname() == s.getpeername()
        assert g._client_handshake(s) is False
        # and without the address check: the echo of the hello is not the hub's answer
        s.settimeout(5.0)
        s.sendall(dist._MAGIC + g.token + struct.pack("<i", 1))
        echo = dist._recv_exact(s, len(dist._MAGIC) + 16)
        assert echo == dist._MAGIC + g.token and echo != dist._MAGIC + g.reply
    finally:
        s.close()


LADDER_WORKER = r'''
import os, sys, warnings
sys.path.insert(0, %(root)r)
from naima_amd import dist
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    comm = dist.from_env("rccl")
assert type(comm).__name__ == "HostComm", type(comm).__name__
rungs = [(r["rung"], r["taken"]) for r in comm.ladder]
assert rungs == [("RCCL all-gather", False), ("host-staged all-gather", True)], comm.ladder
assert "probe" in comm.ladder[0]["why"], comm.ladder
assert any("RCCL communicator unavailable" in str(x.message) for x in w)
import numpy as np
g = comm.allgather(np.full((1, 2), float(comm.rank)))
assert g.shape == (2, 2) and g[1, 0] == 1.0
open(os.path.join(%(out)r, "ok_%%d" %% comm.rank), "w").write(comm.ladder[0]["why"])
'''


def test_two_ranks_without_rccl_take_the_host_staged_rung_and_say_why(tmp_path):
    """world_size 2 on a box where RCCL cannot work (here: no GPU at all): the throw-away probe
    processes fail, EVERY rank learns it over the control plane, both take the host-staged
    all-gather and the communicator says which rung was refused and why -- nobody enters
    ncclCommInitRank, nobody hangs (the reference's Pool has no such failure mode: core.py:446-448)"""
    script = tmp_path / "worker.py"
    script.write_text(LADDER_WORKER % {"root": ROOT, "out": str(tmp_path)})
    port = 31500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r),
                   LOCAL_RANK=str(r), WORLD_SIZE="2", NAIMA_AMD_RCCL_PROBE_TIMEOUT="120")
        env.pop("NAIMA_AMD_COMM", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=300) == 0
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()
