#!/usr/bin/env python
"""bench.py -- walker-steps/s of the ensemble log-probability on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3] [--walkers 512]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    ... bench.py --gpus 8 --scaling strong --workload cfg5 --walkers-total 2048   (BASELINE's
        multi-GPU configurations: cfg5 2048 walkers over 8 GPUs, cfg4 1024 over 4)

One "step" = one ensemble step of the stretch-move sampler = two half-steps =
``walkers`` evaluations of naima's lnprob (model + likelihood) for ``cfg3``, the
configuration BASELINE.json quotes the metric on: RXJ1713 Syn+IC joint fit, 5
parameters, CMB+FIR+NIR seed fields, 64 photon energies, 512 walkers per GPU
(weak scaling, the default: every rank adds its own 512 walkers to the ensemble;
``--scaling strong`` keeps the ensemble fixed and splits it).  Synthetic spectrum (seed
20260929), inputs resident in HBM before the timed region.  The timed region is EXACTLY
K steps between barriers; it is repeated until >= 0.5 s have been timed and ``value`` is
the median region (min / max / first in ``timing``).

Prints ONE JSON line on rank 0 with the contract keys plus
  "roofline"      the dominant kernel, from HIP events recorded around every launch
                  inside the timed region (algorithmic bytes = SURVEY.md 8d figure);
  "fp64_valu"     the same kernel against the FP64 vector peak (the bound that
                  actually applies: the path is transcendental-bound, not HBM-bound);
  "cpu_baseline"  the NumPy oracle (a port of the reference's array-at-a-time
                  algorithm) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# compulsory HBM bytes and FP64 flop-equivalents per walker-step
ALGO = {
    "cfg1": dict(bytes=8 * (3 + 28 + 1), flop_eq=1.0e6),
    "cfg2": dict(bytes=8 * (4 + 179 + 1), flop_eq=2.7e6),
    "cfg3": dict(bytes=8 * (5 + 64 + 2), flop_eq=6.4e6),
    "cfg4": dict(bytes=8 * (6 + 261 + 1), flop_eq=1.0e9),
    "cfg5": dict(bytes=8 * (5 + 28 + 2), flop_eq=0.5e6),
}
# flop-equivalents per launch-walker of the two hot kernels (op-count convention of
# a node of Synchrotron 50 eq., a segment of a table reduction 30 eq.)
# (round 6: the SSC segment's reciprocal is credited with the FIVE instructions it is on this chip
# -- v_rcp_f64 seed + Newton -- not with the convention's 20: at 30 eq. per segment the kernel's
# `frac` came out at 1.006 of the FP64 peak, which is a statement about the convention; 5 + 10 = 15)
SSC_SEG_EQ = 19.0 if os.environ.get("NAIMA_AMD_SSC_TABLE", "1") == "0" else 15.0
KERNEL_FLOP_EQ = {
    "cfg3": {"synchrotron": 64 * 570 * 50.0, "integrate_tables": 64 * 3 * 370 * 30.0},
    "cfg2": {"synchrotron": 179 * 300 * 50.0},
    "cfg1": {"integrate_tables": 28 * 570 * 30.0},
    "cfg5": {"integrate_tables": 28 * 600 * 30.0},
    # (seed energy, photon energy, gamma) segments of the SSC seed integral, 15 eq. each: per
    # walker a reciprocal (5: see SSC_SEG_EQ) + mul, add, sub, cmp, three fma (10).  (The Aharonian-Atoyan
    # kernel and its logarithm come from the table built once per sampler; evaluated per step
    # -- NAIMA_AMD_SSC_TABLE=0 -- a walker's 1/16 share of them is another 4.)
    "cfg4": {"ic_seed_walkers": 100 * 261 * 869 * SSC_SEG_EQ},
}
# profiler category -> kernel symbol in the rocprofv3 kernel trace
# ("half_step" is k_half_step_run when the loop is resident)
KERNEL_SYMBOL = {"half_step": "k_half_step", "integrate_tables": "k_integrate_tables",
                 "synchrotron": "k_synchrotron",
                 "particle_weights": "k_step_front (proposal+packs+weights+We)", "lnprob": "k_lnprobmodel",
                 "integrate_rows": "k_integrate_rows", "ic_seed_walkers": "k_ic_seed_walkers",
                 "tables": "k_table_*", "glue": "k_pack_rows/k_move_*"}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VEC_PEAK_TF = 78.6    # vendor figure quoted in SURVEY.md 8d (256 CU x 128 flop/clk x 2.4 GHz)


def _profiles_by_round(pattern):
    """committed profiles matching ``pattern``, oldest round first (r10 after r9: by the parsed
    round number, not by the name's spelling)"""
    import glob
    import re

    def key(f):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        return (int(m.group(1)) if m else -1, os.path.basename(f))
    return sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=key)


def _last_commit_of(path):
    """short hash of the commit that last touched ``path`` (None outside a git checkout: the GPU
    box runs a snapshot without .git)"""
    import subprocess
    try:
        out = subprocess.run(["git", "-C", ROOT, "log", "-n", "1", "--format=%h", "--", path],
                             capture_output=True, text=True, timeout=10)
        return out.stdout.strip() or None
    except Exception:
        return None


def measured_utilisation(name, resident, waves_per_simd=4):
    """What the vector pipe did, from the committed SQ counter passes of this workload's bench
    command under rocprofv3 (profiles/r*_<name>_counters_per_launch.json, newest round): the
    fraction of the dominant kernel's time its SIMDs spent issuing vector instructions
    (SQ_ACTIVE_INST_VALU x 4 waves per SIMD / SQ_WAVE_CYCLES for the 1024-thread workgroups; this
    is the utilisation figure of an issue-bound kernel -- `fp64_valu` beside it is SURVEY.md 8d's
    op-count convention) and instructions per wave and launch.  None without a profile."""
    files = _profiles_by_round("r*_%s_counters_per_launch.json" % name)
    if not files:
        return None
    want = "half_step_run" if resident else ("ic_seed_walkers" if name == "cfg4" else "half_step")
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None
    best = None
    recorded = d.pop("_source_commit", None)  # (written by scripts/collect_profiles.py at collection)
    for k, v in d.items():
        if want in k and "SQ_ACTIVE_INST_VALU" in v and "SQ_WAVE_CYCLES" in v:
            if best is None or v.get("SQ_WAVE_CYCLES", 0) > best[1].get("SQ_WAVE_CYCLES", 0):
                best = (k, v)
    if best is None:
        return None
    k, v = best
    waves = max(v.get("SQ_WAVES", 1.0), 1.0)
    out = {"kernel": k.split("(")[0].strip(), "source": os.path.basename(files[-1]),
           "from_committed_profile": True,
           # (the tree the counters were measured on: recorded in the file at collection; the commit
           # that last touched the file where there is a .git and no record)
           "source_commit": recorded or _last_commit_of(files[-1]),
           "measured_in_this_run": False,
           "valu_busy": float(waves_per_simd) * v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"],
           "waves_per_simd": waves_per_simd,
           "valu_insts_per_wave_and_launch": v.get("SQ_INSTS_VALU", 0.0) / waves,
           "salu_insts_per_wave_and_launch": v.get("SQ_INSTS_SALU", 0.0) / waves,
           "wait_any_frac": v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"],
           "note": "valu_busy = SQ_ACTIVE_INST_VALU x W / SQ_WAVE_CYCLES, W = the workgroup's waves per "
                   "SIMD (1024 threads: 4, 512: 2 -- one workgroup per CU); a resident launch of the "
                   "profiled command is 40 half-steps"}
    return out


def measured_traffic(name, symbol):
    """HBM-side bytes per launch of ``symbol`` from the committed TCC counter run
    (profiles/*_hbm_counters*.json: FETCH_SIZE and WRITE_SIZE collected in separate
    rocprofv3 --pmc passes, KB per launch; see profiles/README.md), with the gfx950
    correction of MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide coalesced
    reads (16 B per lane: the kernel's table stream) and is doubled; WRITE_SIZE is
    uncalibrated and taken as reported.  None if absent."""
    files = _profiles_by_round("r*_%s_hbm_counters*.json" % name)
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    same = lambda k: symbol in k and ("_run" in k) == ("_run" in symbol)  # noqa: E731
    f = [v for k, v in d.get("fetch", {}).items() if same(k)]
    w = [v for k, v in d.get("write", {}).items() if same(k)]
    if not f:
        return None, None
    return (2.0 * max(f) + (max(w) if w else 0.0)) * 1024.0, os.path.basename(files[-1])


def build_problem(name, na):
    from naima_amd import workloads as W
    from naima_amd.datatable import make_data
    wl = W.WORKLOADS[name]
    model = wl["model"](na)
    p0 = np.asarray(wl["p0"], dtype=float)

    def flux_at_p0(E_eV):
        out = model(p0, {"energy": E_eV * na.u.eV})
        out = out[0] if isinstance(out, tuple) else out
        return out.to("1/(s cm2 eV)").value

    raw = W.build_data(name, flux_at_p0)
    return model, p0, raw, make_data(raw), W.prior_for(name, na), wl["labels"]


def executed_flop_eq(name, raw, coords):
    """flop-equivalents per walker that the hot kernels actually EXECUTE (mean over the
    ensemble ``coords``): the table reductions visit every segment; the synchrotron kernel
    only the (energy, gamma) nodes with x = E/Ec(gamma, B) <= 746 (nh_synchrotron.hip: the
    liveness search), which depends on each walker's B"""
    out = dict(KERNEL_FLOP_EQ.get(name, {}))
    from naima_amd import constants as K
    if name == "cfg4":
        # the SSC seed integral: a segment of the seed axis is integrated only where the
        # Aharonian-Atoyan kernel is non-zero at one of its ends (radiative.py:628-636: the
        # windows q <= 1, q >= 1/(4 gamma^2)); elsewhere the kernel evaluates fic and moves on
        l0, l1 = np.log10(1e8 / K.MEC2_EV), np.log10(5e16 / K.MEC2_EV)
        gam = np.logspace(l0, l1, max(10, int(100 * (l1 - l0))))
        eps = np.logspace(-7, 9, 100) / K.MEC2_EV
        Eg = np.asarray(raw["energy"], dtype=float) * {"eV": 1.0, "keV": 1e3, "MeV": 1e6,
                                                       "GeV": 1e9, "TeV": 1e12}[raw["energy_unit"]] / K.MEC2_EV
        w = Eg[:, None] / gam[None, :]
        ok = (w < 1.0) & (w > 0.0)
        with np.errstate(all="ignore"):
            c1 = np.where(ok, w / (4.0 * gam[None, :] * (1.0 - w)), np.nan)
        qmin = 1.0 / (4.0 * gam[None, :] ** 2)
        live, prev = 0, None
        for e0 in eps:
            q = c1 / e0
            on = ok & (q < 1.0) & (q > qmin)  # (fic is 0 AT q = 1 and half-weighted at qmin)
            if prev is not None:
                live += int((on | prev).sum())
            prev = on
        out["ic_seed_walkers"] = live * SSC_SEG_EQ
        return out
    E_all = np.asarray(raw["energy"], dtype=float) * {"eV": 1.0, "keV": 1e3, "MeV": 1e6, "GeV": 1e9,
                                                      "TeV": 1e12}[raw["energy_unit"]]
    if name in ("cfg1", "cfg3") and os.environ.get("NAIMA_AMD_SORTED_TABLES", "1") != "0" \
            and os.environ.get("NAIMA_AMD_RESIDENT", "1") != "0":
        # the resident loop does not walk the rows of an inverse-Compton table below a column
        # tile's first non-zero one (zero for gamma <= E / mec2, radiative.py:574; columns sorted
        # by that row, 64 to a tile: Context.sorted_tables)
        lo_ic, nseed = {"cfg1": (1e9, 1), "cfg3": (1e11, 3)}[name]
        l0, l1 = np.log10(lo_ic / K.MEC2_EV), np.log10(1e9)
        gic = np.logspace(l0, l1, max(10, int(100 * (l1 - l0))))
        first = np.sort(np.repeat(np.searchsorted(gic, E_all / K.MEC2_EV, side="right"), nseed))
        rows = sum((gic.size - first[q:q + 64].min()) * first[q:q + 64].size
                   for q in range(0, first.size, 64))
        out["integrate_tables"] = float(rows) * 30.0
    if "synchrotron" not in out:
        return out
    # (Eemin, Eemax, nEed) of the workload's Synchrotron grid and the index of B [uG]
    lo, hi, per, iB = {"cfg2": (1e9, 1e15, 50, 3), "cfg3": (1e9, 1e9 * K.MEC2_EV, 100, 3)}[name]
    l0, l1 = np.log10(lo / K.MEC2_EV), np.log10(hi / K.MEC2_EV)
    gam = np.logspace(l0, l1, max(10, int(per * (l1 - l0))))
    E_eV = np.asarray(raw["energy"], dtype=float) * {"eV": 1.0, "keV": 1e3, "MeV": 1e6,
                                                     "GeV": 1e9, "TeV": 1e12}[raw["energy_unit"]]
    B = np.abs(np.asarray(coords)[:, iB]) * 1e-6
    qfac = K.ERG_PER_EV * 2.0 * K.M_E_G * K.C_CGS / (3.0 * K.E_GAUSS * K.HBAR_CGS * B)  # [N]
    x = (E_eV[None, :, None] * qfac[:, None, None]) / gam[None, None, :] ** 2           # [N][nE][nG]
    live = (x <= 746.0).sum(axis=2)               # nodes from the first live one on
    nodes = np.where(live > 0, np.minimum(live + 1, gam.size), 0).sum(axis=1)  # + its left neighbour
    out["synchrotron"] = float(nodes.mean()) * 50.0
    return out


def _cpu_worker(args):
    name, raw, pars, seconds = args
    from oracle import workloads_np as WN
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        WN.lnprob(name, pars[n % len(pars)], raw)
        n += 1
    return n, time.perf_counter() - t0


def usable_cores():
    """cores this process may actually use: affinity mask capped by the cgroup quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    """CPU model name, sockets and logical CPUs of this host (/proc/cpuinfo)"""
    model, phys, n = "unknown", set(), 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                n += 1
            elif line.startswith("physical id"):
                phys.add(line.split(":", 1)[1].strip())
    except OSError:
        pass
    return {"cpu_model": model, "sockets": max(1, len(phys)), "logical_cpus": n}


def cpu_baseline(name, raw, p0, seconds=8.0):
    """the oracle timed single-core and on a Pool over all host cores (the
    reference's own parallel mode, core.py:446-448)"""
    import multiprocessing as mp
    rng = np.random.default_rng(1)
    pars = p0 * (1 + 0.01 * rng.standard_normal((64, p0.size)))
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    n1, t1 = _cpu_worker((name, raw, pars, min(4.0, seconds)))
    cores = usable_cores()
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, [(name, raw, pars, seconds)] * cores)
        wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    return {"value": total / wall, "unit": "walker-steps/s", "cores": cores, "kind": "port",
            "single_core": n1 / t1, **cpu_model(),
            "sample": "%d lnprob evaluations of %s by the NumPy oracle over %d processes in "
                      "%.1f s (+%d on one core in %.1f s)" % (total, name, cores, wall, n1, t1)}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves -- one
    process per GPU, the reference's ``Pool(threads)`` of core.py:446-457 with GPUs for
    workers -- with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way
    ``torch.distributed.run`` sets them (naima_amd.dist.SocketGroup is the control plane: no
    torch).  Rank 0's stdout (the ONE JSON line) is passed through, every rank's stderr too.
    Fewer than N visible devices is an error unless NAIMA_AMD_DEVICE pins every rank to one
    (the rehearsal of the N-rank path on a one-GPU box: RCCL refuses two ranks of one device,
    so the control plane is the host-staged one, NAIMA_AMD_COMM=host)."""
    import signal
    import subprocess
    from naima_amd import _lib
    ndev = _lib.device_count()
    pinned = os.environ.get("NAIMA_AMD_DEVICE") or None
    if ndev < n and pinned is None:
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible.  One rank per GPU needs "
                         "%d; set NAIMA_AMD_DEVICE=<k> to rehearse all ranks on ONE device."
                         % (n, ndev, n))
    import tempfile
    port = _free_port()
    procs = []
    # first-contact hardening: every rank reports how far it has come (one word in a file of its
    # own: "ctx" = GPU context made, "comm" = communicator made, "done"); a rank that stays between
    # "ctx" and "comm" for too long is stuck in communicator creation -- ncclCommInitRank waits
    # for every rank and cannot be interrupted from inside -- and a run that exceeds its budget is
    # stuck somewhere else: every process group started here is ended, the stuck ranks are named,
    # the exit code is non-zero.  (The driver's limit for a line is 1 800 s.)
    comm_limit = float(os.environ.get("NAIMA_AMD_LAUNCH_COMM_TIMEOUT", "300"))
    budget = float(os.environ.get("NAIMA_AMD_BENCH_BUDGET", "1500"))
    pdir = tempfile.mkdtemp(prefix="naima_amd_ranks_")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NAIMA_AMD_SELF_SPAWNED="1",
                   NAIMA_AMD_PROGRESS_FILE=os.path.join(pdir, "rank%d" % r),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if pinned is not None:
            env.setdefault("NAIMA_AMD_COMM", "host")
            env.setdefault("NAIMA_AMD_CU_SHARE", str(n))  # (every rank plans for its share of the CUs)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, cwd=ROOT, start_new_session=True,
                                      stdout=None if r == 0 else subprocess.DEVNULL))

    def progress(r):
        try:
            return open(os.path.join(pdir, "rank%d" % r)).read().split()
        except OSError:
            return []

    def end_all(which):
        for pr in which:  # (exactly the process groups started above)
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, signal.SIGTERM)
                except OSError:
                    pass
        t_end = time.time() + 5.0
        for pr in which:
            while pr.poll() is None and time.time() < t_end:
                time.sleep(0.05)
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass

    rc = 0
    t_start = time.time()
    ctx_seen = {}
    try:
        alive = list(procs)
        while alive:
            for pr in list(alive):
                c = pr.poll()
                if c is None:
                    continue
                alive.remove(pr)
                if c != 0 and rc == 0:
                    rc = c
                    end_all(alive)
            now = time.time()
            stuck = []
            for r, pr in enumerate(procs):
                if pr.poll() is not None:
                    continue
                seen = progress(r)
                if "ctx" in seen and r not in ctx_seen:
                    ctx_seen[r] = now
                if "comm" not in seen and r in ctx_seen and now - ctx_seen[r] > comm_limit:
                    stuck.append(r)
            if stuck and rc == 0:
                # (communicator creation is a rendezvous: a rank waits in it for every other one, so
                # the ranks named first are the ones whose time ran out first, not necessarily the
                # one everybody is waiting for -- hence every rank's last reported stage)
                where = {r: (progress(r) or ["started"])[-1] for r in range(n)}
                sys.stderr.write("bench.py --gpus %d: rank(s) %s have not passed communicator creation "
                                 "within %.0f s of having a GPU context (RCCL probe / ncclCommInitRank); "
                                 "last stage reported by every rank: %s; ending all ranks\n"
                                 % (n, stuck, comm_limit, where))
                rc = 18
                end_all(alive)
            elif now - t_start > budget and rc == 0:
                where = {r: (progress(r) or ["started"])[-1] for r, pr in enumerate(procs)
                         if pr.poll() is None}
                sys.stderr.write("bench.py --gpus %d: the run exceeded its budget of %.0f s; ranks still "
                                 "running and the last stage each reported: %s; ending all ranks\n"
                                 % (n, budget, where))
                rc = 19
                end_all(alive)
            time.sleep(0.05)
    except KeyboardInterrupt:
        rc = 130
        end_all(procs)
    finally:
        import shutil
        shutil.rmtree(pdir, ignore_errors=True)
    raise SystemExit(rc)


def report_progress(stage):
    """one word per stage into the file the rank's launcher watches (launch_ranks)"""
    f = os.environ.get("NAIMA_AMD_PROGRESS_FILE")
    if f:
        try:
            with open(f, "a") as fh:
                fh.write(stage + "\n")
        except OSError:
            pass


def exchange_info(sampler, comm, shared_note, ctx=None):
    """which exchange path the ranks took for the one exchange of a half-step, and why: the
    ladder's rungs in the order they were tried -- records stored into each other's rings by the
    resident loop | one launch + one RCCL all-gather per half-step | the same with the all-gather
    staged through the host -- and who the ranks are (PCI bus id of each rank's GPU; the live
    communicator's own count).  Collective: every rank calls it."""
    dev = getattr(sampler, "_dev", None)
    if dev is None or comm.size == 1:
        return {"path": "none (one rank)", "ranks": comm.size}
    kind = type(comm).__name__
    out = {"ranks": comm.size, "communicator": kind, "rccl_nranks": None}
    if kind == "RcclComm":
        # (what ncclCommCount says of the communicator that exists, not Python's idea of the world)
        live = comm.live_info()
        out["rccl_nranks"], out["rccl_device"] = live["nranks"], live["device"]
    group = getattr(comm, "group", None)
    if group is not None and ctx is not None:
        ids = [p_.decode() for p_ in group.allgather_bytes(ctx.pci_bus_id().encode())]
        out["devices"] = ids
        out["devices_distinct"] = len(set(ids)) == len(ids)
    shared = bool(getattr(dev, "shared", False))
    ladder = [dict(rung="shared_resident_loop", taken=shared,
                   why="" if shared else (getattr(dev, "resident_reason", None) or shared_note or ""))]
    if not shared:
        ladder += list(getattr(comm, "ladder", None) or
                       [dict(rung="RCCL all-gather" if kind == "RcclComm" else "host-staged all-gather",
                             taken=True, why="")])
    out["ladder"] = ladder
    if shared:
        out.update(path="shared_resident_loop", why=shared_note,
                   probe_us_per_exchange=(dev.shared_info or {}).get("probe_us_per_exchange"))
    else:
        out.update(path=("RCCL all-gather" if kind == "RcclComm" else "host-staged all-gather"),
                   in_graph=bool(getattr(dev, "coll_in_graph", False)), why=shared_note)
    return out


def measure(args, ctx, comm, name, scaling, walkers, walkers_total, full):
    """one workload on the ranks of ``comm``: the timed K-step regions (``full``: plus the other
    blob setting, per-kernel HIP events, roofline, CPU baseline) -> the line's dict on rank 0,
    None elsewhere"""
    import math

    import naima_amd as na
    from naima_amd import _lib
    from naima_amd import workloads as W
    from naima_amd.sampler import EnsembleSampler
    rank = comm.rank
    model, p0, raw, data, prior, labels = build_problem(name, na)
    if scaling == "strong":
        nwalkers = walkers_total or W.WORKLOADS[name]["nwalkers"]
        if nwalkers % (2 * comm.size):
            raise SystemExit("--walkers-total %d does not split into two halves over %d GPUs"
                             % (nwalkers, comm.size))
        per_gpu = nwalkers // comm.size
    else:
        per_gpu = walkers or (256 if name in ("cfg4", "cfg5") else W.WORKLOADS[name]["nwalkers"])
        nwalkers = per_gpu * comm.size

    def make_sampler(device, graph, blobs=False):
        return EnsembleSampler(nwalkers, p0.size, na.lnprob, args=[data, model, prior],
                               seed=20260929, comm=comm, naima_style=True, store_blobs=blobs,
                               device=device, use_graph=graph,
                               nan_policy="reject" if args.reject_nan else "raise")

    device = not args.host_loop
    keep_blobs = not args.no_blobs
    shared_note = None
    sampler = make_sampler(device, not args.no_graph, blobs=keep_blobs)
    # naima's initial ensemble: a ball of 10 % of p0 around p0 (core.py:477-481)
    pos = p0 + args.ball * p0 * sampler._rng.normal(size=(nwalkers, p0.size))
    state = sampler.run_mcmc(pos, max(2, args.warmup), store=False)
    ctx.sync()
    if device and comm.size > 1:
        # several GPUs share the ensemble through records stored into each other's memory
        # (device_sampler._create_shared_run probes that before taking the path); should a
        # launch still have given up waiting for a record, every rank drops to one launch and
        # one all-gather per half-step -- agreed here, before anything is timed
        state = sampler.run_mcmc(state, 40, store=False)
        bad = 1.0 if sampler._dev.resident_status() != 0 else 0.0  # (reduced over the ranks below)
        shared_note = ("taken" if sampler._dev.shared else "not available: %s"
                       % getattr(sampler._dev, "resident_reason", "the model's launches are not "
                                 "ones the resident kernel absorbs"))
        if comm.max(bad) > 0:
            shared_note = ("a launch of the rehearsal gave up waiting for a record (workgroups not "
                           "all resident, or a rank fell behind): per-launch loop for this run")
            os.environ["NAIMA_AMD_SHARED"] = "0"
            sampler = make_sampler(device, not args.no_graph, blobs=keep_blobs)
            state = sampler.run_mcmc(pos, max(2, args.warmup), store=False)
            ctx.sync()
    # Spin-up, untimed and reported as config.untimed_spinup_steps: the first ~20 ms after
    # an idle period run measurably slower (device clocks ramp, first replays of the
    # multi-step graph).  A short --warmup is topped up to 160 steps (at most 0.5 s of them,
    # judged by eight steps timed after the warm-up -- the warm-up itself contains one-off
    # set-up); every rank takes the same number so the collectives stay matched.
    spinup = 0
    if device:
        tw = time.perf_counter()
        state = sampler.run_mcmc(state, 8, store=False)
        ctx.sync()
        per_step = comm.max((time.perf_counter() - tw) / 8)
        spinup = 8 + int(min(max(0, 152 - args.warmup), 0.5 / max(per_step, 1e-6)))
        spinup = int(comm.max(spinup))
        if spinup > 8:
            state = sampler.run_mcmc(state, spinup - 8, store=False)

    # ---- the timed region: EXACTLY K ensemble steps, barrier + device sync on both sides.
    # Repeated (every rank the same number of times) until --min-time seconds have been
    # timed: a 20-step region of cfg3 lasts 2 ms, which is noise.
    # (several ranks: the control plane's barrier is a star of Python sockets -- its ranks leave it
    # up to ~0.1 ms apart -- so ranks that share an ensemble through each other's memory line
    # up once more on the device, DeviceLoop.device_barrier; a rank's time runs from there to
    # its own device sync after step K, the closing barrier follows, and the MAX over ranks is
    # the region's time)
    # What a region spends OUTSIDE its kernels (Python ahead of the launch calls, launch latency,
    # the gaps between launches, the synchronisation behind the last one): this rank's host time
    # minus the device span clock over the SAME launches (nh_clock_read: every span of the step
    # loop -- a resident launch with its epilogue, or one half-step's launches -- is stamped on the
    # device by its first workgroup in and its last workgroup out).  The spans lie inside the host
    # interval and do not overlap, so the difference cannot be negative.  The clock is reset ahead
    # of t0 and read behind the region's own sync: nothing is added to the timed launches.
    overheads, spans_seen = [], []

    def timed_region(smp, st):
        comm.barrier()
        if device:
            smp._dev.device_barrier()
        ctx.clock_read(reset=True)  # (synchronises)
        t0 = time.perf_counter()
        st = smp.run_mcmc(st, args.steps, store=not args.no_chain)
        ctx.sync()
        dt_mine = time.perf_counter() - t0
        span_us, nspans = ctx.clock_read()
        overheads.append(dt_mine * 1e6 - span_us)
        spans_seen.append(nspans)
        comm.barrier()
        return comm.max(dt_mine), st

    # (rehearsals of the region, untimed: the graphs a K-step call replays -- the tail of a
    # block of moves is a graph of its own -- are captured here, not inside a timed region)
    # (a K-step call takes its moves from 32-step blocks: where a call starts within a block
    # repeats after 32 / gcd(32, K) calls, and with it the set of graphs)
    rehearsed = 0
    for _ in range(min(16, 32 // math.gcd(32, max(1, args.steps)))):
        _, state = timed_region(sampler, state)
        sampler.reset()
        rehearsed += args.steps
    del overheads[:], spans_seen[:]  # (the rehearsals' are not the timed regions')
    first, state = timed_region(sampler, state)
    acc_frac = float(np.mean(sampler.acceptance_fraction))
    sampler.reset()  # (the chain of a region is dropped before the next one)
    times = [first]
    min_time = args.min_time if full else min(args.min_time, 0.25)
    # every region's time is already the max over ranks: all ranks take the same decisions
    while sum(times) < min_time and len(times) < 1000:
        dt_i, state = timed_region(sampler, state)
        times.append(dt_i)
        sampler.reset()
    dt = float(np.median(times))
    region_overheads, region_spans = list(overheads), list(spans_seen)
    final_coords = np.asarray(state.coords)

    # ---- the same loop with the other blob setting (the reference always stores (flux, We)
    # per walker and step, core.py:450-457, and so does the timed loop above unless
    # --no-blobs): reported in an extra key
    blobs_value = None
    if device and full and not args.no_blobs_run:
        bs = make_sampler(device, not args.no_graph, blobs=not keep_blobs)
        bst = bs.run_mcmc(final_coords, 6, store=False)
        bt = []
        for _ in range(int(min(40, max(1, np.ceil(args.min_time / 2 / max(dt, 1e-6)))))):
            dt_i, bst = timed_region(bs, bst)
            bt.append(dt_i)
            bs.reset()
        blobs_value = nwalkers * args.steps / float(np.median(bt))
        del bs, bst

    # ---- per-kernel HIP-event timing: hipGraph replay hides the launches from
    # events, so the SAME launch sequence is run eagerly (device loop, no graph)
    # with an event pair around every launch, for the same number of steps
    # ... unless the loop IS eager launches: the resident loop (nh_half_step_run) is one launch
    # per block of moves, no graph -- then the timed sampler itself is bracketed, with regions of
    # the same K steps as the timed ones
    resident = device and getattr(sampler._dev, "resident_launches", 0) > 0
    prof_steps = max(args.steps, 50)
    prof, ev_us = {}, 0.0
    if full or comm.size == 1:
        if resident:
            ctx.sync()
            ctx.profile(True)
            ctx.profile_read(reset=True)
            pstate, prof_steps = state, 0
            while prof_steps < 50:
                pstate = sampler.run_mcmc(pstate, args.steps, store=not args.no_chain)
                sampler.reset()
                prof_steps += args.steps
            prof = ctx.profile_read(reset=True)
            ctx.profile(False)
        else:
            prof_sampler = make_sampler(device, False, blobs=keep_blobs)
            pstate = prof_sampler.run_mcmc(final_coords, 2, store=False)
            ctx.sync()
            ctx.profile(True)
            ctx.profile_read(reset=True)
            prof_sampler.run_mcmc(pstate, prof_steps, store=False)
            prof = ctx.profile_read(reset=True)
            ctx.profile(False)
        ev_us = ctx.profile_overhead_us()  # what an event pair adds to every launch
    # (collective: every rank's counts meet here)
    forbidden, nan_rej = int(sampler.prior_forbidden_proposals), int(sampler.nan_proposals)
    proposals_total = int(sampler.steps_total) * int(nwalkers)
    xinfo = exchange_info(sampler, comm, shared_note, ctx)

    if rank != 0:
        return None
    value = nwalkers * args.steps / dt
    info = ctx.info()
    frac_evaluated = 1.0 - forbidden / max(proposals_total, 1)
    out = {
        "metric": "walker-steps/sec (ensemble lnprob evals/s)",
        "value": value, "unit": "walker-steps/s", "n_gpus": comm.size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        # walker-steps whose integrals RAN: a proposal the prior forbids is a walker-step of `value`
        # (it is a proposal of the move, never accepted) but the device loop evaluates none of its
        # integrals, where the reference evaluates the model and discards it (core.py:98-119) --
        # the share is that of every step this sampler made, warm-up included
        "value_evaluated": value * frac_evaluated,
        "config": {"workload": "%s: %s" % (name, {
            "cfg1": "ECPL -> IC(CMB), 28 energies",
            "cfg2": "ECPL -> Synchrotron, 179 energies, 300-pt Ee grid",
            "cfg3": "RXJ1713 Syn+IC joint fit (CMB+FIR+NIR), 5 parameters, 64 energies",
            "cfg4": "Crab Syn+SSC, 261 energies, 869-pt Ee grid, 100 seed energies",
            "cfg5": "PionDecay ECBPL, 28 energies, 600-pt Ep grid"}[name]),
            "scaling": scaling,
            "walkers_per_gpu": per_gpu, "walkers_total": nwalkers, "ndim": int(p0.size),
            "n_energies": int(len(raw["energy"])), "sharding": "walkers/%d" % comm.size,
            "initial_ball": args.ball,
            "device": info["name"], "untimed_spinup_steps": spinup + rehearsed,
            "ranks": comm.size,
            "ranks_started_by": ("bench.py itself (one process per GPU)"
                                 if os.environ.get("NAIMA_AMD_SELF_SPAWNED") == "1" else
                                 "the launcher (WORLD_SIZE)" if comm.size > 1 else "one process"),
            "devices_pinned_to": os.environ.get("NAIMA_AMD_DEVICE"),
            "cu_share": int(os.environ.get("NAIMA_AMD_CU_SHARE", "1")),
            "exchange": xinfo,
            "collective": ("none (one rank)" if not getattr(sampler._dev, "sharded", False) else
                           "none: a mover stores its walker's record into every rank's ring "
                           "(system-scope stores over xGMI, rings mapped through hipIpc); %r"
                           % (sampler._dev.shared_info,) if getattr(sampler._dev, "shared", False) else
                           "RCCL all-gather inside the step graphs" if sampler._dev.coll_in_graph
                           else "all-gather between two graphs per half-step")
            if device else "host loop",
            "shared_resident_loop": shared_note,
            "chain": "discarded (store=False)" if args.no_chain else
            "kept: every step's coords, log-prob%s appended in HBM by the step kernels"
            % (" and blobs" if keep_blobs else "")},
        "timing": {"regions": len(times), "steps_per_region": args.steps,
                   "statistic": "median region (every region: barrier + sync, K steps, sync + "
                                "barrier, max over ranks)",
                   "timed_s_total": float(np.sum(times)),
                   "value_first_region": nwalkers * args.steps / times[0],
                   "value_min": nwalkers * args.steps / max(times),
                   "value_max": nwalkers * args.steps / min(times)},
        "blobs": ("kept: the model spectrum and We/Wp of every walker and step, in HBM"
                  if keep_blobs else "not kept (--no-blobs)"),
        "acceptance_fraction": acc_frac,
        # (emcee stops at the first NaN log-probability, and so does this run unless
        # --reject-nan: the workloads' priors -- workloads.prior_for -- keep the walkers of the
        # 10 % ball off the zero-flux plateau where 10 ** x of a wandered coordinate overflows)
        "nan_policy": "reject" if args.reject_nan else "raise",
        "nan_proposals_rejected": nan_rej,
        # (counted by the kernels over every step the timed sampler made, warm-up included)
        "proposals_forbidden_by_prior": forbidden,
        "proposals_total": proposals_total,
        "loop": ("host" if not device else
                 "device, resident workgroups: one launch of k_half_step_run per block of moves "
                 "(<= 32 steps), walkers handed over by tagged records" if resident else
                 ("device+hipGraph" if sampler._dev.graph is not None else "device")),
    }
    if blobs_value is not None or full:
        out["value_without_blobs" if keep_blobs else "value_store_blobs"] = blobs_value
    out["region_us"] = dt * 1e6
    if region_spans and min(region_spans) > 0:
        # (rank 0's own regions: its host time around the K steps minus its device spans)
        out["region_overhead_us"] = float(np.median(region_overheads))
        out["region_overhead"] = {
            "min_us": float(np.min(region_overheads)), "max_us": float(np.max(region_overheads)),
            "device_spans_per_region": int(np.median(region_spans)),
            "how": "host perf_counter around the K steps (launch calls + sync) minus the device "
                   "span clock of the same launches (nh_clock_read: wall_clock64 stamps by the first "
                   "workgroup in and the last workgroup out of every span) -- non-negative by "
                   "construction"}
    else:
        out["region_overhead_us"] = None  # (a loop whose launches carry no span stamps: host loop)
    if not prof:
        return out
    # dominant kernel by accumulated HIP-event time (one category = one kernel symbol;
    # "glue"/"tables" aggregate several small kernels and are not candidates).  The fixed
    # cost of the event pair (an empty kernel bracketed the same way, minus its ~1 us of
    # execution) is removed so that the figure is comparable with rocprofv3's durations.
    ev_fix = max(ev_us - 1.0, 0.0)
    # (under a profiler that instruments every launch the empty kernel's pair calibrates at
    # tens of microseconds -- more than the kernels' own raw figures: no correction then, the
    # raw event times are reported and the profiler's own durations are the ones to read)
    ev_note = "event pair cost removed"
    if ev_us > 15.0:
        ev_fix = 0.0
        ev_note = ("no correction: an event pair calibrated at %.1f us (launches are being "
                   "instrumented); read the profiler's durations" % ev_us)

    def launch_us(cat):
        return max(prof[cat]["ms"] * 1e3 / prof[cat]["launches"] - ev_fix, 0.1)

    dom = max((k for k in prof if k not in ("glue", "tables")),
              key=lambda k: launch_us(k) * prof[k]["launches"])
    avg_s = launch_us(dom) * 1e-6
    walkers_per_launch = per_gpu / 2.0  # one half-ensemble shard per launch
    if resident:  # one launch = every half-step of a K-step region: walker-steps per launch
        walkers_per_launch = per_gpu * prof_steps / float(prof[dom]["launches"])
    abytes = ALGO[name]["bytes"] * walkers_per_launch
    achieved = abytes / avg_s / 1e9
    dom_symbol = ("k_half_step_run" if resident and dom == "half_step"
                  else KERNEL_SYMBOL.get(dom, dom).split("/")[0].split(" ")[0])
    traffic, traffic_src = measured_traffic(name, dom_symbol)
    base_note = ("FP64-issue-bound path: the HBM fraction is << 1 % by construction (SURVEY.md 8d); "
                 "fp64_valu / valu_utilisation are the figures that apply.")
    if traffic is None:
        note = base_note + " traffic: no committed TCC counter pass for this kernel."
    elif resident:
        note = base_note + (" traffic = L2 memory-side bytes (2 x FETCH_SIZE + WRITE_SIZE) per launch "
                            "of the profiled driver command (a resident launch = every half-step "
                            "of a 20-step region: the tables stay in the L2s, what is left is the "
                            "chain history, the kept blobs and the write-through records)")
    else:
        note = base_note + (" traffic = L2 memory-side bytes (2 x FETCH_SIZE + WRITE_SIZE): every "
                            "launch starts with cold L2s, so each of the 8 XCDs pulls its own "
                            "copy of the emission table (Infinity-Cache hits)")
    out["roofline"] = {"bound": "hbm",
                       "kernel": ("k_half_step_run (nh_half_step_run: one launch per block of moves)"
                                  if resident and dom == "half_step"
                                  else KERNEL_SYMBOL.get(dom, dom)), "achieved": achieved,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "traffic": traffic, "traffic_source": traffic_src,
                       "traffic_from_committed_profile": traffic is not None,
                       "avg_launch_us": avg_s * 1e6,
                       "avg_launch_us_raw_events": prof[dom]["ms"] * 1e3 / prof[dom]["launches"],
                       "event_pair_overhead_us": ev_us, "event_correction": ev_note,
                       "algorithmic_bytes_per_launch": abytes,
                       "walkers_per_launch": walkers_per_launch,
                       "us_per_half_step": (avg_s * 1e6 / (walkers_per_launch / (per_gpu / 2.0))),
                       "note": note}
    out["kernels_us_per_launch"] = {KERNEL_SYMBOL.get(k, k): round(launch_us(k), 2) for k in prof}
    out["kernel_launches"] = {KERNEL_SYMBOL.get(k, k): v["launches"] for k, v in prof.items()}
    fp = {}
    executed = executed_flop_eq(name, raw, final_coords)
    flop_all = dict(KERNEL_FLOP_EQ.get(name, {}))
    # a category the profiler saw under its own name is credited to its own kernel (cfg4's SSC
    # seed integral next to the staged plan's two k_half_step launches); what runs INSIDE the
    # one-launch kernel -- table reductions, synchrotron nodes -- is credited to it, together
    if "half_step" in prof:
        inside = {c: v for c, v in executed.items() if c not in prof}
        if inside:
            executed = {c: v for c, v in executed.items() if c in prof}
            executed["half_step"] = sum(inside.values())
            flop_all["half_step"] = sum(flop_all[c] for c in inside)
    for cat, kflop in executed.items():
        if cat in prof:
            t = launch_us(cat) * 1e-6
            tf = kflop * walkers_per_launch / t / 1e12
            fp[KERNEL_SYMBOL[cat]] = {"achieved": tf, "frac": tf / FP64_VEC_PEAK_TF,
                                      "avg_launch_us": t * 1e6,
                                      "flop_eq_per_walker": kflop,
                                      "flop_eq_per_walker_all_nodes": flop_all[cat]}
    if fp:
        out["fp64_valu"] = {"peak": FP64_VEC_PEAK_TF, "unit": "TFLOP-eq/s", "kernels": fp,
                            # (the vendor peak is one FP64 vector instruction per 4 cycles and SIMD at
                            # 2.4 GHz; scripts/ubench measures 5.2-5.6 cycles of the nominal clock for
                            # v_add / v_mul / v_fma_f64 on this chip under load, profiles/README.md)
                            "sustained_issue_note": "scripts/ubench: v_fma_f64 5.6, v_mul_f64 5.4, "
                                                    "v_add_f64 5.2 cycles per wave at the nominal 2.4 GHz "
                                                    "-- 0.71-0.77 of the quoted peak is what the pipe "
                                                    "sustains",
                            "convention": "SURVEY.md 8d: transcendental = 20 flop-eq; "
                                          "Synchrotron node 50, table-reduction segment 30.  "
                                          "EXECUTED nodes only: the synchrotron kernel skips "
                                          "every (energy, gamma) node with E/Ec > 746 "
                                          "(exp(-x) == 0 in double) -- counted on the host "
                                          "from the final ensemble's B, mean over walkers; "
                                          "the SSC seed kernel (cfg4) is credited with the "
                                          "seed-axis segments inside the Aharonian-Atoyan "
                                          "kernel's windows, %g eq. each (the reciprocal "
                                          "credited with its five instructions on this chip, not "
                                          "the convention's 20: an op count, not pipe utilisation "
                                          "-- valu_utilisation.valu_busy is that)" % SSC_SEG_EQ}
    default_walkers = 256 if name in ("cfg4", "cfg5") else W.WORKLOADS[name]["nwalkers"]
    if per_gpu == default_walkers and comm.size == 1:  # (the configuration the profiles were taken on)
        thr = (getattr(sampler._dev, "resident_info", None) or {}).get("threads", 1024) if resident else 1024
        util = measured_utilisation(name, resident, max(1, int(thr) // 256))
        if util:
            out["valu_utilisation"] = util
    if full and not args.no_cpu and comm.size == 1:
        out["cpu_baseline"] = cpu_baseline(name, raw, p0, args.cpu_seconds)
    return out


# BASELINE.json's multi-GPU configurations: a fixed ensemble split over the GPUs
BASELINE_SPLIT = {4: ("cfg4", 1024), 8: ("cfg5", 2048)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--walkers", type=int, default=None, help="walkers per GPU")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --walkers per GPU (default); strong: --walkers-total split over the GPUs")
    ap.add_argument("--walkers-total", type=int, default=None,
                    help="strong scaling: size of the whole ensemble (default: the workload's "
                         "BASELINE figure -- cfg5 2048, cfg4 1024, cfg3 512)")
    ap.add_argument("--baseline-split", choices=["auto", "on", "off"], default="auto",
                    help="after the main line, BASELINE.json's fixed-ensemble configuration for this "
                         "number of GPUs (4: cfg4 / 1024 walkers, 8: cfg5 / 2048; strong scaling) as "
                         "the key `baseline_split` of the same JSON line.  auto: when --gpus is 4 or 8 "
                         "and the main line is the default weak cfg3")
    ap.add_argument("--ball", type=float, default=0.1,
                    help="relative spread of the initial ensemble around p0 (naima: 10 %%, core.py:477-481)")
    ap.add_argument("--min-time", type=float, default=0.5,
                    help="repeat the K-step timed region until this many seconds have been timed")
    ap.add_argument("--no-blobs", action="store_true",
                    help="the timed loop does not keep the blobs (emcee and the reference always do: "
                         "(flux, We) per walker and step, core.py:450-457)")
    ap.add_argument("--no-blobs-run", action="store_true",
                    help="skip the extra measurement with the opposite blob setting")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--reject-nan", action="store_true",
                    help="treat a NaN log-probability as a rejected proposal and count it; the "
                         "default is emcee's: ValueError on the first one")
    ap.add_argument("--no-chain", action="store_true",
                    help="do not keep the chain (emcee's store=False); default keeps it in HBM")
    ap.add_argument("--host-loop", action="store_true",
                    help="drive the step loop from the host (no device-resident ensemble)")
    ap.add_argument("--no-graph", action="store_true", help="device loop without hipGraph replay")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        # no launcher: this process becomes one (`python bench.py --gpus N` is a complete run)
        launch_ranks(args.gpus)

    from naima_amd import _lib, dist

    if world > 1:
        # (a rank of several: whatever blocks for ever ends the rank, with a message, well inside
        # the driver's 1 800 s -- its launcher then ends the others)
        dist._Watchdog("the whole bench run", float(os.environ.get("NAIMA_AMD_BENCH_BUDGET", "1500"))).__enter__()
    ctx = _lib.get_context()  # raises when libnaima_hip.so or the GPU is missing
    report_progress("ctx")
    if os.environ.get("NAIMA_AMD_TEST_STALL_BEFORE_COMM") == os.environ.get("RANK", "0"):
        time.sleep(3600)  # (tests: this rank never reaches communicator creation)
    comm = dist.from_env(os.environ.get("NAIMA_AMD_COMM", "rccl"))  # host: test hook / one-GPU rehearsal
    report_progress("comm")
    if comm.size != args.gpus:
        raise SystemExit("--gpus %d but the communicator has %d rank(s)" % (args.gpus, comm.size))
    if comm.size > 1:
        # one rank per GPU: the ranks' PCI bus ids are distinct, unless NAIMA_AMD_DEVICE pinned
        # them to one device on purpose (the one-GPU rehearsal)
        ids = [p_.decode() for p_ in comm.group.allgather_bytes(ctx.pci_bus_id().encode())]
        if len(set(ids)) != len(ids) and os.environ.get("NAIMA_AMD_DEVICE") and \
                not os.environ.get("NAIMA_AMD_CU_SHARE"):
            # the one-GPU rehearsal under a launcher that does not know about it (bench.py's own sets
            # this): every rank plans for its share of the device's CUs, or the ranks' resident
            # launches cannot all be resident together and the rehearsal falls back to the last rung
            import collections
            os.environ["NAIMA_AMD_CU_SHARE"] = str(max(collections.Counter(ids).values()))
        if len(set(ids)) != len(ids) and not os.environ.get("NAIMA_AMD_DEVICE"):
            raise SystemExit("bench.py --gpus %d: ranks share a GPU (PCI bus ids %s) although "
                             "NAIMA_AMD_DEVICE does not pin them; LOCAL_RANK / the visible devices "
                             "are not what one rank per GPU needs" % (args.gpus, ids))
    out = measure(args, ctx, comm, args.workload, args.scaling, args.walkers, args.walkers_total,
                  full=True)
    split = BASELINE_SPLIT.get(comm.size)
    want_split = split is not None and (
        args.baseline_split == "on" or
        (args.baseline_split == "auto" and args.workload == "cfg3" and args.scaling == "weak"))
    if want_split:
        # BASELINE.json's configuration for this many GPUs, strong scaling, in the same processes
        try:
            sub = measure(args, ctx, comm, split[0], "strong", None, split[1], full=False)
        except Exception as e:  # (the main line stands on its own)
            sub = {"error": "%s: %s" % (type(e).__name__, e)} if comm.rank == 0 else None
        if out is not None:
            out["baseline_split"] = sub
    report_progress("done")
    if comm.rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
