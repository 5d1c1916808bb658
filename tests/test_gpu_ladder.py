"""The exchange ladder of a multi-rank run, one rung at a time, with two ranks bench.py starts
itself on the ONE GPU of the test box (NAIMA_AMD_DEVICE=0):

    records stored into each other's rings by the resident loop   (shared_resident_loop)
      -> refused ->  one launch + one RCCL all-gather per half-step (RCCL all-gather)
      -> unavailable -> the same with the all-gather staged through the host

RCCL refuses two ranks of one device, so on this box its rung can only be seen REFUSED -- by the
throw-away probe processes that build the communicator first (dist.run_rccl_probe), which is the
mechanism that keeps a hanging ncclCommInitRank out of the run.  The line says which rung was
taken and why the ones before it were not (config.exchange.ladder).  And the launcher's watchdog:
a rank that never reaches communicator creation ends the run with the rank named, not with the
driver's 1 800 s limit.  The reference's parallel entry: core.py:446-457, 533-536 (Pool(threads)).

No assertion here depends on a clock (a watchdog that does not fire runs into the subprocess timeout)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv, env, timeout=900):
    e = {k: v for k, v in os.environ.items()
         if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NAIMA_AMD_DEVICE", "NAIMA_AMD_COMM")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=e,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def _line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


ARGS = ["--gpus", "2", "--steps", "10", "--warmup", "5", "--no-cpu", "--min-time", "0.05",
        "--no-blobs-run"]


def test_first_rung_rings():
    d = _line(_bench(ARGS, {"NAIMA_AMD_DEVICE": "0", "NH_RUN_SPIN_LIMIT": str(1 << 24)}))
    x = d["config"]["exchange"]
    assert x["path"] == "shared_resident_loop" and x["ladder"][0] == dict(
        rung="shared_resident_loop", taken=True, why="")
    assert len(x["devices"]) == 2 and x["devices"][0] == x["devices"][1]  # (pinned to one GPU)
    assert x["devices_distinct"] is False and d["config"]["devices_pinned_to"] == "0"


def test_rings_refused_then_rccl_unavailable_then_host_staged():
    """every rung in turn: the ring probe refused (fault injection), RCCL asked for and found
    unavailable by the probe processes (two ranks of one device), the host-staged all-gather
    taken -- and the line says so, rung by rung"""
    d = _line(_bench(ARGS, {"NAIMA_AMD_DEVICE": "0", "NAIMA_AMD_COMM": "rccl",
                            "NAIMA_AMD_LADDER_REFUSE": "ring", "NAIMA_AMD_RCCL_PROBE_TIMEOUT": "60",
                            "NH_RUN_SPIN_LIMIT": str(1 << 24)}))
    x = d["config"]["exchange"]
    assert x["path"] == "host-staged all-gather" and x["communicator"] == "HostComm"
    assert x["rccl_nranks"] is None
    rungs = [(r["rung"], r["taken"]) for r in x["ladder"]]
    assert rungs == [("shared_resident_loop", False), ("RCCL all-gather", False),
                     ("host-staged all-gather", True)], x["ladder"]
    assert "NAIMA_AMD_LADDER_REFUSE" in x["ladder"][0]["why"]
    assert "probe" in x["ladder"][1]["why"]  # (who found RCCL unavailable, and its complaint)
    assert d["n_gpus"] == 2 and d["value"] > 0.0


def test_rings_refused_host_staged_by_request():
    d = _line(_bench(ARGS, {"NAIMA_AMD_DEVICE": "0", "NAIMA_AMD_LADDER_REFUSE": "ring",
                            "NH_RUN_SPIN_LIMIT": str(1 << 24)}))
    x = d["config"]["exchange"]
    assert x["path"] == "host-staged all-gather"
    assert [r["taken"] for r in x["ladder"]] == [False, False, True]
    assert "NAIMA_AMD_COMM=host" in x["ladder"][1]["why"]


def test_a_rank_that_never_reaches_the_communicator_ends_the_run_by_name():
    """rank 1 stalls between its GPU context and communicator creation (what a rank stuck in
    ncclCommInitRank looks like from outside): bench.py's launcher ends every rank, says which
    ranks' time ran out and what stage every rank last reported, and exits non-zero -- in seconds,
    not at the driver's limit (the subprocess timeout is the only clock here)"""
    p = _bench(ARGS, {"NAIMA_AMD_DEVICE": "0", "NAIMA_AMD_TEST_STALL_BEFORE_COMM": "1",
                      "NAIMA_AMD_LAUNCH_COMM_TIMEOUT": "8"}, timeout=300)
    assert p.returncode == 18, (p.returncode, p.stderr[-2000:])
    # (communicator creation is a rendezvous: rank 0 waits in it for rank 1, and whose time runs out
    # first depends on who had its GPU context first -- the message names those AND every rank's
    # last reported stage; the stalled rank is the one that is not past "ctx")
    import ast
    import re
    m = re.search(r"rank\(s\) \[([0-9, ]+)\] have not passed communicator creation", p.stderr)
    assert m and len(m.group(1).split(",")) >= 1, p.stderr[-2000:]
    w = re.search(r"last stage reported by every rank: (\{[^}]*\})", p.stderr)
    assert w, p.stderr[-2000:]
    stages = ast.literal_eval(w.group(1))
    assert set(stages) == {0, 1} and stages[1] in ("ctx", "started") and "comm" not in stages.values(), stages
    assert p.stdout.strip() == ""


def test_run_budget_ends_a_run_that_overstays():
    p = _bench(ARGS, {"NAIMA_AMD_DEVICE": "0", "NAIMA_AMD_TEST_STALL_BEFORE_COMM": "0",
                      "NAIMA_AMD_BENCH_BUDGET": "6"}, timeout=300)
    assert p.returncode in (17, 19), (p.returncode, p.stderr[-2000:])
    assert "budget" in p.stderr or "the whole bench run" in p.stderr
