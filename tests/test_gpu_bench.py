"""bench.py as the driver runs it: the ONE JSON line and its contract keys, and `--gpus N` as a
complete run by itself -- bench.py starts its own N ranks when no launcher has (one process per
GPU: the reference's ``Pool(threads)`` of core.py:446-457 with GPUs for workers), here rehearsed
with both ranks pinned to the one GPU of the test box.

NO ASSERTION IN THIS FILE DEPENDS ON A CLOCK (round 5's driver run died on one: a +-5 us band on a
figure that swings by +-100 us).  What is checked is the line's contract -- keys, arithmetic
identities, which path was taken -- and a rate is only ever compared with zero.  A rate that looks
low is a `warnings.warn`, never a failure; conftest.py collects this file LAST."""
import warnings
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv, env=None, timeout=900):
    e = {k: v for k, v in os.environ.items()
         if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NAIMA_AMD_DEVICE")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    return p


def _line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py owes the driver exactly ONE line on stdout: %r" % lines
    return json.loads(lines[0])


def test_one_gpu_line_has_the_contract_keys():
    d = _line(_bench(["--steps", "20", "--warmup", "5", "--min-time", "0.15", "--cpu-seconds", "1.5",
                      "--no-blobs-run"]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline", "value_evaluated", "region_overhead_us"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("cfg3") and d["config"]["walkers_total"] == 512
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0.0 < r["frac"] < 1.0 and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "k_half_step_run" in r["kernel"]
    for k, v in d["fp64_valu"]["kernels"].items():
        assert 0.0 < v["frac"] <= 1.0, (k, v)  # (an op-count convention, but never above the peak)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert 0.0 < d["value_evaluated"] <= d["value"]
    # what a region spends outside its kernels: device time stamps of the timed launches
    # themselves against the host clock around them -- cannot be negative by construction (no
    # band: it is a clock)
    assert d["region_overhead_us"] >= 0.0 and d["region_us"] > 0.0
    assert d["value"] > 0.0
    assert abs(d["value"] - 512 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks appear (here both on
    device 0, NAIMA_AMD_DEVICE), share cfg3's ensemble through each other's rings, and rank 0
    prints the one line with n_gpus == 2"""
    p = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu", "--min-time", "0.3",
                "--no-blobs-run"],
               env={"NAIMA_AMD_DEVICE": "0", "NH_RUN_SPIN_LIMIT": str(1 << 24)})
    d = _line(p)
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2
    assert d["config"]["walkers_total"] == 1024 and d["config"]["walkers_per_gpu"] == 512
    assert d["config"]["ranks_started_by"].startswith("bench.py itself")
    x = d["config"]["exchange"]
    assert x["ranks"] == 2 and x["path"] in ("shared_resident_loop", "host-staged all-gather"), x
    assert x["path"] == "shared_resident_loop", x  # (two processes on one GPU map each other's rings)
    # weak scaling rehearsed on ONE GPU: two ranks of 512 walkers each share its 256 CUs, so the
    # whole job runs at about the one-GPU rate of a 1024-walker ensemble (r05: 13.2 M walker-steps/s);
    # which PATH ran is asserted above -- the rate itself is a clock and only warns
    assert d["value"] > 0.0
    if d["value"] < 6.0e6:
        warnings.warn("two self-started ranks on one GPU: %.3g walker-steps/s (expected ~1.3e7)" % d["value"])


def test_bench_shares_a_fixed_ensemble_between_its_own_ranks():
    """strong scaling over self-started ranks: cfg3's 512 walkers SHARED by two ranks on the one
    GPU -- the configuration of profiles/r05_bench_cfg3_shared_two_ranks_one_gpu.json (12.6 M
    walker-steps/s there).  The path is asserted; the rate only warns"""
    p = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu", "--min-time", "0.3",
                "--no-blobs-run", "--scaling", "strong", "--walkers-total", "512"],
               env={"NAIMA_AMD_DEVICE": "0", "NH_RUN_SPIN_LIMIT": str(1 << 24)})
    d = _line(p)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["walkers_total"] == 512
    assert d["config"]["exchange"]["path"] == "shared_resident_loop", d["config"]["exchange"]
    assert d["value"] > 0.0
    if d["value"] < 6.0e6:
        warnings.warn("512 walkers shared by two ranks on one GPU: %.3g walker-steps/s" % d["value"])


def test_bench_refuses_more_ranks_than_devices():
    """--gpus N on a node with fewer than N visible devices is an error, not a silent one-rank
    run (unless NAIMA_AMD_DEVICE says the ranks are meant to share one)"""
    from naima_amd import _lib
    n = _lib.device_count()
    p = _bench(["--gpus", str(n + 1), "--steps", "2", "--warmup", "2", "--no-cpu"], timeout=120)
    assert p.returncode != 0 and "visible" in (p.stderr + p.stdout)


def test_cfg4_line_never_claims_more_than_the_peak():
    """cfg4 (the staged plan: two k_half_step launches around the SSC seed integral): every
    `frac` of the line is a fraction -- round 5's line credited the SSC kernel with 1.006 of the
    FP64 peak (an op-count convention that counted a five-instruction reciprocal as 20) -- and the
    span clock covers the whole half-step (the first launch opens it, the last closes it)"""
    d = _line(_bench(["--workload", "cfg4", "--walkers", "64", "--steps", "4", "--warmup", "2",
                      "--no-cpu", "--min-time", "0.05", "--no-blobs-run"]))
    assert d["config"]["workload"].startswith("cfg4")
    assert 0.0 < d["roofline"]["frac"] < 1.0
    for k, v in d["fp64_valu"]["kernels"].items():
        assert 0.0 < v["frac"] <= 1.0, (k, v)
    assert d["region_overhead_us"] >= 0.0
    assert d["region_overhead"]["device_spans_per_region"] == 8  # (one per half-step)
