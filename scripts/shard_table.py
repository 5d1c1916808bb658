"""Single-GPU projection of the multi-GPU targets (BASELINE: cfg5 2048 walkers over 8 GPUs,
cfg4 1024 over 4; cfg3 for the headline): ms per ensemble step of the SHARDED device loop
(NAIMA_AMD_FORCE_SHARDED=1, a one-rank RCCL communicator: split graphs or in-graph
all-gather, exactly the code a rank of an R-GPU job runs) for local shards of
1024 / 512 / 256 / 128 walkers per half-step, next to the fused single-GPU loop on the same
ensemble.  Implied upper bound on the R-GPU speed-up of a W-walker ensemble:
t_fused(W) / t_sharded(W / R) -- an upper bound because a one-rank all-gather costs less
than one over xGMI.  Writes one JSON document to stdout."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29733")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd import _lib, dist  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

ctx = _lib.get_context()
os.environ["NAIMA_AMD_FORCE_SHARDED"] = "1"
rccl = dist.from_env("rccl")
assert type(rccl).__name__ == "RcclComm"
local = dist.LocalComm()


def measure(name, nwalkers, comm, steps, sharded):
    os.environ["NAIMA_AMD_FORCE_SHARDED"] = "1" if sharded else "0"
    model, p0, raw, data, prior, labels = build_problem(name, na)
    s = EnsembleSampler(nwalkers, p0.size, na.lnprob, args=[data, model, prior], seed=20260929,
                        comm=comm, naima_style=True, store_blobs=False, device=True)
    pos = p0 + 0.1 * p0 * s._rng.normal(size=(nwalkers, p0.size))
    st = s.run_mcmc(pos, 168, store=False)
    ctx.sync()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        st = s.run_mcmc(st, steps, store=False)
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    mode = "fused" if not s._dev.sharded else ("in-graph" if s._dev.coll_in_graph else "split")
    del s
    return float(np.median(ts)) / steps * 1e3, mode


out = {"note": __doc__.split("Writes")[0].strip(), "device": ctx.info()["name"], "rows": []}
for name, steps in (("cfg5", 400), ("cfg3", 400), ("cfg1", 400), ("cfg2", 200)):
    for half in (1024, 512, 256, 128, 64):
        nw = 2 * half
        ms_f, _ = measure(name, nw, local, steps, False)
        ms_s, mode = measure(name, nw, rccl, steps, True)
        out["rows"].append({"workload": name, "walkers_per_half_step": half, "ensemble": nw,
                            "ms_per_step_fused_single_gpu": round(ms_f, 5),
                            "ms_per_step_sharded_one_rank": round(ms_s, 5),
                            "sharded_mode": mode,
                            "us_per_half_step_sharded": round(ms_s * 500, 2)})
        print(out["rows"][-1], file=sys.stderr, flush=True)
rows = {(r["workload"], r["walkers_per_half_step"]): r for r in out["rows"]}
proj = []
for name, total in (("cfg5", 2048), ("cfg3", 2048), ("cfg3", 512), ("cfg2", 2048), ("cfg1", 2048)):
    one = rows.get((name, total // 2))
    if one is None:
        continue
    for R in (2, 4, 8):
        sh = rows.get((name, total // 2 // R))
        if sh is None:
            continue
        proj.append({"workload": name, "walkers_total": total, "gpus": R,
                     "speedup_upper_bound": round(one["ms_per_step_fused_single_gpu"]
                                                  / sh["ms_per_step_sharded_one_rank"], 2)})
out["projection"] = proj
print(json.dumps(out, indent=1))
