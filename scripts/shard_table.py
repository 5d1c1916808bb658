"""Single-GPU projection of the multi-GPU targets (BASELINE: cfg5 2048 walkers over 8 GPUs,
cfg4 1024 over 4; cfg3 for the headline): ms per ensemble step of the one-GPU device loop (the
resident loop wherever it applies) on ensembles of 2048 ... 128 walkers, next to the SHARDED
per-launch loop (NAIMA_AMD_FORCE_SHARDED=1, a one-rank RCCL communicator: the fallback a rank
runs when the shared resident loop is not available) on the same ensemble.
A rank of an R-GPU run over W walkers that share one ensemble (nh_half_step_run_create_shared)
runs the workgroups of a one-GPU resident loop over W / R walkers -- same geometry, same
dependency chain per half-step, records stored into R rings instead of one -- so
t_one(W) / t_one(W / R) bounds the R-GPU speed-up of the shared loop from above (xGMI latency
on the hand-off and the 6 % measured for two ranks sharing ONE GPU come off it), and
t_one(W) / t_sharded(W / R) that of the per-launch fallback.  """
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
    st = s.run_mcmc(pos, 168 if name != "cfg4" else 12, store=False)
    ctx.sync()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        st = s.run_mcmc(st, steps, store=False)
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    mode = ("resident" if s._dev.resident_launches else "fused") if not s._dev.sharded else \
        ("in-graph" if s._dev.coll_in_graph else "split")
    del s
    return float(np.median(ts)) / steps * 1e3, mode


out = {"note": __doc__.strip(), "device": ctx.info()["name"], "rows": []}
for name, steps in (("cfg5", 400), ("cfg3", 400), ("cfg1", 400), ("cfg2", 200), ("cfg4", 24)):
    for half in ((512, 256, 128, 64) if name == "cfg4" else (1024, 512, 256, 128, 64)):
        nw = 2 * half
        ms_f, mode_f = measure(name, nw, local, steps, False)
        ms_s, mode = measure(name, nw, rccl, steps, True)
        out["rows"].append({"workload": name, "walkers_per_half_step": half, "ensemble": nw,
                            "ms_per_step_fused_single_gpu": round(ms_f, 5), "one_gpu_mode": mode_f,
                            "ms_per_step_sharded_one_rank": round(ms_s, 5),
                            "sharded_mode": mode,
                            "us_per_half_step_sharded": round(ms_s * 500, 2)})
        print(out["rows"][-1], file=sys.stderr, flush=True)
rows = {(r["workload"], r["walkers_per_half_step"]): r for r in out["rows"]}
proj = []
for name, total in (("cfg5", 2048), ("cfg3", 2048), ("cfg3", 512), ("cfg2", 2048), ("cfg1", 2048),
                    ("cfg4", 1024)):  # (BASELINE: cfg4's 1024 walkers over 4 GPUs)
    one = rows.get((name, total // 2))
    if one is None:
        continue
    for R in (2, 4, 8):
        sh = rows.get((name, total // 2 // R))
        if sh is None:
            continue
        proj.append({"workload": name, "walkers_total": total, "gpus": R,
                     "speedup_upper_bound_shared_resident_loop":
                         round(one["ms_per_step_fused_single_gpu"] / sh["ms_per_step_fused_single_gpu"], 2),
                     "speedup_upper_bound_per_launch_fallback":
                         round(one["ms_per_step_fused_single_gpu"] / sh["ms_per_step_sharded_one_rank"], 2)})
out["projection"] = proj
print(json.dumps(out, indent=1))
