"""gpurun_out/<tag>prof (scripts/gpu_profile.sh on the GPU box) -> profiles/<tag>_* (tracked)

    python scripts/collect_profiles.py [tag, default r04]
"""
import glob
import json
import os
import shutil
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, D = os.path.join(R, "gpurun_out", TAG + "prof"), os.path.join(R, "profiles")
COPY = {"bench_n1_default.json": "bench_n1_default.json",
        "bench_n1_per_launch_kernel.json": "bench_n1_per_launch_kernel.json",
        "bench_cfg5_per_launch_kernel.json": "bench_cfg5_per_launch_kernel.json",
        "bench_cfg2_per_launch_kernel.json": "bench_cfg2_per_launch_kernel.json",
        "bench_cfg5_strong2048_n1_per_launch_kernel.json": "bench_cfg5_strong2048_n1_per_launch_kernel.json",
        "cfg3_stats_command.txt": "cfg3_kernel_stats_command.txt",
        "stamps_cfg3.txt": "cfg3_resident_loop_phase_stamps.txt",
        "stamps_cfg5.txt": "cfg5_resident_loop_phase_stamps.txt",
        "bench_cfg3_w256.json": "bench_cfg3_w256.json",
        "bench_cfg5_strong2048_n1.json": "bench_cfg5_strong2048_n1.json"}
for f in ("bench_cfg3_w1024.json", "bench_cfg3_w2048.json", "bench_cfg3_w1024_per_launch_kernel.json",
          "bench_cfg3_w2048_per_launch_kernel.json", "bench_cfg3_shared_two_ranks_one_gpu.json",
          "bench_cfg3_shared_two_ranks_one_gpu_steps100.json", "bench_cfg3_one_process_steps100.json",
          "shard_table.json", "bench_cfg3_weak_two_ranks_one_gpu.json",
          "bench_cfg3_weak_four_ranks_one_gpu_baseline_split.json", "bench_cfg4_strong1024_n1.json",
          "region_host_latency.txt"):
    COPY[f] = f
for w in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"):
    COPY["%s_stats_kernel_stats.csv" % w] = "%s_kernel_stats.csv" % w
    COPY["%s_stats_bench.json" % w] = "%s_bench_under_rocprof.json" % w
    if w != "cfg3":
        COPY["bench_%s.json" % w] = "bench_%s.json" % w
for w in ("cfg3", "cfg5", "cfg2", "cfg4"):
    COPY["%s_counters_per_launch.json" % w] = "%s_counters_per_launch.json" % w
COPY["cfg3w2048_counters_per_launch.json"] = "cfg3_w2048_counters_per_launch.json"
def head_commit():
    import subprocess
    try:
        return subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True,
                              text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


HEAD = head_commit()  # (the tree the GPU box ran: the profile run is made on a committed tree)
for f in ("bench_cfg3_w4096.json", "bench_cfg3_w1024_serial_turns.json", "bench_cfg3_w2048_serial_turns.json",
          "bench_cfg3_w4096_serial_turns.json", "bench_ladder_rings_refused_rccl_unavailable_host_staged.json",
          "stamps_cfg3_w2048.txt", "per_launch_kernel_phase_stamps_cfg3.txt"):
    COPY[f] = f
for a, b in COPY.items():
    src = os.path.join(S, a)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(D, "%s_%s" % (TAG, b)))
    else:
        print("missing:", a)
# fabric-side traffic of the half-step kernels in the layout bench.py reads (KB per launch)
for w in ("cfg3", "cfg5"):
    f = os.path.join(S, "%s_counters_per_launch.json" % w)
    if not os.path.exists(f):
        continue
    c = json.load(open(f))
    out = {"_note": "rocprofv3 --kernel-trace --pmc, one counter group per pass (scripts/gpu_profile.sh); "
                    "FETCH_SIZE / WRITE_SIZE in KB per launch as reported: bench.py doubles FETCH_SIZE "
                    "(gfx950 reports half the bytes of 16-byte-per-lane reads, MI355X_MICROARCH.md)",
           "fetch": {}, "write": {}, "tcc_hit": {}, "tcc_miss": {}, "launches": {}}
    for k, v in c.items():
        for src, dst in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"), ("TCC_HIT_sum", "tcc_hit"),
                         ("TCC_MISS_sum", "tcc_miss")):
            if src in v:
                out[dst][k] = v[src]
                out["launches"][k] = v.get("launches_" + src)
    json.dump(out, open(os.path.join(D, "%s_%s_hbm_counters.json" % (TAG, w)), "w"), indent=1)
    print(w, {k: v for k, v in out["fetch"].items() if "half_step" in k}, {k: v for k, v in out["write"].items() if "half_step" in k})

# which tree the counter files were measured on (bench.py's valu_utilisation.source_commit: the GPU box
# runs a snapshot without .git, so the commit is recorded here, at collection)
for f in glob.glob(os.path.join(D, "%s_*_counters_per_launch.json" % TAG)):
    d = json.load(open(f))
    d["_source_commit"] = HEAD
    json.dump(d, open(f, "w"), indent=1)
