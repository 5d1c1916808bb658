"""gpurun_out/r2prof (scripts/gpu_r2_profile.sh on the GPU box) -> profiles/r02_* (tracked)"""
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, D = os.path.join(R, "gpurun_out", "r2prof"), os.path.join(R, "profiles")
COPY = {
    "bench_n1_default.json": "r02_bench_n1_default.json",
    "bench_under_rocprof.json": "r02_cfg3_bench_under_rocprof.json",
    "stats_kernel_stats.csv": "r02_cfg3_kernel_stats.csv",
    "stats_command.txt": "r02_cfg3_kernel_stats_command.txt",
    "counters_per_launch.json": "r02_cfg3_counters_per_launch.json",
    "stamps_cfg3.txt": "r02_cfg3_half_step_phase_stamps.txt",
    "stamps_cfg5.txt": "r02_cfg5_half_step_phase_stamps.txt",
    "shard_table.json": "r02_shard_table.json",
    "cfg4stats_kernel_stats.csv": "r02_cfg4_kernel_stats.csv",
    "cfg4_bench_under_rocprof.json": "r02_cfg4_bench_under_rocprof.json",
    "bench_cfg1.json": "r02_bench_cfg1.json", "bench_cfg2.json": "r02_bench_cfg2.json",
    "bench_cfg2_nosplit.json": "r02_bench_cfg2_nosplit.json",
    "bench_cfg4.json": "r02_bench_cfg4.json", "bench_cfg5.json": "r02_bench_cfg5.json",
    "bench_cfg3_ball0005.json": "r02_bench_cfg3_ball0005.json",
    "bench_cfg3_w256_split.json": "r02_bench_cfg3_w256_split.json",
    "bench_cfg3_w256_nosplit.json": "r02_bench_cfg3_w256_nosplit.json",
    "bench_cfg5_strong2048_n1.json": "r02_bench_cfg5_strong2048_n1.json",
}
for a, b in COPY.items():
    if os.path.exists(os.path.join(S, a)) and os.path.getsize(os.path.join(S, a)) > 0:
        shutil.copyfile(os.path.join(S, a), os.path.join(D, b))
    else:
        print("missing:", a)
# TCC traffic of the half-step kernel in the layout bench.py reads (KB per launch)
c = json.load(open(os.path.join(S, "counters_per_launch.json")))
old = json.load(open(os.path.join(D, "r02_cfg3_hbm_counters.json")))
out = {"_note": old["_note"], "fetch": {}, "write": {}, "tcc_hit": {}, "tcc_miss": {}}
for k, v in c.items():
    for src, dst in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"), ("TCC_HIT_sum", "tcc_hit"),
                     ("TCC_MISS_sum", "tcc_miss")):
        if src in v:
            out[dst][k] = v[src]
json.dump(out, open(os.path.join(D, "r02_cfg3_hbm_counters.json"), "w"), indent=1)
print("k_half_step fetch KB", out["fetch"].get("k_half_step"), "write KB", out["write"].get("k_half_step"))
