# gpurun_out/r2prof -> profiles/r02_* (the tracked copies)
O=gpurun_out/r2prof; P=profiles
cp $O/bench_n1_default.json $P/r02_bench_n1_default.json
cp $O/stats_kernel_stats.csv $P/r02_cfg3_kernel_stats.csv
cp $O/bench_under_rocprof.json $P/r02_cfg3_bench_under_rocprof.json
cp $O/counters_per_launch.json $P/r02_cfg3_counters_per_launch.json
cp $O/stamps_cfg3.txt $P/r02_cfg3_half_step_phase_stamps.txt
cp $O/stamps_cfg5.txt $P/r02_cfg5_half_step_phase_stamps.txt
for w in 1 2 4 5; do cp $O/bench_cfg$w.json $P/r02_bench_cfg$w.json; done
cp $O/bench_cfg3_ball0005.json $P/r02_bench_cfg3_ball0005.json
cp $O/bench_cfg5_strong2048_n1.json $P/r02_bench_cfg5_strong2048_n1.json
cp $O/shard_table.json $P/r02_shard_table.json
python - <<'PY'
import json
c = json.load(open('gpurun_out/r2prof/counters_per_launch.json'))
old = json.load(open('profiles/r02_cfg3_hbm_counters.json'))
out = {"_note": old["_note"], "fetch": {}, "write": {}, "tcc_hit": {}, "tcc_miss": {}}
for k, v in c.items():
    if "FETCH_SIZE" in v: out["fetch"][k] = v["FETCH_SIZE"]
    if "WRITE_SIZE" in v: out["write"][k] = v["WRITE_SIZE"]
    if "TCC_HIT_sum" in v: out["tcc_hit"][k] = v["TCC_HIT_sum"]
    if "TCC_MISS_sum" in v: out["tcc_miss"][k] = v["TCC_MISS_sum"]
json.dump(out, open('profiles/r02_cfg3_hbm_counters.json', 'w'), indent=1)
h = c["k_half_step"]
print({k: round(v, 1) for k, v in h.items() if not k.startswith("launches")})
PY
