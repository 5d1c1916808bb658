# round 6: the profile entries that depend on the two-walkers-in-flight instance, refreshed on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06deep; rm -rf $O; mkdir -p $O
for n in 1024 2048 4096; do
  timeout 600 python bench.py --workload cfg3 --walkers $n --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/bench_cfg3_w$n.json 2>> $O/err.log
  NH_RUN_PIPELINE=0 timeout 600 python bench.py --workload cfg3 --walkers $n --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/bench_cfg3_w${n}_serial_turns.json 2>> $O/err.log
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/bench_cfg3_w512_same_box.json 2>> $O/err.log
NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py cfg3 2048 > $O/stamps_cfg3_w2048.txt 2>&1
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o cfg3w2048_sqa -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --workload cfg3 --walkers 2048 --no-cpu --no-blobs-run --min-time 0.1 > $O/cfg3w2048_sqa_bench.json 2> $O/err_sqa.log )
python - <<PY
import csv, collections, glob, json
O = "$O"
res = {}
for f in sorted(glob.glob(O + '/cfg3w2048_*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    for k in agg:
        for c in agg[k]:
            res.setdefault(k, {})[c] = agg[k][c] / n[k][c]
            res[k]["launches_" + c] = n[k][c]
json.dump(res, open(O + '/cfg3w2048_counters_per_launch.json', 'w'), indent=1)
for k, v in res.items():
    if 'half_step_run' in k: print(k, 'busy', 4 * v['SQ_ACTIVE_INST_VALU'] / v['SQ_WAVE_CYCLES'])
for f in sorted(glob.glob(O + '/bench_*.json')):
    d = json.load(open(f)); print(f.split('/')[-1], round(d['value']), round(d['value_evaluated']), round(d['region_overhead_us'], 1))
PY
timeout 2400 python scripts/shard_table.py > $O/shard_table.json 2> $O/shard_table.err
head -12 $O/stamps_cfg3_w2048.txt
