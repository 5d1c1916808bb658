# round-2 evidence for profiles/: kernel-trace stats of the default bench command, TCC traffic
# (separate passes), SQ counters, the per-phase stamps, bench lines of every workload, shard table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r2_gputest.log 2>&1
tail -4 gpurun_out/r2_gputest.log
O=gpurun_out/r2prof
rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run"
bash scripts/gpu_r2_stats.sh
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- $CMD > /dev/null 2> $O/err_fetch.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O -o write -- $CMD > /dev/null 2> $O/err_write.log
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O -o tcc -- $CMD > /dev/null 2> $O/err_tcc.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o sqa -- $CMD > /dev/null 2> $O/err_sqa.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $O -o sqb -- $CMD > /dev/null 2> $O/err_sqb.log
python - <<'PY'
import csv, collections, glob, json
O = "gpurun_out/r2prof"
res = {}
for f in sorted(glob.glob(O + '/*_counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in rows:
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    for k in agg:
        for c in agg[k]:
            res.setdefault(k, {})[c] = agg[k][c] / n[k][c]
            res[k]["launches_" + c] = n[k][c]
json.dump(res, open(O + '/counters_per_launch.json', 'w'), indent=1)
for k, v in res.items():
    if 'half_step' in k:
        print(k, json.dumps({c: round(x, 1) for c, x in v.items() if not c.startswith("launches")}))
PY
# cfg4 (Crab Syn+SSC): per-kernel totals of its three-launch loop + the SSC seed kernel
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o cfg4stats -- python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run > $O/cfg4_bench_under_rocprof.json 2> $O/err_cfg4stats.log
head -8 $O/cfg4stats_kernel_stats.csv | cut -c1-200
head -12 $O/stats_kernel_stats.csv | cut -c1-200
cut -c1-600 $O/bench_under_rocprof.json
# un-profiled bench lines
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1_default.json 2> $O/err_bench.log
for w in cfg1 cfg2 cfg5; do timeout 600 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu > $O/bench_$w.json 2>> $O/err_bench.log; done
timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run > $O/bench_cfg4.json 2>> $O/err_bench.log
timeout 600 python bench.py --workload cfg3 --walkers 256 --steps 200 --warmup 20 --no-cpu > $O/bench_cfg3_w256_split.json 2>> $O/err_bench.log
NH_HS_SPLIT=1 timeout 600 python bench.py --workload cfg3 --walkers 256 --steps 200 --warmup 20 --no-cpu > $O/bench_cfg3_w256_nosplit.json 2>> $O/err_bench.log
NH_HS_SPLIT=1 timeout 600 python bench.py --workload cfg2 --steps 200 --warmup 20 --no-cpu > $O/bench_cfg2_nosplit.json 2>> $O/err_bench.log
timeout 600 python bench.py --workload cfg5 --scaling strong --walkers-total 2048 --steps 100 --warmup 10 --no-cpu --no-blobs-run > $O/bench_cfg5_strong2048_n1.json 2>> $O/err_bench.log
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu --ball 0.005 > $O/bench_cfg3_ball0005.json 2>> $O/err_bench.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2prof/bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], round(d['value']), d['ms_per_step'], d.get('value_without_blobs', d.get('value_store_blobs')), d['kernels_us_per_launch'])
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 300 python scripts/hs_stamps.py cfg3 512 0.1 400 20260929 > $O/stamps_cfg3.txt 2>&1
timeout 300 python scripts/hs_stamps.py cfg5 256 0.1 400 20260929 > $O/stamps_cfg5.txt 2>&1
timeout 1200 python scripts/shard_table.py > $O/shard_table.json 2> $O/shard_table.err
tail -3 $O/err_bench.log
