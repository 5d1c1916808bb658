# instruction-cache / scalar-cache / latency counters of the half-step kernel: W=cfg5 bash scripts/gpu_r2_sq2.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=${W:-cfg5}
O=gpurun_out/sq2_$W
rm -rf $O; mkdir -p $O
CMD="python bench.py --workload $W --walkers 512 --steps 20 --warmup 5 --no-cpu --no-blobs-run --ball 0.005"
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ --output-format csv -d $O -o a -- $CMD > /dev/null 2> $O/err_a.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES --output-format csv -d $O -o b -- $CMD > /dev/null 2> $O/err_b.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --output-format csv -d $O -o c -- $CMD > /dev/null 2> $O/err_c.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM --output-format csv -d $O -o d -- $CMD > /dev/null 2> $O/err_d.log
python - <<PY
import csv, collections, glob, json
O = "$O"
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
json.dump(res, open(O + '/counters_per_launch.json', 'w'), indent=1)
for k, v in res.items():
    if 'half_step' in k:
        print(k, json.dumps({c: round(x, 1) for c, x in v.items()}))
PY
for f in a b c d; do grep -i "error\|invalid\|fail" $O/err_$f.log | head -2; done
