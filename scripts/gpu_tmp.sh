cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/icache; rm -rf $O; mkdir -p $O
for w in cfg3 cfg5; do
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O -o ${w}_sqd -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --workload $w --no-cpu --no-blobs-run --min-time 0.1 > $O/${w}_bench.json 2> $O/err_$w.log ); echo "$w exit $?"
done
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/*_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in rows:
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    for k in agg:
        if 'half_step' in k or 'epilogue' in k:
            print(f.split('/')[-1][:8], k, {c: round(agg[k][c]/n[k][c]) for c in agg[k]}, dict(n[k]))
PY
