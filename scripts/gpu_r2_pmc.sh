# counters of the half-step kernel: fabric-side traffic, L2 hit/miss, SQ busy/wait (separate passes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=${1:-cfg3}
mkdir -p gpurun_out/pmc2
CMD="python bench.py --workload $W --steps 40 --warmup 10 --no-cpu --no-blobs-run --ball 0.005 --min-time 0.01"
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc2 -o $name -- $CMD > /dev/null 2> gpurun_out/pmc2/err_$name.log
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run sqa SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run sqb SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
python - <<'PY'
import csv, collections, glob, json
res = {}
for f in sorted(glob.glob('gpurun_out/pmc2/*_counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in rows:
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    for k in agg:
        for c in agg[k]:
            res.setdefault(k, {})[c] = agg[k][c] / n[k][c]
for k, v in res.items():
    if 'half_step' in k or 'integrate' in k or 'synchrotron' in k or 'step_front' in k:
        print(k[:50], json.dumps({c: round(x, 1) for c, x in v.items()}))
json.dump(res, open('gpurun_out/pmc2/counters_%s.json' % "W", 'w'), indent=1)
PY
