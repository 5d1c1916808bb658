# SQ counters of the half-step kernel for one workload: W=cfg5 bash scripts/gpu_r2_sq.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=${W:-cfg5}
O=gpurun_out/sq_$W
rm -rf $O; mkdir -p $O
CMD="python bench.py --workload $W --walkers 512 --steps 20 --warmup 5 --no-cpu --no-blobs-run --ball 0.005"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o sqa -- $CMD > /dev/null 2> $O/err_sqa.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $O -o sqb -- $CMD > /dev/null 2> $O/err_sqb.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH --output-format csv -d $O -o sqc -- $CMD > /dev/null 2> $O/err_sqc.log
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
tail -2 $O/err_sqc.log
