cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc -o $name -- python scripts/kbench.py 256 > /dev/null 2> gpurun_out/pmc/err_$name.log
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run b SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC
run c SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc/[a-c]_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.defaultdict(set)
    for r in rows:
        k=r['Kernel_Name'][:34]
        if 'integrate_tables' not in k and 'synchrotron' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); nd[k].add(r['Dispatch_Id'])
    for k,v in agg.items():
        n=len(nd[k]); print(f[-28:-23], k, {c: round(x/n) for c,x in v.items()})
PY
