cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() { # name counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc -o $name -- python scripts/kbench.py 256 > /dev/null 2> gpurun_out/pmc/err_$name.log
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run b SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run e GRBM_GUI_ACTIVE GRBM_COUNT
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc/[a-e]_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.defaultdict(set)
    for r in rows:
        k=r['Kernel_Name'][:34]
        if 'integrate_tables' not in k and 'synchrotron' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); nd[k].add(r['Dispatch_Id'])
    for k,v in agg.items():
        n=len(nd[k]); print(f[-28:-23], k, {c: round(x/n) for c,x in v.items()})
PY
tail -3 gpurun_out/pmc/err_c.log
