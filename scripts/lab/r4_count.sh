# instruction counts of the resident kernel by kind of work item (rocprofv3 --pmc, cfg3)
# (needs a library built with: NH_OUT=naima_amd/variants/lab.so naima_amd/csrc/build.sh -DNH_LAB; NAIMA_AMD_LIB points at it)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4count; rm -rf $O; mkdir -p $O
run() { name=$1; shift
  ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $O -o $name -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run --min-time 0.1 --reject-nan > $O/${name}_bench.json 2> $O/err_$name.log )
}
run s2 NH_RUN_SYN2=1
run s1 NH_RUN_SYN2=0
run s2_nosyn NH_RUN_SYN2=1 NH_RUN_DEBUG_SKIP=1
run s2_notab NH_RUN_SYN2=1 NH_RUN_DEBUG_SKIP=2
run s2_none NH_RUN_SYN2=1 NH_RUN_DEBUG_SKIP=3
run s1_notab NH_RUN_SYN2=0 NH_RUN_DEBUG_SKIP=2
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/*_counter_collection.csv")):
    agg = collections.defaultdict(float); n = collections.Counter(); dur = []
    for r in csv.DictReader(open(f)):
        if 'half_step_run' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
            if r['Counter_Name'] == 'SQ_WAVES': dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    if not n: continue
    m = {k: agg[k] / n[k] for k in agg}
    w = m['SQ_WAVES']
    print(f.split('/')[-1][:14].ljust(14), 'launches', n['SQ_WAVES'], 'us', round(sum(dur) / len(dur) / 1e3, 1), 'VALU/wave/halfstep', round(m['SQ_INSTS_VALU'] / w / 40, 1),
          'SALU', round(m['SQ_INSTS_SALU'] / w / 40, 1), 'LDS', round(m['SQ_INSTS_LDS'] / w / 40, 1), 'valu_busy', round(4 * m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES'], 3))
PY
