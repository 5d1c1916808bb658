cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc3
CMD="python scripts/hs_stamps.py cfg3 512"
for v in "" "-DHS_SKIP_TAB" "-DHS_SKIP_SYN" "-DHS_SKIP_TAB -DHS_SKIP_SYN"; do
  bash naima_amd/csrc/build.sh $v > /dev/null 2>&1
  tag=$(echo "x$v" | tr -d ' -' )
  NH_HS_DEBUG=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d gpurun_out/pmc3 -o $tag -- $CMD > /dev/null 2> gpurun_out/pmc3/err_$tag.log
  python - "$tag" <<'PY'
import csv, collections, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open('gpurun_out/pmc3/%s_counter_collection.csv' % tag)))
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    if 'k_half_step' in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print(tag, {c: round(agg[c] / n[c] / 4096, 1) for c in agg}, "(per wave)")
PY
done
bash naima_amd/csrc/build.sh > /dev/null 2>&1
