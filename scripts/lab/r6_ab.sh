# A/B of library variants on ONE box: VARS="a.so b.so" (under naima_amd/variants), alternating, R rounds
# (round 6: walkers per workload in WL="cfg3:512 cfg3:2048"; EXTRA = further bench.py flags)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ab_${TAG:-x}; mkdir -p $O
for r in $(seq 1 ${R:-3}); do
  for w in ${WL:-cfg3:512}; do
    for v in $VARS; do
      n=${w%%:*}; k=${w##*:}
      NAIMA_AMD_LIB=naima_amd/variants/$v timeout 300 python bench.py --workload $n --walkers $k --steps ${STEPS:-20} --warmup 5 --no-cpu --no-blobs-run --min-time 0.4 $EXTRA > $O/b.json 2> $O/b.err
      python -c "import json; d=json.load(open('$O/b.json')); print('$v $n $k', round(d['value']), 'us/half-step', round(d['roofline']['us_per_half_step'],2), 'ovh', round(d['region_overhead_us'],1))" || tail -3 $O/b.err
    done
  done
done
