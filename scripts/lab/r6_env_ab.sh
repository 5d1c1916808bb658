# A/B of an environment knob on one library: KNOB=NH_RUN_PIPELINE VALS="0 1" WL="cfg2:1024"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6env; mkdir -p $O
for r in $(seq 1 ${R:-3}); do for w in $WL; do for v in $VALS; do
  n=${w%%:*}; k=${w##*:}
  env $KNOB=$v NAIMA_AMD_LIB=${LIB:-naima_amd/libnaima_hip.so} timeout 300 python bench.py --workload $n --walkers $k --steps 20 --warmup 5 --no-cpu --no-blobs-run --min-time 0.4 > $O/b.json 2> $O/b.err
  python -c "import json; d=json.load(open('$O/b.json')); print('$KNOB=$v $n $k', round(d['value']), 'us/half-step', round(d['roofline']['us_per_half_step'],2))" || tail -3 $O/b.err
done; done; done
