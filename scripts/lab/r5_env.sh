# A/B of environment knobs on ONE box: ENVS="A=1 B=2|C=3|" (settings separated by |, empty = default), R rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5env_${TAG:-x}; mkdir -p $O
IFS='|' read -ra SETS <<< "$ENVS"
for r in $(seq 1 ${R:-2}); do
  for s in "${SETS[@]}" ""; do
    for w in ${WL:-cfg3:512}; do
      n=${w%%:*}; k=${w##*:}
      env $s timeout 300 python bench.py --workload $n --walkers $k --steps ${STEPS:-20} --warmup 5 --no-cpu --no-blobs-run --min-time 0.4 > $O/b.json 2> $O/b.err
      python -c "import json; d=json.load(open('$O/b.json')); print('[$s] $n $k', round(d['value']), 'us/half-step', round(d['roofline']['us_per_half_step'],2), 'ovh', round(d['region_overhead_us'],1))" || tail -3 $O/b.err
    done
  done
done
