# round 5 kernel experiments: phase stamps + short bench lines (+ the loop parity tests with T=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5exp_${TAG:-x}; mkdir -p $O
for w in ${WL:-cfg3:512}; do
  n=${w%%:*}; k=${w##*:}
  NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py $n $k > $O/stamps_${n}_$k.txt 2>&1
  head -22 $O/stamps_${n}_$k.txt
  timeout 300 python bench.py --workload $n --walkers $k --steps 20 --warmup 5 --no-cpu --no-blobs-run --min-time 0.4 > $O/bench_${n}_$k.json 2> $O/bench_${n}_$k.err
  python -c "import json; d=json.load(open('$O/bench_${n}_$k.json')); print('$n $k', round(d['value']), 'us/half-step', round(d['roofline']['us_per_half_step'],2), 'overhead', round(d['region_overhead_us'],1))" || tail -5 $O/bench_${n}_$k.err
done
if [ "${T:-0}" = "1" ]; then
  (timeout 1500 python -m pytest tests/test_gpu_loops.py -x -q -m gpu -k "${K:-benchmarks_size or resident_loop_equals or gives_up}") > $O/tests.log 2>&1; tail -4 $O/tests.log
fi
