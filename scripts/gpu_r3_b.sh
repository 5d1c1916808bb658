# round 3, second call: parity tests again; the profiler workaround with the driver's exact
# command; the ensemble-age effect; workgroup size of the table-only instance
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3b
rm -rf $O; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_gpu_loops.py -m gpu -q -s -k "benchmarks or rejected") > $O/tests.log 2>&1
grep -E "proposals|passed|failed|Error|error" $O/tests.log | cut -c1-300 | tail -20
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o drv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$O/drv.json 2> $GRAFT_REPO_ROOT/$O/drv.err ); echo "driver command under rocprofv3: exit $?"
head -3 $O/drv_kernel_stats.csv | cut -c1-200
timeout 900 python scripts/ensemble_age.py cfg2 256 8 400 > $O/age_cfg2.log 2>&1; cat $O/age_cfg2.log | cut -c1-250
timeout 900 python scripts/ensemble_age.py cfg3 512 6 400 > $O/age_cfg3.log 2>&1; cat $O/age_cfg3.log | cut -c1-250
for t in 1024 512 256 128; do
  for w in "--walkers 256" "--scaling strong --walkers-total 2048"; do
    NH_HS_THREADS=$t timeout 300 python bench.py --workload cfg5 $w --steps 100 --warmup 10 --no-cpu --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg5 threads $t', d['config']['walkers_total'], round(d['value']/1e6,3), 'M/s', d['ms_per_step'], d['kernels_us_per_launch'])
"
  done
done
