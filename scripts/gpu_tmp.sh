cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; rm -rf $O; mkdir -p $O
(time timeout 1800 python -m pytest tests/test_gpu_loops.py -m gpu -q -s -k "resident or split") > $O/tests.log 2>&1; grep -E "resident ==|passed|failed|FAILED|Error" $O/tests.log | cut -c1-250
for w in "cfg2" "cfg3 --walkers 256" "cfg1"; do
 for r in 0 1; do
  NAIMA_AMD_RESIDENT=$r timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$w resident=$r', d['config']['walkers_total'], round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline'].get('us_per_half_step'))
"
 done
done
