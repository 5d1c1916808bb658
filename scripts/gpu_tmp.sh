cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; rm -rf $O; mkdir -p $O
(time timeout 1800 python -m pytest tests/test_gpu_loops.py -m gpu -q) > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-250
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')
"
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3 200-step regions', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')
"
