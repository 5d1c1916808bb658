cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_loops.py -m gpu -q -x -k "not cfg4" > gpurun_out/r2/tests_l.log 2>&1; echo rc=$?
tail -3 gpurun_out/r2/tests_l.log
timeout 300 python scripts/hs_stamps.py cfg5 256 2>&1 | tail -20 | head -2
for w in cfg3 cfg5; do
timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu --ball 0.005 --no-blobs-run 2> gpurun_out/r2/bench_$w.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['config']['workload'][:5], round(d['value']/1e6,3), d['ms_per_step'], d['kernels_us_per_launch'])
"
done
echo "--- driver-like default run (ball 0.1)"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2/bench_n1_driverlike.json 2> gpurun_out/r2/bench_n1.err; echo rc=$?; tail -3 gpurun_out/r2/bench_n1.err
python -c "
import json
d=json.load(open('gpurun_out/r2/bench_n1_driverlike.json'))
print(d['value'], d['ms_per_step'], d.get('value_without_blobs', d.get('value_without_blobs', d.get('value_store_blobs'))), d['timing'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['fp64_valu']['kernels'], d['acceptance_fraction'])"
timeout 300 python scripts/ballprobe.py cfg3 2>&1 | grep "ball 0.100" | tail -2
