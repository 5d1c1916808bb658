cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests/test_gpu_loops.py tests/test_gpu_properties.py tests/test_gpu_parity.py -m gpu -q -x -k "not cfg4" > gpurun_out/r2/tests_i.log 2>&1; echo rc=$?
tail -5 gpurun_out/r2/tests_i.log
timeout 300 python scripts/hs_stamps.py cfg3 512 2>&1 | tail -20 | head -4
for w in cfg3 cfg5 cfg1 cfg2; do
timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu --ball 0.005 --no-blobs-run 2> gpurun_out/r2/bench_$w.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['config']['workload'][:5], round(d['value']/1e6,3), d['ms_per_step'], d['kernels_us_per_launch'])
"
tail -3 gpurun_out/r2/bench_$w.err
done
