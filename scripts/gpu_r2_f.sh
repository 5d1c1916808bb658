cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests/test_gpu_loops.py tests/test_gpu_properties.py -m gpu -q -x > gpurun_out/r2/tests_f.log 2>&1; echo rc=$?
tail -15 gpurun_out/r2/tests_f.log
for a in "cfg3 512" "cfg5 256" "cfg1 32" "cfg2 256"; do timeout 300 python scripts/hs_stamps.py $a 2>&1 | tail -3; done
for w in cfg3 cfg5 cfg1 cfg2; do
timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu --ball 0.005 2> gpurun_out/r2/bench_$w.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['config']['workload'][:5], d['value'], d['ms_per_step'], d.get('value_without_blobs', d.get('value_without_blobs', d.get('value_store_blobs'))), d['kernels_us_per_launch'])
"
tail -3 gpurun_out/r2/bench_$w.err
done
