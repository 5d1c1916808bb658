# quick loop: the loop / parity tests that cover the half-step kernel + headline and cfg2 bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5q
(timeout 900 python -m pytest tests/test_gpu_loops.py tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -x -q) > gpurun_out/s5q/test.log 2>&1; tail -4 gpurun_out/s5q/test.log
for w in cfg5 cfg1 cfg5 cfg1; do
timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu --no-blobs-run --min-time 0.3 > gpurun_out/s5q/$w.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/s5q/$w.json')); print('$w', round(d['value']), d['kernels_us_per_launch'])"
done
