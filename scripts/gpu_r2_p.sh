cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_loops.py -m gpu -q -x -k "sharded or two_ranks or loop" > gpurun_out/r2/tests_p.log 2>&1; echo rc=$?
tail -15 gpurun_out/r2/tests_p.log
NAIMA_AMD_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29755 RANK=0 WORLD_SIZE=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu 2> gpurun_out/r2/bench_sharded.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('sharded 1-rank:', round(d['value']/1e6,3), d['ms_per_step'], d.get('value_without_blobs'), d['config']['collective'], d['kernels_us_per_launch'])"
tail -3 gpurun_out/r2/bench_sharded.err
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('single:', round(d['value']/1e6,3), d['ms_per_step'], d.get('value_without_blobs'), d['blobs'])"
