cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1200 python -m pytest tests/test_gpu_general.py tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -q -x > gpurun_out/r2/tests_b.log 2>&1; echo rc=$?
tail -5 gpurun_out/r2/tests_b.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2/bench_n1_driverlike.json 2> gpurun_out/r2/bench_n1.err; echo rc=$?; tail -3 gpurun_out/r2/bench_n1.err
cut -c1-1500 gpurun_out/r2/bench_n1_driverlike.json
timeout 900 python scripts/shard_table.py > gpurun_out/r2/shard_table.json 2> gpurun_out/r2/shard_table.err; echo rc=$?
tail -30 gpurun_out/r2/shard_table.err
