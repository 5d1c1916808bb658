cd $GRAFT_REPO_ROOT
O=gpurun_out/ssc; rm -rf $O; mkdir -p $O
(time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "ssc") > $O/tests.log 2>&1; grep -E "tabulated vs|passed|failed|Error|error" $O/tests.log | cut -c1-250
(timeout 900 python -m pytest tests/test_gpu_loops.py -m gpu -q -k "cfg4") > $O/loops.log 2>&1; tail -3 $O/loops.log | cut -c1-250
timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-330 $O/bench_cfg4.json
