cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ladder.py tests/test_gpu_bench.py -m gpu -q > $O/t.log 2>&1; tail -40 $O/t.log | cut -c1-400
