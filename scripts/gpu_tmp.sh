cd $GRAFT_REPO_ROOT
O=gpurun_out/t14; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_loops.py -m gpu -q -x -k "shared_ensemble") > $O/tests.log 2>&1; tail -30 $O/tests.log | cut -c1-300
