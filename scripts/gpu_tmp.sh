cd $GRAFT_REPO_ROOT
O=gpurun_out/t12; rm -rf $O; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_general.py tests/test_gpu_models.py tests/test_gpu_random.py tests/test_gpu_parity.py -m gpu -q -x) > $O/tests.log 2>&1; tail -25 $O/tests.log | cut -c1-300
