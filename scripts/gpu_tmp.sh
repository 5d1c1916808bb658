cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r; rm -rf $O; mkdir -p $O
(time timeout 1800 python -m pytest tests/test_gpu_general.py tests/test_gpu_models.py tests/test_gpu_parity.py -m gpu -q) > $O/tests.log 2>&1; tail -40 $O/tests.log | cut -c1-250
