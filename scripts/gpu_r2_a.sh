cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 2400 python -m pytest tests/test_gpu_loops.py tests/test_gpu_models.py tests/test_gpu_properties.py tests/test_gpu_parity.py -m gpu -q -x --durations=15 > gpurun_out/r2/tests_a.log 2>&1; echo rc=$?
tail -40 gpurun_out/r2/tests_a.log
