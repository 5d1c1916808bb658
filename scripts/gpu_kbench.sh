cd $GRAFT_REPO_ROOT
python scripts/kbench.py 256 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25
