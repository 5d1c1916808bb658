cd $GRAFT_REPO_ROOT
python scripts/kbench.py 256 2>&1 | tail -8
for c in 4 8 16; do echo -n "SYN C=$c "; NH_SYN_C=$c python scripts/kbench.py 256 2>&1 | grep -E "synchrotron"; done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
