cd $GRAFT_REPO_ROOT
python scripts/kbench.py 256 2>&1 | grep "integrate(IC\|synchrotron"
timeout 900 python -m pytest tests -m gpu -x -q > /tmp/t.log 2>&1; echo rc=$?; grep -E "passed|failed|rror|assert" /tmp/t.log | tail -8
for i in 1 2; do timeout 600 python bench.py --no-cpu 2>&1 | tail -1 | cut -c60-110; done
