cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > /tmp/t.log 2>&1; echo rc=$?; grep -E "passed|failed|rror|assert|^E " /tmp/t.log | tail -25
