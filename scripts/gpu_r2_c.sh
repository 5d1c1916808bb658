cd $GRAFT_REPO_ROOT
timeout 600 python scripts/ballprobe.py cfg3 2>&1 | tail -40
