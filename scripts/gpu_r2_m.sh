cd $GRAFT_REPO_ROOT
timeout 300 python scripts/nanprobe.py 2>&1 | tail -4
timeout 300 python scripts/hs_stamps.py cfg3 512 0.005 40 2>&1 | tail -3
timeout 300 python scripts/hs_stamps.py cfg3 512 0.1 3000 2>&1 | tail -3
