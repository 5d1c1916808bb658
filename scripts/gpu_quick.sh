cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 320 --warmup 32 --no-cpu 2>&1 | tail -1
