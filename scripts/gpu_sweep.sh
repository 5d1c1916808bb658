cd $GRAFT_REPO_ROOT
for i in 1 2; do echo "chain   $(timeout 600 python bench.py --no-cpu 2>&1 | tail -1 | cut -c60-100)"; echo "nochain $(timeout 600 python bench.py --no-cpu --no-chain 2>&1 | tail -1 | cut -c60-100)"; done
timeout 600 python bench.py --no-cpu --steps 50 --warmup 5 2>&1 | tail -1 | cut -c1-1200
