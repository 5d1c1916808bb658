cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 900 python bench.py --steps 200 --warmup 20 2>&1 | tee gpurun_out/bench_n1.json | tail -2 | cut -c1-1500
