set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 900 python bench.py --steps 100 --warmup 10 2>&1 | tee gpurun_out/bench_n1.json | tail -5
