# what the driver does at round end: the GPU test suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
(time timeout 1800 python -m pytest tests -m gpu -x -q) > gpurun_out/final/gputest.log 2>&1; tail -3 gpurun_out/final/gputest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; cut -c1-330 gpurun_out/final/bench.json
