# what the driver does at round end: the GPU test suite (cold GPU, first command of the lease),
# smoke(), the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${OUT:-final}; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/gputest.log 2>&1; tail -5 $O/gputest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json
