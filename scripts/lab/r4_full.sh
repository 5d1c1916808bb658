cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4d}; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
