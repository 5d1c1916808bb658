# round 5: what the driver does at round end + bench.py starting its own ranks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5check; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/gputest.log 2>&1; tail -5 $O/gputest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
NAIMA_AMD_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/bench_g2.json 2> $O/bench_g2.err; cut -c1-300 $O/bench_g2.json; tail -3 $O/bench_g2.err
