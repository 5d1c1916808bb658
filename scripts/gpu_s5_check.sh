# full GPU test suite + the default bench line (the state of HEAD on the box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
(time timeout 1200 python -m pytest tests -m gpu -x -q) > gpurun_out/s5/gputest.log 2>&1
tail -5 gpurun_out/s5/gputest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/s5/bench_default.json 2> gpurun_out/s5/bench_default.err
cut -c1-400 gpurun_out/s5/bench_default.json
timeout 300 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run > gpurun_out/s5/bench_cfg4.json 2> gpurun_out/s5/bench_cfg4.err
cut -c1-300 gpurun_out/s5/bench_cfg4.json
