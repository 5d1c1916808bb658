# refresh of the judged artifacts: default bench line, counters + kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
tail -c 600 gpurun_out/bench_n1.json
bash scripts/gpu_traffic.sh 2>&1 | tail -20
