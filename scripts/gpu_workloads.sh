# the other BASELINE workloads (parity-test cases; not the bench line)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in cfg1 cfg2 cfg5; do
  timeout 900 python bench.py --workload $c --steps 100 --warmup 10 > gpurun_out/bench_$c.json 2> gpurun_out/err_$c.log
  tail -1 gpurun_out/bench_$c.json | cut -c1-200
done
timeout 1500 python bench.py --workload cfg4 --steps 10 --warmup 2 > gpurun_out/bench_cfg4.json 2> gpurun_out/err_cfg4.log
tail -1 gpurun_out/bench_cfg4.json | cut -c1-200
