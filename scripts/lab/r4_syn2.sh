# round 4: the log-domain synchrotron items (nh_syn2.h) -- direct test, loop tests, bench A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_loops.py -x -q -k "log_domain" > $O/t_syn2.log 2>&1; tail -15 $O/t_syn2.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_syn2.json 2> $O/bench_syn2.err; cut -c1-200 $O/bench_syn2.json
NH_RUN_SYN2=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_syn1.json 2> $O/bench_syn1.err; cut -c1-200 $O/bench_syn1.json
timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu > $O/bench_cfg2_syn2.json 2>> $O/err.log; cut -c1-200 $O/bench_cfg2_syn2.json
NH_RUN_SYN2=0 timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu > $O/bench_cfg2_syn1.json 2>> $O/err.log; cut -c1-200 $O/bench_cfg2_syn1.json
