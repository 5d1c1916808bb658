# round 4, first GPU pass: the suite on the edited resident kernel, the default bench line, and
# an A/B of the narrow-table items' nodes per trip (HS_RUN_PK = 6: 8 VGPRs spilled; 5: none)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/gputest.log 2>&1; tail -3 $O/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
for w in cfg5 cfg1; do timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu > $O/bench_${w}_pk6.json 2>> $O/err.log; cut -c1-160 $O/bench_${w}_pk6.json; done
timeout 600 python bench.py --workload cfg5 --scaling strong --walkers-total 2048 --steps 100 --warmup 10 --no-cpu --no-blobs-run > $O/bench_cfg5_2048_pk6.json 2>> $O/err.log; cut -c1-160 $O/bench_cfg5_2048_pk6.json
bash naima_amd/csrc/build.sh -DHS_RUN_PK=5 > $O/build_pk5.log 2>&1; tail -1 $O/build_pk5.log
for w in cfg5 cfg1; do timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu > $O/bench_${w}_pk5.json 2>> $O/err.log; cut -c1-160 $O/bench_${w}_pk5.json; done
timeout 600 python bench.py --workload cfg5 --scaling strong --walkers-total 2048 --steps 100 --warmup 10 --no-cpu --no-blobs-run > $O/bench_cfg5_2048_pk5.json 2>> $O/err.log; cut -c1-160 $O/bench_cfg5_2048_pk5.json
