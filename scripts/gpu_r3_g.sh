cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3g
rm -rf $O; mkdir -p $O
(time timeout 1800 python -m pytest tests/test_gpu_loops.py -m gpu -q -k "resident or benchmarks_ball or benchmarks_size or more_walkers") > $O/tests_loops.log 2>&1; tail -6 $O/tests_loops.log | cut -c1-300
timeout 300 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>$O/bench_cfg3.err | tee $O/bench_cfg3.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', d['timing']['value_min'], d['timing']['value_max'], d['kernels_us_per_launch'], d['kernel_launches'], d['roofline'].get('us_per_half_step'))
" || tail -3 $O/bench_cfg3.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run --min-time 0.1 > $GRAFT_REPO_ROOT/$O/tl.json 2> $GRAFT_REPO_ROOT/$O/tl.err ); echo "trace exit $?"
python scripts/region_timeline.py $O 0.6 30 | cut -c1-160
NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py cfg3 512 > $O/stamps_cfg3.txt 2>&1; cat $O/stamps_cfg3.txt | cut -c1-200
timeout 600 python scripts/nan_hunt.py cfg5 256 > $O/nan_cfg5.log 2>&1; tail -30 $O/nan_cfg5.log | cut -c1-250
timeout 600 python scripts/nan_hunt.py cfg2 256 > $O/nan_cfg2.log 2>&1; tail -30 $O/nan_cfg2.log | cut -c1-250
