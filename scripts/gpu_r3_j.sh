cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3j
rm -rf $O; mkdir -p $O
(time timeout 1800 python -m pytest tests/test_gpu_loops.py -m gpu -q) > $O/tests_loops.log 2>&1; tail -5 $O/tests_loops.log | cut -c1-300
for w in cfg3 cfg5 cfg1; do
timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>$O/bench_$w.err | tee $O/bench_$w.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$w', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', d['kernels_us_per_launch'], d['kernel_launches'], d['roofline'].get('us_per_half_step'))
" || tail -3 $O/bench_$w.err
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run --min-time 0.1 > $GRAFT_REPO_ROOT/$O/tl.json 2> $GRAFT_REPO_ROOT/$O/tl.err ); echo "trace exit $?"
python scripts/region_timeline.py $O 0.6 12 | cut -c1-160
(time timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_loops.py) > $O/tests_rest.log 2>&1; tail -4 $O/tests_rest.log | cut -c1-300
