cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3f
rm -rf $O; mkdir -p $O
(time timeout 1800 python -m pytest tests/test_gpu_loops.py -m gpu -q) > $O/tests_loops.log 2>&1; tail -8 $O/tests_loops.log | cut -c1-300
for g in 0 2; do
NH_RUN_ORDER=$g NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py cfg3 512 > $O/stamps_cfg3_o$g.txt 2>&1; cat $O/stamps_cfg3_o$g.txt | cut -c1-200
NH_RUN_ORDER=$g timeout 300 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>$O/bench_cfg3_o$g.err | tee $O/bench_cfg3_o$g.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3 order=$g', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', d['timing']['value_min'], d['timing']['value_max'], d['kernels_us_per_launch'], d['roofline'].get('us_per_half_step'))
" || tail -3 $O/bench_cfg3_o$g.err
done
NAIMA_AMD_RESIDENT=1 timeout 300 python bench.py --workload cfg3 --steps 200 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3 200-step regions', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')
"
NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py cfg5 256 > $O/stamps_cfg5.txt 2>&1; cat $O/stamps_cfg5.txt | cut -c1-200
for w in cfg5 cfg1; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$w', d['config']['walkers_total'], round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')
"
done
