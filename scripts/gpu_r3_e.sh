cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3e
rm -rf $O; mkdir -p $O
(time timeout 1800 python -m pytest tests/test_gpu_loops.py -m gpu -q -x) > $O/tests_loops.log 2>&1; tail -5 $O/tests_loops.log | cut -c1-300
for g in 0 1; do
NH_HS_GRADE=$g NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py cfg3 512 > $O/stamps_cfg3_g$g.txt 2>&1; cat $O/stamps_cfg3_g$g.txt | cut -c1-200
NH_HS_GRADE=$g timeout 300 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>$O/bench_cfg3_g$g.err | tee $O/bench_cfg3_g$g.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3 grade=$g', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', d['timing']['value_min'], d['timing']['value_max'])
" || tail -3 $O/bench_cfg3_g$g.err
done
NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py cfg5 256 > $O/stamps_cfg5.txt 2>&1; cat $O/stamps_cfg5.txt | cut -c1-200
timeout 600 python scripts/nan_hunt.py cfg5 256 > $O/nan_cfg5.log 2>&1; tail -30 $O/nan_cfg5.log | cut -c1-250
timeout 600 python scripts/nan_hunt.py cfg2 256 > $O/nan_cfg2.log 2>&1; tail -30 $O/nan_cfg2.log | cut -c1-250
for w in cfg5 cfg1 cfg2; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$w', d['config']['walkers_total'], round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')
"
done
