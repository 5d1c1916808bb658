# round 3, third call: the resident loop's first run (parity + speed + stamps), the host path's NaNs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3c
rm -rf $O; mkdir -p $O
(time timeout 900 python -m pytest tests/test_gpu_loops.py -m gpu -q -s -x -k "resident") > $O/tests_resident.log 2>&1
grep -E "resident ==|passed|failed|Error|error|assert" $O/tests_resident.log | cut -c1-300 | tail -12
for w in cfg3 cfg5 cfg1; do
  for r in 0 1; do
    NAIMA_AMD_RESIDENT=$r timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>$O/bench_${w}_$r.err | tee $O/bench_${w}_$r.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$w resident=$r', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', d['kernels_us_per_launch'], d['kernel_launches'])
" || tail -3 $O/bench_${w}_$r.err
  done
done
timeout 600 python scripts/nan_hunt.py cfg5 256 > $O/nan_cfg5.log 2>&1; tail -30 $O/nan_cfg5.log | cut -c1-250
timeout 600 python scripts/nan_hunt.py cfg2 256 > $O/nan_cfg2.log 2>&1; tail -30 $O/nan_cfg2.log | cut -c1-250
(time timeout 1500 python -m pytest tests/test_gpu_loops.py -m gpu -q -x) > $O/tests_loops.log 2>&1; tail -5 $O/tests_loops.log | cut -c1-300
