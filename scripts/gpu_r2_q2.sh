cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_loops.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
for w in "cfg3 --walkers 512" "cfg5 --walkers 512" "cfg1 --walkers 512" "cfg2 --walkers 512"; do
  set -- $w
  timeout 300 python bench.py --workload $1 $2 $3 --steps 200 --warmup 20 --no-cpu --ball 0.005 --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  ', d['config']['workload'][:5], round(d['value']/1e6,3), 'M/s', d['ms_per_step'], d['kernels_us_per_launch'])
"
done
timeout 300 python scripts/hs_stamps.py cfg5 256 0.005 40 1 2>&1 | grep -v "^  wave" | tail -3 | cut -c1-600
