cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_loops.py -m gpu -q -x -k "not cfg4" 2>&1 | tail -2
timeout 300 python scripts/hs_stamps.py cfg3 512 0.005 40 1 2>&1 | grep -v "^  wave" | tail -7
timeout 300 python scripts/hs_stamps.py cfg5 256 0.005 40 1 2>&1 | grep -v "^  wave" | tail -4
for w in cfg3 cfg5; do
timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu --ball 0.005 --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['config']['workload'][:5], round(d['value']/1e6,3), d['ms_per_step'], d['kernels_us_per_launch'])
"
done
