cd $GRAFT_REPO_ROOT
IFS='|' read -ra V <<< "${VARIANTS}"
for f in "${V[@]}"; do
  bash naima_amd/csrc/build.sh $f > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  echo "== [$f]"
  timeout 300 python bench.py --workload cfg3 --steps 200 --warmup 20 --no-cpu --ball 0.005 --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  ', d['config']['workload'][:5], round(d['value']/1e6,3), 'M/s', d['ms_per_step'], d['kernels_us_per_launch'])
"
  timeout 300 python scripts/hs_stamps.py cfg3 512 0.005 40 1 1 2>&1 | grep "^block 0 start" | cut -c1-260
done
bash naima_amd/csrc/build.sh > /dev/null 2>&1
