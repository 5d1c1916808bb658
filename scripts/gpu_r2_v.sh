# A/B of kernel build variants on ONE box: VARIANTS="flags1|flags2|..." (empty string = default build)
cd $GRAFT_REPO_ROOT
IFS='|' read -ra V <<< "${VARIANTS:-|-DHS_NO_PREFETCH}"
one() {
  for w in "cfg3 --walkers 512" "cfg5 --walkers 512" "cfg1 --walkers 512" "cfg2 --walkers 512"; do
    set -- $w
    timeout 300 python bench.py --workload $1 $2 $3 --steps 200 --warmup 20 --no-cpu --ball 0.005 --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  ', d['config']['workload'][:5], round(d['value']/1e6,3), 'M/s', d['ms_per_step'], d['kernels_us_per_launch'])
"
  done
}
for rep in 1 2; do
for f in "${V[@]}"; do
  bash naima_amd/csrc/build.sh $f > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  echo "== [$f] rep $rep"
  one
done
done
bash naima_amd/csrc/build.sh > /dev/null 2>&1
