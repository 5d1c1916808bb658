# one bench line per workload (and the per-launch kernel's), value + kernel times only
cd $GRAFT_REPO_ROOT
one() { tag=$1; shift; timeout 300 python bench.py "$@" --no-cpu 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-28s %12.0f  %s' % ('$tag', d['value'], d.get('kernels_us_per_launch')))"; }
for w in ${WL:-cfg3 cfg2 cfg5 cfg1 cfg4}; do
  one $w --workload $w --steps 20 --warmup 5
  if [ -n "$PERLAUNCH" ]; then NAIMA_AMD_RESIDENT=0 one "$w per-launch" --workload $w --steps 20 --warmup 5; fi
done
true
