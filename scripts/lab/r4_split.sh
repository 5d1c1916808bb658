cd $GRAFT_REPO_ROOT
b() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,3), round(d['roofline']['us_per_half_step'],2), d['loop'][:40])"; }
for k in 1 2; do
  echo -n "cfg5/256 kmax=$k: "; NH_HS_SPLIT_MIN_WORK=0 NH_HS_SPLIT=$k timeout 300 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
  echo -n "cfg5/256 analytic kmax=$k: "; 
done
for k in 1 2 4 8; do
  echo -n "cfg1/32 kmax=$k: "; NH_HS_SPLIT_MIN_WORK=0 NH_HS_SPLIT=$k timeout 300 python bench.py --workload cfg1 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
  echo -n "cfg5/64 kmax=$k: "; NH_HS_SPLIT_MIN_WORK=0 NH_HS_SPLIT=$k timeout 300 python bench.py --workload cfg5 --walkers 64 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
done
for k in 2 4 8; do
  echo -n "cfg3/128 kmax=$k: "; NH_HS_SPLIT=$k timeout 300 python bench.py --workload cfg3 --walkers 128 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
done
