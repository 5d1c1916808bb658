cd $GRAFT_REPO_ROOT
b() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,3), round(d['roofline']['us_per_half_step'],2))"; }
for r in 0 1; do
  echo -n "cfg5 rebalance=$r: "; NH_RUN_REBALANCE=$r timeout 300 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
  echo -n "cfg1 rebalance=$r: "; NH_RUN_REBALANCE=$r timeout 300 python bench.py --workload cfg1 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
  echo -n "cfg5/2048 rebalance=$r: "; NH_RUN_REBALANCE=$r timeout 300 python bench.py --workload cfg5 --scaling strong --walkers-total 2048 --steps 100 --warmup 10 --no-cpu --no-blobs-run 2>/dev/null | b
  echo -n "cfg3 rebalance=$r: "; NH_RUN_REBALANCE=$r timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
  echo -n "cfg3 seg48 nodes48 rebalance=$r: "; NH_HS_SEG=48 NH_RUN_SYN_NODES=48 NH_RUN_REBALANCE=$r timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
done
echo -n "cfg3 per-launch seg32: "; NAIMA_AMD_RESIDENT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
echo -n "cfg3 per-launch seg48: "; NH_HS_SEG=48 NAIMA_AMD_RESIDENT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
