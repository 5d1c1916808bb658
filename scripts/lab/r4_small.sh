cd $GRAFT_REPO_ROOT
b() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,3), round(d['roofline']['us_per_half_step'],2))"; }
for w in 128 256; do
  echo -n "cfg3/$w syn2=0: "; NH_RUN_SYN2=0 timeout 300 python bench.py --workload cfg3 --walkers $w --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
  for n in 0 4 8 12 16 24; do
    echo -n "cfg3/$w syn2=1 nodes=$n: "; if [ $n = 0 ]; then timeout 300 python bench.py --workload cfg3 --walkers $w --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b; else NH_RUN_SYN_NODES=$n timeout 300 python bench.py --workload cfg3 --walkers $w --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b; fi
  done
done
echo -n "cfg2/256 syn2=1 nodes 16/24: "; NH_RUN_SYN_NODES=16 timeout 300 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b; NH_RUN_SYN_NODES=24 timeout 300 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | b
