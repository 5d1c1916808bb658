cd $GRAFT_REPO_ROOT
for seg in 32 48 64 96 128 192; do for n in 32 48; do
  echo -n "cfg3 NH_HS_SEG=$seg syn_nodes=$n: "; NH_HS_SEG=$seg NH_RUN_SYN_NODES=$n timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,3), round(d['roofline']['us_per_half_step'],2))"
done; done
for seg in 16 32 64; do
  echo -n "cfg2 NH_HS_SEG=$seg: "; NH_HS_SEG=$seg timeout 300 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,3), round(d['roofline']['us_per_half_step'],2))"
done
