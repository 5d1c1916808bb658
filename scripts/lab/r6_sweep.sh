# round 6: item lengths again, now that an item's preamble is 16 ints of LDS (round 5)
cd $GRAFT_REPO_ROOT
run() { python bench.py --workload ${WL:-cfg3} --walkers ${NW:-512} --steps 20 --warmup 5 --no-cpu --no-blobs-run --min-time 0.4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,3), round(d['roofline']['us_per_half_step'],2))"; }
for rep in 1 2; do
for seg in 24 32 48; do for n in 24 32 48; do for g in 0 1; do
  echo -n "cfg3/512 rep$rep SEG=$seg syn_nodes=$n grade=$g: "; NH_HS_SEG=$seg NH_RUN_SYN_NODES=$n NH_HS_GRADE=$g run
done; done; done; done
