cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
python scripts/lab/r4_syn2_debug.py cfg3 512 2>&1 | grep -v Warning | head -9
python scripts/lab/r4_syn2_debug.py cfg2 256 2>&1 | grep -v Warning | sed -n 1,5p
for n in 8 12 16 20 24 32 48; do for o in 2 0; do
  echo -n "cfg3 syn_nodes=$n order=$o: "; NH_RUN_SYN_NODES=$n NH_RUN_ORDER=$o timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,3), round(d['roofline']['us_per_half_step'],2))"
done; done
for n in 4 6 8 10 16; do
  echo -n "cfg2 syn_nodes=$n: "; NH_RUN_SYN_NODES=$n timeout 300 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,3), round(d['roofline']['us_per_half_step'],2))"
done
