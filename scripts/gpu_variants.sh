# build variants of the half-step kernel ON the GPU box and time them (one session = one box = one clock)
cd $GRAFT_REPO_ROOT
run() {
  bash naima_amd/csrc/build.sh $1 > /dev/null 2>&1 || { echo "build failed: $1"; return; }
  for a in "cfg3 512"; do
    echo "== [$1] [$2] $a"
    env $2 timeout 300 python scripts/hs_stamps.py $a 2>&1 | tail -1
    env $2 timeout 300 python bench.py --workload cfg3 --steps 200 --warmup 20 --no-cpu --ball 0.005 --no-blobs-run 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   bench', round(d['value']/1e6,3), 'M/s', d['kernels_us_per_launch'])"
  done
}
run "" "A=1"
run "-DHS_SKIP_TAB" "A=1"
run "-DHS_SKIP_SYN" "A=1"
run "-DHS_ORDER=1" "A=1"
run "-DHS_ORDER=2" "A=1"
run "-DHS_SYN_NODES=20" "A=1"
run "-DHS_SYN_NODES=5" "A=1"
run "" "NH_HS_THREADS=512"
bash naima_amd/csrc/build.sh > /dev/null 2>&1
