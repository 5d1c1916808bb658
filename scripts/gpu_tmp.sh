cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; wl=$1; shift
env "$@" timeout 300 python bench.py $wl --steps 20 --warmup 5 --no-cpu --min-time 0.4 > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.load(open('$O/$tag.json')); print('$tag', round(d['value']), round(d['value_without_blobs']), d['kernels_us_per_launch'].get('k_half_step'))
except Exception as e: print('$tag ERR', e)
PY
}
run base "--workload cfg3" A=1
run hs24 "--workload cfg3" NH_HS_SYN_NODES=24
run run10 "--workload cfg3" NH_RUN_SYN_NODES=10
run run30 "--workload cfg3" NH_RUN_SYN_NODES=30
run run40 "--workload cfg3" NH_RUN_SYN_NODES=40
run hs40 "--workload cfg3" NH_HS_SYN_NODES=40
run base2 "--workload cfg3" A=1
