cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep; rm -rf $O; mkdir -p $O
run() { tag=$1; shift
env "$@" timeout 300 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run --min-time 0.2 > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.load(open('$O/$tag.json')); print('$tag', round(d['value']), d['kernels_us_per_launch'])
except Exception as e: print('$tag ERR', e)
PY
}
run base A=1
run W1 NH_INT_W=1
run W4 NH_INT_W=4
run C8 NH_INT_C=8
run W1C8 NH_INT_W=1 NH_INT_C=8
run W4C8 NH_INT_W=4 NH_INT_C=8
run S2 NH_INT_SPLIT=2
run W1S2 NH_INT_W=1 NH_INT_SPLIT=2
run W4C4 NH_INT_W=4 NH_INT_C=4
