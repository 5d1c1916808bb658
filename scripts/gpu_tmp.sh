cd $GRAFT_REPO_ROOT
O=gpurun_out/syn; rm -rf $O; mkdir -p $O
for v in "16 64" "8 64" "8 44" "8 33" "16 44" "16 33" "4 64" "8 22"; do set -- $v
NH_SYN_C=$1 NH_SYN_TW=$2 timeout 300 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run --min-time 0.2 > $O/b_$1_$2.json 2> $O/b_$1_$2.err
python - <<PY
import json
d=json.load(open('$O/b_$1_$2.json'))
print('$1 $2', round(d['value']), d['kernels_us_per_launch'])
PY
done
