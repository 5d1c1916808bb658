cd $GRAFT_REPO_ROOT
O=gpurun_out/syn2; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -x) > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-250
for w in cfg3 cfg2; do
timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu --min-time 0.4 > $O/$w.json 2> $O/$w.err
python - <<PY
import json
d=json.load(open('$O/$w.json')); print('$w', round(d['value']), round(d['value_without_blobs']), d['kernels_us_per_launch'])
PY
done
