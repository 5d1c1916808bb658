cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_loops.py -m gpu -x -q -k "gives_up or replay" > $O/t1.log 2>&1; tail -3 $O/t1.log
timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -x -q > $O/t2.log 2>&1; tail -3 $O/t2.log
for w in "cfg3 512" "cfg5 256" "cfg2 256" "cfg1 32" "cfg4 256"; do set -- $w
 timeout 600 python bench.py --workload $1 --walkers $2 --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/b_$1.json 2> $O/b_$1.err
 python - <<PY
import json
d=json.load(open("$O/b_$1.json"))
print("$1", round(d["value"]/1e6,3), "M  overhead", d.get("region_overhead_us"), d.get("region_overhead"), "region", round(d["region_us"],1))
PY
done
NAIMA_AMD_RESIDENT=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/b_cfg3_pl.json 2> $O/b_cfg3_pl.err
python -c "
import json
d=json.load(open('$O/b_cfg3_pl.json'))
print('cfg3 per-launch', round(d['value']/1e6,3), 'M  overhead', d.get('region_overhead_us'), d.get('region_overhead'), 'region', round(d['region_us'],1))"
