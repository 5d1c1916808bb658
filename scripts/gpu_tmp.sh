cd $GRAFT_REPO_ROOT
O=gpurun_out/t13; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_loops.py -m gpu -q -x) > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-300
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/one_$i.json 2> $O/one.err
python -c "import json; d=json.load(open('$O/one_$i.json')); print(round(d['value']), d['ms_per_step'], d['kernels_us_per_launch'])"; done
timeout 600 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/cfg5.json 2> $O/one.err
python -c "import json; d=json.load(open('$O/cfg5.json')); print(round(d['value']), d['ms_per_step'], d['kernels_us_per_launch'])"
