cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r2/tests_n.log 2>&1; echo rc=$?
tail -4 gpurun_out/r2/tests_n.log
echo "--- driver-like default run (ball 0.1)"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2/bench_n1_driverlike.json 2> gpurun_out/r2/bench_n1.err; echo rc=$?; tail -3 gpurun_out/r2/bench_n1.err
python -c "
import json
d=json.load(open('gpurun_out/r2/bench_n1_driverlike.json'))
print(d['value'], d['ms_per_step'], d.get('value_without_blobs', d.get('value_without_blobs', d.get('value_store_blobs'))), d['timing'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['fp64_valu']['kernels'], d['acceptance_fraction'], d['cpu_baseline'])"
