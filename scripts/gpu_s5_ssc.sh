# the SSC kernel: its parity tests, the cfg4 bench line, its kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s5ssc
rm -rf $O; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -x -q -k "cfg4 or inverse_compton or ssc or seed") > $O/test.log 2>&1
tail -5 $O/test.log
timeout 300 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/s5ssc/bench_cfg4.json"))
print(d["value"], d["ms_per_step"], d["kernels_us_per_launch"], json.dumps(d.get("fp64_valu"))[:400])
PY
