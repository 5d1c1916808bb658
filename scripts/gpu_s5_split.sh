# K workgroups per walker: the loop tests, then bench lines with and without the split
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s5split
rm -rf $O; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_gpu_loops.py tests/test_gpu_general.py tests/test_gpu_parity.py -m gpu -x -q -k "loop or device or sampler or shard or rank") > $O/test.log 2>&1
tail -6 $O/test.log
for w in cfg2 cfg5 cfg1; do
  for k in 1 8; do
    NH_HS_SPLIT=$k timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu --no-blobs-run > $O/bench_${w}_k$k.json 2>> $O/err.log
  done
done
for k in 1 8; do
  NH_HS_SPLIT=$k timeout 300 python bench.py --workload cfg3 --walkers 256 --steps 200 --warmup 20 --no-cpu --no-blobs-run > $O/bench_cfg3w256_k$k.json 2>> $O/err.log
done
NH_HS_SPLIT=8 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/bench_default.json 2>> $O/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/s5split/bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 5), d['kernels_us_per_launch'], d['config'].get('walkers_total'))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 $O/err.log
