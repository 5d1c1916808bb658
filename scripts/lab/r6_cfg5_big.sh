# round 6: BASELINE's cfg5 ensemble (2048 walkers) on ONE GPU under other plans
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python bench.py --workload cfg5 --scaling strong --walkers-total 2048 --steps 100 --warmup 10 --no-cpu --no-blobs-run --min-time 0.4 2>/tmp/e.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,2), round(d['value_evaluated']/1e6,2), d['loop'][:40], d['kernels_us_per_launch'])
except Exception as e: print('ERR', e); print(open('/tmp/e.log').read()[-400:])
"; }
echo -n "default: "; run X=1
echo -n "per_wg 2: "; run NH_RUN_MAX_PER_WG=2
echo -n "threads 512: "; run NH_HS_THREADS=512
echo -n "threads 512 per_wg 2: "; run NH_HS_THREADS=512 NH_RUN_MAX_PER_WG=2
echo -n "threads 512 per_wg 4: "; run NH_HS_THREADS=512 NH_RUN_MAX_PER_WG=4
echo -n "threads 1024 per_wg 4: "; run NH_HS_THREADS=1024 NH_RUN_MAX_PER_WG=4
echo -n "threads 128: "; run NH_HS_THREADS=128
echo -n "per-launch: "; run NAIMA_AMD_RESIDENT=0
python - <<PY
import os
os.environ["NH_HS_DEBUG"]="0"
PY
