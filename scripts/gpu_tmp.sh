cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; rm -rf $O; mkdir -p $O
(time timeout 1800 python -m pytest tests/test_gpu_loops.py -m gpu -q -s -k "more_walkers or resident") > $O/tests.log 2>&1; grep -E "passed|failed|FAILED|Error" $O/tests.log | cut -c1-250
for w in "--walkers-total 2048" "--walkers-total 1024" "--walkers-total 4096"; do
 for r in 0 1; do
  NAIMA_AMD_RESIDENT=$r timeout 300 python bench.py --workload cfg5 --scaling strong $w --steps 20 --warmup 5 --no-cpu --no-blobs-run 2>$O/err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg5 resident=$r', d['config']['walkers_total'], round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline'].get('us_per_half_step'), d['loop'][:25])
" || tail -2 $O/err.log
 done
done
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import naima_amd as na
from bench import build_problem
from naima_amd.sampler import EnsembleSampler
model, p0, raw, data, prior, labels = build_problem("cfg5", na)
s = EnsembleSampler(2048, p0.size, na.lnprob, args=[data, model, prior], seed=1, naima_style=True, store_blobs=True, device=True, nan_policy="reject")
pos = p0 * (1 + 0.01 * np.random.default_rng(3).standard_normal((2048, p0.size)))
st = s.run_mcmc(pos, 8); st = s.run_mcmc(st, 20)
d = s._dev
print("cfg5/2048: resident launches", d.resident_launches, getattr(d, "resident_info", None), getattr(d, "resident_reason", None), d._plan["hs"]["threads"], d._plan["hs"]["lds_bytes"])
PY
