# GPU-side timeline of 20-step regions (rocprofv3 kernel trace): every kernel's start / end
# relative to the region's first kernel ->  gpurun_out/timeline/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/timeline
mkdir -p $O
cat > /tmp/_tl.py <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import naima_amd as na
from bench import build_problem
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem("cfg3", na)
s = EnsembleSampler(512, p0.size, na.lnprob, args=[data, model, prior], seed=1, naima_style=True,
                    store_blobs=True, device=True, use_graph=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(512, p0.size))
st = s.run_mcmc(pos, 100, store=False)
ctx.sync()
for _ in range(40):
    ctx.sync()
    st = s.run_mcmc(st, 20, store=True)
    ctx.sync()
    s.reset()
    time.sleep(0.002)
print("resident launches", s._dev.resident_launches, "failed", getattr(s._dev, "resident_failed_launches", None))
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o tl -- python /tmp/_tl.py > $O/out.log 2> $O/err.log < /dev/null )
tail -2 $O/out.log
f=$(find $O -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]) for r in csv.DictReader(open(sys.argv[1]))))
# regions are separated by the 2 ms sleeps: print the last three
groups, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][1] > 1_000_000:
        groups.append(cur); cur = []
    cur.append(r)
groups.append(cur)
for g in groups[-3:]:
    t0 = g[0][0]
    print("region: %d kernels, first start -> last end %.1f us" % (len(g), (g[-1][1] - t0) / 1e3))
    for a, b, n in g:
        print("   %8.1f -> %8.1f  (%7.1f us)  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, n))
PY
