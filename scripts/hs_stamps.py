"""per-phase wall-clock stamps of the one-launch half-step kernel (NH_HS_DEBUG=1)"""
import os, sys
os.environ["NH_HS_DEBUG"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
model, p0, raw, data, prior, labels = build_problem(name, na)
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=1, naima_style=True,
                    store_blobs=False, device=True, use_graph=False)
pos = p0 + 0.005 * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, 40, store=False)
ctx.sync()
hs = s._dev._plan["hs"]
print(name, nw, "threads", hs["threads"], "blocks", hs["blocks"], "lds", hs["lds_bytes"])
acc = []
for rep in range(20):
    st = s.run_mcmc(st, 1, store=False)
    out = np.zeros(256, dtype=np.int64)
    _lib._chk(_lib._lib.nh_half_step_stamps(ctx.h, hs["plan"], out.ctypes.data))
    acc.append(out[:128].reshape(8, 16)[:, :12].astype(float))
    last = out
a = np.array(acc)  # [rep][block][phase]
d = (a - a[:, :, :1]) / 100.0  # us since the block's start
names = ["start", "prefetch issued", "proposal", "packs", "weights+live", "moments", "items done",
         "barrier", "reduce", "lik+accept", "barrier", "end"]
m = np.median(d, axis=0)
for b in (0, 3, 7):
    print("block", b, " ".join("%s=%.2f" % (n, v) for n, v in zip(names, m[b])))
t0 = last[0]
print("block 0 per wave: item-phase start / end (us since block start), table items, syn items")
for w in range(hs["threads"] // 64):
    print("  wave %2d: %.2f -> %.2f  tab %d syn %d" % (w, (last[176 + w] - t0) / 100.0, (last[128 + w] - t0) / 100.0, last[144 + w], last[160 + w]))
