"""micro-benchmark of the hot entry points at cfg3 shapes (HIP events on the stream)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from naima_amd import _lib
from naima_amd.constants import MEC2_EV, MEC2_ERG, ERG_TO_EV
from naima_amd.radiative import BaseElectron, _dlog
u = na.u
ctx = _lib.get_context()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
E = np.sort(np.concatenate([np.geomspace(550, 11200, 36), np.geomspace(0.33e12, 170e12, 28)]))
nE = E.size
rows = np.zeros((N, 8)); rows[:, 0] = 10 ** (33 + 0.01 * rng.standard_normal(N)); rows[:, 1] = 1e13
rows[:, 2] = 2.5 + 0.02 * rng.standard_normal(N); rows[:, 3] = 48e12; rows[:, 4] = 1.0
B = (12 + 0.1 * rng.standard_normal(N)) * 1e-6

def timeit(fn, reps=50):
    fn(); ctx.sync()
    ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop() / reps * 1e3  # us

def weights(lo):
    gam = BaseElectron._gam_between(lo * u.GeV, 1e9 * na.constants.mec2, 100)
    gd = ctx.const(gam); ed = ctx.const((gam * MEC2_ERG) * ERG_TO_EV)
    w, dlw = ctx.empty((N, gam.size)), ctx.empty((N, gam.size))
    rd = ctx.array(rows)
    f = lambda: ctx.call("nh_particle_weights", 1, rd, N, ed, gd, gam.size, MEC2_EV, w, dlw, None)
    return gam, gd, w, dlw, ctx.grid_logratio(gd), f

res = {}
gam, gd, w, dlw, lx, f = weights(1.0)
res["weights(570)"] = timeit(f)
Ed = ctx.const(E); Bd = ctx.array(B); out = ctx.empty((N, nE))
res["synchrotron"] = timeit(lambda: ctx.call("nh_synchrotron", w, dlw, Bd, 1, N, gd, lx, gam.size, Ed, nE, out, nE))
gam2, gd2, w2, dlw2, lx2, f2 = weights(100.0)
res["weights(370)"] = timeit(f2)
nK = 3 * nE
Kt, dKt = ctx.empty((gam2.size, nK)), ctx.empty((gam2.size, nK))
def tabs():
    for j, T in enumerate((2.72548, 30.0, 3000.0)):
        ctx.call("nh_table_ic_planck", gd2, gam2.size, Ed, nE, T, -1.0, Kt.ptr + 8 * j * nE, dKt.ptr + 8 * j * nE, nK)
res["tables(3 seeds)"] = timeit(tabs, 10)
NS = int(os.environ.get("KB_NSPLIT", "1"))
out2 = ctx.empty((NS * N, nK))
res["integrate(IC nK=192)"] = timeit(lambda: ctx.call("nh_integrate_tables", w2, dlw2, N, gam2.size, lx2, Kt, dKt, nK, None, out2, nK, 1, NS))
gam3, gd3, w3, dlw3, lx3, f3 = weights(1000.0); f3()
K = gam3 * MEC2_ERG
Kd, dKd = ctx.const(K), ctx.const(_dlog(K)); out3 = ctx.empty((N, 1))
res["integrate(We nK=1)"] = timeit(lambda: ctx.call("nh_integrate_tables", w3, dlw3, N, gam3.size, lx3, Kd, dKd, 1, None, out3, 1, 0, 1))
for k, v in res.items():
    print("%-24s %8.2f us" % (k, v))
print("checksum", float(out.get().sum()), float(out2.get().sum()), float(out3.get().sum()))
