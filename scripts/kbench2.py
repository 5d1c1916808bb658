"""in-situ-like alternation of the hot kernels (run under rocprofv3 --kernel-trace --stats)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from naima_amd import _lib
from naima_amd.constants import MEC2_EV, MEC2_ERG, ERG_TO_EV
from naima_amd.radiative import BaseElectron
u = na.u
ctx = _lib.get_context()
N = 256
mode = sys.argv[1] if len(sys.argv) > 1 else "wis"
rng = np.random.default_rng(0)
E = np.sort(np.concatenate([np.geomspace(550, 11200, 36), np.geomspace(0.33e12, 170e12, 28)]))
nE = E.size
rows = np.zeros((N, 8)); rows[:, 0] = 10 ** (33 + 0.01 * rng.standard_normal(N)); rows[:, 1] = 1e13
rows[:, 2] = 2.5 + 0.02 * rng.standard_normal(N); rows[:, 3] = 48e12; rows[:, 4] = 1.0
B = (12 + 0.1 * rng.standard_normal(N)) * 1e-6
rd = ctx.array(rows)
def grid(lo):
    gam = BaseElectron._gam_between(lo * u.GeV, 1e9 * na.constants.mec2, 100)
    gd = ctx.const(gam); ed = ctx.const((gam * MEC2_ERG) * ERG_TO_EV)
    w, dlw = ctx.empty((N, gam.size)), ctx.empty((N, gam.size))
    f = lambda: ctx.call("nh_particle_weights", 1, rd, N, ed, gd, gam.size, MEC2_EV, w, dlw, None)
    return gam, gd, w, dlw, ctx.grid_logratio(gd), f
gam, gd, w, dlw, lx, f1 = grid(1.0)
gam2, gd2, w2, dlw2, lx2, f2 = grid(100.0)
Ed = ctx.const(E); Bd = ctx.array(B); out = ctx.empty((N, nE))
nK = 3 * nE
Kt, dKt = ctx.empty((gam2.size, nK)), ctx.empty((gam2.size, nK))
for j, T in enumerate((2.72548, 30.0, 3000.0)):
    ctx.call("nh_table_ic_planck", gd2, gam2.size, Ed, nE, T, -1.0, Kt.ptr + 8 * j * nE, dKt.ptr + 8 * j * nE, nK)
out2 = ctx.empty((N, nK))
f1(); f2(); ctx.sync()
for _ in range(60):
    if "w" in mode:
        f1(); f2()
    if "i" in mode:
        ctx.call("nh_integrate_tables", w2, dlw2, N, gam2.size, lx2, Kt, dKt, nK, None, out2, nK, 1, 1)
    if "s" in mode:
        ctx.call("nh_synchrotron", w, dlw, Bd, 1, N, gd, lx, gam.size, Ed, nE, out, nE)
ctx.sync()
