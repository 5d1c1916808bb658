"""do the synchrotron kernel and the table reduction overlap when launched on two streams?"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from naima_amd import _lib
from naima_amd.constants import MEC2_EV, MEC2_ERG, ERG_TO_EV
from naima_amd.radiative import BaseElectron
u = na.u
ctx = _lib.get_context()
L = _lib._lib
N = 256
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
    ctx.call("nh_particle_weights", 1, rd, N, ed, gd, gam.size, MEC2_EV, w, dlw, None)
    return gam, gd, w, dlw, ctx.grid_logratio(gd)
gam, gd, w, dlw, lx = grid(1.0)
gam2, gd2, w2, dlw2, lx2 = grid(100.0)
Ed = ctx.const(E); Bd = ctx.array(B); out = ctx.empty((N, nE))
nK = 3 * nE
Kt, dKt = ctx.empty((gam2.size, nK)), ctx.empty((gam2.size, nK))
for j, T in enumerate((2.72548, 30.0, 3000.0)):
    ctx.call("nh_table_ic_planck", gd2, gam2.size, Ed, nE, T, -1.0, Kt.ptr + 8 * j * nE, dKt.ptr + 8 * j * nE, nK)
out2 = ctx.empty((2 * N, nK))
def syn(): ctx.call("nh_synchrotron", w, dlw, Bd, 1, N, gd, lx, gam.size, Ed, nE, out, nE)
def ic(): ctx.call("nh_integrate_tables", w2, dlw2, N, gam2.size, lx2, Kt, dKt, nK, None, out2, nK, 1, 2)
def serial():
    ic(); syn()
def forked():
    _lib._chk(L.nh_stream_fork(ctx.h, 0)); ic()
    _lib._chk(L.nh_stream_switch(ctx.h, -1)); syn()
    _lib._chk(L.nh_stream_join(ctx.h))
for name, fn in (("serial", serial), ("two streams", forked), ("serial", serial), ("two streams", forked)):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    ctx.sync()
    print("%-12s %.2f us per (IC + synchrotron)" % (name, (time.perf_counter() - t0) / 200 * 1e6))
