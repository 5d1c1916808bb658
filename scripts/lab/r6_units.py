"""round 6: how many weight units each of a walker's two workgroups has (rows split), NH_HS_DEBUG=1"""
import ctypes as C, os, sys
import numpy as np
os.environ.setdefault("NH_HS_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import naima_amd as na
from bench import build_problem
from naima_amd import _lib
from naima_amd.sampler import EnsembleSampler
name, nw = sys.argv[1], int(sys.argv[2])
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem(name, na)
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=20260929, naima_style=True, store_blobs=True, device=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, 8, store=False)
st = s.run_mcmc(st, 32, store=False)
raw_ = np.zeros((256 * 64 * 8 + 64 * 4 * 16,), dtype=np.int64)
_lib._chk(_lib._lib.nh_half_step_run_stamps(ctx.h, s._dev._run, raw_.ctypes.data_as(C.c_void_p)))
b = 256 * 64 * 8
print(name, nw, s._dev.resident_info)
for part in range(2):
    print("part", part, "units", raw_[b + 3 * 16 + 4 + part])
    for g in range(2):
        v = int(raw_[b + 7 * 16 + part * 4 + 2 * g])
        print("   grid", g, "rows", v & 0xffffffff, (v >> 32) & 0xffff, "r0" if g == 0 else "", (int(raw_[b + 7 * 16 + part * 4 + 1]) >> 48))
