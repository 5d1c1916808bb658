"""which walkers make the half-step kernel slow after naima's 10 % initial ball?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na
from bench import build_problem
from naima_amd import _lib, constants as K
from naima_amd.sampler import EnsembleSampler
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem("cfg3", na)
s = EnsembleSampler(512, p0.size, na.lnprob, args=[data, model, prior], seed=20260929, naima_style=True,
                    store_blobs=False, device=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(512, p0.size))
st = s.run_mcmc(pos, 4000, store=False)
c = np.asarray(st.coords); lp = np.asarray(st.log_prob)
l0, l1 = np.log10(1e9 / K.MEC2_EV), 9.0
gam = np.logspace(l0, l1, max(10, int(100 * (l1 - l0))))
E = np.asarray(raw["energy"], float) * 1e3
B = np.abs(c[:, 3]) * 1e-6
qfac = K.ERG_PER_EV * 2.0 * K.M_E_G * K.C_CGS / (3.0 * K.E_GAUSS * K.HBAR_CGS * B)
x = (E[None, :, None] * qfac[:, None, None]) / gam[None, None, :] ** 2
live = (x <= 746.0).sum(axis=2).sum(axis=1)
order = np.argsort(-live)
print("median live nodes", np.median(live), "max", live.max())
for i in order[:12]:
    print("walker %3d live %6d lnp %10.4g pars %s" % (i, live[i], lp[i], np.array2string(c[i], precision=4)))
print("walkers with lnp < -1000:", int((lp < -1000).sum()), " with live > 1.2 median:", int((live > 1.2 * np.median(live)).sum()))
