"""Eight stamps per wave inside a slice (library built with -DHSR_FINE=1|2 -DHSR_FINE_PART=0|1; workgroup
blockIdx.x == 0 of that part, iterations 32 .. 39): median time of each stamp per wave, relative to
stamp REF's earliest wave of the same iteration (the slice-start stamp is shared by a walker's parts).

    NAIMA_AMD_LIB=naima_amd/variants/fine2p0.so NH_HS_DEBUG=1 python scripts/lab/r6_fine.py cfg5 256 [REF]
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("NH_HS_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd import _lib  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ref = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem(name, na)
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=20260929,
                    naima_style=True, store_blobs=True, device=True)
pos = p0 + 0.1 * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, 8, store=False)
st = s.run_mcmc(st, 400, store=False)
st = s.run_mcmc(st, 32, store=True)
dev = s._dev
raw = np.zeros((256 * 64 * 8 + 64 * 4 * 16,), dtype=np.int64)
_lib._chk(_lib._lib.nh_half_step_run_stamps(ctx.h, dev._run, raw.ctypes.data_as(C.c_void_p)))
f = raw[256 * 64 * 8:256 * 64 * 8 + 8 * 8 * 16].reshape(8, 8, 16).astype(float) / 100.0
f[f == 0] = np.nan
t0 = np.nanmin(f[:, ref, :], axis=1)
print(name, nw, os.environ.get("NAIMA_AMD_LIB"), dev.resident_info)
print("stamp   " + " ".join("w%-4d" % w for w in range(16)))
for k in range(8):
    rel = f[:, k, :] - t0[:, None]
    with np.errstate(all="ignore"):
        med = np.nanmedian(rel, axis=0)
    print("F%d     " % k + " ".join("%5.2f" % v if np.isfinite(v) else "   - " for v in med))
