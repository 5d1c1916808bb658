"""Per-phase timeline of the resident half-step loop (k_half_step_run) from the wall-clock
stamps the kernel writes under NH_HS_DEBUG=1.

    NH_HS_DEBUG=1 python scripts/run_stamps.py cfg3 512 [ball] [seed]
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("NH_HS_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naima_amd as na  # noqa: E402
from bench import build_problem  # noqa: E402
from naima_amd import _lib  # noqa: E402
from naima_amd.sampler import EnsembleSampler  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ball = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 20260929
ctx = _lib.get_context()
model, p0, raw, data, prior, labels = build_problem(name, na)
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=seed,
                    naima_style=True, store_blobs=True, device=True)
pos = p0 + ball * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, 8, store=False)
st = s.run_mcmc(st, 400, store=False)
st = s.run_mcmc(st, 32, store=True)  # the launch whose stamps are read: 64 slices
dev = s._dev
assert dev._run, getattr(dev, "resident_reason", "resident loop not taken")
raw = np.zeros((256 * 64 * 8 + 64 * 4 * 16,), dtype=np.int64)
_lib._chk(_lib._lib.nh_half_step_run_stamps(ctx.h, dev._run, raw.ctypes.data_as(C.c_void_p)))
buf = raw[:256 * 64 * 8].reshape(256, 64, 8)
wst = raw[256 * 64 * 8:].reshape(64, 4, 16).astype(float) / 100.0
print(name, nw, "walkers;", dev.resident_info)
G = min(256, dev.resident_info["grid"])
t = buf[:G].astype(float) / 100.0  # us
t[buf[:G] == 0] = np.nan  # (a stamp nobody wrote: phase A made AHEAD -- two walkers in flight -- leaves none)
with np.errstate(invalid="ignore"):
    # (... or leaves an OLDER launch's in its slot: a stamp earlier than its turn's first one)
    stale = t[:, :, 1:] < t[:, :, :1]
t[:, :, 1:][stale] = np.nan
names = ["records in", "packs + barrier 1", "weights + barrier 2", "tid 0's items", "barrier 3",
         "spectra summed (barrier 4)", "record published"]
it = np.arange(4, 60)  # (steady state)
d = np.diff(t[:, :, :], axis=2)[:, it, :]
print("phase durations, us (median over %d workgroups x %d slices | 10 %% | 90 %%):" % (G, len(it)))
for k, nm in enumerate(names):
    v = d[:, :, k].ravel()
    n_all = v.size
    v = v[np.isfinite(v)]
    if v.size == 0:
        print("  %-28s      -       -       -   (no turn wrote this stamp)" % nm)
        continue
    print("  %-28s %6.2f  %6.2f  %6.2f%s" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90),
                                           "" if v.size == n_all else "   (%d %% of the turns: the others had phase A made ahead)"
                                           % round(100.0 * v.size / n_all)))
per = np.diff(t[:, :, 0], axis=1)[:, it[:-1]]
print("slice period (start to start): median %.2f, 10 %% %.2f, 90 %% %.2f us"
      % (np.median(per), np.percentile(per, 10), np.percentile(per, 90)))
tot = t[:, 60, 0] - t[:, 4, 0]
print("56 slices of a workgroup: median %.1f us = %.2f us per slice; launch start skew %.2f us"
      % (np.median(tot), np.median(tot) / 56, t[:, 0, 0].max() - t[:, 0, 0].min()))
wait = d[:, :, 0]
wait = wait[np.isfinite(wait)]
if wait.size:
    print("waiting for records: mean %.2f us, fraction of slices > 2 us: %.3f" % (wait.mean(), (wait > 2).mean()))

# workgroup 0: arrival of each wave at the barriers, relative to the slice's start (us), slices 8..40
t0 = t[0, :, 0]
for k, nm in enumerate(["barrier 1", "barrier 2", "barrier 3"]):
    rel = np.array([wst[i, k, :] - t0[i] for i in range(8, 40)])
    print("workgroup 0, waves reach %s at (median over 32 slices), us: %s" % (nm, " ".join("%5.1f" % v for v in np.median(rel, axis=0))))
