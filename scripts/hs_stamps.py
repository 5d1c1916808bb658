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
ball = float(sys.argv[3]) if len(sys.argv) > 3 else 0.005
burn = int(sys.argv[4]) if len(sys.argv) > 4 else 40
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
graph = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
model, p0, raw, data, prior, labels = build_problem(name, na)
s = EnsembleSampler(nw, p0.size, na.lnprob, args=[data, model, prior], seed=seed, naima_style=True,
                    store_blobs=False, device=True, use_graph=graph)
pos = p0 + ball * p0 * s._rng.normal(size=(nw, p0.size))
st = s.run_mcmc(pos, burn, store=False)
ctx.sync()
hs = s._dev._plan["hs"]
print(name, nw, "threads", hs["threads"], "blocks", hs["blocks"], "lds", hs["lds_bytes"])
acc = []
for rep in range(20):
    st = s.run_mcmc(st, 1, store=False)
    out = np.zeros(67840, dtype=np.int64)
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
print("block 0: each wave's arrival at the barrier that ends the weights phase (us):",
      [round((last[192 + w] - t0) / 100.0, 2) for w in range(hs["threads"] // 64)])
print("block 0: each wave before its LDS fills / at the FIRST barrier (us):",
      [(round((last[224 + w] - t0) / 100.0, 2), round((last[208 + w] - t0) / 100.0, 2)) for w in range(hs["threads"] // 64)])
nb = hs["blocks"]
st_, en_ = last[256:256 + nb].astype(float), last[1280:1280 + nb].astype(float)
dur = (en_ - st_) / 100.0
t00 = st_.min()
print("all %d workgroups: duration median %.2f max %.2f us; first start -> last end %.2f us; start skew %.2f us"
      % (nb, np.median(dur), dur.max(), (en_.max() - t00) / 100.0, (st_.max() - t00) / 100.0))
worst = np.argsort(-dur)[:6]
c = np.asarray(st.coords)
print("slowest workgroups:", [(int(w), round(float(dur[w]), 1)) for w in worst])
allst = last[2304:2304 + 16 * nb].reshape(nb, 16)[:, :10].astype(float)
pw = last[18688:18688 + nb * 48].reshape(nb, 16, 3)
for w in worst[:2]:
    print("  workgroup %d:" % w, " ".join("%s=%.2f" % (n, v) for n, v in zip(names, (allst[w] - allst[w, 0]) / 100.0)))
    print("     per wave (end us, tab, syn):", [(round((pw[w, q, 0] - allst[w, 0]) / 100.0, 1), int(pw[w, q, 1]), int(pw[w, q, 2] & 255)) for q in range(16)],
          "nA", int(pw[w, 0, 2] >> 8 & 4095), "Cd", int(pw[w, 0, 2] >> 20))
stage = s._dev._plan.get("stage")
if stage is not None:
    # the staged plan's other launch (Context._stage_a): its phases the same way
    import ctypes as C
    th, bl, ld = C.c_int(), C.c_int(), C.c_longlong()
    _lib._chk(_lib._lib.nh_half_step_info(stage["plan"], C.byref(th), C.byref(bl), C.byref(ld)))
    out = np.zeros(67840, dtype=np.int64)
    _lib._chk(_lib._lib.nh_half_step_stamps(ctx.h, stage["plan"], out.ctypes.data))
    print("stage A: threads", th.value, "blocks", bl.value, "lds", ld.value)
    a1 = out[:128].reshape(8, 16)[:, :12].astype(float)
    for b in (0, 3, 7):
        print("  block", b, " ".join("%s=%.2f" % (n, v) for n, v in zip(names, (a1[b] - a1[b, 0]) / 100.0)))
    nb1 = bl.value
    st1, en1 = out[256:256 + nb1].astype(float), out[1280:1280 + nb1].astype(float)
    print("  all %d workgroups: duration median %.2f max %.2f us; first start -> last end %.2f us"
          % (nb1, np.median(en1 - st1) / 100.0, (en1 - st1).max() / 100.0, (en1.max() - st1.min()) / 100.0))
    print("  block 0 per wave: item-phase start / end, table items, syn items:",
          [(round((out[176 + w] - out[0]) / 100.0, 1), round((out[128 + w] - out[0]) / 100.0, 1), int(out[144 + w]), int(out[160 + w])) for w in range(th.value // 64)])
if name == "cfg3":
    from naima_amd import constants as K
    l0, l1 = np.log10(1e9 / K.MEC2_EV), 9.0
    gam = np.logspace(l0, l1, max(10, int(100 * (l1 - l0))))
    E = np.asarray(raw["energy"], float) * 1e3
    B = np.abs(c[:, 3]) * 1e-6
    qfac = K.ERG_PER_EV * 2.0 * K.M_E_G * K.C_CGS / (3.0 * K.E_GAUSS * K.HBAR_CGS * B)
    x = (E[None, :, None] * qfac[:, None, None]) / gam[None, None, :] ** 2
    live = (x <= 746.0).sum(axis=2).sum(axis=1)
    Emin = gam[0] * K.MEC2_EV
    with np.errstate(all="ignore"):
        t = (Emin / (10 ** c[:, 2] * 1e12)) ** c[:, 4]
    zero_w = ~(t < 745)
    lp = np.asarray(st.log_prob)
    print("walkers: median live %d; live > 1.2 median: %d of which zero-weight %d; stuck (lnp < -1000): %d" % (
        np.median(live), (live > 1.2 * np.median(live)).sum(),
        ((live > 1.2 * np.median(live)) & zero_w).sum(), (lp < -1000).sum()))
    heavy = np.nonzero((live > 1.2 * np.median(live)) & ~zero_w)[0]
    for i in heavy[:8]:
        print("   heavy, weights not zero: walker %d live %d lnp %.6g pars %s" % (i, live[i], lp[i], np.array2string(c[i], precision=4)))
