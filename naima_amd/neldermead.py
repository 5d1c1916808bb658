"""Nelder-Mead with the candidate points of every iteration evaluated as ONE batch.

naima's prefit (core.py:163-217) minimises -lnprob with a Nelder-Mead variant that uses
RELATIVE tolerances (extern/minimize.py:47-217) and calls the model once per point.  On
the GPU a model evaluation costs the same for 1 or for 4 parameter vectors, so the
reflection, expansion and both contraction points of an iteration are evaluated
together (speculatively) and the initial simplex and every shrink as one batch each.
The decisions taken on those values, the convergence test and the way function
evaluations are COUNTED (only the ones the sequential algorithm would have made) are
the sequential algorithm's, so the simplex follows exactly the same path and stops at
the same place.
"""
import numpy as np

RHO, CHI, PSI, SIGMA = 1.0, 2.0, 0.5, 0.5   # extern/minimize.py:91-94
NONZDELT, ZDELT = 0.05, 0.00025             # extern/minimize.py:107-108


def minimize_batched(fbatch, x0, xtol=1e-4, ftol=1e-4, maxiter=None, maxfev=None):
    """``fbatch(X)`` maps an (m, N) array of points to m function values.
    Returns dict(x, fun, nfev, nit, status, success) with scipy's meaning of status
    (0 converged, 1 maxfev reached, 2 maxiter reached)."""
    x0 = np.asarray(x0, dtype=float).ravel()
    N = x0.size
    maxiter = N * 200 if maxiter is None else maxiter
    maxfev = N * 200 if maxfev is None else maxfev
    sim = np.tile(x0, (N + 1, 1))
    for k in range(N):
        sim[k + 1, k] = (1 + NONZDELT) * x0[k] if x0[k] != 0 else ZDELT
    fsim = np.asarray(fbatch(sim), dtype=float)
    nfev = N + 1
    order = np.argsort(fsim)
    sim, fsim = sim[order], fsim[order]
    nit = 1
    with np.errstate(divide="ignore", invalid="ignore"):
        while nfev < maxfev and nit < maxiter:
            if (np.max(np.abs((sim[1:] - sim[0]) / sim[0])) <= xtol
                    and np.max(np.abs((fsim[0] - fsim[1:]) / fsim[0])) <= ftol):
                break
            xbar = np.add.reduce(sim[:-1], 0) / N
            worst = sim[-1]
            cand = np.stack([(1 + RHO) * xbar - RHO * worst,               # reflection
                             (1 + RHO * CHI) * xbar - RHO * CHI * worst,   # expansion
                             (1 + PSI * RHO) * xbar - PSI * RHO * worst,   # outside contraction
                             (1 - PSI) * xbar + PSI * worst])              # inside contraction
            fxr, fxe, fxc, fxcc = np.asarray(fbatch(cand), dtype=float)
            nfev += 1
            shrink = False
            if fxr < fsim[0]:
                nfev += 1
                if fxe < fxr:
                    sim[-1], fsim[-1] = cand[1], fxe
                else:
                    sim[-1], fsim[-1] = cand[0], fxr
            elif fxr < fsim[-2]:
                sim[-1], fsim[-1] = cand[0], fxr
            else:
                nfev += 1
                if fxr < fsim[-1]:
                    if fxc <= fxr:
                        sim[-1], fsim[-1] = cand[2], fxc
                    else:
                        shrink = True
                elif fxcc < fsim[-1]:
                    sim[-1], fsim[-1] = cand[3], fxcc
                else:
                    shrink = True
            if shrink:
                sim[1:] = sim[0] + SIGMA * (sim[1:] - sim[0])
                fsim[1:] = np.asarray(fbatch(sim[1:]), dtype=float)
                nfev += N
            order = np.argsort(fsim)
            sim, fsim = sim[order], fsim[order]
            nit += 1
    status = 1 if nfev >= maxfev else (2 if nit >= maxiter else 0)
    return dict(x=sim[0], fun=float(np.min(fsim)), nfev=nfev, nit=nit, status=status,
                success=status == 0)
