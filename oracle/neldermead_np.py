"""Test infrastructure: the sequential Nelder-Mead of naima's prefit, one function call
per point, restated from /root/reference/src/naima/extern/minimize.py:47-217 (relative
xtol/ftol; reflection rho=1, expansion chi=2, contraction psi=0.5, shrink sigma=0.5; the
initial simplex perturbs each coordinate by 5 %, or sets 0.00025 where it is zero).
Only tests may import it: it checks naima_amd.neldermead.minimize_batched."""
import numpy as np


def minimize_sequential(func, x0, xtol=1e-4, ftol=1e-4, maxiter=None, maxfev=None):
    calls = [0]

    def f(x):
        calls[0] += 1
        return func(x)

    x0 = np.asarray(x0, dtype=float).ravel()
    N = len(x0)
    maxiter = N * 200 if maxiter is None else maxiter
    maxfev = N * 200 if maxfev is None else maxfev
    sim = np.zeros((N + 1, N))
    fsim = np.zeros(N + 1)
    sim[0] = x0
    fsim[0] = f(x0)
    for k in range(N):
        y = x0.copy()
        y[k] = 1.05 * y[k] if y[k] != 0 else 0.00025
        sim[k + 1] = y
        fsim[k + 1] = f(y)
    ind = np.argsort(fsim)
    fsim, sim = fsim[ind], sim[ind]
    it = 1
    with np.errstate(divide="ignore", invalid="ignore"):
        while calls[0] < maxfev and it < maxiter:
            if (np.max(np.abs((sim[1:] - sim[0]) / sim[0])) <= xtol
                    and np.max(np.abs((fsim[0] - fsim[1:]) / fsim[0])) <= ftol):
                break
            xbar = np.add.reduce(sim[:-1], 0) / N
            xr = 2 * xbar - sim[-1]
            fxr = f(xr)
            shrink = False
            if fxr < fsim[0]:
                xe = 3 * xbar - 2 * sim[-1]
                fxe = f(xe)
                if fxe < fxr:
                    sim[-1], fsim[-1] = xe, fxe
                else:
                    sim[-1], fsim[-1] = xr, fxr
            elif fxr < fsim[-2]:
                sim[-1], fsim[-1] = xr, fxr
            elif fxr < fsim[-1]:
                xc = 1.5 * xbar - 0.5 * sim[-1]
                fxc = f(xc)
                if fxc <= fxr:
                    sim[-1], fsim[-1] = xc, fxc
                else:
                    shrink = True
            else:
                xcc = 0.5 * xbar + 0.5 * sim[-1]
                fxcc = f(xcc)
                if fxcc < fsim[-1]:
                    sim[-1], fsim[-1] = xcc, fxcc
                else:
                    shrink = True
            if shrink:
                for j in range(1, N + 1):
                    sim[j] = sim[0] + 0.5 * (sim[j] - sim[0])
                    fsim[j] = f(sim[j])
            ind = np.argsort(fsim)
            sim, fsim = sim[ind], fsim[ind]
            it += 1
    status = 1 if calls[0] >= maxfev else (2 if it >= maxiter else 0)
    return dict(x=sim[0], fun=float(np.min(fsim)), nfev=calls[0], nit=it, status=status,
                success=status == 0)
