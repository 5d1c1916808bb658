// nh_pdist.h -- the particle distribution functions of models.py:88-407 as device code
// (shared by the weights kernels of nh_core.hip and the Kelner+06 quadrature)
#pragma once
#include "nh_common.h"

struct pd_par { double A, e0, al, ec, be, eb, a2; };

// exp(d) - 1 with d = beta * ln(E2/E1): a few per cent on naima's default grids, but
// the grid density is a user parameter (nEed = 10 gives d = 0.23), so no series here
__device__ __forceinline__ double pd_expm1_small(double d) { return expm1(d); }

// One node of a walker's particle spectrum: n(E) as the reference evaluates it
// (models.py:88-92, 157-161, 234-238, 330-335, 402-407; x**p as exp(p ln x), 1e-14)
// and the log-ratio of the SHAPE to the next node, ln f(E2)/f(E1), assembled from
// small pieces with lr = ln(E2/E1):  power laws -> -alpha lr;  cutoff ->
// -(t2 - t1) = -t1 expm1(beta lr);  log-parabola -> -alpha lr - beta lr (l1 + l2).
// Inputs are logarithms: lxx = ln(E/e_0), lxc = ln(E/e_cutoff), lkb = ln(e_break/e_0);
// b1, b2 say whether this node / the next one lie below the break.
__device__ __forceinline__ void pd_core(int kind, const pd_par& p, double lxx, double lxc,
                                        double lkb, bool b1, bool b2, double lr, double& n,
                                        double& dsh) {
  switch (kind) {
    case NH_PD_POWERLAW:
      n = p.A * exp(-p.al * lxx);
      dsh = -p.al * lr;
      break;
    case NH_PD_ECPL: {
      const double t = exp(p.be * lxc);
      n = p.A * exp(-p.al * lxx - t);
      dsh = -p.al * lr - t * pd_expm1_small(p.be * lr);
    } break;
    case NH_PD_BROKENPL:
    case NH_PD_ECBPL: {
      const double lK = (p.a2 - p.al) * lkb;
      double ex = (b1 ? 0.0 : lK) - (b1 ? p.al : p.a2) * lxx;
      if (b1 == b2) {
        dsh = -(b1 ? p.al : p.a2) * lr;
      } else {  // the one segment that straddles the break
        dsh = (b2 ? 0.0 : lK) - (b1 ? 0.0 : lK) -
              ((b2 ? p.al : p.a2) * (lxx + lr) - (b1 ? p.al : p.a2) * lxx);
      }
      if (kind == NH_PD_ECBPL) {
        const double t = exp(p.be * lxc);
        ex -= t;
        dsh -= t * pd_expm1_small(p.be * lr);
      }
      n = p.A * exp(ex);
    } break;
    default: {  // NH_PD_LOGPARABOLA
      n = p.A * exp((-p.al - p.be * lxx) * lxx);
      dsh = -p.al * lr - p.be * lr * (lxx + (lxx + lr));
    } break;
  }
}

__device__ __forceinline__ bool pd_has_cutoff(int kind) {
  return kind == NH_PD_ECPL || kind == NH_PD_ECBPL;
}
__device__ __forceinline__ bool pd_has_break(int kind) {
  return kind == NH_PD_BROKENPL || kind == NH_PD_ECBPL;
}

// the same from energies (three logarithms per node)
__device__ __forceinline__ void pd_node(int kind, const pd_par& p, double E, double E2,
                                        double lr, double& n, double& dsh) {
  const double lxx = log(E / p.e0);
  const double lxc = pd_has_cutoff(kind) ? log(E / p.ec) : 0.0;
  const double lkb = pd_has_break(kind) ? log(p.eb / p.e0) : 0.0;
  pd_core(kind, p, lxx, lxc, lkb, E < p.eb, E2 < p.eb, lr, n, dsh);
}

