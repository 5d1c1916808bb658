// nh_pdist.h -- the particle distribution functions of models.py:88-407 as device code
// (shared by the weights kernels of nh_core.hip and the Kelner+06 quadrature)
#pragma once
#include "nh_common.h"

struct pd_par { double A, e0, al, ec, be, eb, a2; };

// fma(a, b, C) with the constant C in a SCALAR register pair.  Left to itself the compiler writes a
// Horner step whose addend is a 64-bit literal as v_mov_b32 x 2 + v_fmac_f64 -- the addend of the
// two-operand form is its destination, so the constant is built in vector registers first: three
// vector instructions per coefficient where one does (the weights' two exponentials and the
// cut-off's expm1 are 19 such steps per node).  s_mov_b32 x 2 go to the scalar unit, which the
// waves of a SIMD that are bound by their vector instructions do not miss.
__device__ __forceinline__ double pd_fma_sc(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}

// exp(d) - 1 with d = beta * ln(E2/E1): a few per cent on naima's default grids (100 nodes
// per decade: d = 0.023 beta) -- ten Taylor terms are exact to 3e-17 below |d| = 0.1; the
// grid density is a user parameter (nEed = 10 gives d = 0.23), so a wave that sees a
// larger |d| anywhere takes the library function instead
__device__ __forceinline__ double pd_expm1_small(double d) {
  if (__builtin_amdgcn_ballot_w64(fabs(d) >= 0.1) != 0ull) {
    asm volatile("" ::: "memory");  // keep the library call in the branch
    return expm1(d);
  }
  double p = fma(d, 2.7557319223985893e-07 /* 1/10! */, 2.7557319223985888e-06);
  p = pd_fma_sc(p, d, 2.4801587301587302e-05);
  p = pd_fma_sc(p, d, 1.9841269841269841e-04);
  p = pd_fma_sc(p, d, 1.3888888888888889e-03);
  p = pd_fma_sc(p, d, 8.3333333333333332e-03);
  p = pd_fma_sc(p, d, 4.1666666666666664e-02);
  p = pd_fma_sc(p, d, 1.6666666666666666e-01);
  p = fma(p, d, 0.5);
  p = fma(p, d, 1.0);
  return p * d;
}

// exp(x), any x: Cody-Waite reduction + degree-13 Taylor + v_ldexp_f64 (gradual underflow,
// overflow to inf through ldexp); 1 ulp on |r| <= ln2/2.  About half the instructions
// of the library exp, which the weights kernels call two to three times per node.
__device__ __forceinline__ double pd_exp(double x) {
  const double xc = fmin(fmax(x, -1100.0), 1100.0);  // keeps the exponent an int
  const double kf = rint(xc * 1.4426950408889634);
  double r = fma(-kf, 6.93147180369123816490e-01, xc);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;
  p = fma(p, r, 2.08767569878681e-09);
  p = fma(p, r, 2.505210838544172e-08);
  p = fma(p, r, 2.755731922398589e-07);
  p = fma(p, r, 2.755731922398589e-06);
  p = fma(p, r, 2.48015873015873e-05);
  p = fma(p, r, 1.984126984126984e-04);
  p = fma(p, r, 1.388888888888889e-03);
  p = fma(p, r, 8.333333333333333e-03);
  p = fma(p, r, 4.166666666666666e-02);
  p = fma(p, r, 1.666666666666667e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return (x == x) ? ldexp(p, (int)kf) : x;  // NaN in, NaN out
}

// the same with a 64-entry table T64[j] = 2^(j/64) (LDS) and a degree-5 polynomial: 17
// instructions instead of 25 (the weights take two to three exponentials per node)
__device__ __forceinline__ double pd_exp_tab(double x, const double* __restrict__ T64) {
  const double xc = fmin(fmax(x, -1100.0), 1100.0);
  const double kf = rint(xc * 92.33248261689366);   // 64 / ln 2
  double r = fma(-kf, 0.010830424696223417, xc);    // ln2/64: 36 leading bits ...
  r = fma(-kf, 2.572804622327669e-14, r);           // ... and the rest
  double p = fma(r, 8.3333333333333332e-03, 4.1666666666666664e-02);
  p = pd_fma_sc(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const int k = (int)kf;
  return (x == x) ? ldexp(T64[k & 63] * p, k >> 6) : x;  // NaN in, NaN out
}

// the same with the table's LDS BYTE ADDRESS (a pointer into dynamic LDS handed to a function that
// is not inlined is a flat pointer, and getting the local address back out of it costs a scalar
// load from a per-kernel offset table at the head of every exponential)
typedef __attribute__((address_space(3))) const double pd_lds_cd;
__device__ __forceinline__ double pd_exp_tab_lds(double x, unsigned t64) {
  const double xc = fmin(fmax(x, -1100.0), 1100.0);
  const double kf = rint(xc * 92.33248261689366);   // 64 / ln 2
  double r = fma(-kf, 0.010830424696223417, xc);    // ln2/64: 36 leading bits ...
  r = fma(-kf, 2.572804622327669e-14, r);           // ... and the rest
  double p = fma(r, 8.3333333333333332e-03, 4.1666666666666664e-02);
  p = pd_fma_sc(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const int k = (int)kf;
  unsigned j = (unsigned)(k & 63);
  asm("" : "+v"(j));  // (v_and + v_lshl_add, not shift + mask + add)
  const double tj = *(pd_lds_cd*)(unsigned long long)(t64 + (j << 3));
  return (x == x) ? ldexp(tj * p, k >> 6) : x;  // NaN in, NaN out
}

// One node of a walker's particle spectrum: n(E) as the reference evaluates it
// (models.py:88-92, 157-161, 234-238, 330-335, 402-407; x**p as exp(p ln x), 1e-14)
// and the log-ratio of the SHAPE to the next node, ln f(E2)/f(E1), assembled from
// small pieces with lr = ln(E2/E1):  power laws -> -alpha lr;  cutoff ->
// -(t2 - t1) = -t1 expm1(beta lr);  log-parabola -> -alpha lr - beta lr (l1 + l2).
// Inputs are logarithms: lxx = ln(E/e_0), lxc = ln(E/e_cutoff), lkb = ln(e_break/e_0);
// b1, b2 say whether this node / the next one lie below the break.
// T64 != nullptr: exponentials through pd_exp_tab (t64_lds != 0: through pd_exp_tab_lds).
__device__ __forceinline__ void pd_core(int kind, const pd_par& p, double lxx, double lxc,
                                        double lkb, bool b1, bool b2, double lr, double& n,
                                        double& dsh, const double* __restrict__ T64 = nullptr,
                                        double* __restrict__ lnn = nullptr, bool full = true,
                                        unsigned t64_lds = 0) {
  auto pd_exp = [T64, t64_lds](double v) {
    return t64_lds ? pd_exp_tab_lds(v, t64_lds) : (T64 ? pd_exp_tab(v, T64) : ::pd_exp(v));
  };
  double ex;  // ln(n / A): what the log-domain consumers (nh_syn2.h) take instead of n
  switch (kind) {
    case NH_PD_POWERLAW:
      ex = -p.al * lxx;
      dsh = -p.al * lr;
      break;
    case NH_PD_ECPL: {
      const double t = pd_exp(p.be * lxc);
      ex = -p.al * lxx - t;
      dsh = full ? -p.al * lr - t * pd_expm1_small(p.be * lr) : 0.0;
    } break;
    case NH_PD_BROKENPL:
    case NH_PD_ECBPL: {
      const double lK = (p.a2 - p.al) * lkb;
      ex = (b1 ? 0.0 : lK) - (b1 ? p.al : p.a2) * lxx;
      if (b1 == b2) {
        dsh = -(b1 ? p.al : p.a2) * lr;
      } else {  // the one segment that straddles the break
        dsh = (b2 ? 0.0 : lK) - (b1 ? 0.0 : lK) -
              ((b2 ? p.al : p.a2) * (lxx + lr) - (b1 ? p.al : p.a2) * lxx);
      }
      if (kind == NH_PD_ECBPL) {
        const double t = pd_exp(p.be * lxc);
        ex -= t;
        if (full) dsh -= t * pd_expm1_small(p.be * lr);
      }
    } break;
    default: {  // NH_PD_LOGPARABOLA
      ex = (-p.al - p.be * lxx) * lxx;
      dsh = -p.al * lr - p.be * lr * (lxx + (lxx + lr));
    } break;
  }
  // (full == false, wave-uniform: the caller wants ln(n / A) only -- the log-domain synchrotron
  // items of nh_syn2.h -- and is spared the exponential and the cut-off's expm1)
  n = full ? p.A * pd_exp(ex) : 0.0;
  if (lnn) *lnn = ex;
  // a cutoff energy so far below the grid that (E/e_c)^beta overflows makes the log-ratio
  // -inf (the node itself is an exact 0): kept finite, because the reductions form 1/dl
  // before they look at the nodes (0 x 1/inf must stay 0, not become NaN)
  dsh = fmin(fmax(dsh, -NH_DL_ZERO), NH_DL_ZERO);
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

