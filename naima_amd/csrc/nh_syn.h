// nh_syn.h -- the per-node arithmetic of Synchrotron._spectrum (radiative.py:300-340):
// shared by k_synchrotron (nh_synchrotron.hip) and the fused half-step kernel
#pragma once
#include "nh_common.h"

// d = a*b + c as the three-address v_fma_f64.  The compiler prefers the two-address
// v_fmac_f64 and then copies the (loop-invariant) coefficient into the destination
// first: one v_mov_b64 per Horner step, 17 of the ~110 instructions of a node.
__device__ __forceinline__ double nh_fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// the same with the addend in a scalar register pair (one SGPR source is allowed): the
// coefficients of a polynomial then cost no vector registers at all
__device__ __forceinline__ double nh_fma3s(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}

// 1/sqrt(a), 1/a for well-scaled a (1 <= a < 1e5 here): single-precision seed
// (v_cvt + v_rsq_f32/v_rcp_f32 + v_cvt, 9 cycles against 18 for the f64 instruction,
// same 2^-23 accuracy) + two Newton steps
__device__ __forceinline__ double nh_rsqrt(double a) {
  double y = (double)__builtin_amdgcn_rsqf((float)a);
  const double h = 0.5 * a;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

__device__ __forceinline__ double nh_rcp2f(double x) {
  double r = (double)__builtin_amdgcn_rcpf((float)x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

// exp(-x) for 0 <= x <= 746 (gradual underflow through v_ldexp_f64)
__device__ __forceinline__ double nh_exp_neg(double x) {
  const double t = -x;
  const double kf = rint(t * 1.4426950408889634);
  double r = fma(-kf, 6.93147180369123816490e-01, t);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;           // 1/13!
  p = nh_fma3s(p, r, 2.08767569878681e-09);     // 1/12!
  p = nh_fma3s(p, r, 2.505210838544172e-08);    // 1/11!
  p = nh_fma3s(p, r, 2.755731922398589e-07);    // 1/10!
  p = nh_fma3s(p, r, 2.755731922398589e-06);    // 1/9!
  p = nh_fma3s(p, r, 2.48015873015873e-05);     // 1/8!
  p = nh_fma3s(p, r, 1.984126984126984e-04);    // 1/7!
  p = nh_fma3s(p, r, 1.388888888888889e-03);    // 1/6!
  p = nh_fma3s(p, r, 8.333333333333333e-03);    // 1/5!
  p = nh_fma3s(p, r, 4.166666666666666e-02);    // 1/4!
  p = nh_fma3s(p, r, 1.666666666666667e-01);    // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kf);
}

// P(x) of AKP10 Eq. D7 (radiative.py:300-311) from cb = cbrt(x)
__device__ __forceinline__ double syn_P(double cb) {
  const double cb2 = cb * cb;
  const double cb4 = cb2 * cb2;
  const double rs = nh_rsqrt(fma(3.4, cb2, 1.0));
  const double gt2 = fma(0.347, cb4, fma(2.210, cb2, 1.0));
  const double gt3 = fma(0.217, cb4, fma(1.353, cb2, 1.0));
  return (1.808 * cb) * rs * (gt2 * nh_rcp2f(gt3));
}

// ln(P2/P1) for neighbouring nodes
__device__ __forceinline__ double syn_dlnP(double P1, double P2) {
  const double s = (P2 - P1) * nh_rcp2f(P2 + P1);
  const double s2 = s * s;
  if (__builtin_amdgcn_ballot_w64(s2 > 9e-4) != 0ull) {  // coarse grid
    asm volatile("" ::: "memory");  // keep the division and the logarithm in the branch
    return log(P2 / P1);
  }
  double a = fma(s2, 1.0 / 9.0, 1.0 / 7.0);
  a = fma(a, s2, 0.2);
  a = fma(a, s2, 1.0 / 3.0);
  a = fma(a, s2, 1.0);
  return 2.0 * s * a;
}


// ---------------------------------------------------------------------------------------
// The same arithmetic on a diet, for the half-step kernel (issue-bound: every VALU slot of a
// SIMD is taken, so instructions per node are what its time is made of).
// ---------------------------------------------------------------------------------------
// exp(x) for -1100 <= x <= 709 with a 64-entry table T64[j] = 2^(j/64) (in LDS):
// x = (64 e + j) ln2/64 + r, |r| <= ln2/128 = 0.0054, exp(r) by a degree-5 polynomial
// (next term r^6/720 = 3.5e-17): 15 VALU instructions and one LDS read against 19 + 13
// s_nop for the degree-13 Horner form (the inline-asm FMAs with SGPR coefficients make the
// assembler pad every one of them).  Gradual underflow through v_ldexp_f64.
__device__ __forceinline__ double nh_exp_tab(double x, const double* __restrict__ T64) {
  const double kf = rint(x * 92.33248261689366);    // 64 / ln 2
  double r = fma(-kf, 0.010830424696223417, x);     // ln2/64: 36 leading bits ...
  r = fma(-kf, 2.572804622327669e-14, r);           // ... and the rest
  double p = fma(r, 8.3333333333333332e-03, 4.1666666666666664e-02);
  p = fma(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const int k = (int)kf;
  return ldexp(T64[k & 63] * p, k >> 6);
}

// P(x) of AKP10 Eq. D7 from cb = cbrt(x) with ONE reciprocal square root for both
// 1/sqrt(1 + 3.4 cb^2) and 1/gt3:  P = 1.808 cb gt2 rsqrt(gt3^2 (1 + 3.4 cb^2))
// (ONE Newton step on the single-precision seed: 1.5 e0^2 = 2e-14 relative on P, next to the
// 1.4e-13 of the three-term ln(P2/P1) below -- three instructions of a node's 73)
__device__ __forceinline__ double nh_rsqrt1(double a) {
  const double y = (double)__builtin_amdgcn_rsqf((float)a);
  return y * fma(-(0.5 * a) * y, y, 1.5);
}
__device__ __forceinline__ double syn_P1(double cb) {
  const double cb2 = cb * cb;
  const double cb4 = cb2 * cb2;
  const double t34 = fma(3.4, cb2, 1.0);
  const double gt2 = fma(0.347, cb4, fma(2.210, cb2, 1.0));
  const double gt3 = fma(0.217, cb4, fma(1.353, cb2, 1.0));
  return ((1.808 * cb) * gt2) * nh_rsqrt1((gt3 * gt3) * t34);
}

// ln(P2/P1) for neighbouring nodes: 2 atanh(s), s = (P2-P1)/(P2+P1).  Naima's default grids
// (100 nodes per decade) have s^2 < 1e-4, where three terms are exact to 1.4e-13 relative
// (5e-16 absolute); up to s^2 = 9e-4 five terms; coarser grids take the logarithm.
// (out of line: the library logarithm and the IEEE division are ~150 instructions that would sit
// in the middle of the hot loop's instruction stream, for grids coarser than naima ever uses)
__device__ __attribute__((noinline)) double syn_log_ratio(double P1, double P2) {
  return log(P2 / P1);
}
__device__ __forceinline__ double syn_dlnP1(double P1, double P2) {
  const double s = (P2 - P1) * nh_rcp1f(P2 + P1);
  const double s2 = s * s;
  if (__builtin_amdgcn_ballot_w64(s2 > 1e-4) != 0ull) {
    asm volatile("" ::: "memory");  // keep the slower forms in the branch
    if (__builtin_amdgcn_ballot_w64(s2 > 9e-4) != 0ull) return syn_log_ratio(P1, P2);
    double a = fma(s2, 1.0 / 9.0, 1.0 / 7.0);
    a = fma(a, s2, 0.2);
    a = fma(a, s2, 1.0 / 3.0);
    a = fma(a, s2, 1.0);
    return 2.0 * s * a;
  }
  double a = fma(s2, 0.2, 1.0 / 3.0);
  a = fma(a, s2, 1.0);
  return (s + s) * a;
}
