// nh_syn2.h -- Synchrotron._spectrum's integrand (radiative.py:300-340) in the log domain, on a
// comb: the resident loop's synchrotron work item (k_half_step_run, nh_persist.hip).
//
// The direct form (nh_hs.h: hs_syn_item) costs ~69 vector instructions per grid node: the AKP10
// function P(x) = Gtilde(x) e^x (one reciprocal square root, 19), exp(-x) (16), and -- because
// trapz_loglog (utils.py:336-345) needs the LOG-ratio of neighbouring nodes to 1e-13 -- the
// three-term atanh of (P2 - P1) / (P2 + P1) (14) before the segment itself (12).  The kernel is
// bound by vector issue, so instructions per node are what its time is made of.
//
// Here every node is ONE exponent
//     E_i = ln w_i + g(t_i) - x_i,     g(t) = ln(Gtilde(x) e^x),  t = ln x,
// its integrand u_i = exp(E_i) and the segment's log-ratio ln(u2/u1) = E_2 - E_1 EXACTLY (one
// subtraction of neighbouring exponents -- consistent with u_2 / u_1 to the last place of the
// exponential, which is what the cancellation in (u2 - u1) / ln(u2/u1) needs; what an error of E
// itself does is move u by that relative amount and no more).  g(t) is smooth and slowly varying
// (t/3 + const for small x, a constant for large x), so it is tabulated; and because naima's
// particle grid is np.logspace (radiative.py:147-154: ln gamma_i = ln gamma_0 + i lx), the nodes of
// one (walker, photon energy) pair lie on a COMB in t,
//     t_i = ln q - 2 ln gamma_i = T_top - (z + i) delta,   delta = 2 lx,
// so with table pieces of m comb steps (h = m delta ~ 0.16; degree 5, <= 1.2e-12 absolute on g)
// aligned to the same comb, node i sits in piece (Z + i) / m at the local coordinate
// ((Z + i) mod m + f) / m with Z = floor(z), f = z - Z: no per-node index arithmetic, no
// logarithm, the coefficients of a piece are read once per m nodes, and the lanes of a wave --
// (energy, chunk) pairs whose chunks start on piece boundaries -- walk the pieces in step.
// Everything is kept in units of ln2 / 128 (E' = E 128 / ln2), so that the exponential is
// rint, a subtraction, a degree-4 polynomial and a 128-entry table 2^(j/128).
//
// ~37 vector instructions per node: x = cb^3 from the grid's cube roots (3), Horner (5), E' (2),
// exponential (12), log-ratio (2), segment (hs_seg_pre: 10), bookkeeping (3).
//
// Taken when the plan's synchrotron grid is log-uniform to 1e-11 (checked on the host when the
// loop is created; any other grid keeps hs_syn_item) -- then the segment width ln(g2/g1) is one
// number too.  Exact zeros of the reference: a node with x > 746 has Gtilde = 0 there
// (exp(-x) underflows) and is never the first node of a range, except for the <= m - 1 nodes a
// range's start is aligned down by; those carry u ~ w e^-746 or less, 1e-324 of the integral.
// A zero weight is ln w = -1e300: u = 0 exactly and the segment to either neighbour an exact 0,
// as utils.py:347-348 has it (HS_S2_FLOOR below).  A negative
// amplitude is a sign in front of the integral (the integrand is linear in it); a negative
// magnetic field makes ln q a NaN and the spectrum with it, as it makes the reference's.
#pragma once
#include "nh_common.h"

#define HS_S2_TTOP 7.0       // top of the table in t = ln x: above ln(746 e^h) for every h used
#define HS_S2_TBOT (-46.0)   // below it g = ln 1.808 + t / 3 to 5e-14: the last piece, linear
#define HS_S2_GUARD 16       // guard nodes either side of the walker's ln w and the grid's cube roots
#define HS_S2_TBITS 10       // the exponential's table: 2^(j / 1024), 8 KB of LDS -- a degree-3 remainder (128 entries: degree 4)
#define HS_S2_TN (1 << HS_S2_TBITS)
#define HS_S2_LAMBDA 1477.3197218702985  // 1024 / ln 2
#define HS_S2_C 0.00067690154351557159L  // ln 2 / 1024
// ln w of a zero weight and of the guard nodes: hs_exp128 gives an exact 0 (the integer conversion
// saturates), and the log-ratio to a finite neighbour is beyond the single-precision range, where
// nh_rcp1f returns 0 -- the segment between a zero node and its neighbour is an exact 0, as
// utils.py:347-348 has it (two zero nodes: the series branch, 0 x lx)
#define HS_S2_FLOOR (-1.0e300)
#define HS_S2_DEG 5
#define HS_S2_STRIDE 6       // doubles per piece
#ifndef HS_S2_INLINE
#define HS_S2_INLINE __forceinline__
#endif

struct hs_syn2_par {  // wave-uniform
  int lm, P, nG, pad;     // log2 of the nodes per piece | the last (linear) piece | grid nodes
  double ilx, th, im, lml;  // (ln2/128) / lx | 2^-10 / lx | 1 / m | (m - 1) / m
};

typedef double hs_s2_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const hs_s2_d2 hs_s2_lds_d2;
typedef __attribute__((address_space(3))) const double hs_s2_lds_d;

__device__ __forceinline__ double hs_s2_ld(unsigned addr) {
  return *(hs_s2_lds_d*)(unsigned long long)addr;
}

// 2^(v / 128) for finite v (NaN in, NaN out; gradual underflow, overflow and the saturating integer
// conversion through ldexp).  t128: LDS byte address of 2^(j/128), j < 128.  The remainder's
// polynomial is degree 4 on |r| ln2/128 <= 0.0027: 1.2e-15 relative -- it has to be that good,
// because the truncation errors of two neighbouring nodes are NOT the same function of anything
// and (u2 - u1) / ln(u2/u1) divides their difference by |dl| >= 2^-10 (with the 64-entry table
// of nh_exp_tab and this degree: 4e-14 / 1e-3 = 4e-11 on the segments around the integrand's peak,
// measured)
// fma(a, b, c) as ONE three-operand v_fma_f64 whatever it would have written: in the items' inner loop, with the coefficients parked in
// vector registers across it, every other Horner step came out as v_mov_b64 (a copy of the
// coefficient) + v_fmac_f64 (which overwrites its addend)
__device__ __forceinline__ double hs_fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ double hs_fma3c(double a, double b, double c_scalar) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_scalar));
  return d;
}
__device__ __forceinline__ double hs_fma3s(double a, double b_scalar, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_scalar), "v"(c));
  return d;
}
__device__ __forceinline__ double hs_exp128(double v, unsigned t128) {
  const double kf = rint(v);
  const double r = v - kf;  // |r| <= 1/2 (exact)
  double p = hs_fma3s(r, 5.169222938345892e-11, 2.2909784980688163e-07);  // c^3/6, c^2/2,  c = ln2/1024
  p = hs_fma3c(p, r, 6.769015435155716e-04);                              // c
  p = fma(p, r, 1.0);
  const int k = (int)kf;
  // (v_and + v_lshl_add: left to itself the compiler shifts first, masks with 0x3f8 and adds -- three)
  unsigned j = (unsigned)(k & (HS_S2_TN - 1));
  asm("" : "+v"(j));
  return ldexp(hs_s2_ld(t128 + (j << 3)) * p, k >> HS_S2_TBITS);
}

// floor(x / d) for 0 <= x < 2^20, 1 <= d < 2^12 with d's reciprocal given in single precision: a
// multiplication, a conversion and one correction either way instead of the ~25 instructions of
// an integer division (a work item starts with two of them per lane)
__device__ __forceinline__ int hs_div_small(int x, int d, float rd) {
  int q = (int)((float)x * rd);
  const int r = x - q * d;
  q += r >= d ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}

// One work item: 64 (live photon energy, chunk) pairs.  Per live energy a (compacted, as in
// hs_syn_item): ai0[a] its first live node, Zs[a] the comb index of node 0, and in sq (nEs
// apart) q | Lambda ln(q) / 3 | f / m | CS1 (signed by the amplitude).
// a_lw / a_ig: LDS byte addresses of node 0 of the walker's Lambda ln(w) and of the grid's
// 1/gamma^2 (HS_S2_GUARD entries either side); a_tab: the table, HS_S2_STRIDE doubles per
// piece, 16-byte aligned; a_t128: 2^(j/128), j < 128.
__device__ HS_S2_INLINE void hs_syn2_item(int ix, int lane, int nA, int Cd, int nEs,
                                             const hs_syn2_par& S, const int* ai0, const int* Zs,
                                             const double* sq, unsigned a_lw, unsigned a_ig,
                                             unsigned a_tab, unsigned a_t128, double* part_s) {
  const int vt = ix * 64 + lane;
  const int ch = hs_div_small(vt, nA, __builtin_amdgcn_rcpf((float)nA));  // (nA, Cd: wave-uniform)
  const int a = vt - ch * nA;
  if (ch >= Cd) return;
  const int lm = S.lm, m = 1 << lm;
  const int Z = Zs[a];
  const int k0 = ((Z + ai0[a]) >> lm) << lm;  // the range starts on a piece boundary
  const int kend = Z + S.nG - 1;              // the grid's last node
  int per = hs_div_small(kend - k0 + Cd, Cd, __builtin_amdgcn_rcpf((float)Cd));  // nodes k0 .. kend over Cd chunks ...
  per = ((per + m - 1) >> lm) << lm;          // ... of whole pieces
  const int kb = k0 + ch * per;
  // The segments' log-ratios stay in the exponents' own unit (1/128 octave): dl' = dE ilx has the
  // same ilx for every segment of a log-uniform grid, so the sum is taken over (u2 - u1) / dE and
  // multiplied by 1 / ilx once (one multiplication per node less); |dl'| < th = 2^-10 / lx  <=>
  // |dE| < th / ilx = 2^-10 128 / ln 2, whatever the grid.
  double thE = 0x1p-10 * HS_S2_LAMBDA;
  asm volatile("" : "+s"(thE));  // (a scalar operand of the comparison, not a vector register kept across the loops)
  double acc = 0.0;
  if (kb <= kend) {
    const int n = min(per, kend - kb + 1);
    const int groups = (n + m - 1) >> lm;     // (the last chunk's last piece runs on into the guards)
    const double qx = sq[a], Kc = sq[nEs + a], lam0 = sq[2 * nEs + a];
    double c0, c1, c2, c3, c4, c5;
    auto coefs = [&](int p, int kfirst, double& lam) {  // piece p, whose first node is kfirst
      const int pe = min(p, S.P);
      const unsigned at = a_tab + (unsigned)pe * (8u * HS_S2_STRIDE);
      const hs_s2_d2 v01 = *(hs_s2_lds_d2*)(unsigned long long)at;
      const hs_s2_d2 v23 = *(hs_s2_lds_d2*)(unsigned long long)(at + 16u);
      const hs_s2_d2 v45 = *(hs_s2_lds_d2*)(unsigned long long)(at + 32u);
      c0 = v01.x + Kc; c1 = v01.y; c2 = v23.x; c3 = v23.y; c4 = v45.x; c5 = v45.y;
      lam = fma((double)(kfirst - (pe << lm)), S.im, lam0);  // (= f / m inside the table)
    };
    auto node = [&](unsigned pl, unsigned pg, double lam, double& E, double& u) {
      const double x = qx * hs_s2_ld(pg);  // (q / gamma^2: one product -- it was the cube of cbrt(q) cbrt(1/gamma^2), three)
      double g = fma(c5, lam, c4);
      g = fma(g, lam, c3);
      g = fma(g, lam, c2);
      g = fma(g, lam, c1);
      g = fma(g, lam, c0);
      E = fma(x, -HS_S2_LAMBDA, g) + hs_s2_ld(pl);
      u = hs_exp128(E, a_t128);
    };
    // the start node kb - 1: the last node of the piece before
    double lam, E1, u1;
    coefs((kb >> lm) - 1, kb - m, lam);
    const int i0 = kb - 1 - Z;
    unsigned pl = a_lw + 8u * (unsigned)i0, pg = a_ig + 8u * (unsigned)i0;
    node(pl, pg, lam + S.lml, E1, u1);
    int kf = kb;
    for (int gi = 0; gi < groups; ++gi, kf += m) {
      coefs(kf >> lm, kf, lam);
      for (int j = 0; j < m; j += 2) {  // (m is even: two nodes per trip, two chains to interleave)
        double EA, uA, EB, uB;
        node(pl + 8u, pg + 8u, lam, EA, uA);
        node(pl + 16u, pg + 16u, lam + S.im, EB, uB);
        acc = hs_seg_pre(acc, u1, uA, EA - E1, thE);
        acc = hs_seg_pre(acc, uA, uB, EB - EA, thE);
        E1 = EB;
        u1 = uB;
        lam += 2.0 * S.im;
        pl += 16u;
        pg += 16u;
      }
    }
  }
  // (linear in u: CS1 once per thread -- and the segments' common 1 / ilx, see thE)
  // (ilx is a kernel argument: as a loop invariant its reciprocal is formed once per launch, kept in
  // a vector register pair across every phase of every slice, and spilled -- opaque, it is six
  // instructions per item)
  double ilx_here = S.ilx;
  asm volatile("" : "+v"(ilx_here));
  part_s[ch * nEs + a] = acc * (sq[3 * nEs + a] * nh_rcp(ilx_here));
}

// ---- the table of nh_syn2.h, built on the host when the loop is created -------------------------
// rho(t) = ln(Gtilde(x) e^x) - t / 3 - ln 1.808,  x = e^t   (radiative.py:300-311); the table holds
// Lambda (rho + ln 1.808)
static inline long double hs_s2_rho(long double t) {
  const long double s = expl(2.0L * t / 3.0L);
  return logl((1.0L + 2.210L * s + 0.347L * s * s) /
              ((1.0L + 1.353L * s + 0.217L * s * s) * sqrtl(1.0L + 3.4L * s)));
}
// piece p covers t in (T_top - (p + 1) h, T_top - p h]; its polynomial in lambda = (T_top - t) / h - p
// interpolates Lambda rho at the six Chebyshev nodes of [0, 1] (<= 1.2e-12 absolute for h <= 0.226)
static inline void hs_s2_piece(int p, long double h, double* c /*[HS_S2_STRIDE]*/) {
  const int n = HS_S2_DEG + 1;
  long double A[HS_S2_DEG + 1][HS_S2_DEG + 2];
  for (int r = 0; r < n; ++r) {
    const long double lam = 0.5L * (1.0L + cosl((2 * r + 1) * 3.14159265358979323846264L / (2 * n)));
    long double pw = 1.0L;
    for (int k = 0; k < n; ++k) { A[r][k] = pw; pw *= lam; }
    A[r][n] = (long double)HS_S2_LAMBDA * (hs_s2_rho((long double)HS_S2_TTOP - (p + lam) * h) + logl(1.808L));
  }
  for (int k = 0; k < n; ++k) {  // Gaussian elimination, partial pivoting
    int piv = k;
    for (int r = k + 1; r < n; ++r) if (fabsl(A[r][k]) > fabsl(A[piv][k])) piv = r;
    for (int q = 0; q <= n; ++q) { const long double t = A[k][q]; A[k][q] = A[piv][q]; A[piv][q] = t; }
    for (int r = k + 1; r < n; ++r) {
      const long double f = A[r][k] / A[k][k];
      for (int q = k; q <= n; ++q) A[r][q] -= f * A[k][q];
    }
  }
  long double sol[HS_S2_DEG + 1];
  for (int k = n - 1; k >= 0; --k) {
    long double v = A[k][n];
    for (int q = k + 1; q < n; ++q) v -= A[k][q] * sol[q];
    sol[k] = v / A[k][k];
    c[k] = (double)sol[k];
  }
}

// What a kernel needs to run hs_syn2_item on a log-uniform grid (radiative.py:147-154 makes it
// one): the table's pieces followed by Lambda (ln gamma_i / 3 + ln scale) per node, and the comb's
// constants.  false: the grid is not log-uniform to 1e-12 (or too coarse / too fine for pieces of
// 2 .. 16 steps): the direct form stays.  (k_half_step's plan; the resident loop's creation holds
// the same steps.)
struct hs_s2_host {
  hs_syn2_par par;
  double invd, z0;           // 1 / (2 lx);  (T_top + 2 ln gamma_0) / (2 lx)
  std::vector<double> data;  // (P + 1) HS_S2_STRIDE table entries | nG node constants
};
static inline bool hs_s2_prepare(const double* gam, int nG, double scale, hs_s2_host& o) {
  bool ok = nG >= 2 * HS_S2_GUARD && gam[0] > 0.0 && scale > 0.0;
  long double lx = 0.0L, dev = 0.0L;
  if (ok) {
    const long double l0 = logl((long double)gam[0]);
    lx = (logl((long double)gam[nG - 1]) - l0) / (nG - 1);
    for (int i = 0; i < nG && ok; ++i) {
      ok = gam[i] > 0.0;
      if (ok) dev = fmaxl(dev, fabsl(logl((long double)gam[i]) - l0 - i * lx));
    }
    ok = ok && lx > 0.0L && dev <= 1e-12L;
  }
  int lm = 0;
  if (ok) {
    lm = (int)lrint(log2(0.16 / (double)(2.0L * lx)));
    ok = lm >= 1 && lm <= 4;
  }
  if (!ok) return false;
  const int m = 1 << lm;
  const long double h = m * 2.0L * lx;
  const int P = (int)ceill(((long double)HS_S2_TTOP - (long double)HS_S2_TBOT) / h);
  o.data.assign((size_t)(P + 1) * HS_S2_STRIDE + nG, 0.0);
  for (int pc = 0; pc < P; ++pc) hs_s2_piece(pc, h, &o.data[(size_t)pc * HS_S2_STRIDE]);
  o.data[(size_t)P * HS_S2_STRIDE] = (double)((long double)HS_S2_LAMBDA * logl(1.808L));
  for (int i = 0; i < nG; ++i)
    o.data[(size_t)(P + 1) * HS_S2_STRIDE + i] =
        (double)((long double)HS_S2_LAMBDA * (logl((long double)gam[i]) / 3.0L + logl((long double)scale)));
  o.par.lm = lm; o.par.P = P; o.par.nG = nG; o.par.pad = 0;
  o.par.ilx = (double)(HS_S2_C / lx);  // (ln 2 / 1024) / lx
  o.par.th = (double)((long double)NH_SEG_SMALL_POS / lx);
  o.par.im = 1.0 / m;
  o.par.lml = (double)(m - 1) / m;
  o.invd = (double)(1.0L / (2.0L * lx));
  o.z0 = (double)(((long double)HS_S2_TTOP + 2.0L * logl((long double)gam[0])) / (2.0L * lx));
  return true;
}
