// nh_lnprob.h -- core.lnprobmodel + priors + core.lnprob epilogue (core.py:64-121) and the
// stretch move's accept for ONE walker, executed by one wave.  Shared by k_lnprobmodel
// (nh_core.hip) and by the synchrotron kernel's epilogue (nh_synchrotron.hip), which runs
// it for its own walker right after producing the last spectrum the likelihood needs.
#pragma once
#include "nh_common.h"

struct nh_comps {
  nh_comp c[NH_MAX_COMP];
  int n;
};

struct nh_lnprob_args {
  nh_comps cs;
  int N, nE;
  const double* conv; const double* flux; const double* elo; const double* ehi;
  const int* ul; const double* cl; const double* lp;
  nh_prior_pack pri;
  double* model_out; double* lnl;
  nh_accept mv;
  int* nan_count;  // the context's counter of NaN log-probabilities met by an accept (or NULL)
};

__device__ __forceinline__ double nh_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// local_comp >= 0: component local_comp is taken from local[k] (LDS: the values the
// calling block has just produced) instead of memory
__device__ __forceinline__ void nh_lnprob_wave(const nh_lnprob_args& A, int wi, int lane,
                                               const double* local, int local_comp) {
  const nh_accept& mv = A.mv;
  const int nE = A.nE;
  // the accept's inputs are requested first so that their latency hides behind the sum
  int me = 0, pa = 0;
  double mz = 1.0, mlnu = 0.0, mold = 0.0, cpa = 0.0, cme = 0.0;
  if (mv.coords) {
    const int g = mv.lo + wi;
    const double* r = mv.blk + (long long)mv.cursor[0] * 3 * mv.ns;
    const int* idx = reinterpret_cast<const int*>(r + 2 * mv.ns);
    me = idx[g];
    pa = idx[mv.ns + g];
    mz = r[g];
    mlnu = r[mv.ns + g];
    mold = mv.logp[me];
    if (lane < mv.ndim) {
      cpa = mv.coords[(long long)pa * mv.ndim + lane];
      cme = mv.coords[(long long)me * mv.ndim + lane];
    }
  }
  // prior terms and the confidence-level table are requested before the sum as well
  double prior = 0.0;
  const bool has_prior = A.lp || A.pri.n > 0;
  if (has_prior && lane == 0) prior = (A.lp ? A.lp[wi] : 0.0) + nh_prior_sum(A.pri, wi);
  const double cl_lane = lane < nE ? A.cl[lane] : 0.0;
  double acc = 0.0;
  int nviol = 0, nul = 0;
  for (int k = lane; k < nE; k += 64) {
    double m = 0.0;
    for (int j = 0; j < A.cs.n; ++j) {
      const double v = (j == local_comp) ? local[k]
                                         : A.cs.c[j].ptr[(long long)wi * A.cs.c[j].ld + k];
      m += A.cs.c[j].scale * v;
    }
    if (A.model_out) A.model_out[(long long)wi * nE + k] = m;
    const double mc = m * A.conv[k];
    const double f = A.flux[k];
    if (A.ul[k]) {
      nul += 1;
      nviol += (mc > f) ? 1 : 0;
    } else {
      const double d = mc - f;
      const double sg = (d > 0.0) ? A.ehi[k] : A.elo[k];
      acc += -(d * d) / (2.0 * (sg * sg));
    }
  }
  // one reduction for both counters (16 bits each: a table has far fewer than 65536
  // upper limits), interleaved with the sum of squares so that the cross-lane latencies overlap
  int cnt = nviol | (nul << 16);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double a2 = __shfl_down(acc, off, 64);
    const int c2 = __shfl_down(cnt, off, 64);
    acc += a2;
    cnt += c2;
  }
  // quirk kept from core.py:89-92: cl is indexed by the violation count
  cnt = __shfl(cnt, 0, 64);
  nviol = cnt & 0xffff;
  nul = cnt >> 16;
  const double clv = (nviol < 64 && nviol < nE) ? __shfl(cl_lane, nviol, 64) : A.cl[nviol];
  if (lane == 0) {
    if (nul > 0) acc += (double)nviol * log(1.0 - clv);
    if (has_prior)  // core.py:115-119: a forbidden walker keeps the prior value
      acc = isinf(prior) ? prior : acc + prior;
    A.lnl[wi] = acc;
  }
  if (mv.coords) {  // nh_move_accept for this walker (emcee RedBlueMove.propose)
    acc = __shfl(acc, 0, 64);
    const double d = (mv.ndim - 1.0) * log(mz) + acc - mold;
    const bool ok = mlnu < d;  // NaN compares false, as numpy
    if (ok && lane < mv.ndim) mv.coords[(long long)me * mv.ndim + lane] = cpa - (cpa - cme) * mz;
    if (lane == 0) {
      const int g = mv.lo + wi;
      if (ok) {
        mv.logp[me] = acc;
        if (mv.naccepted) mv.naccepted[me] += 1;
      }
      mv.accepted[g] = ok ? 1 : 0;
      if (mv.sel) mv.sel[g] = me;
      // emcee raises "Probability function returned NaN" here; a launch cannot, it counts
      if (acc != acc && A.nan_count) atomicAdd(A.nan_count, 1);
    }
  }
}

// The same for nE <= 64 (lane = energy index) in two halves, for a producer kernel that
// runs the likelihood as its epilogue: everything that does not depend on the producer's
// own spectrum -- the other components, the data columns, the prior terms, the move's
// random numbers and coordinates: three dependent round trips to memory other launches
// wrote -- is requested when the kernel STARTS and waits in registers.
struct nh_lnprob_pre {
  double other, conv, flux, elo, ehi, prior, cl_lane;
  double mz, mlnu, mold, cpa, cme;
  int ul, me, pa;
};

// three stages, each one round trip: the caller places them between its own phases so
// that no wave ever waits for the previous stage's answer at a barrier
__device__ __forceinline__ void nh_lnprob64_prefetch_a(nh_lnprob_pre& P, const nh_lnprob_args& A,
                                                       int wi, int lane, int local_comp) {
  P.mz = 1.0;
  if (A.mv.coords) P.me = A.mv.cursor[0];  // the slice index, until stage b
  if ((A.lp || A.pri.n > 0) && lane == 0) P.prior = (A.lp ? A.lp[wi] : 0.0) + nh_prior_sum(A.pri, wi);
  if (lane < A.nE) {
    P.cl_lane = A.cl[lane];
    for (int j = 0; j < A.cs.n; ++j)
      if (j != local_comp) P.other += A.cs.c[j].scale * A.cs.c[j].ptr[(long long)wi * A.cs.c[j].ld + lane];
    P.conv = A.conv[lane];
    P.flux = A.flux[lane];
    P.elo = A.elo[lane];
    P.ehi = A.ehi[lane];
    P.ul = A.ul[lane];
  }
}

__device__ __forceinline__ void nh_lnprob64_prefetch_b(nh_lnprob_pre& P, const nh_lnprob_args& A,
                                                       int wi) {
  const nh_accept& mv = A.mv;
  if (!mv.coords) return;
  const int g = mv.lo + wi;
  const double* r = mv.blk + (long long)P.me * 3 * mv.ns;
  const int* idx = reinterpret_cast<const int*>(r + 2 * mv.ns);
  P.me = idx[g];
  P.pa = idx[mv.ns + g];
  P.mz = r[g];
  P.mlnu = r[mv.ns + g];
}

__device__ __forceinline__ void nh_lnprob64_prefetch_c(nh_lnprob_pre& P, const nh_lnprob_args& A,
                                                       int lane) {
  const nh_accept& mv = A.mv;
  if (!mv.coords) return;
  P.mold = mv.logp[P.me];
  if (lane < mv.ndim) {
    P.cpa = mv.coords[(long long)P.pa * mv.ndim + lane];
    P.cme = mv.coords[(long long)P.me * mv.ndim + lane];
  }
  P.mlnu -= (mv.ndim - 1.0) * log(P.mz);  // ln U' - (ndim-1) ln z: off the epilogue's path
}

__device__ __forceinline__ void nh_lnprob64_finish(const nh_lnprob_args& A, const nh_lnprob_pre& P,
                                                   int wi, int lane, const double* local,
                                                   int local_comp) {
  const nh_accept& mv = A.mv;
  const int nE = A.nE;
  double acc = 0.0;
  int nviol = 0, nul = 0;
  if (lane < nE) {
    const double m = P.other + A.cs.c[local_comp].scale * local[lane];
    const double mc = m * P.conv;
    if (P.ul) {
      nul = 1;
      nviol = (mc > P.flux) ? 1 : 0;
    } else {
      const double d = mc - P.flux;
      const double sg = (d > 0.0) ? P.ehi : P.elo;
      acc = -(d * d) / (2.0 * (sg * sg));
    }
  }
  int cnt = nviol | (nul << 16);  // both counters in one reduction, interleaved with the sum
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double a2 = __shfl_down(acc, off, 64);
    const int c2 = __shfl_down(cnt, off, 64);
    acc += a2;
    cnt += c2;
  }
  cnt = __shfl(cnt, 0, 64);
  nviol = cnt & 0xffff;
  nul = cnt >> 16;
  const double clv = (nviol < 64 && nviol < nE) ? __shfl(P.cl_lane, nviol, 64) : A.cl[nviol];
  const bool has_prior = A.lp || A.pri.n > 0;
  if (lane == 0) {
    if (nul > 0) acc += (double)nviol * log(1.0 - clv);
    if (has_prior) acc = isinf(P.prior) ? P.prior : acc + P.prior;
    A.lnl[wi] = acc;
  }
  if (mv.coords) {
    acc = __shfl(acc, 0, 64);
    const bool ok = P.mlnu < acc - P.mold;  // ln U' < (ndim-1) ln z + lnp(q) - lnp(s)
    if (ok && lane < mv.ndim)
      mv.coords[(long long)P.me * mv.ndim + lane] = P.cpa - (P.cpa - P.cme) * P.mz;
    if (lane == 0) {
      const int g = mv.lo + wi;
      if (ok) {
        mv.logp[P.me] = acc;
        if (mv.naccepted) mv.naccepted[P.me] += 1;
      }
      mv.accepted[g] = ok ? 1 : 0;
      if (mv.sel) mv.sel[g] = P.me;
      if (acc != acc && A.nan_count) atomicAdd(A.nan_count, 1);
    }
  }
}

// host side: fill the by-value argument block from the C-ABI arguments
static inline void nh_lnprob_fill(nh_lnprob_args& A, const nh_comps& cs, int N, int nE,
                                  const double* conv, const double* flux, const double* elo,
                                  const double* ehi, const int* ul, const double* cl,
                                  const double* lp, const nh_prior* terms, int nterms,
                                  double* model_out, double* lnl, const nh_accept* mv) {
  A.cs = cs; A.N = N; A.nE = nE; A.conv = conv; A.flux = flux; A.elo = elo; A.ehi = ehi;
  A.ul = ul; A.cl = cl; A.lp = lp; A.model_out = model_out; A.lnl = lnl;
  A.pri.n = nterms;
  for (int j = 0; j < nterms; ++j) A.pri.t[j] = terms[j];
  nh_accept z = {};
  A.mv = mv ? *mv : z;
  A.nan_count = nullptr;  // (the caller's context sets it where an accept rides in the launch)
}
