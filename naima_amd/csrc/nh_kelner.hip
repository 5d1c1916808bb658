// nh_kelner.hip -- PionDecayKelner06._spectrum (radiative.py:1543-1767): Kelner, Aharonian
// & Bugayov 2006 parametrisation of pp -> pi0 -> gamma gamma.
//
// The reference integrates with adaptive QUADPACK (scipy.integrate.quad, epsrel = 1e-3)
// one photon energy at a time.  Here every (walker, photon energy) pair is one wave that
// evaluates the same integrals with a fixed composite Gauss-Legendre rule after a change
// of variable that makes the integrands smooth and exponentially decaying:
//   E_gamma >= Etrans (Eq. 71/72, :1665-1684):  int_0^1 sigma J F dx/x,   x = exp(-t)
//       -> int_0^T sigma(Ep) J(Ep) F(x, Ep) dt,  Ep = E_gamma exp(t)
//   E_gamma <  Etrans (delta-functional, :1693-1714):
//       2 int_{Epimin}^inf q_pi / sqrt(Epi^2 - m_pi^2) dEpi,  Epi = m_pi cosh(s)
//       -> 2 int_{s_min}^{s_min+T} q_pi(m_pi cosh s) ds      (the root singularity is gone)
// T = 64 (the integrands fall like exp(-(alpha-1) t)), 128 panels x 8 points.  The result
// is the CONVERGED integral: it differs from the reference's number by the reference's
// own quadrature error (measured 4e-5, bound 1e-3); tests/test_oracle.py pins both.
#include "nh_pdist.h"

namespace {

constexpr double K06_KPI = 0.17;                       // radiative.py:1689
constexpr double K06_MP_TEV = NH_M_P_GEV * 1e-3;       // :1690
constexpr double K06_MPI_TEV = 1.349766e-4;            // :1691
constexpr double K06_ETH_TEV = 1.22e-3;                // :1643
constexpr int K06_PANELS = 128;
constexpr double K06_H = 0.5;

__constant__ double GLX[8] = {-0.9602898564975363, -0.7966664774136267, -0.5255324099163290,
                              -0.1834346424956498, 0.1834346424956498,  0.5255324099163290,
                              0.7966664774136267,  0.9602898564975363};
__constant__ double GLW[8] = {0.1012285362903763, 0.2223810344533745, 0.3137066458778873,
                              0.3626837833783620, 0.3626837833783620, 0.3137066458778873,
                              0.2223810344533745, 0.1012285362903763};

// KAB06 Eq. 73, 79 (radiative.py:1625-1647), cm^2
__device__ __forceinline__ double k06_sigma_inel(double Ep) {
  const double L = log(Ep);
  double s = 34.3 + 1.88 * L + 0.25 * L * L;
  if (Ep <= 0.1) {
    const double r = K06_ETH_TEV / Ep, r2 = r * r;
    const double f = 1.0 - r2 * r2;
    s *= f * f * nh_heaviside(Ep - K06_ETH_TEV);
  }
  return s * 1e-27;
}

// particles per TeV at Ep [TeV]
__device__ __forceinline__ double k06_J(int kind, const pd_par& p, double Ep_TeV) {
  const double E = Ep_TeV * 1e12;
  double n, dsh;
  pd_node(kind, p, E, E, 0.0, n, dsh);
  return n * 1e12;
}

// KAB06 Eq. 58-61 (radiative.py:1597-1623) at x = exp(-t): F1*F2 rearranged so that the
// 1/ln(x) and 1/(1 - x^beta) factors cancel analytically (regular as t -> 0):
//   F = (B/x) G^3 [ G + (4 beta t xb / D)(1 + k (1 - 2 xb) G) ],
//   xb = x^beta, D = 1 + k xb (1 - xb), G = (1 - xb)/D
__device__ __forceinline__ double k06_Fgamma(double t, double Ep) {
  const double L = log(Ep);
  const double B = 1.30 + 0.14 * L + 0.011 * L * L;
  const double beta = 1.0 / (1.79 + 0.11 * L + 0.008 * L * L);
  const double k = 1.0 / (0.801 + 0.049 * L + 0.014 * L * L);
  const double xb = exp(-beta * t);
  const double omx = -expm1(-beta * t);
  const double D = 1.0 + k * xb * omx;
  const double G = omx / D;
  return B * exp(t) * (G * G * G) * (G + (4.0 * beta * t * xb / D) * (1.0 + k * (1.0 - 2.0 * xb) * G));
}

__device__ __forceinline__ double k06_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return __shfl(v, 0, 64);
}

// mode 0: full calculation at E_gamma; mode 1: delta-functional (nhat = 1); mode 2: Wp
__device__ double k06_integral(int mode, int kind, const pd_par& p, double Eg, int lane) {
  double t0 = 0.0;
  if (mode == 1) {
    const double Epimin = Eg + K06_MPI_TEV * K06_MPI_TEV / (4.0 * Eg);
    t0 = acosh(fmax(Epimin / K06_MPI_TEV, 1.0));
  }
  double acc = 0.0;
  // Wp's integrand falls only like exp(-(alpha-2) t): eight times the range
  const int panels = mode == 2 ? 8 * K06_PANELS : K06_PANELS;
  for (int n = lane; n < panels * 8; n += 64) {
    const int pan = n >> 3, q = n & 7;
    const double t = t0 + (pan + 0.5 * (1.0 + GLX[q])) * K06_H;
    double f;
    if (mode == 0) {
      const double Ep = Eg * exp(t);
      f = k06_sigma_inel(Ep) * k06_J(kind, p, Ep) * k06_Fgamma(t, Ep);
    } else if (mode == 1) {
      const double Ep0 = K06_MP_TEV + K06_MPI_TEV * cosh(t) / K06_KPI;
      f = k06_sigma_inel(Ep0) * k06_J(kind, p, Ep0);
    } else {
      const double E = K06_ETH_TEV * exp(t);  // int E J dE = int E^2 J dt
      f = E * E * k06_J(kind, p, E);
    }
    if (!(f == f) || isinf(f)) f = 0.0;  // overflowed tails contribute nothing
    acc += GLW[q] * f;
  }
  acc = k06_wave_sum(acc) * (0.5 * K06_H);
  if (mode == 0) return NH_C_CGS * acc;
  if (mode == 1) return 2.0 * NH_C_CGS / K06_KPI * acc;
  return acc;
}

__global__ __launch_bounds__(256) void k_pion_kelner06(int kind, const double* __restrict__ params,
                                                        int N, const double* __restrict__ E_eV,
                                                        int nE, double Etrans_TeV, int mixed,
                                                        double* __restrict__ out, int ldo,
                                                        double* __restrict__ nhat_out,
                                                        double* __restrict__ wp_out) {
  extern __shared__ double res[];  // [nE + 3]: spectra | full(Etrans) | delta(Etrans) | Wp
  const int wi = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double* pr = params + (long long)wi * NH_PD_NPAR;
  const pd_par p = {pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6]};
  for (int task = wv; task < nE + 3; task += 4) {
    double v = 0.0;
    if (task < nE) {
      const double Eg = E_eV[task] * 1e-12;
      v = k06_integral(Eg >= Etrans_TeV ? 0 : 1, kind, p, Eg, lane);
    } else if (task == nE) {
      if (mixed) v = k06_integral(0, kind, p, Etrans_TeV, lane);
    } else if (task == nE + 1) {
      if (mixed) v = k06_integral(1, kind, p, Etrans_TeV, lane);
    } else if (wp_out) {
      v = k06_integral(2, kind, p, 0.0, lane);
    }
    if (lane == 0) res[task] = v;
  }
  __syncthreads();
  // nhat makes the delta-functional branch meet the full one at Etrans (:1743-1748)
  const double nhat = mixed ? res[nE] / res[nE + 1] : 1.0;
  for (int k = threadIdx.x; k < nE; k += blockDim.x) {
    const bool hi = E_eV[k] * 1e-12 >= Etrans_TeV;
    out[(long long)wi * ldo + k] = (hi ? res[k] : res[k] * nhat) * 1e-12;  // 1/(s TeV) -> 1/(s eV)
  }
  if (threadIdx.x == 0) {
    if (nhat_out) nhat_out[wi] = nhat;
    if (wp_out) wp_out[wi] = res[nE + 2];
  }
}

}  // namespace

extern "C" int nh_pion_kelner06(nh_ctx* c, int kind, const double* params, int N,
                                const double* E_eV, int nE, double Etrans_eV, int mixed,
                                double* out, int ldo, double* nhat_out, double* wp_TeV_out) {
  NH_REQUIRE(c && params && E_eV && out, "NULL pointer");
  NH_REQUIRE(kind >= NH_PD_POWERLAW && kind <= NH_PD_LOGPARABOLA, "unknown particle distribution kind");
  NH_REQUIRE(N >= 0 && nE >= 1 && ldo >= nE && Etrans_eV > 0.0, "bad sizes");
  NH_REQUIRE((size_t)(nE + 3) * sizeof(double) <= 60 * 1024, "too many photon energies per call");
  if (N == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_TABLES);
  hipLaunchKernelGGL(k_pion_kelner06, dim3((unsigned)N), dim3(256), (nE + 3) * sizeof(double),
                     c->stream, kind, params, N, E_eV, nE, Etrans_eV * 1e-12, mixed, out, ldo,
                     nhat_out, wp_TeV_out);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}
