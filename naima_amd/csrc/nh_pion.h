// nh_pion.h -- the Kafexhiu+14 pp -> pi0 -> gamma differential cross section
// (radiative.py:1179-1482) and the FITPACK bicubic spline evaluation behind the reference's
// LookupTable (radiative.py:1770-1797), as device functions: shared by the table kernels
// (nh_tables.hip) and the general per-walker-grid kernel (nh_general.hip)
#pragma once
#include "nh_common.h"

struct pp_model {
  double a[5];    // Table IV   (radiative.py:1179-1183)
  double f[4];    // Table V hi (radiative.py:1194-1197)
  double b[3];    // Table VII  (radiative.py:1202-1205)
  double etrans;  // radiative.py:1209
};

__device__ __forceinline__ double pp_sigma_inel(double Tp) {  // radiative.py:1215-1233
  double L = log(Tp / NH_T_TH_GEV);
  double s = 30.7 - 0.96 * L + 0.18 * (L * L);
  s *= pow(1.0 - pow(NH_T_TH_GEV / Tp, 1.9), 3.0);
  return s * 1e-27;
}

static __device__ double pp_sigma_pi_lo(double Tp) {  // radiative.py:1235-1266
  const double mp = NH_M_P_GEV, mpi = NH_M_PI_GEV, Mres = 1.1883, Gres = 0.2264;
  double s = 2.0 * mp * (Tp + 2.0 * mp);
  double gamma = sqrt(Mres * Mres * (Mres * Mres + Gres * Gres));
  double K = sqrt(8.0) * Mres * Gres * gamma;
  K = K / (NH_PI * sqrt(Mres * Mres + gamma));
  double fBW = mp * K;
  double d = (sqrt(s) - mp) * (sqrt(s) - mp) - Mres * Mres;
  fBW = fBW / (d * d + Mres * Mres * Gres * Gres);
  double t = s - mpi * mpi - 4.0 * mp * mp;
  double mu = sqrt(t * t - 16.0 * mpi * mpi * mp * mp);
  mu = mu / (2.0 * mpi * sqrt(s));
  double s1 = 7.66e-3 * pow(mu, 1.95) * (1.0 + mu + pow(mu, 5.0)) * pow(fBW, 1.86);
  double s2 = 5.7 / (1.0 + exp(-9.3 * (Tp - 1.4)));
  if (Tp < 0.56) s2 = 0.0;
  return (s1 + s2) * 1e-27;
}

__device__ __forceinline__ double pp_sigma_pi_mid(double Tp) {  // radiative.py:1268-1275
  double Qp = (Tp - NH_T_TH_GEV) / NH_M_P_GEV;
  return pp_sigma_inel(Tp) * (-6e-3 + 0.237 * Qp - 0.023 * (Qp * Qp));
}

__device__ __forceinline__ double pp_sigma_pi_hi(double Tp, const double* a) {  // 1277-1286
  double csip = (Tp - 3.0) / NH_M_P_GEV;
  double m1 = a[0] * pow(csip, a[3]) * (1.0 + exp(-a[1] * pow(csip, a[4])));
  double m2 = 1.0 - exp(-a[2] * pow(csip, 0.25));
  return pp_sigma_inel(Tp) * (m1 * m2);
}

__device__ __forceinline__ double pp_EpimaxLAB(double Tp) {  // radiative.py:1325-1336
  const double mp = NH_M_P_GEV, mpi = NH_M_PI_GEV;
  double s = 2.0 * mp * (Tp + 2.0 * mp);
  double EpiCM = (s - 4.0 * mp * mp + mpi * mpi) / (2.0 * sqrt(s));
  double PpiCM = sqrt(EpiCM * EpiCM - mpi * mpi);
  double gCM = (Tp + 2.0 * mp) / sqrt(s);
  double betaCM = sqrt(1.0 - 1.0 / (gCM * gCM));
  return gCM * (EpiCM + PpiCM * betaCM);
}

static __device__ double pp_diffsigma(double Ep, double Eg, const pp_model& M, const double* aG4,
                               int nuc) {
  const double mp = NH_M_P_GEV, mpi = NH_M_PI_GEV;
  const double Tp = Ep - mp;
  // --- sigma_pi, radiative.py:1288-1304
  double spi;
  if (Tp < 2.0) spi = pp_sigma_pi_lo(Tp);
  else if (Tp < 5.0) spi = pp_sigma_pi_mid(Tp);
  else if (Tp < M.etrans) spi = pp_sigma_pi_hi(Tp, aG4);
  else spi = pp_sigma_pi_hi(Tp, M.a);
  // --- Amax, radiative.py:1306-1367
  const double EpimaxLAB = pp_EpimaxLAB(Tp);
  double Amax;
  if (Tp < 1.0) {
    Amax = 5.9 * spi / EpimaxLAB;
  } else {
    double b1, b2, b3;
    if (Tp < 5.0) { b1 = 9.53; b2 = 0.52; b3 = 0.054; }
    else if (Tp < M.etrans) { b1 = 9.13; b2 = 0.35; b3 = 9.7e-3; }
    else { b1 = M.b[0]; b2 = M.b[1]; b3 = M.b[2]; }
    double th = Tp / mp;
    double lt = log(th);
    Amax = b1 * pow(th, -b2) * exp(b3 * (lt * lt)) * spi / mp;
  }
  // --- F(Tp, Egamma), radiative.py:1369-1438 (later ranges override earlier)
  double F = 0.0;
  {
    double lam, alp, bet, gm;
    bool inr = true;
    double q = (Tp - 1.0) / mp;
    double mu = 1.25 * pow(q, 1.25) * exp(-1.25 * q);
    if (Tp > M.etrans) { lam = M.f[0]; alp = M.f[1]; bet = M.f[2]; gm = M.f[3]; }
    else if (Tp > 20.0 && Tp <= 100.0) { lam = 3.0; alp = 0.5; bet = 4.2; gm = 1.0; }
    else if (Tp > 4.0 && Tp <= 20.0) { lam = 3.0; alp = 1.0; bet = 1.5 * mu + 4.95; gm = mu + 1.50; }
    else if (Tp > 1.0 && Tp <= 4.0) { lam = 3.0; alp = 1.0; bet = mu + 2.45; gm = mu + 1.45; }
    else if (Tp >= NH_T_TH_GEV && Tp <= 1.0) {
      lam = 1.0; alp = 1.0; bet = 3.29 - pow(Tp / mp, -1.5) / 5.0; gm = 0.0;
    } else { inr = false; lam = alp = bet = gm = 0.0; }
    if (inr) {
      double gpi = EpimaxLAB / mpi;  // radiative.py:1338-1345
      double bpi = sqrt(1.0 - 1.0 / (gpi * gpi));
      double Egmax = (mpi / 2.0) * gpi * (1.0 + bpi);
      double Yg = Eg + mpi * mpi / (4.0 * Eg);
      double Ygmax = Egmax + mpi * mpi / (4.0 * Egmax);
      double Xg = (Yg - mpi) / (Ygmax - mpi);
      if (Xg > 1.0) Xg = 1.0;
      double Cc = lam * mpi / Ygmax;
      F = pow(1.0 - pow(Xg, alp), bet);
      F = F / pow(1.0 + Xg / Cc, gm);
    }
  }
  double ds = Amax * F;
  if (nuc) {  // radiative.py:1455-1482
    const double sRpp = 10.0 * NH_PI * 1e-27;
    double sin_ = pp_sigma_inel(Tp);
    double f = sin_ / pp_sigma_inel(1e3);
    double G = 1.0 + log(f > 1.0 ? f : 1.0);
    double eps = (Tp > NH_T_TH_GEV) ? 1.37 + (0.29 + 0.1) * sRpp * G / sin_ : 0.0;
    if (Tp > NH_T_TH_GEV && Tp < 1.0) eps = 1.9141;
    ds *= eps;
  }
  return ds;
}


// --- LookupTable (radiative.py:1770-1797): FITPACK bispev for kx = ky = 3 ----
__device__ __forceinline__ int bspl_locate(const double* __restrict__ t, int n, double& x) {
  // fpbisp: clamp to [t[3], t[n-4]], then the knot interval t[l] <= x < t[l+1]
  double tb = t[3], te = t[n - 4];
  if (x < tb) x = tb;
  if (x > te) x = te;
  int lo = 3, hi = n - 5;  // largest l in [3, n-5] with t[l] <= x
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (t[mid] <= x) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ void bspl_basis(const double* __restrict__ t, int l, double x,
                                           double h[4]) {
  // fpbspl with k = 3 (de Boor / Cox recurrence)
  double hh[3];
  h[0] = 1.0;
  for (int j = 1; j <= 3; ++j) {
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.0;
    for (int i = 1; i <= j; ++i) {
      double tli = t[l + i], tlj = t[l + i - j];
      if (tli == tlj) { h[i] = 0.0; continue; }
      double f = hh[i - 1] / (tli - tlj);
      h[i - 1] = h[i - 1] + f * (tli - x);
      h[i] = f * (x - tlj);
    }
  }
}


// one value of the spline through 10**lut at (Ep [GeV], Egamma [GeV])
__device__ __forceinline__ double pp_lut_value(double Ep, double Eg, const double* __restrict__ tx,
                                               int ntx, const double* __restrict__ ty, int nty,
                                               const double* __restrict__ cf) {
  double x = log10(Ep);
  double y = log10(Eg);
  int lxk = bspl_locate(tx, ntx, x);
  int lyk = bspl_locate(ty, nty, y);
  double hx[4], hy[4];
  bspl_basis(tx, lxk, x, hx);
  bspl_basis(ty, lyk, y, hy);
  const int nky1 = nty - 4;
  double sp = 0.0;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b)
      sp = sp + cf[(long long)(lxk - 3 + a) * nky1 + (lyk - 3 + b)] * hx[a] * hy[b];
  return sp;
}

static inline pp_model pp_get_model(int m) {
  static const pp_model models[4] = {
      {{0.728, 0.596, 0.491, 0.2503, 0.117}, {3.0, 0.5, 4.9, 1.0}, {9.13, 0.35, 9.7e-3}, 100.0},
      {{0.652, 0.0016, 0.488, 0.1928, 0.483}, {3.5, 0.5, 4.0, 1.0}, {9.06, 0.3795, 0.01105}, 50.0},
      {{5.436, 0.254, 0.072, 0.075, 0.166}, {3.55, 0.5, 3.6, 1.0}, {10.77, 0.412, 0.01264}, 100.0},
      {{0.908, 0.0009, 6.089, 0.176, 0.448}, {3.55, 0.5, 4.5, 1.0}, {13.16, 0.4419, 0.01439}, 100.0}};
  return models[m];
}

