// nh_ic.h -- the thermal inverse-Compton kernel of Khangulyan+14 (radiative.py:547-607,
// G12/G34 345-367) and the Aharonian-Atoyan kernel on a monochromatic / tabulated seed
// (radiative.py:609-655) as device functions: shared by the table builder (nh_tables.hip) and
// the general per-walker-grid kernel (nh_general.hip)
#pragma once
#include "nh_common.h"

__device__ __forceinline__ double ic_g(double x, double al, double a, double be, double b) {
  return 1.0 / (a * pow(x, al) / (1.0 + b * pow(x, be)) + 1.0);
}
__device__ __forceinline__ double ic_G34(double x, double al, double a, double be, double b,
                                         double cc) {
  double G = NH_PI26 * ((1.0 + cc * x) / (1.0 + NH_PI26 * cc * x)) * exp(-x);
  return G * ic_g(x, al, a, be, b);
}
__device__ __forceinline__ double ic_G12(double x, double al, double a, double be, double b) {
  double G = (NH_PI26 + x) * exp(-x);
  return G * ic_g(x, al, a, be, b);
}


// sigma(gamma, E_gamma) of Eq. 14 (isotropic, theta < 0) / Eq. 11 (anisotropic) times the
// (T'/gamma)^2 prefactor; eg = E_gamma / mec2, Tp = T in mec2 (radiative.py:557-574, 586-607)
__device__ __forceinline__ double ic_planck_K(double g, double eg, double Tp, double theta) {
  const double z = eg / g;
  double cs;
  if (theta < 0.0) {
    double x = z / (1.0 - z) / (4.0 * g * Tp);
    cs = z * z / (2.0 * (1.0 - z)) * ic_G34(x, 0.606, 0.443, 1.481, 0.540, 0.319) +
         ic_G34(x, 0.461, 0.726, 1.457, 0.382, 6.620);
  } else {
    double tt = 2.0 * g * Tp * (1.0 - cos(theta));
    double x = z / (1.0 - z) / tt;
    cs = z * z / (2.0 * (1.0 - z)) * ic_G12(x, 0.857, 0.153, 1.840, 0.254) +
         ic_G12(x, 0.691, 1.330, 1.668, 0.534);
  }
  double pref = (Tp / g) * (Tp / g);
  pref *= NH_IC_PLANCK_NORM;
  const bool ok = (eg < g) && (g > 1.0);
  return ok ? pref * cs : 0.0;
}

// ---------------------------------------------------------------------------
// row 8: Aharonian & Atoyan 81 Eq. 22 (radiative.py:609-655)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double ic_fic_windowed(double e0, double g, double eg) {
  double b = 4.0 * e0 * g;
  double wq = eg / g;
  double q = wq / (b * (1.0 - wq));
  double bq = b * q;
  double fic = 2.0 * q * log(q) + (1.0 + 2.0 * q) * (1.0 - q) +
               0.5 * (bq * bq) * (1.0 - q) / (1.0 + bq);
  double gi = fic * nh_heaviside(1.0 - q) * nh_heaviside(q - 1.0 / (4.0 * (g * g)));
  return (gi != gi) ? 0.0 : gi;  // gamint[isnan] = 0, radiative.py:636
}

// inner reduction over the seed spectrum for one (E_k, gamma_i): trapz_loglog of
// fic*n_ph/eps0 over eps0 (radiative.py:638-640), in the u/l form of nh_seg_term
__device__ __forceinline__ double ic_seed_inner(const double* __restrict__ se,
                                                const double* __restrict__ sd, int ns, double g,
                                                double eg) {
  if (ns == 1) {
    double e0 = se[0] / NH_MEC2_EV;
    double dens = sd[0] / NH_MEC2_EV;  // eV/cm3 -> mec2/cm3, radiative.py:642
    return ic_fic_windowed(e0, g, eg) * (dens / (e0 * e0));
  }
  double acc = 0.0;
  double e1 = se[0] / NH_MEC2_EV;
  double u1 = ic_fic_windowed(e1, g, eg) * (sd[0] * NH_MEC2_EV);  // y*x = fic*n_ph
  for (int s = 1; s < ns; ++s) {
    double e2 = se[s] / NH_MEC2_EV;
    double u2 = ic_fic_windowed(e2, g, eg) * (sd[s] * NH_MEC2_EV);
    acc += nh_seg_term(u1, u2, log(fabs(u2 / u1)), log(e2 / e1));
    e1 = e2; u1 = u2;
  }
  return acc;
}

