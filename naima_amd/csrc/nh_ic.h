// nh_ic.h -- the thermal inverse-Compton kernel of Khangulyan+14 (radiative.py:547-607,
// G12/G34 345-367) as one device function: shared by the table builder (nh_tables.hip) and
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
