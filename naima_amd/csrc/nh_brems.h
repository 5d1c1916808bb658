// nh_brems.h -- Baring+99 bremsstrahlung cross sections (radiative.py:838-928), shared by the
// table kernel (nh_tables.hip: one table per grid) and the general kernel (nh_general.hip: a
// grid per walker, the cross section evaluated at every node of every walker)
#pragma once
#include "nh_common.h"

__device__ __forceinline__ double br_sigma_1(double g, double eps) {
  double s1 = 4.0 * (NH_R0_CM * NH_R0_CM) * NH_ALPHA_FS / eps;
  double s2 = 1.0 + (1.0 / 3.0 - eps / g) * (1.0 - eps / g);
  double s3 = log(2.0 * g * (g - eps) / eps) - 0.5;
  if (g < eps) s3 = 0.0;
  return s1 * s2 * s3;
}

__device__ __forceinline__ double br_sigma_2(double g, double eps) {
  double s0 = (NH_R0_CM * NH_R0_CM) * NH_ALPHA_FS / (3.0 * eps);
  double e2 = eps * eps, e3 = e2 * eps;
  double v;
  if (eps <= 0.5) {
    v = 16.0 * (1.0 - eps + e2) * log(g / eps) + (-1.0 / e2 + 3.0 / eps - 4.0 - 4.0 * eps - 8.0 * e2) +
        (-2.0 * (1.0 - 2.0 * eps) * log(1.0 - 2.0 * eps)) *
            (1.0 / (4.0 * e3) - 1.0 / (2.0 * e2) + 3.0 / eps - 2.0 + 4.0 * eps);
  } else {
    v = (2.0 / eps) * ((4.0 - 1.0 / eps + 1.0 / (4.0 * e2)) * log(2.0 * g) +
                       (-2.0 + 2.0 / eps - 5.0 / (8.0 * e2)));
  }
  return s0 * v * nh_heaviside(g - eps);
}

__device__ __forceinline__ double br_F(double x, double g) {  // A6, A7
  double g2 = g * g;
  double beta = sqrt(1.0 - 1.0 / g2);
  double B = 1.0 + 0.5 * (g2 - 1.0);
  double Cc = 10.0 * x * g * beta * (2.0 + g * beta);
  Cc = Cc / (1.0 + x * x * (g2 - 1.0));
  double tmx = 2.0 - x;
  double F1 = (17.0 - 3.0 * x * x / (tmx * tmx) - Cc) * sqrt(1.0 - x);
  double F2 = 12.0 * tmx - 7.0 * x * x / tmx - 3.0 * (x * x) * (x * x) / (tmx * tmx * tmx);
  double F3 = log((1.0 + sqrt(1.0 - x)) / sqrt(x));
  return B * F1 + F2 * F3;
}

// electron-electron (radiative.py:873-880 relativistic A1 + A4, :898-908 non-relativistic A5,
// switching at 2 MeV, :914) and electron-ion (sigma_1) cross sections, cm2 per unit eps
__device__ __forceinline__ void br_sigma(double g, double eps, double& see, double& sep) {
  const double gtrans = 2e6 / NH_MEC2_EV;  // 2 MeV, radiative.py:914
  const double s1 = br_sigma_1(g, eps);
  if (g <= gtrans) {  // non-relativistic, A5 (radiative.py:898-908)
    double s0 = 4.0 * (NH_R0_CM * NH_R0_CM) * NH_ALPHA_FS / (15.0 * eps);
    double x = 4.0 * eps / (g * g - 1.0);
    see = s0 * br_F(x, g);
    if (eps >= 0.25 * (g * g - 1.0)) see = 0.0;
    if (g < 1.0) see = 0.0;
  } else {  // relativistic, A1 + A4 (radiative.py:873-880)
    double A = 1.0 - 8.0 / 3.0 * pow(g - 1.0, 0.2) / (g + 1.0) * pow(eps / g, 1.0 / 3.0);
    see = (s1 + br_sigma_2(g, eps)) * A;
  }
  sep = s1;
}
