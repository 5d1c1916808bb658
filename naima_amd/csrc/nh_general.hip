// nh_general.hip -- the GENERAL electron path: every walker has its own particle grid.
//
// When Eemin / Eemax are fit parameters (the reference takes any keyword as per-call state,
// radiative.py:280, 430) the grid of radiative.py:147-154
//     gamma = logspace(log10(Eemin/mec2), log10(Eemax/mec2), max(10, int(nEed * decades)))
// differs from walker to walker -- in its limits AND in its number of nodes -- so nothing
// walker-independent can be tabulated: the emission kernel is evaluated at every (node,
// photon energy) of every walker.  One workgroup per (walker, component): it builds the
// walker's grid and weights in LDS (models.py eval + radiative.py:156-160), then integrates
//     Synchrotron._spectrum                       (radiative.py:282-342), or
//     InverseCompton on one thermal seed field    (radiative.py:547-607, 657-687), or
//     Bremsstrahlung's e-e / e-ion emissivities   (radiative.py:838-989)
// with trapz_loglog (utils.py:285-355).  The same holds when a seed field's temperature or
// angle is a fit parameter (the Khangulyan kernel then differs per walker even on a shared
// grid): T and theta are lazy per-walker scalars too.  The parameters stay in HBM, so the
// device-resident step loop no longer falls back to the host for such models.
#include "nh_brems.h"
#include "nh_ic.h"
#include "nh_pdist.h"
#include "nh_pion.h"
#include "nh_syn.h"

struct gen_args {
  int kind, N, what, nseed;
  const double* rows;   // [N][NH_PD_NPAR] particle-distribution rows
  nh_lazy emin, emax;   // per walker, in the unit the caller's Quantity carries ...
  double emin_erg, emax_erg;  // ... and that unit in erg (astropy's own factor)
  nh_lazy nEed;         // nodes per decade, per walker
  nh_lazy B;            // synchrotron: magnetic field [G]
  nh_lazy T[NH_MAX_COMP], theta[NH_MAX_COMP];  // IC: seed temperatures [K], angles [rad] (< 0: isotropic)
  const double* E_eV; int nE;
  double* out; int ldo;  // out[w*ldo + c*nE + k]
  int nmax;              // LDS capacity in nodes
  const double* seed_E; const double* seed_d; int ns;  // what = 4: a monochromatic / tabulated seed
  long long seed_ld;     // 0: every walker has the same densities; else walker w's are seed_d + w seed_ld (SSC)
  int* status;           // [0]: the largest node count asked for when it exceeds nmax;
                         // [1]: evaluations whose node count sat on an int() boundary
};

__device__ __forceinline__ double gen_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// term of one segment from the integrand at its two nodes: u = x y, dl = ln(u2/u1) formed as
// ln(w2/w1) + ln(K2/K1) (log of a RATIO of neighbouring nodes: no cancellation)
__device__ __forceinline__ double gen_term(double w1, double w2, double dlw, double K1, double K2,
                                           double lx) {
  const double u1 = w1 * K1, u2 = w2 * K2;
  if (u1 == 0.0 || u2 == 0.0) return 0.0;  // utils.py:347-348
  const double dl = dlw + log(K2 / K1);
  return nh_seg_term<true>(u1, u2, dl, lx);
}

__global__ __launch_bounds__(256) void k_general_electron(gen_args A) {
  extern __shared__ double sm[];
  double* gam = sm;               // [nmax]
  double* wv = sm + A.nmax;       // w = gamma * n(gamma) [1/mec2 -> per unit gamma]
  double* dw = sm + 2 * A.nmax;   // ln(w[i+1]/w[i])
  double* lxs = sm + 3 * A.nmax;  // ln(gamma[i+1]/gamma[i])
  double* part = sm + 4 * A.nmax;  // [4][64]
  __shared__ int s_n;
  __shared__ double s_l0, s_step, s_l1;
  const int wi = blockIdx.x, comp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wvi = tid >> 6;
  if (tid == 0) {
    // radiative.py:147-154 (the limits in units of mec2; int() truncates).  The reference's
    // expression, operation for operation: (value / mec2[erg]) * (the unit in erg) -- what
    // astropy reduces Eemin / mec2 to -- so that the limits are the host path's bit for bit.
    const double gmin = (nh_lazy_eval(A.emin, wi) / NH_MEC2_ERG_) * A.emin_erg;
    const double gmax = (nh_lazy_eval(A.emax, wi) / NH_MEC2_ERG_) * A.emax_erg;
    const double l0 = log10(gmin), l1 = log10(gmax);
    const double v = nh_lazy_eval(A.nEed, wi) * (l1 - l0);
    int n = (int)v;
    if (!(n >= 10)) n = 10;  // (also NaN limits)
    // int() of a float: numpy's log10 and this one may differ in the last place, which decides
    // the count only when v sits within rounding (~1e-13) of an integer.  Such evaluations are
    // counted -- the caller is told (Context.check_general) -- never silently different.
    if (v >= 10.0 && fabs(v - rint(v)) < 1e-9) atomicAdd(A.status + 1, 1);
    if (n > A.nmax) {
      atomicMax(A.status, n);
      n = 0;
    }
    s_n = n;
    s_l0 = l0;
    s_l1 = l1;
    s_step = n > 1 ? (l1 - l0) / (n - 1) : 0.0;
  }
  __syncthreads();
  const int n = s_n;
  const double* pr = A.rows + (long long)wi * NH_PD_NPAR;
  const pd_par p = {pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6]};
  double* orow = A.out + (long long)wi * A.ldo + (long long)comp * A.nE;
  if (n == 0) {  // grid too long for the workgroup's LDS: NaN, and the status word says so
    for (int k = tid; k < A.nE; k += blockDim.x) orow[k] = NAN;
    return;
  }
  // ---- the walker's grid: np.logspace = 10 ** (start + i step), last node exactly 10 ** stop
  for (int i = tid; i < n; i += blockDim.x)
    gam[i] = exp10(i + 1 < n ? s_l0 + i * s_step : s_l1);
  __syncthreads();
  // ---- weights (models.py eval on E = gamma mec2 in eV; nelec in 1/mec2: radiative.py:156-160)
  for (int i = tid; i < n; i += blockDim.x) {
    const bool last = i + 1 >= n;
    const double g = gam[i], g2 = last ? g : gam[i + 1];
    const double E = (g * NH_MEC2_ERG_) * NH_ERG_TO_EV_, E2 = (g2 * NH_MEC2_ERG_) * NH_ERG_TO_EV_;
    const double lr = last ? 0.0 : log(g2 / g);
    double nn, dsh;
    pd_node(A.kind, p, E, E2, last ? 0.0 : log(E2 / E), nn, dsh);
    nn *= NH_MEC2_EV;
    wv[i] = g * nn;
    dw[i] = last ? 0.0 : lr + dsh;
    lxs[i] = lr;
  }
  __syncthreads();
  const int nseg = n - 1;
  const int per = (nseg + 3) / 4;
  const int s0 = wvi * per, s1 = min(nseg, s0 + per);
  if (A.what == 2) {
    // ---- We = trapz_loglog(gamma nelec, gamma mec2) over the walker's own grid, erg --------
    // (radiative.py:162-195: u = x y = (gamma mec2)(gamma nelec); ln(K2/K1) = ln(g2/g1))
    double acc = 0.0;
    for (int sg = tid; sg < nseg; sg += blockDim.x) {
      const double u1 = wv[sg] * (gam[sg] * NH_MEC2_ERG_), u2 = wv[sg + 1] * (gam[sg + 1] * NH_MEC2_ERG_);
      acc += (u1 == 0.0 || u2 == 0.0) ? 0.0 : nh_seg_term<true>(u1, u2, dw[sg] + lxs[sg], lxs[sg]);
    }
    acc = gen_wave_sum(acc);
    if (lane == 0) part[wvi] = acc;
    __syncthreads();
    if (tid == 0) orow[0] = (part[0] + part[1]) + (part[2] + part[3]);
    return;
  }
  if (A.what == 0) {
    // ---- Synchrotron._spectrum -------------------------------------------------------
    const double Bw = nh_lazy_eval(A.B, wi);
    const double qfac = NH_ERG_PER_EV * (2.0 * (NH_M_E_G * NH_C_CGS)) /
                        (3.0 * NH_E_GAUSS * NH_HBAR_CGS * Bw);
    for (int k0 = 0; k0 < A.nE; k0 += 64) {
      const int k = k0 + lane;
      double acc = 0.0;
      if (k < A.nE && s0 < s1) {
        const double q = A.E_eV[k] * qfac;
        // Gtilde(x) = P(x) exp(-x); beyond x = 746 it is exactly 0 in double
        auto G = [&](double g) {
          const double x = q / (g * g);
          return x <= 746.0 ? syn_P(cbrt(x)) * nh_exp_neg(x) : 0.0;
        };
        double K1 = G(gam[s0]);
        for (int s = s0; s < s1; ++s) {
          const double K2 = G(gam[s + 1]);
          acc += gen_term(wv[s], wv[s + 1], dw[s], K1, K2, lxs[s]);
          K1 = K2;
        }
      }
      part[wvi * 64 + lane] = acc;
      __syncthreads();
      if (wvi == 0 && k < A.nE) {
        const double E_erg = A.E_eV[k] * NH_ERG_PER_EV;
        // CS1 = sqrt(3) e^3 B / (2 pi m_e c^2 hbar E), then 1/(s erg) -> 1/(s eV)
        const double cs1 = (1.7320508075688772 * (NH_E_GAUSS * NH_E_GAUSS * NH_E_GAUSS) * Bw) /
                           (2.0 * NH_PI * NH_M_E_G * (NH_C_CGS * NH_C_CGS) * NH_HBAR_CGS * E_erg);
        orow[k] = (part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane]) * cs1 *
                  NH_ERG_PER_EV;
      }
      __syncthreads();
    }
  } else if (A.what == 4) {
    // ---- InverseCompton on a monochromatic (ns = 1: energy density, eV/cm3) or tabulated
    // (1/(eV cm3) at ns energies) isotropic seed, the same for every walker: the inner
    // trapz_loglog over the seed's energies at every (node, photon energy), radiative.py:609-655;
    // the caller applies Eph / E (radiative.py:684-687) ----------------------------------------
    for (int k0 = 0; k0 < A.nE; k0 += 64) {
      const int k = k0 + lane;
      double acc = 0.0;
      if (k < A.nE && s0 < s1) {
        const double eg = A.E_eV[k] / NH_MEC2_EV;
        auto Ks = [&](double g) {
          return ic_seed_inner(A.seed_E, A.seed_d + (long long)wi * A.seed_ld, A.ns, g, eg) *
                 ((3.0 / 4.0) * NH_SIGT_LIT * NH_C_CGS / (g * g));  // radiative.py:650-653
        };
        double K1 = Ks(gam[s0]);
        for (int s = s0; s < s1; ++s) {
          const double K2 = Ks(gam[s + 1]);
          acc += gen_term(wv[s], wv[s + 1], dw[s], K1, K2, lxs[s]);
          K1 = K2;
        }
      }
      part[wvi * 64 + lane] = acc;
      __syncthreads();
      if (wvi == 0 && k < A.nE)
        orow[k] = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
      __syncthreads();
    }
  } else if (A.what == 3) {
    // ---- Bremsstrahlung: comp 0 = int(n sigma_ee), comp 1 = int(n sigma_1), per eV (the caller
    // applies n0 c and the abundance weights, radiative.py:949-987; the Baring+99 fits go
    // negative near their edges: the signed segment term) ----------------------------------
    for (int k0 = 0; k0 < A.nE; k0 += 64) {
      const int k = k0 + lane;
      double acc = 0.0;
      if (k < A.nE && s0 < s1) {
        const double eps = A.E_eV[k] / NH_MEC2_EV;
        auto Kb = [&](double g) {
          double see, sep;
          br_sigma(g, eps, see, sep);
          return (comp == 0 ? see : sep) / NH_MEC2_EV;
        };
        double K1 = Kb(gam[s0]);
        for (int s = s0; s < s1; ++s) {
          const double K2 = Kb(gam[s + 1]);
          acc += gen_term(wv[s], wv[s + 1], dw[s], K1, K2, lxs[s]);
          K1 = K2;
        }
      }
      part[wvi * 64 + lane] = acc;
      __syncthreads();
      if (wvi == 0 && k < A.nE)
        orow[k] = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
      __syncthreads();
    }
  } else {
    // ---- InverseCompton on thermal seed `comp` (the caller applies uf * Eph / E) --------
    const double Tp = nh_lazy_eval(A.T[comp], wi) * NH_K_TO_MEC2;
    const double th = nh_lazy_eval(A.theta[comp], wi);
    for (int k0 = 0; k0 < A.nE; k0 += 64) {
      const int k = k0 + lane;
      double acc = 0.0;
      if (k < A.nE && s0 < s1) {
        const double eg = A.E_eV[k] / NH_MEC2_EV;
        double K1 = ic_planck_K(gam[s0], eg, Tp, th);
        for (int s = s0; s < s1; ++s) {
          const double K2 = ic_planck_K(gam[s + 1], eg, Tp, th);
          acc += gen_term(wv[s], wv[s + 1], dw[s], K1, K2, lxs[s]);
          K1 = K2;
        }
      }
      part[wvi * 64 + lane] = acc;
      __syncthreads();
      if (wvi == 0 && k < A.nE)
        orow[k] = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
      __syncthreads();
    }
  }
}

static int general_electron(nh_ctx* c, int kind, const double* rows, int N,
                            const nh_lazy* Eemin, double Eemin_unit_erg,
                            const nh_lazy* Eemax, double Eemax_unit_erg,
                            const nh_lazy* nEed, int what, const nh_lazy* B_G,
                            const nh_lazy* seed_T, const nh_lazy* seed_theta, int nseed,
                            const double* seed_E, const double* seed_d, int ns,
                            const double* E_eV, int nE, double* out, int ldo, int nmax,
                            int* status, long long seed_ld = 0) {
  NH_REQUIRE(c && rows && Eemin && Eemax && nEed && out && status, "NULL pointer");
  NH_REQUIRE(kind >= NH_PD_POWERLAW && kind <= NH_PD_LOGPARABOLA, "unknown particle distribution kind");
  NH_REQUIRE(N >= 0 && nmax >= 10 && Eemin_unit_erg > 0 && Eemax_unit_erg > 0, "bad sizes");
  NH_REQUIRE(what == 2 || (E_eV && nE >= 1), "photon energies missing");
  NH_REQUIRE(nEed->base || nEed->a > 0.0, "nEed must be positive");
  NH_REQUIRE(what == 0 ? (B_G != nullptr)
                       : (what == 2 || what == 3 || (what == 4 && seed_E && seed_d && ns >= 1) ||
                          (what == 1 && seed_T && seed_theta && nseed >= 1 && nseed <= NH_MAX_COMP)),
             "bad component");
  if (what == 2) nE = 1;
  const int ncomp = what == 1 ? nseed : (what == 3 ? 2 : 1);
  NH_REQUIRE(ldo >= ncomp * nE, "ldo too small");
  if (N == 0) return NH_OK;
  gen_args A;
  memset(&A, 0, sizeof(A));
  A.kind = kind; A.N = N; A.what = what; A.nseed = ncomp; A.rows = rows;
  A.emin = *Eemin; A.emax = *Eemax; A.nEed = *nEed;
  A.emin_erg = Eemin_unit_erg; A.emax_erg = Eemax_unit_erg;
  if (what == 0) A.B = *B_G;
  for (int s = 0; s < ncomp && what == 1; ++s) {
    NH_REQUIRE(seed_T[s].base || seed_T[s].a > 0.0, "seed temperature must be positive");
    A.T[s] = seed_T[s];
    A.theta[s] = seed_theta[s];
  }
  A.E_eV = E_eV; A.nE = nE; A.out = out; A.ldo = ldo; A.nmax = nmax; A.status = status;
  A.seed_E = seed_E; A.seed_d = seed_d; A.ns = ns; A.seed_ld = seed_ld;
  const size_t lds = ((size_t)4 * nmax + 256) * sizeof(double);
  NH_REQUIRE(lds <= 150 * 1024, "nmax does not fit in LDS (at most ~4700 nodes)");
  if (lds > 64 * 1024)
    NH_CHECK_HIP(hipFuncSetAttribute((const void*)k_general_electron,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  nh_prof_scope ps(c, what == 0 ? NH_K_SYNCHROTRON : NH_K_INTEGRATE);
  // (what == 2 reads no photon energies: any valid pointer)
  hipLaunchKernelGGL(k_general_electron, dim3((unsigned)N, (unsigned)ncomp), dim3(256), lds,
                     c->stream, A);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_general_electron(nh_ctx* c, int kind, const double* rows, int N,
                                   const nh_lazy* Eemin, double Eemin_unit_erg,
                                   const nh_lazy* Eemax, double Eemax_unit_erg,
                                   const nh_lazy* nEed, int what, const nh_lazy* B_G,
                                   const nh_lazy* seed_T, const nh_lazy* seed_theta, int nseed,
                                   const double* E_eV, int nE, double* out, int ldo, int nmax,
                                   int* status) {
  NH_REQUIRE(what >= 0 && what <= 3, "what = 0 .. 3");
  return general_electron(c, kind, rows, N, Eemin, Eemin_unit_erg, Eemax, Eemax_unit_erg, nEed, what,
                          B_G, seed_T, seed_theta, nseed, nullptr, nullptr, 0, E_eV, nE, out, ldo,
                          nmax, status);
}

// InverseCompton on ONE monochromatic or tabulated isotropic seed field that is the same for
// every walker, over every walker's own grid (k_general_electron, what = 4)
extern "C" int nh_general_electron_seed(nh_ctx* c, int kind, const double* rows, int N,
                                        const nh_lazy* Eemin, double Eemin_unit_erg,
                                        const nh_lazy* Eemax, double Eemax_unit_erg,
                                        const nh_lazy* nEed, const double* seed_E,
                                        const double* seed_dens, int ns, const double* E_eV, int nE,
                                        double* out, int ldo, int nmax, int* status) {
  return general_electron(c, kind, rows, N, Eemin, Eemin_unit_erg, Eemax, Eemax_unit_erg, nEed, 4,
                          nullptr, nullptr, nullptr, 0, seed_E, seed_dens, ns, E_eV, nE, out, ldo,
                          nmax, status);
}

// ... and the same with a photon density PER WALKER (seed_dens[w * seed_ld + s], seed_ld >= ns):
// the synchrotron-self-Compton seed of examples/CrabNebula_SynSSC.py:29-45 -- each walker's own
// synchrotron photons -- over each walker's own grid (InverseCompton takes any keyword per call,
// radiative.py:430; the inner integral is radiative.py:609-655)
extern "C" int nh_general_electron_seed_rows(nh_ctx* c, int kind, const double* rows, int N,
                                             const nh_lazy* Eemin, double Eemin_unit_erg,
                                             const nh_lazy* Eemax, double Eemax_unit_erg,
                                             const nh_lazy* nEed, const double* seed_E,
                                             const double* seed_dens, long long seed_ld, int ns,
                                             const double* E_eV, int nE, double* out, int ldo,
                                             int nmax, int* status) {
  NH_REQUIRE(seed_ld >= ns && ns >= 1, "seed_ld must be at least the number of seed energies");
  return general_electron(c, kind, rows, N, Eemin, Eemin_unit_erg, Eemax, Eemax_unit_erg, nEed, 4,
                          nullptr, nullptr, nullptr, 0, seed_E, seed_dens, ns, E_eV, nE, out, ldo,
                          nmax, status, seed_ld);
}


// ---------------------------------------------------------------------------------------------
// The GENERAL proton path: Epmin / Epmax / nEpd per walker (radiative.py:1002-1055, 1495-1536).
// One workgroup per walker builds the walker's proton grid
//     Ep = logspace(log10 Epmin, log10 Epmax, max(10, int(nEpd * log10(Epmax / Epmin))))   [GeV]
// (the spectrum's grid, radiative.py:1002-1009: the count from the logarithm of the RATIO; Wp
// between explicit limits, :1047-1053, from the DIFFERENCE of the logarithms -- `count_mode`)
// and its weights Ep J(Ep) in LDS, then integrates
//   what = 0: the Kafexhiu+14 cross section evaluated at every (node, photon energy)
//   what = 1: the look-up table's FITPACK spline at every (node, photon energy)
//   what = 2: Wp = trapz_loglog(Ep J, Ep), GeV
// with trapz_loglog (negative spline values and sign changes as utils.py:336-348 treats them).
// ---------------------------------------------------------------------------------------------
struct genp_args {
  int kind, N, what, count_mode;
  const double* rows;
  nh_lazy emin, emax;        // in the caller's unit ...
  double emin_GeV, emax_GeV; // ... and that unit in GeV (astropy's factor)
  nh_lazy nEpd;
  const double* E_eV; int nE;
  double* out; int ldo;
  int nmax; int* status;
  pp_model M, G4; int nuc;
  const double* tx; const double* ty; const double* cf; int ntx, nty;
};

__global__ __launch_bounds__(256) void k_general_proton(genp_args A) {
  extern __shared__ double sm[];
  double* Ep = sm;                // [nmax]
  double* wv = sm + A.nmax;       // Ep * J(Ep)
  double* dw = sm + 2 * A.nmax;   // ln(w[i+1]/w[i])
  double* lxs = sm + 3 * A.nmax;  // ln(Ep[i+1]/Ep[i])
  double* part = sm + 4 * A.nmax; // [4][64]
  __shared__ int s_n;
  __shared__ double s_l0, s_step, s_l1;
  const int wi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wvi = tid >> 6;
  if (tid == 0) {
    const double lo = nh_lazy_eval(A.emin, wi) * A.emin_GeV;
    const double hi = nh_lazy_eval(A.emax, wi) * A.emax_GeV;
    const double l0 = log10(lo), l1 = log10(hi);
    const double dec = A.count_mode == 0 ? log10(hi / lo) : (l1 - l0);
    const double v = nh_lazy_eval(A.nEpd, wi) * dec;
    int n = (int)v;
    if (!(n >= 10)) n = 10;
    if (v >= 10.0 && fabs(v - rint(v)) < 1e-9) atomicAdd(A.status + 1, 1);  // (see k_general_electron)
    if (n > A.nmax) {
      atomicMax(A.status, n);
      n = 0;
    }
    s_n = n;
    s_l0 = l0;
    s_l1 = l1;
    s_step = n > 1 ? (l1 - l0) / (n - 1) : 0.0;
  }
  __syncthreads();
  const int n = s_n;
  const double* pr = A.rows + (long long)wi * NH_PD_NPAR;
  const pd_par p = {pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6]};
  double* orow = A.out + (long long)wi * A.ldo;
  const int nout = A.what == 2 ? 1 : A.nE;
  if (n == 0) {
    for (int k = tid; k < nout; k += blockDim.x) orow[k] = NAN;
    return;
  }
  for (int i = tid; i < n; i += blockDim.x)
    Ep[i] = exp10(i + 1 < n ? s_l0 + i * s_step : s_l1);
  __syncthreads();
  // weights: J = n(Ep [eV]) per eV -> per GeV (radiative.py:1011-1015)
  for (int i = tid; i < n; i += blockDim.x) {
    const bool last = i + 1 >= n;
    const double e1 = Ep[i], e2 = last ? e1 : Ep[i + 1];
    const double E = e1 * 1e9, E2 = e2 * 1e9;
    const double lr = last ? 0.0 : log(e2 / e1);
    double nn, dsh;
    pd_node(A.kind, p, E, E2, last ? 0.0 : log(E2 / E), nn, dsh);
    nn *= 1e9;
    wv[i] = e1 * nn;
    dw[i] = last ? 0.0 : lr + dsh;
    lxs[i] = lr;
  }
  __syncthreads();
  const int nseg = n - 1;
  if (A.what == 2) {
    double acc = 0.0;
    for (int sg = tid; sg < nseg; sg += blockDim.x) {
      const double u1 = wv[sg] * Ep[sg], u2 = wv[sg + 1] * Ep[sg + 1];
      acc += (u1 == 0.0 || u2 == 0.0) ? 0.0 : nh_seg_term<true>(u1, u2, dw[sg] + lxs[sg], lxs[sg]);
    }
    acc = gen_wave_sum(acc);
    if (lane == 0) part[wvi] = acc;
    __syncthreads();
    if (tid == 0) orow[0] = (part[0] + part[1]) + (part[2] + part[3]);
    return;
  }
  const int per = (nseg + 3) / 4;
  const int s0 = wvi * per, s1 = min(nseg, s0 + per);
  for (int k0 = 0; k0 < A.nE; k0 += 64) {
    const int k = k0 + lane;
    double acc = 0.0;
    if (k < A.nE && s0 < s1) {
      const double Eg = A.E_eV[k] * 1e-9;
      auto K = [&](double ep) {
        return A.what == 0 ? pp_diffsigma(ep, Eg, A.M, A.G4.a, A.nuc)
                           : pp_lut_value(ep, Eg, A.tx, A.ntx, A.ty, A.nty, A.cf);
      };
      double K1 = K(Ep[s0]);
      for (int s = s0; s < s1; ++s) {
        const double K2 = K(Ep[s + 1]);
        acc += gen_term(wv[s], wv[s + 1], dw[s], K1, K2, lxs[s]);
        K1 = K2;
      }
    }
    part[wvi * 64 + lane] = acc;
    __syncthreads();
    if (wvi == 0 && k < A.nE)
      orow[k] = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
    __syncthreads();
  }
}

extern "C" int nh_general_proton(nh_ctx* c, int kind, const double* rows, int N,
                                 const nh_lazy* Epmin, double Epmin_unit_GeV,
                                 const nh_lazy* Epmax, double Epmax_unit_GeV,
                                 const nh_lazy* nEpd, int count_mode, int what, int hiE, int nuc,
                                 const double* tx, int ntx, const double* ty, int nty,
                                 const double* cf, const double* E_eV, int nE, double* out,
                                 int ldo, int nmax, int* status) {
  NH_REQUIRE(c && rows && Epmin && Epmax && nEpd && out && status, "NULL pointer");
  NH_REQUIRE(kind >= NH_PD_POWERLAW && kind <= NH_PD_LOGPARABOLA, "unknown particle distribution kind");
  NH_REQUIRE(N >= 0 && nmax >= 10 && Epmin_unit_GeV > 0 && Epmax_unit_GeV > 0, "bad sizes");
  NH_REQUIRE(what >= 0 && what <= 2 && (count_mode == 0 || count_mode == 1), "bad mode");
  NH_REQUIRE(what == 2 || (E_eV && nE >= 1 && ldo >= nE), "photon energies missing");
  NH_REQUIRE(what != 0 || (hiE >= NH_PP_GEANT4 && hiE <= NH_PP_QGSJET), "unknown hiEmodel");
  NH_REQUIRE(what != 1 || (tx && ty && cf && ntx >= 8 && nty >= 8), "spline missing");
  NH_REQUIRE(nEpd->base || nEpd->a > 0.0, "nEpd must be positive");
  if (N == 0) return NH_OK;
  genp_args A;
  memset(&A, 0, sizeof(A));
  A.kind = kind; A.N = N; A.what = what; A.count_mode = count_mode; A.rows = rows;
  A.emin = *Epmin; A.emax = *Epmax; A.emin_GeV = Epmin_unit_GeV; A.emax_GeV = Epmax_unit_GeV;
  A.nEpd = *nEpd;
  A.E_eV = E_eV; A.nE = what == 2 ? 1 : nE; A.out = out; A.ldo = ldo; A.nmax = nmax; A.status = status;
  if (what == 0) { A.M = pp_get_model(hiE); A.G4 = pp_get_model(NH_PP_GEANT4); A.nuc = nuc; }
  A.tx = tx; A.ty = ty; A.cf = cf; A.ntx = ntx; A.nty = nty;
  const size_t lds = ((size_t)4 * nmax + 256) * sizeof(double);
  NH_REQUIRE(lds <= 150 * 1024, "nmax does not fit in LDS (at most ~4700 nodes)");
  if (lds > 64 * 1024)
    NH_CHECK_HIP(hipFuncSetAttribute((const void*)k_general_proton,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  nh_prof_scope ps(c, NH_K_INTEGRATE);
  hipLaunchKernelGGL(k_general_proton, dim3((unsigned)N), dim3(256), lds, c->stream, A);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}
