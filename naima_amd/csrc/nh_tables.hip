// nh_tables.hip -- walker-independent emission kernels K(E_k, x_i) written as
// transposed tables Kt[i][k] (+ ln|Kt|) for nh_integrate_tables.
//
// In the reference these matrices are rebuilt for every walker because emcee
// calls lnprob one walker at a time (core.py:450-457); they depend only on the
// grids, the photon energies and the seed-field / target parameters, so a
// batched call builds them ONCE per call (n_E*n_x elements, microseconds) and
// spends the per-walker work in the reduction.  Nothing is cached across calls.
#include "nh_common.h"
#include <algorithm>
#include <vector>
#include "nh_brems.h"
#include "nh_ic.h"
#include "nh_pion.h"
#include "nh_syn.h"
#include "nh_lnprob.h"

#define NH_TAB_PROLOGUE                                                  \
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      \
  if (idx >= (long long)nG * nE) return;                                 \
  const int i = (int)(idx / nE), k = (int)(idx % nE);

#define NH_TAB_AT ((long long)i * ld + k)

// second pass of every builder: dlnKt[i][k] = ln|K[i+1][k] / K[i][k]| -- the
// log-ratio of ADJACENT nodes, accurate to ~2 ulp however large ln K itself is
__global__ __launch_bounds__(256) void k_table_dlog(const double* __restrict__ Kt, int nG, int nE,
                                                     int ld, double* __restrict__ dlnKt) {
  NH_TAB_PROLOGUE
  double v = 0.0;
  if (i + 1 < nG) {
    const double k1 = Kt[NH_TAB_AT], k2 = Kt[(long long)(i + 1) * ld + k];
    // a zero node ends the power-law segment (utils.py:347-348): marked for nh_seg_pos
    v = (k1 == 0.0 || k2 == 0.0) ? NH_DL_ZERO : log(fabs(k2 / k1));
  }
  dlnKt[NH_TAB_AT] = v;
}

static int table_dlog(nh_ctx* c, const double* Kt, int nG, int nE, int ld, double* dlnKt) {
  long long tot = (long long)nG * nE;
  hipLaunchKernelGGL(k_table_dlog, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                     Kt, nG, nE, ld, dlnKt);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// rows 6,7: Khangulyan+14 Eq. 14 / Eq. 11: nh_ic.h
__global__ __launch_bounds__(256) void k_table_ic_planck(const double* __restrict__ gam, int nG,
                                                          const double* __restrict__ E_eV,
                                                          int nE, double T_K, double theta,
                                                          double* __restrict__ Kt, int ld) {
  NH_TAB_PROLOGUE
  Kt[NH_TAB_AT] = ic_planck_K(gam[i], E_eV[k] / NH_MEC2_EV, T_K * NH_K_TO_MEC2, theta);
}

extern "C" int nh_table_ic_planck(nh_ctx* c, const double* gam, int nG, const double* E_eV,
                                  int nE, double T_K, double theta_rad, double* Kt,
                                  double* lnKt, int ld) {
  NH_REQUIRE(c && gam && E_eV && Kt && lnKt && nG >= 2 && nE >= 1, "bad argument");
  NH_REQUIRE(T_K > 0.0, "seed temperature must be positive");
  nh_prof_scope ps(c, NH_K_TABLES);
  long long tot = (long long)nG * nE;
  hipLaunchKernelGGL(k_table_ic_planck, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     c->stream, gam, nG, E_eV, nE, T_K, theta_rad, Kt, ld);
  NH_CHECK_HIP(hipGetLastError());
  return table_dlog(c, Kt, nG, nE, ld, lnKt);
}

// row 8: Aharonian & Atoyan 81 Eq. 22 (radiative.py:609-655): ic_fic_windowed / ic_seed_inner in nh_ic.h
__global__ __launch_bounds__(256) void k_table_ic_seed(const double* __restrict__ gam, int nG,
                                                        const double* __restrict__ E_eV, int nE,
                                                        const double* __restrict__ se,
                                                        const double* __restrict__ sd, int ns,
                                                        double* __restrict__ Kt, int ld) {
  NH_TAB_PROLOGUE
  const double g = gam[i];
  const double eg = E_eV[k] / NH_MEC2_EV;
  double gi = ic_seed_inner(se, sd, ns, g, eg);
  gi *= (3.0 / 4.0) * NH_SIGT_LIT * NH_C_CGS / (g * g);  // radiative.py:650-653
  Kt[NH_TAB_AT] = gi;
}

extern "C" int nh_table_ic_seed(nh_ctx* c, const double* gam, int nG, const double* E_eV, int nE,
                                const double* seed_E, const double* seed_dens, int ns,
                                double* Kt, double* lnKt, int ld) {
  NH_REQUIRE(c && gam && E_eV && seed_E && seed_dens && Kt && lnKt, "NULL pointer");
  NH_REQUIRE(nG >= 2 && nE >= 1 && ns >= 1, "bad sizes");
  nh_prof_scope ps(c, NH_K_TABLES);
  long long tot = (long long)nG * nE;
  hipLaunchKernelGGL(k_table_ic_seed, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     c->stream, gam, nG, E_eV, nE, seed_E, seed_dens, ns, Kt, ld);
  NH_CHECK_HIP(hipGetLastError());
  return table_dlog(c, Kt, nG, nE, ld, lnKt);
}

// ---------------------------------------------------------------------------
// SSC: the seed density depends on the walker (examples/CrabNebula_SynSSC.py:29-31), so
// the (n_s x n_E x n_gam) double reduction of radiative.py:609-655 + 684 cannot be
// tabulated.  What CAN be shared is the Aharonian-Atoyan kernel fic(eps0_s, gamma_i,
// E_k): it does not depend on the walker.  One wave = 64 photon energies (lanes) x W
// walkers (register-blocked) x one chunk of gamma; per (s, i, k) the kernel and the
// log-ratio ln|fic_{s+1}/fic_s| are evaluated ONCE and applied to the W walkers, whose
// seed densities n_w(eps0_s) (pre-scaled) and log-ratios are wave-uniform scalars.
// The gamma range is additionally split over gridDim.y; the partial sums are reduced
// deterministically by k_ssc_finish.
//
// The kernel is bound by instruction issue (no memory traffic to speak of), so the seed-axis
// step is written for its instruction count -- ~57 VALU instructions per node + 10 per
// walker against ~400 for 8 walkers in the first version, whose ISA showed where they went: a
// double-double library log per node (110), two scalar loads WITH their wait and ten address
// instructions per walker, exec-mask branches for the window's Heaviside factors:
//   * the walkers' densities and log-ratios are transposed by k_ssc_prep to [group][s][W]:
//     two 64-byte scalar loads per seed node fetch them for eight walkers of the wave (W = 16:
//     the second eight's are on their way while the first eight are worked on);
//   * the log-ratios are divided by ln(eps_{s+1}/eps_s) there, so that a segment term is
//     (u2 - u1) / dl' -- no multiplication by lx per walker (dl' = dl / lx; the series
//     threshold |dl| < 2^-10 becomes |dl'| < 2^-10 / lx, a scalar per seed node);
//   * ln fic by ssc_log (fdlibm's degree-7 atanh form, 30 instructions, <= 2 ulp: the
//     difference of the logarithms of neighbouring nodes is O(1));
//   * the window as ONE product (1 - q)(q - qmin) > 0 (qmin < 1: never both negative), the
//     reference's value 0.5 at an edge behind a wave-uniform branch nobody takes.
// ---------------------------------------------------------------------------
#define SSC_REC(W) (2 * (W) + 8)  // doubles per (group, seed node) record; W walkers per group
// Everything a wave needs at seed node s, as ONE record behind ONE pointer (three wide scalar
// loads at fixed offsets; separate arrays cost four address computations per node and the
// scalar registers to hold them):
//   [0, W)    n_w(eps0_s) of the group's W walkers, 1/(mec2 cm3)       (radiative.py:639)
//   [W, 2W)   ln(n_w(eps0_s)/n_w(eps0_{s-1})) / lx of the segment that ENDS at s
//   [2W, 2W+4)  lx = ln(eps0_s/eps0_{s-1}), 1/lx, 2^-10/lx (series threshold in units of
//               dl' = dl/lx), 0     -- record 0: 1/eps0_0, 2 ln eps0_0 in the first two
//   [2W+4, 2W+6)  1/eps0_{s+1}, 2 ln eps0_{s+1} (in mec2): the NEXT node's fic operands
// A walker whose seed density is zero at EVERY node has an SSC spectrum of exactly 0 (every
// segment of the inner trapz_loglog has a zero node: utils.py:347-348) -- and that is what the
// step loop hands over for every proposal its prior forbids (the half-step kernel integrates
// nothing for those: a third of cfg4's proposals while its walkers sit against the prior's
// bounds).  The walkers with something to integrate are packed into the first groups:
//   ord[0] = their number, ord[1 + p] = the walker at position p (the others behind them, in
//   order), ord[1 + N + w] = 1 where walker w is one of them.
// One workgroup: N is an ensemble's half at most.
__global__ __launch_bounds__(1024) void k_ssc_order(const double* __restrict__ sd, int N, int ns,
                                                    int* __restrict__ ord) {
  __shared__ int cnt[1024];
  __shared__ int nlive_s;
  const int t = threadIdx.x, T = blockDim.x, lane = t & 63;
  int* live = ord + 1 + N;
  // a wave per walker, its lanes across the seed nodes (one round trip per 64 nodes)
  // (four walkers' loads in flight per wave: the kernel is round trips and nothing else)
  const int nwv = T >> 6;
  for (int w = t >> 6; w < N; w += 4 * nwv) {
    bool any[4] = {false, false, false, false};
    for (int s = lane; s < ns; s += 64) {
      double v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = sd[(long long)min(w + q * nwv, N - 1) * ns + s];
#pragma unroll
      for (int q = 0; q < 4; ++q) any[q] = any[q] || v[q] != 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool some = __builtin_amdgcn_ballot_w64(any[q]) != 0ull;
      if (lane == 0 && w + q * nwv < N) live[w + q * nwv] = some ? 1 : 0;
    }
  }
  __syncthreads();  // (the flags are this workgroup's own stores)
  const int per = (N + T - 1) / T;  // consecutive walkers per thread: the order is kept
  const int w_lo = min(t * per, N), w_hi = min(w_lo + per, N);
  int mine = 0;
  for (int w = w_lo; w < w_hi; ++w) mine += live[w];
  cnt[t] = mine;
  __syncthreads();
  // exclusive scan of the counts: one wave, 1024 / 64 = 16 counts per lane at most
  if (t < 64) {
    const int chunk = (T + 63) / 64;
    int sum = 0;
    for (int q = 0; q < chunk; ++q) sum += t * chunk + q < T ? cnt[t * chunk + q] : 0;
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (t >= off) incl += v;
    }
    int run = incl - sum;
    for (int q = 0; q < chunk; ++q)
      if (t * chunk + q < T) { const int c = cnt[t * chunk + q]; cnt[t * chunk + q] = run; run += c; }
    if (t == 63) { nlive_s = incl; ord[0] = incl; }
  }
  __syncthreads();
  int pl = cnt[t];                       // walkers with a density before mine
  int pd = nlive_s + (w_lo - cnt[t]);    // ... and where my first one without goes
  for (int w = w_lo; w < w_hi; ++w) {
    if (live[w]) ord[1 + pl++] = w;
    else ord[1 + pd++] = w;
  }
}

template <int SSC_W>
__global__ void k_ssc_prep(const double* __restrict__ se, const double* __restrict__ sd, int N,
                           int ns, const int* __restrict__ ord, double* __restrict__ rec) {
  constexpr int REC = SSC_REC(SSC_W);
  static_assert((SSC_W & (SSC_W - 1)) == 0 && REC % 4 == 0, "record layout");
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = (N + SSC_W - 1) / SSC_W;
  if (idx >= (long long)groups * ns * REC) return;
  const int slot = (int)(idx % REC);
  const int s = (int)((idx / REC) % ns);
  const int grp = (int)(idx / ((long long)REC * ns));
  if (grp * SSC_W >= ord[0]) return;  // (a group without a walker to integrate: never read)
  double v = 0.0;
  if (slot < 2 * SSC_W) {
    const long long w = ord[1 + min(grp * SSC_W + (slot & (SSC_W - 1)), N - 1)];
    const double b = sd[w * ns + s];
    if (slot < SSC_W) {
      v = b * NH_MEC2_EV;
    } else if (s > 0) {  // a zero node ends the power-law segment: marked as in the tables
      const double a = sd[w * ns + s - 1];
      v = (a == 0.0 || b == 0.0) ? NH_DL_ZERO : log(fabs(b / a)) / log(se[s] / se[s - 1]);
      // (last mantissa bit SET, k_ssc_table's log-ratios have it CLEAR: the sum of the two that
      // k_ic_seed_walkers_tab divides by is then never exactly 0 -- see there)
      if (v != 0.0 && fabs(v) < INFINITY) v = __longlong_as_double(__double_as_longlong(v) | 1ll);
    }
  } else if (slot < 2 * SSC_W + 4) {
    const int q = slot - 2 * SSC_W;
    if (s > 0) {
      const double lx = log(se[s] / se[s - 1]);
      v = q == 0 ? lx : q == 1 ? 1.0 / lx : q == 2 ? NH_SEG_SMALL_POS / lx : 0.0;
    } else if (q == 0) {
      v = NH_MEC2_EV / se[0];
    } else if (q == 1) {
      v = 2.0 * log(se[0] / NH_MEC2_EV);
    }
  } else if (s + 1 < ns) {
    const int q = slot - 2 * SSC_W - 4;
    const double e = se[s + 1] / NH_MEC2_EV;
    v = q == 0 ? 1.0 / e : q == 1 ? 2.0 * log(e) : 0.0;
  }
  rec[idx] = v;
}

// fic of Eq. 22 along the seed axis for fixed (gamma, E_gamma): with w = E_gamma/gamma,
//   q = w / (4 eps0 gamma (1 - w)) = c1 / eps0,   b q = w / (1 - w) =: B  (no eps0 in it),
//   fic = 2 q ln q + (1 + 2 q)(1 - q) + (B^2 / (2 (1 + B))) (1 - q)
//       = q (2 ln c1 - 2 ln eps0) + (1 - q)(2 q + 1 + hB),
// so one seed node costs six instructions: the divisions and the logarithm of
// ic_fic_windowed are taken once per (gamma, E_gamma).  Windows and the NaN -> 0 rule as there.
struct ssc_gk { double c1, lnc1x2, ohB, hB, qmin; bool valid; };

__device__ __forceinline__ ssc_gk ssc_setup(double g, double eg) {
  ssc_gk r;
  const double wq = eg / g, omw = 1.0 - wq;
  r.valid = omw > 0.0 && wq > 0.0;  // else q <= 0 or infinite: log(q) NaN -> 0 (radiative.py:636)
  r.c1 = wq / (4.0 * g * omw);
  r.lnc1x2 = r.valid ? 2.0 * log(r.c1) : 0.0;
  const double B = wq / omw;
  r.hB = 0.5 * (B * B) / (1.0 + B);
  r.ohB = 1.0 + r.hB;
  r.qmin = 1.0 / (4.0 * (g * g));
  return r;
}

// (ie0 = 1 / eps0, le0x2 = 2 ln eps0)
__device__ __forceinline__ double ssc_fic(const ssc_gk& r, double ie0, double le0x2) {
  const double q = r.c1 * ie0;
  const double omq = 1.0 - q;
  const double f = fma(omq, fma(2.0, q, r.ohB), q * (r.lnc1x2 - le0x2));
  // heaviside(1 - q) heaviside(q - qmin), radiative.py:633-634: both factors 1 inside the
  // window; they cannot both be negative (qmin <= 1/4), so inside <=> their product > 0
  const double p = omq * (q - r.qmin);
  double fw = (r.valid && p > 0.0) ? f : 0.0;
  if (__builtin_amdgcn_ballot_w64(p == 0.0) != 0ull) {  // on an edge: heaviside(0) = 0.5
    asm volatile("" ::: "memory");
    const double win = nh_heaviside(omq) * nh_heaviside(q - r.qmin);
    fw = r.valid ? f * win : 0.0;
  }
  return fw;
}

// ln x for a normal x > 0: x = 2^e m, m in [sqrt(1/2), sqrt(2)), s = (m - 1)/(m + 1),
// ln m = 2 s + s R(s^2) with fdlibm's degree-7 R (|error| < 2^-58 before rounding); ~2 ulp.
// The library's log carries a double-double through all of it for the last half ulp: 110
// instructions against 30.
__device__ __forceinline__ double ssc_log(double x, double L0, double L1, double L2, double L3,
                                          double L4, double L5, double L6) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double s = (m - 1.0) * nh_rcp2f(m + 1.0);
  const double z = s * s;
  double R = fma(z, L0, L1);
  R = fma(R, z, L2);
  R = fma(R, z, L3);
  R = fma(R, z, L4);
  R = fma(R, z, L5);
  R = fma(R, z, L6);
  const double lnm = fma(s, R * z, s + s);
  const double ef = (double)e;
  return fma(ef, 6.93147180369123816490e-01, fma(ef, 1.90821492927058770002e-10, lnm));
}

typedef double ssc_d4 __attribute__((ext_vector_type(4)));
// Mapping: one wave = ONE photon energy x 64 consecutive nodes of the gamma grid (63
// segments; the tiles overlap by one node) x W walkers.  The first version put 64 photon
// energies on the lanes: a tile of cfg4's 261 energies spans five decades, so the kernel's
// window in eps0 (its lower edge moves with E/gamma^2) differed from lane to lane and a wave
// walked the union -- 65 % of the lane-slots it paid for were inside their window (counted on
// the host for cfg4's grids).  64 neighbouring gamma nodes span 0.64 decades: 93 %.  The
// outer trapz_loglog over gamma (radiative.py:684) then runs ACROSS the lanes: each lane
// fetches its right neighbour's inner integral with one shuffle, the 63 segment terms meet
// in a wave sum; the tiles' partial sums are added by k_ssc_finish in a fixed order.
#define SSC_TILE 63  // segments of the gamma grid per wave
template <int C, int W>
__global__ __launch_bounds__(64 * C) void k_ic_seed_walkers(
    const double* __restrict__ w, const double* __restrict__ dlw, int N,
    const double* __restrict__ gam, const double* __restrict__ lx, int nG,
    const double* __restrict__ E_eV, int nE, const double* __restrict__ rec, int ns, int ntile,
    const int* __restrict__ ord, double* __restrict__ partial) {
  constexpr int REC = SSC_REC(W);
  const int lane = threadIdx.x & 63;
  const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x % ntile, grp = blockIdx.x / ntile;
  const int k = blockIdx.y * C + ch;  // this wave's photon energy (wave-uniform)
  if (k >= nE || grp * W >= ord[0]) return;  // (ord: k_ssc_order)
  const double eg = E_eV[k] / NH_MEC2_EV;
  const int w0 = grp * W;
  const int i = tile * SSC_TILE + lane;  // this lane's node
  const bool node = i < nG;
  const bool seg = lane < SSC_TILE && i + 1 < nG;  // ... and the segment that starts there
  const int ic = node ? i : nG - 1;
  const double g = gam[ic];
  const double* __restrict__ rg = rec + (size_t)grp * ns * REC;  // this group's records
  // the coefficients of ssc_log in scalar registers: v_fma_f64 takes them as they are (from
  // vector registers the compiler copies each one into the destination of a v_fmac first)
  double L0 = 1.479819860511658591e-01, L1 = 1.531383769920937332e-01;
  double L2 = 1.818357216161805012e-01, L3 = 2.222219843214978396e-01;
  double L4 = 2.857142874366239149e-01, L5 = 3.999999999940941908e-01;
  double L6 = 6.666666666666735130e-01;
  asm volatile("" : "+s"(L0), "+s"(L1), "+s"(L2), "+s"(L3), "+s"(L4), "+s"(L5), "+s"(L6));
#define SSC_LOG(x) ssc_log(x, L0, L1, L2, L3, L4, L5, L6)
  // inner reduction over the seed spectrum for W walkers at once
  double in[W], u1[W];
  ssc_gk gk = ssc_setup(g, eg);
  gk.valid = gk.valid && node;
  double f1 = ssc_fic(gk, rg[2 * W], rg[2 * W + 1]);
  // (the logarithm on every lane, a zero patched afterwards: behind a select the compiler
  // puts it in a branch of its own, and the scalar loads of the step behind that branch)
  double lf1 = SSC_LOG(fabs(f1)) + (f1 == 0.0 ? -INFINITY : 0.0);
#pragma unroll
  for (int j = 0; j < W; ++j) { in[j] = 0.0; u1[j] = f1 * rg[j]; }
  const double* rp = rg;
  double ie = rp[2 * W + 4], le = rp[2 * W + 5];  // the fic operands, one step ahead
  for (int s = 1; s < ns; ++s) {
    rp += REC;
    const double ien = rp[2 * W + 4], len = rp[2 * W + 5];
    const double f2 = ssc_fic(gk, ie, le);
    ie = ien;
    le = len;
    // both zero for a whole wave (outside every lane's window): nothing to add
    if (__builtin_amdgcn_ballot_w64(f1 != 0.0 || f2 != 0.0) != 0ull) {
      double sd8[W], dl8[W];
#pragma unroll
      for (int j = 0; j < W; ++j) {
        sd8[j] = rp[j];
        dl8[j] = rp[W + j];
      }
      const ssc_d4 Lv = *reinterpret_cast<const ssc_d4*>(rp + 2 * W);  // (one load, one wait)
      const double lxv = Lv.x, ilx = Lv.y, thr = Lv.z;
      const double lf2 = SSC_LOG(fabs(f2)) + (f2 == 0.0 ? -INFINITY : 0.0);
      // +-inf / NaN where a node is zero -> +-1e300: the reciprocal underflows to 0 and the
      // segment contributes (u2 - u1) 0 = 0 (utils.py:347-348) without a test per walker
      const double dlf = fmin(fmax(lf2 - lf1, -NH_DL_ZERO), NH_DL_ZERO) * ilx;
      lf1 = lf2;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const double uo = u1[j], io = in[j];
        const double u2 = f2 * sd8[j];
        const double dl = dlf + dl8[j];
        double t = fma(u2 - uo, nh_rcp1f(dl), io);
        // |dl| < 2^-10 (rare: the segment at the peak of u): the series of nh_seg_pos
        const bool small = fabs(dl) < thr;
        if (__builtin_amdgcn_ballot_w64(small) != 0ull) {
          asm volatile("" ::: "memory");  // keep this a branch
          const double d = dl * lxv;
          double f = fma(d, 8.333333333333333e-03, 4.166666666666666e-02);
          f = fma(f, d, 1.666666666666667e-01);
          f = fma(f, d, 0.5);
          f = fma(f, d, 1.0);
          t = small ? fma(uo * lxv, f, io) : t;
        }
        in[j] = t;
        u1[j] = u2;
      }
    } else {
      lf1 = -INFINITY;
#pragma unroll
      for (int j = 0; j < W; ++j) u1[j] = 0.0;
    }
    f1 = f2;
  }
  // outer segments (i, i+1) of trapz_loglog(nelec*gamint, gam), radiative.py:684: across lanes
  const double pref = (3.0 / 4.0) * NH_SIGT_LIT * NH_C_CGS / (g * g);  // radiative.py:650-653
  const double lxi = lx[seg ? i : 0];
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int wj = ord[1 + min(w0 + j, N - 1)];  // the walker at this position of the packed order
    const size_t row = (size_t)wj * (size_t)nG;
    const double Kv = in[j] * pref;
    const double Kn = __shfl_down(Kv, 1, 64);
    const double wi = w[row + ic], wn = w[row + (seg ? i + 1 : ic)];
    const double dl = dlw[row + (seg ? i : 0)] + SSC_LOG(fabs(Kn * nh_rcp(Kv)));
    double t = nh_seg_term(wi * Kv, wn * Kn, dl, lxi);
    t = nh_wave_sum(seg ? t : 0.0);
    if (lane == 0 && w0 + j < N) partial[((size_t)tile * N + wj) * nE + k] = t;
  }
#undef SSC_LOG
}

// ---------------------------------------------------------------------------
// The same reduction with the kernel TABULATED.  fic(eps0_s, gamma_i, E_k), its logarithmic
// slope along the seed axis and the window of seed nodes in which it is non-zero depend on
// the three grids only -- not on the walker, not on the step -- and 288 GB of HBM hold them
// for any fit: n_E x n_gamma x n_s x 16 bytes (cfg4: 261 x 14 tiles x 100 x 1 KB = 374 MB),
// built once per sampler by k_ssc_table with the very instruction sequence k_ic_seed_walkers
// runs per step (ssc_fic, ssc_log: the two kernels give the same bits).  The per-step kernel
// then is the walkers' arithmetic alone: per seed node ONE 16-byte load per lane (a wave reads
// 1 KB contiguous; the C waves of a workgroup are C walker groups at the same (E, gamma tile),
// so one of them misses and the others hit the CU's L1) instead of ~80 instructions of kernel
// + logarithm, and the loop runs over the tile's window [s_lo, s_hi) instead of testing every
// node for it (45 % of cfg4's (gamma, E, s) triples are outside).
//   table = int2 win[n_E * ntile] | int order[n_E * ntile] (padded to 256 B) |
//           double2 F[n_E * ntile][n_s][64]:
//   F[..][s][lane] = { fic at seed node s,  ln(fic_s / fic_{s-1}) / ln(eps_s / eps_{s-1}) }
//   order = the (E, gamma tile) pairs by decreasing window length: a workgroup's time is
//   proportional to it (0 ... n_s - 1 segments), and the longest ones dispatched first leave
//   the short ones to fill the end of the launch
// ---------------------------------------------------------------------------
typedef double ssc_d2 __attribute__((ext_vector_type(2)));

static inline size_t ssc_table_win_bytes(int nE, int ntile) {  // win + order
  return (((size_t)nE * ntile * (sizeof(int2) + sizeof(int))) + 255) & ~(size_t)255;
}

__global__ __launch_bounds__(512) void k_ssc_table(const double* __restrict__ gam, int nG,
                                                   const double* __restrict__ E_eV, int nE,
                                                   const double* __restrict__ se, int ns,
                                                   int ntile, ssc_d2* __restrict__ F,
                                                   int2* __restrict__ win) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x;
  const int k = blockIdx.y * 8 + (threadIdx.x >> 6);
  if (k >= nE) return;
  const double eg = E_eV[k] / NH_MEC2_EV;
  const int i = tile * SSC_TILE + lane;
  const bool node = i < nG;
  const double g = gam[node ? i : nG - 1];
  const double L0 = 1.479819860511658591e-01, L1 = 1.531383769920937332e-01;
  const double L2 = 1.818357216161805012e-01, L3 = 2.222219843214978396e-01;
  const double L4 = 2.857142874366239149e-01, L5 = 3.999999999940941908e-01;
  const double L6 = 6.666666666666735130e-01;
  ssc_gk gk = ssc_setup(g, eg);
  gk.valid = gk.valid && node;
  ssc_d2* __restrict__ Fp = F + ((size_t)(k * ntile + tile) * ns) * 64 + lane;
  // (the operands as k_ssc_prep writes them for k_ic_seed_walkers)
  double f1 = ssc_fic(gk, NH_MEC2_EV / se[0], 2.0 * log(se[0] / NH_MEC2_EV));
  double lf1 = ssc_log(fabs(f1), L0, L1, L2, L3, L4, L5, L6) + (f1 == 0.0 ? -INFINITY : 0.0);
  ssc_d2 e0 = {f1, 0.0};
  Fp[0] = e0;
  int s_lo = ns, s_hi = 0;
  for (int s = 1; s < ns; ++s) {
    const double e = se[s] / NH_MEC2_EV;
    const double f2 = ssc_fic(gk, 1.0 / e, 2.0 * log(e));
    const double lf2 = ssc_log(fabs(f2), L0, L1, L2, L3, L4, L5, L6) + (f2 == 0.0 ? -INFINITY : 0.0);
    const double ilx = 1.0 / log(se[s] / se[s - 1]);
    // The log-ratio with its last mantissa bit CLEAR, and 2^-60 for an exact 0 (two equal
    // entries: the difference of the integrand's nodes is then 0 as well): with the walkers'
    // log-ratios' last bit SET (k_ssc_prep) the sum of the two cannot cancel to exactly 0, and
    // k_ic_seed_walkers_tab, which adds (u2 - u1) / dl to its sum BEFORE it looks at dl, never
    // adds a NaN.
    double dq = fmin(fmax(lf2 - lf1, -NH_DL_ZERO), NH_DL_ZERO) * ilx;
    dq = dq == 0.0 ? 0x1p-60 : __longlong_as_double(__double_as_longlong(dq) & ~1ll);
    ssc_d2 v = {f2, dq};
    Fp[(size_t)s * 64] = v;
    if (__builtin_amdgcn_ballot_w64(f1 != 0.0 || f2 != 0.0) != 0ull) {
      s_lo = min(s_lo, s);
      s_hi = s + 1;
    }
    f1 = f2;
    lf1 = lf2;
  }
  if (lane == 0) win[k * ntile + tile] = make_int2(s_lo, s_hi);
}

// blockIdx.x -> (tile of (E, gamma), chunk of C walker groups): consecutive workgroups go to
// different XCDs, so the chunks of one tile sit 8 apart -- one XCD's L2 serves all of them
template <int C, int W>
__global__ __launch_bounds__(64 * C) void k_ic_seed_walkers_tab(
    const double* __restrict__ w, const double* __restrict__ dlw, int N,
    const double* __restrict__ gam, const double* __restrict__ lx, int nG, int nE,
    const double* __restrict__ rec, int ns, int ntile, int groups, int ngchunk,
    const ssc_d2* __restrict__ F, const int2* __restrict__ win, const int* __restrict__ order,
    const int* __restrict__ ord, double* __restrict__ partial) {
  constexpr int REC = SSC_REC(W);
  const int lane = threadIdx.x & 63;
  const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bx = blockIdx.x & 7, by = blockIdx.x >> 3;
  const int slot = (by / ngchunk) * 8 + bx;
  const int grp = (by % ngchunk) * C + ch;
  // (ord: k_ssc_order -- the groups behind the last walker with a seed density have nothing to do)
  if (slot >= nE * ntile || grp >= groups || grp * W >= ord[0]) return;
  const int tk = __builtin_amdgcn_readfirstlane(order[slot]);
  const int k = tk / ntile, tile = tk - k * ntile;
  const int w0 = grp * W;
  const int i = tile * SSC_TILE + lane;  // this lane's node
  const bool node = i < nG;
  const bool seg = lane < SSC_TILE && i + 1 < nG;  // ... and the segment that starts there
  const int ic = node ? i : nG - 1;
  const double g = gam[ic];
  const int2 wn = win[tk];
  const int s_lo = __builtin_amdgcn_readfirstlane(wn.x);
  const int s_hi = __builtin_amdgcn_readfirstlane(wn.y);
  double in[W], u1[W];
#pragma unroll
  for (int j = 0; j < W; ++j) in[j] = 0.0;
  if (s_lo < s_hi) {
    const ssc_d2* __restrict__ Fp = F + (size_t)tk * ns * 64 + lane;
    const double* rp = rec + ((size_t)grp * ns + (s_lo - 1)) * REC;
    const double f0 = Fp[(size_t)(s_lo - 1) * 64].x;
#pragma unroll
    for (int j = 0; j < W; ++j) u1[j] = f0 * rp[j];
    ssc_d2 nx = Fp[(size_t)s_lo * 64];
    for (int s = s_lo; s < s_hi; ++s) {
      rp += REC;
      const ssc_d2 cur = nx;
      nx = Fp[(size_t)min(s + 1, ns - 1) * 64];  // one node ahead
      double sd8[W], dl8[W];
#pragma unroll
      for (int j = 0; j < W; ++j) {
        sd8[j] = rp[j];
        dl8[j] = rp[W + j];
      }
      const ssc_d4 Lv = *reinterpret_cast<const ssc_d4*>(rp + 2 * W);
      const double lxv = Lv.x, thr = Lv.z;
      const double f2 = cur.x, dlf = cur.y;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const double uo = u1[j];
        const double u2 = f2 * sd8[j];
        const double dl = dlf + dl8[j];
        const double diff = u2 - uo, r = nh_rcp1f(dl);
        // (accumulated IN PLACE: with the sum's old value kept for the rare branch below, the
        // compiler computed into a temporary and paid a 64-bit move per walker and node to put
        // the result back where the loop carries it -- one instruction of eleven)
        in[j] = fma(diff, r, in[j]);
        // |dl| < 2^-10 (rare: the segment at the peak of u): the series of nh_seg_pos instead of
        // the term just added (taken off again: the sum moves by an ulp there; dl is never an
        // exact 0, whose reciprocal would have made the sum a NaN: k_ssc_table)
        const bool small = fabs(dl) < thr;
        if (__builtin_amdgcn_ballot_w64(small) != 0ull) {
          asm volatile("" ::: "memory");  // keep this a branch
          const double d = dl * lxv;
          double f = fma(d, 8.333333333333333e-03, 4.166666666666666e-02);
          f = fma(f, d, 1.666666666666667e-01);
          f = fma(f, d, 0.5);
          f = fma(f, d, 1.0);
          in[j] = small ? fma(uo * lxv, f, fma(-diff, r, in[j])) : in[j];
        }
        u1[j] = u2;
      }
    }
  }
  double L0 = 1.479819860511658591e-01, L1 = 1.531383769920937332e-01;
  double L2 = 1.818357216161805012e-01, L3 = 2.222219843214978396e-01;
  double L4 = 2.857142874366239149e-01, L5 = 3.999999999940941908e-01;
  double L6 = 6.666666666666735130e-01;
  // outer segments (i, i+1) of trapz_loglog(nelec*gamint, gam), radiative.py:684: across lanes
  const double pref = (3.0 / 4.0) * NH_SIGT_LIT * NH_C_CGS / (g * g);  // radiative.py:650-653
  const double lxi = lx[seg ? i : 0];
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int wj = ord[1 + min(w0 + j, N - 1)];  // the walker at this position of the packed order
    const size_t row = (size_t)wj * (size_t)nG;
    const double Kv = in[j] * pref;
    const double Kn = __shfl_down(Kv, 1, 64);
    const double wi = w[row + ic], wnx = w[row + (seg ? i + 1 : ic)];
    const double dl = dlw[row + (seg ? i : 0)] +
                      ssc_log(fabs(Kn * nh_rcp(Kv)), L0, L1, L2, L3, L4, L5, L6);
    double t = nh_seg_term(wi * Kv, wnx * Kn, dl, lxi);
    t = nh_wave_sum(seg ? t : 0.0);
    if (lane == 0 && w0 + j < N) partial[((size_t)tile * N + wj) * nE + k] = t;
  }
}

__global__ void k_ssc_finish(const double* __restrict__ partial, int nsuper, int N, int nE,
                             const double* __restrict__ E_eV, const int* __restrict__ ord,
                             double* __restrict__ out, int ldo) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)N * nE) return;
  int wi = (int)(idx / nE), k = (int)(idx % nE);
  double s = 0.0;
  // (a walker without a seed density: nobody wrote its partial sums -- an exact 0, see k_ssc_order.
  // One that shares a group with walkers that have one was integrated to the same 0.)
  if (ord[1 + N + wi])
    for (int y = 0; y < nsuper; ++y) s += partial[((long long)y * N + wi) * nE + k];
  const double E = E_eV[k];
  // lum = uf*Eph*integral, spec = lum/E   (uf = 1), radiative.py:676-687
  out[(long long)wi * ldo + k] = s * (E / NH_MEC2_EV) / E;
}

extern "C" int nh_ic_seed_walkers(nh_ctx* c, const double* w, const double* dlw, int N,
                                  const double* gam, const double* lx, int nG,
                                  const double* E_eV, int nE, const double* seed_E,
                                  const double* seed_dens, int ns, double* out, int ldo) {
  NH_REQUIRE(c && w && dlw && gam && lx && E_eV && seed_E && seed_dens && out, "NULL pointer");
  NH_REQUIRE(N >= 0 && nG >= 2 && nE >= 1 && ns >= 2 && ldo >= nE, "bad sizes");
  NH_REQUIRE((long long)N * nG < (1LL << 31) && (long long)N * ns < (1LL << 31),
             "arrays too large for 32-bit element offsets");
  if (N == 0) return NH_OK;
  // W walkers share a wave's kernel evaluations: 16 where there are that many (the second
  // eight walkers' scalars are fetched while the first eight are worked on), else 8
  constexpr int C = 8;
  const int W = N > 8 ? 16 : 8;
  const int groups = (N + W - 1) / W;
  const int nsuper = (nG - 1 + SSC_TILE - 1) / SSC_TILE;  // tiles of the gamma grid
  const size_t nd = (size_t)groups * ns * SSC_REC(W);
  const size_t nord = ((size_t)(1 + 2 * (size_t)N) * sizeof(int) + 7) / 8;  // (in doubles)
  const size_t need = (nd + SSC_REC(W) + (size_t)nsuper * N * nE + nord) * sizeof(double);
  void* sc = nullptr;
  int rc = nh_scratch(c, need, &sc);
  if (rc) return rc;
  double* rec = static_cast<double*>(sc);
  double* partial = rec + nd + SSC_REC(W);
  int* ord = reinterpret_cast<int*>(partial + (size_t)nsuper * N * nE);
  nh_prof_scope ps(c, NH_K_SSC);
  const dim3 gp((unsigned)((nd + 255) / 256)), gk(groups * nsuper, (nE + C - 1) / C);
  hipLaunchKernelGGL(k_ssc_order, dim3(1), dim3(1024), 0,
                     c->stream, seed_dens, N, ns, ord);
  if (W == 16) {
    hipLaunchKernelGGL(k_ssc_prep<16>, gp, dim3(256), 0, c->stream, seed_E, seed_dens, N, ns, ord, rec);
    hipLaunchKernelGGL((k_ic_seed_walkers<C, 16>), gk, dim3(64 * C), 0, c->stream, w, dlw, N, gam,
                       lx, nG, E_eV, nE, rec, ns, nsuper, ord, partial);
  } else {
    hipLaunchKernelGGL(k_ssc_prep<8>, gp, dim3(256), 0, c->stream, seed_E, seed_dens, N, ns, ord, rec);
    hipLaunchKernelGGL((k_ic_seed_walkers<C, 8>), gk, dim3(64 * C), 0, c->stream, w, dlw, N, gam,
                       lx, nG, E_eV, nE, rec, ns, nsuper, ord, partial);
  }
  long long tot = (long long)N * nE;
  hipLaunchKernelGGL(k_ssc_finish, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                     partial, nsuper, N, nE, E_eV, ord, out, ldo);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" long long nh_ssc_table_bytes(int nG, int nE, int ns) {
  if (nG < 2 || nE < 1 || ns < 2) return 0;
  const int ntile = (nG - 1 + SSC_TILE - 1) / SSC_TILE;
  return (long long)(ssc_table_win_bytes(nE, ntile) +
                     (size_t)nE * ntile * ns * 64 * sizeof(ssc_d2));
}

extern "C" int nh_ssc_table(nh_ctx* c, const double* gam, int nG, const double* E_eV, int nE,
                            const double* seed_E, int ns, void* table) {
  NH_REQUIRE(c && gam && E_eV && seed_E && table, "NULL pointer");
  NH_REQUIRE(nG >= 2 && nE >= 1 && ns >= 2, "bad sizes");
  const int ntile = (nG - 1 + SSC_TILE - 1) / SSC_TILE;
  nh_prof_scope ps(c, NH_K_TABLES);
  int2* win = static_cast<int2*>(table);
  ssc_d2* F = reinterpret_cast<ssc_d2*>(static_cast<char*>(table) + ssc_table_win_bytes(nE, ntile));
  hipLaunchKernelGGL(k_ssc_table, dim3(ntile, (nE + 7) / 8), dim3(512), 0, c->stream, gam, nG,
                     E_eV, nE, seed_E, ns, ntile, F, win);
  NH_CHECK_HIP(hipGetLastError());
  // longest windows first (once per table: a round trip through the host is fine here)
  const int ntk = nE * ntile;
  std::vector<int2> hw((size_t)ntk);
  NH_CHECK_HIP(hipMemcpyAsync(hw.data(), win, (size_t)ntk * sizeof(int2), hipMemcpyDeviceToHost,
                              c->stream));
  NH_CHECK_HIP(hipStreamSynchronize(c->stream));
  std::vector<int> ord((size_t)ntk);
  for (int t = 0; t < ntk; ++t) ord[(size_t)t] = t;
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
    return hw[(size_t)a].y - hw[(size_t)a].x > hw[(size_t)b].y - hw[(size_t)b].x;
  });
  NH_CHECK_HIP(hipMemcpyAsync(win + ntk, ord.data(), (size_t)ntk * sizeof(int),
                              hipMemcpyHostToDevice, c->stream));
  NH_CHECK_HIP(hipStreamSynchronize(c->stream));
  return NH_OK;
}

extern "C" int nh_ic_seed_walkers_tab(nh_ctx* c, const double* w, const double* dlw, int N,
                                      const double* gam, const double* lx, int nG,
                                      const double* E_eV, int nE, const double* seed_E,
                                      const double* seed_dens, int ns, const void* table,
                                      double* out, int ldo) {
  NH_REQUIRE(c && w && dlw && gam && lx && E_eV && seed_E && seed_dens && table && out,
             "NULL pointer");
  NH_REQUIRE(N >= 0 && nG >= 2 && nE >= 1 && ns >= 2 && ldo >= nE, "bad sizes");
  NH_REQUIRE((long long)N * nG < (1LL << 31) && (long long)N * ns < (1LL << 31),
             "arrays too large for 32-bit element offsets");
  if (N == 0) return NH_OK;
  constexpr int C = 8;
  const int W = N > 8 ? 16 : 8;
  const int groups = (N + W - 1) / W;
  const int ngchunk = (groups + C - 1) / C;
  const int ntile = (nG - 1 + SSC_TILE - 1) / SSC_TILE;
  NH_REQUIRE((long long)nE * ntile < (1LL << 27), "too many (energy, tile) pairs");
  const size_t nd = (size_t)groups * ns * SSC_REC(W);
  const size_t nord = ((size_t)(1 + 2 * (size_t)N) * sizeof(int) + 7) / 8;  // (in doubles)
  const size_t need = (nd + SSC_REC(W) + (size_t)ntile * N * nE + nord) * sizeof(double);
  void* sc = nullptr;
  int rc = nh_scratch(c, need, &sc);
  if (rc) return rc;
  double* rec = static_cast<double*>(sc);
  double* partial = rec + nd + SSC_REC(W);
  int* ord = reinterpret_cast<int*>(partial + (size_t)ntile * N * nE);
  const int2* win = static_cast<const int2*>(table);
  const int* order = reinterpret_cast<const int*>(win + (size_t)nE * ntile);
  const ssc_d2* F = reinterpret_cast<const ssc_d2*>(static_cast<const char*>(table) +
                                                    ssc_table_win_bytes(nE, ntile));
  nh_prof_scope ps(c, NH_K_SSC);
  const dim3 gp((unsigned)((nd + 255) / 256));
  const unsigned tk8 = (unsigned)((nE * ntile + 7) / 8);
  const dim3 gk(tk8 * ngchunk * 8);
  hipLaunchKernelGGL(k_ssc_order, dim3(1), dim3(1024), 0,
                     c->stream, seed_dens, N, ns, ord);
  if (W == 16) {
    hipLaunchKernelGGL(k_ssc_prep<16>, gp, dim3(256), 0, c->stream, seed_E, seed_dens, N, ns, ord, rec);
    hipLaunchKernelGGL((k_ic_seed_walkers_tab<C, 16>), gk, dim3(64 * C), 0, c->stream, w, dlw, N,
                       gam, lx, nG, nE, rec, ns, ntile, groups, ngchunk, F, win, order, ord, partial);
  } else {
    hipLaunchKernelGGL(k_ssc_prep<8>, gp, dim3(256), 0, c->stream, seed_E, seed_dens, N, ns, ord, rec);
    hipLaunchKernelGGL((k_ic_seed_walkers_tab<C, 8>), gk, dim3(64 * C), 0, c->stream, w, dlw, N,
                       gam, lx, nG, nE, rec, ns, ntile, groups, ngchunk, F, win, order, ord, partial);
  }
  long long tot = (long long)N * nE;
  hipLaunchKernelGGL(k_ssc_finish, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                     partial, ntile, N, nE, E_eV, ord, out, ldo);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// ---------------------------------------------------------------------------
// row 10: Baring+99 bremsstrahlung cross sections (radiative.py:838-928), cm2/eV
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_table_brems(const double* __restrict__ gam, int nG,
                                                      const double* __restrict__ E_eV, int nE,
                                                      double* __restrict__ Kee,
                                                      double* __restrict__ Kep, int ld) {
  NH_TAB_PROLOGUE
  double see, s1;
  br_sigma(gam[i], E_eV[k] / NH_MEC2_EV, see, s1);
  Kee[NH_TAB_AT] = see / NH_MEC2_EV;
  Kep[NH_TAB_AT] = s1 / NH_MEC2_EV;
}

extern "C" int nh_table_brems(nh_ctx* c, const double* gam, int nG, const double* E_eV, int nE,
                              double* Kt_ee, double* lnKt_ee, double* Kt_ep, double* lnKt_ep, int ld) {
  NH_REQUIRE(c && gam && E_eV && Kt_ee && lnKt_ee && Kt_ep && lnKt_ep && nG >= 2 && nE >= 1,
             "bad argument");
  nh_prof_scope ps(c, NH_K_TABLES);
  long long tot = (long long)nG * nE;
  hipLaunchKernelGGL(k_table_brems, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                     gam, nG, E_eV, nE, Kt_ee, Kt_ep, ld);
  NH_CHECK_HIP(hipGetLastError());
  int rc = table_dlog(c, Kt_ee, nG, nE, ld, lnKt_ee);
  return rc ? rc : table_dlog(c, Kt_ep, nG, nE, ld, lnKt_ep);
}

// ---------------------------------------------------------------------------
// row 9: Kafexhiu+14 pp -> pi0 -> gamma differential cross section
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_table_pion_analytic(const double* __restrict__ Ep,
                                                              int nG,
                                                              const double* __restrict__ E_eV,
                                                              int nE, pp_model M, pp_model G4,
                                                              int nuc, double* __restrict__ Kt, int ld) {
  NH_TAB_PROLOGUE
  double ds = pp_diffsigma(Ep[i], E_eV[k] * 1e-9, M, G4.a, nuc);
  Kt[NH_TAB_AT] = ds;
}

extern "C" int nh_table_pion_analytic(nh_ctx* c, const double* Ep_GeV, int nG,
                                      const double* E_eV, int nE, int hiE, int nuc, double* Kt,
                                      double* lnKt, int ld) {
  NH_REQUIRE(c && Ep_GeV && E_eV && Kt && lnKt && nG >= 2 && nE >= 1, "bad argument");
  NH_REQUIRE(hiE >= NH_PP_GEANT4 && hiE <= NH_PP_QGSJET, "unknown hiEmodel");
  nh_prof_scope ps(c, NH_K_TABLES);
  long long tot = (long long)nG * nE;
  hipLaunchKernelGGL(k_table_pion_analytic, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     c->stream, Ep_GeV, nG, E_eV, nE, pp_get_model(hiE),
                     pp_get_model(NH_PP_GEANT4), nuc, Kt, ld);
  NH_CHECK_HIP(hipGetLastError());
  return table_dlog(c, Kt, nG, nE, ld, lnKt);
}

__global__ __launch_bounds__(256) void k_table_pion_lut(const double* __restrict__ Ep, int nG,
                                                         const double* __restrict__ E_eV, int nE,
                                                         const double* __restrict__ tx, int ntx,
                                                         const double* __restrict__ ty, int nty,
                                                         const double* __restrict__ cf,
                                                         double* __restrict__ Kt, int ld) {
  NH_TAB_PROLOGUE
  const double sp = pp_lut_value(Ep[i], E_eV[k] * 1e-9, tx, ntx, ty, nty, cf);
  Kt[NH_TAB_AT] = sp;
}

extern "C" int nh_table_pion_lut(nh_ctx* c, const double* Ep_GeV, int nG, const double* E_eV,
                                 int nE, const double* tx, int ntx, const double* ty, int nty,
                                 const double* cf, double* Kt, double* lnKt, int ld) {
  NH_REQUIRE(c && Ep_GeV && E_eV && tx && ty && cf && Kt && lnKt, "NULL pointer");
  NH_REQUIRE(nG >= 2 && nE >= 1 && ntx >= 8 && nty >= 8, "bad sizes");
  nh_prof_scope ps(c, NH_K_TABLES);
  long long tot = (long long)nG * nE;
  hipLaunchKernelGGL(k_table_pion_lut, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     c->stream, Ep_GeV, nG, E_eV, nE, tx, ntx, ty, nty, cf, Kt, ld);
  NH_CHECK_HIP(hipGetLastError());
  return table_dlog(c, Kt, nG, nE, ld, lnKt);
}
