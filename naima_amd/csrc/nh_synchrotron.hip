// nh_synchrotron.hip -- Synchrotron._spectrum (radiative.py:282-342) batched
// over walkers.  The emissivity kernel Gtilde(E/Ec(gamma,B)) depends on the
// walker through B, so nothing can be tabulated: every (walker, E_k, gamma_i)
// node needs Gtilde(x) = P(x) exp(-x) in FP64.  The kernel is FP64-VALU bound;
// HBM traffic is the (w, dlw) rows and the output only.
//
// Mapping: see k_synchrotron below (live-energy compaction per block).
//
// Instruction diet (the first version spent ~400 FP64 instructions per node in
// OCML cbrt/sqrt/div/exp/log):
//   cbrt(x)     = cbrt(q) * gamma_i^(-2/3): one cube root per THREAD, the grid part
//                 is tabulated in LDS together with 1/gamma^2 and its difference;
//   P(x)        = ONE reciprocal square root (single-precision v_rsq seed + two Newton steps)
//                 for 1/sqrt(1 + 3.4 cb^2) and 1/gt3 together (nh_syn.h: syn_P1);
//   exp(-x)     = 64-entry 2^(j/64) table in LDS + degree-5 polynomial + v_ldexp_f64;
//   ln(P2/P1)   = 2 atanh(s), s = (P2-P1)/(P2+P1), 3-term series on naima's default grid
//                 density, 5 terms up to s^2 = 9e-4, log() only on coarser grids.
// (the node arithmetic of the half-step kernel, nh_halfstep.hip; round 1's degree-13 Horner
// exp and separate rsqrt + reciprocal cost ~115 instructions per node, this ~75)
#include "nh_lnprob.h"
#include "nh_syn.h"
#include <cstdlib>

// One block = one walker x one tile of 64 photon energies.  Many (energy, gamma)
// pairs cannot contribute at all: exp(-E/Ec) is exactly 0 in double for x > 746, which
// for a given energy removes every node below gamma0 = sqrt(q/746) -- and removes
// TeV energies altogether (the reference computes, and trapz_loglog discards, exact
// zeros there).  The block therefore
//   1. finds, per energy, the first node i0 that can contribute (binary search in the
//      LDS copy of 1/gamma^2) and the number nA of live energies;
//   2. compacts the live energies and spreads each one's [i0 - 1, nseg) over its share of
//      ALL the block's threads:
//      thread t -> live energy t % nA, chunk t / nA (Cd = T / nA chunks);
//   3. reduces the Cd partial sums per energy in LDS; dead energies get 0.
constexpr int SYN_MAXCH = 64;  // chunks per energy (LDS: part[SYN_MAXCH][64])

// EPI: the block goes on to evaluate the likelihood (+ priors, + the stretch move's
// accept) of ITS walker -- nh_lnprob_wave -- with its own spectrum taken from LDS.  Only
// for one tile per walker (nE <= 64), when this launch is the last producer.
template <int C, bool EPI>
__global__ __launch_bounds__(64 * C) void k_synchrotron(
    const double* __restrict__ w, const double* __restrict__ dlw, const double* __restrict__ B,
    int ldB, int N, const double* __restrict__ gam, const double* __restrict__ lx, int nG,
    const double* __restrict__ E_eV, int nE, double* __restrict__ out, int ldo, int tw,
    nh_lnprob_args L, int syn_comp) {
  extern __shared__ double smem[];  // ig2[nG] | dig2[nG] | ig23[nG] | part[SYN_MAXCH][64]
  double* ig2 = smem;
  double* dig2 = smem + nG;
  double* ig23 = smem + 2 * nG;
  double* part = smem + 3 * nG;
  __shared__ int amap[64];   // compacted live energy -> energy index
  __shared__ int ai0[64];    //                       -> its first segment that can contribute
  __shared__ int s_nA;
  __shared__ double synv[64];  // EPI: this walker's spectrum, by energy index
  __shared__ double T64[64];   // 2^(j/64): the table of nh_exp_tab
  constexpr int T = 64 * C;
  const int tid = threadIdx.x;
  // EPI: the LAST wave (the first one is on the critical path of the liveness search)
  // asks for everything the likelihood needs besides this kernel's
  // own spectrum now; the answers wait in registers until the epilogue
  nh_lnprob_pre PRE = {};
  const bool epw = EPI && tid >= T - 64;  // the epilogue's wave
  const int el = tid - (T - 64);          // its lane = energy index
  if (epw) nh_lnprob64_prefetch_a(PRE, L, blockIdx.x, el, syn_comp);
  for (int i = tid; i < nG; i += T) {
    const double g = gam[i];
    const double v = 1.0 / (g * g);
    ig2[i] = v;
    ig23[i] = cbrt(v);
    double d = 0.0;
    if (i + 1 < nG) {
      const double r = g / gam[i + 1];  // 1/g2^2 - 1/g1^2 without cancellation
      d = v * (r * r - 1.0);
    }
    dig2[i] = d;
  }
  if (tid == 0) s_nA = 0;
  if (tid >= T - 64) T64[tid - (T - 64)] = exp2((double)(tid - (T - 64)) * 0.015625);
  __syncthreads();

  // a tile = tw (<= 64) photon energies, INTERLEAVED over the ktiles tiles (energy
  // k = lane*ktiles + tile): live and dead energies come in runs (X-ray points live, TeV
  // points dead), so neighbouring tiles get the same share of live ones
  const int ktiles = (nE + tw - 1) / tw;
  const int tile = blockIdx.x % ktiles, wi = blockIdx.x / ktiles;
  const double Bw = B[(long long)wi * ldB];
  // x = E/Ec,  Ec = 3 e hbar B gamma^2 / (2 m_e c)         radiative.py:331-334
  const double qfac = NH_ERG_PER_EV * (2.0 * (NH_M_E_G * NH_C_CGS)) /
                      (3.0 * NH_E_GAUSS * NH_HBAR_CGS * Bw);

  // ---- 1. liveness of the tile's energies (first wave) ----------------------
  if (epw) nh_lnprob64_prefetch_b(PRE, L, wi);
  if (tid < 64) {
    const int k = tid < tw ? tid * ktiles + tile : nE;
    int i0 = nG;
    if (k < nE) {
      const double q = E_eV[k] * qfac;
      // first i with q*ig2[i] <= 746 (ig2 decreases with i)
      int lo = 0, hi = nG;
      while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (q * ig2[mid] <= 746.0) hi = mid; else lo = mid + 1;
      }
      i0 = lo;
    }
    const bool live = i0 < nG;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
    if (live) {
      const int pos = __popcll(m & ((1ull << tid) - 1ull));
      amap[pos] = k;
      ai0[pos] = i0;  // (from the first LIVE node on: the segment to its left contributes an exact 0)
    }
    if (tid == 0) s_nA = __popcll(m);
    if (k < nE && !live) out[(long long)wi * ldo + k] = 0.0;
    if (EPI && k < nE && !live) synv[k] = 0.0;
  }
  __syncthreads();
  const int nA = s_nA;
  if (epw) nh_lnprob64_prefetch_c(PRE, L, el);
  if (nA == 0) {
    if (epw) nh_lnprob64_finish(L, PRE, wi, el, synv, syn_comp);
    return;
  }
  const int nseg = nG - 1;

  // ---- 2. (live energy, chunk) per thread ------------------------------------
  const int Cd = min(T / nA, SYN_MAXCH);
  const int a = tid % nA, ch = tid / nA;
  double acc = 0.0;
  if (ch < Cd) {
    // every energy spreads ITS live range over its Cd threads (the first live node moves
    // up with the photon energy: 65 nodes between 0.55 and 11 keV on the default grid)
    const int sbeg = ai0[a];
    const int per = (nseg - sbeg + Cd - 1) / Cd;
    const int s0 = sbeg + ch * per;
    const int s1 = min(nseg, s0 + per);
    const int k = amap[a];
    const double E_erg = E_eV[k] * NH_ERG_PER_EV;
    // CS1 = sqrt(3) e^3 B / (2 pi m_e c^2 hbar E)          radiative.py:319-328
    const double cs1 = (1.7320508075688772 * (NH_E_GAUSS * NH_E_GAUSS * NH_E_GAUSS) * Bw) /
                       (2.0 * NH_PI * NH_M_E_G * (NH_C_CGS * NH_C_CGS) * NH_HBAR_CGS * E_erg);
    const double q = E_eV[k] * qfac;
    const double cbq = cbrt(q);
    const double* wr = w + (long long)wi * nG;
    const double* dwr = dlw + (long long)wi * nG;
    if (s0 < s1) {
      // The node arithmetic of the half-step kernel (nh_syn.h: table-driven exp, one reciprocal
      // square root for P, 3-term atanh for ln(P2/P1) on naima's default grid density, the
      // non-negative segment form): ~69 instructions per node against ~115.  No branch and no
      // select: a range starts at its first live node (x <= 746), so no node of it is dead; a
      // zero weight gives u = 0 by itself; two nodes per trip are one straight block.
      auto node = [&](int sn, double& u, double& P) {
        const double x = q * ig2[sn];
        const double wn = wr[sn];
        const double Pv = syn_P1(cbq * ig23[sn]);
        const double ev = nh_exp_tab(-x, T64);  // (x <= 746 from the first live node on)
        u = wn * (Pv * ev);  // gamma nelec dNdE / CS1, :335-338 (every node of the range is live)
        P = Pv;
      };
      double u1, P1;
      node(s0, u1, P1);
      int s = s0;
      for (; s + 2 <= s1; s += 2) {
        double uA, PA, uB, PB;
        node(s + 1, uA, PA);
        node(s + 2, uB, PB);
        // ln|u2/u1| = ln(w2/w1) + ln(P2/P1) - (x2 - x1); unused when a node is 0
        const double dlA = dwr[s] + syn_dlnP1(P1, PA) - q * dig2[s];
        const double dlB = dwr[s + 1] + syn_dlnP1(PA, PB) - q * dig2[s + 1];
        acc += nh_seg_pos<false>(u1, uA, dlA, lx[s]);  // P(x) exp(-x) >= 0: one sign
        acc += nh_seg_pos<false>(uA, uB, dlB, lx[s + 1]);
        u1 = uB;
        P1 = PB;
      }
      if (s < s1) {
        double uA, PA;
        node(s + 1, uA, PA);
        const double dlA = dwr[s] + syn_dlnP1(P1, PA) - q * dig2[s];
        acc += nh_seg_pos<false>(u1, uA, dlA, lx[s]);
      }
    }
    part[ch * 64 + a] = acc * cs1;  // the terms are linear in u: CS1 once per thread
  }
  __syncthreads();
  // ---- 3. per-energy reduction -------------------------------------------------
  if (tid < nA) {
    double sum = 0.0;
    for (int j = 0; j < Cd; ++j) sum += part[j * 64 + tid];
    sum *= NH_ERG_PER_EV;  // 1/(s erg) -> 1/(s eV), :340
    out[(long long)wi * ldo + amap[tid]] = sum;
    if (EPI) synv[amap[tid]] = sum;
  }
  if (EPI) {
    __syncthreads();
    if (epw) nh_lnprob64_finish(L, PRE, wi, el, synv, syn_comp);
  }
}

static int launch_synchrotron(nh_ctx* c, const double* w, const double* dlw, const double* B_G,
                              int ldB, int N, const double* gam, const double* lx, int nG,
                              const double* E_eV, int nE, double* out, int ldo,
                              const nh_lnprob_args* L, int syn_comp) {
  NH_REQUIRE(c && w && dlw && B_G && gam && lx && E_eV && out, "NULL pointer");
  NH_REQUIRE(N >= 0 && nG >= 2 && nE >= 1 && ldo >= nE && ldB >= 1, "bad sizes");
  if (N == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_SYNCHROTRON);
  const int nseg = nG - 1;
  int C = nseg >= 256 ? 16 : (nseg >= 64 ? 8 : 4);
  // tile width: 64 measured best at every (C, tw) tried on cfg3 (C=16: 20.0 / 24.2 / 28.9 us
  // for tw = 64 / 32 / 22; C=8: 21.3 / 20.7 / 20.7) -- the kernel is bound by its total
  // instruction count, not by how the blocks are cut
  int tw = 64;
  // ... for one tile per walker.  Several tiles per walker (cfg4: 261 data + 100 seed
  // energies, 128 walkers) go faster as 512-thread workgroups of ~33 energies, two per CU, one's
  // set-up and reduction beside the other's nodes: 63.0 -> 53.3 us mean of cfg4's two launches
  // (C, tw = 8, 44: 58.3; 8, 22: 57.4; 16, 33: 61.6; 4, 64: 84.4)
  if (nE > 64 && !L) {
    C = min(C, 8);
    const int kt = (nE + 32) / 33;
    tw = (nE + kt - 1) / kt;
  }
  static const int ov_tw = nh_env_int("NH_SYN_TW", 0);
  if (ov_tw > 0) tw = ov_tw;
  if (L) tw = 64;
  NH_REQUIRE(tw >= 1 && tw <= 64, "bad tile width");
  const int ktiles = (nE + tw - 1) / tw;
  NH_REQUIRE(!L || ktiles == 1, "the likelihood epilogue needs all energies in one tile");
  const unsigned blocks = (unsigned)(ktiles * N);
  if ((long long)blocks * C > 16384 && C > 4) C /= 2;  // plenty of waves: longer chunks
  static const int ov_C = nh_env_int("NH_SYN_C", 0);
  if (ov_C > 0) C = ov_C;
  size_t shm = (size_t)(3 * nG + SYN_MAXCH * 64) * sizeof(double);
  NH_REQUIRE(shm <= 150 * 1024, "electron grid too long for the LDS staging");
  nh_lnprob_args none = {};
  const nh_lnprob_args& A = L ? *L : none;
#define NH_LAUNCH_SYN_E(CC, EE)                                                               \
  do {                                                                                        \
    if (shm > 64 * 1024)                                                                      \
      NH_CHECK_HIP(hipFuncSetAttribute((const void*)k_synchrotron<CC, EE>,                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
    hipLaunchKernelGGL((k_synchrotron<CC, EE>), dim3(blocks), dim3(64 * CC), shm, c->stream, w, \
                       dlw, B_G, ldB, N, gam, lx, nG, E_eV, nE, out, ldo, tw, A, syn_comp);   \
  } while (0)
#define NH_LAUNCH_SYN(CC) \
  do { if (L) NH_LAUNCH_SYN_E(CC, true); else NH_LAUNCH_SYN_E(CC, false); } while (0)
  switch (C) {
    case 16: NH_LAUNCH_SYN(16); break;
    case 8: NH_LAUNCH_SYN(8); break;
    default: NH_LAUNCH_SYN(4); break;
  }
#undef NH_LAUNCH_SYN
#undef NH_LAUNCH_SYN_E
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_synchrotron(nh_ctx* c, const double* w, const double* dlw, const double* B_G,
                              int ldB, int N, const double* gam, const double* lx, int nG,
                              const double* E_eV, int nE, double* out, int ldo) {
  return launch_synchrotron(c, w, dlw, B_G, ldB, N, gam, lx, nG, E_eV, nE, out, ldo, nullptr, -1);
}

extern "C" int nh_synchrotron_lnprob(nh_ctx* c, const double* w, const double* dlw,
                                     const double* B_G, int ldB, int N, const double* gam,
                                     const double* lx, int nG, const double* E_eV, int nE,
                                     double* out, int ldo, const nh_comp* comps, int ncomp,
                                     int syn_comp, const double* conv, const double* flux,
                                     const double* err_lo, const double* err_hi, const int* ul,
                                     const double* cl, const double* lp, const nh_prior* terms,
                                     int nterms, double* total, const nh_accept* mv) {
  NH_REQUIRE(c && comps && conv && flux && err_lo && err_hi && ul && cl && total, "NULL pointer");
  NH_REQUIRE(nE <= 64, "the likelihood epilogue needs nE <= 64 (one tile per walker)");
  NH_REQUIRE(ncomp >= 1 && ncomp <= NH_MAX_COMP && syn_comp >= 0 && syn_comp < ncomp,
             "bad components");
  NH_REQUIRE(nterms >= 0 && nterms <= NH_MAX_PRIOR && (nterms == 0 || terms), "bad prior terms");
  NH_REQUIRE(comps[syn_comp].ptr == out && comps[syn_comp].ld == ldo,
             "component syn_comp must be this launch's output");
  if (mv) {
    NH_REQUIRE(mv->coords && mv->logp && mv->blk && mv->cursor && mv->accepted &&
                   mv->ns >= 1 && mv->ndim >= 1 && mv->ndim <= 64 && mv->lo >= 0 &&
                   mv->lo + N <= mv->ns, "bad accept block");
  }
  nh_comps cs;
  cs.n = ncomp;
  for (int j = 0; j < ncomp; ++j) {
    NH_REQUIRE(comps[j].ptr && comps[j].ld >= nE, "bad component");
    cs.c[j] = comps[j];
  }
  nh_lnprob_args A;
  nh_lnprob_fill(A, cs, N, nE, conv, flux, err_lo, err_hi, ul, cl, lp, terms, nterms, nullptr,
                 total, mv);
  A.nan_count = c->nan_word;
  return launch_synchrotron(c, w, dlw, B_G, ldB, N, gam, lx, nG, E_eV, nE, out, ldo, &A, syn_comp);
}
