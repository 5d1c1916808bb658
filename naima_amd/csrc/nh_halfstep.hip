// nh_halfstep.hip -- ONE launch per half-step of the ensemble sampler.
//
// One workgroup per proposed walker carries that walker from the stretch-move proposal to
// the accept (emcee StretchMove.get_proposal / RedBlueMove.propose; reference call sites
// core.py:128, 450-457) through everything naima evaluates in between (core.py:97-121):
//
//   chain history of the step the previous launch closed | proposal | the model's parameter
//   packs | particle weights on every grid of the model (models.py eval, radiative.py:156-160,
//   1011-1015) | We/Wp (radiative.py:165,193) | every table reduction trapz_loglog(n K, x)
//   (radiative.py:684, 949-970, 1530) | Synchrotron._spectrum (radiative.py:282-342) |
//   lnprobmodel + priors + lnprob (core.py:34-121) | accept | cursor
//
// The three launches of the round-1 loop (k_step_front -> k_integrate_tables ->
// k_synchrotron+likelihood) handed w/dlw (4.9 MB per launch) and the spectra to each other
// through HBM and paid two kernel boundaries plus a chain of dependent ~1 us reads of
// freshly written data at the head of each.  Here the weights never leave LDS, the
// spectra meet in LDS, and the only global traffic of a workgroup is the walker-independent
// emission table (L2-resident, streamed once per walker) and a few hundred bytes of state.
//
// Inside a workgroup the table reductions (latency/L2-bound) and the synchrotron nodes
// (FP64-VALU-bound) are cut into wave-sized work items that the waves pull from one LDS
// counter, alternating between the two kinds, so that memory-bound and issue-bound waves
// share every SIMD.
//
// Races: a launch only writes the coordinates / log-probabilities of ITS walkers (the
// active half of the slice); partners come from the complementary half, which no workgroup
// of this launch writes.  The history row is copied by the workgroup that owns the walker
// (before its own accept) and, for the complementary half, by the workgroup with the same
// index (nobody writes those rows).  The kernel boundary is the ensemble-wide barrier
// between half-steps.
#include "nh_front.h"
#include "nh_lnprob.h"
#include "nh_syn.h"

#define HS_MAX_TAB 4
#define HS_SYN_NODES 10  // synchrotron nodes per thread and work item (besides the start node)

struct hs_tab {
  const double* Kt; const double* dlnKt; const double* scale; double* out;
  int grid, nK, ldo, nonneg, spec_off, tiles, item0, chunks;
};

struct hs_syn {
  const double* E_eV; const double* B; double* out;
  int grid, nE, ldo, bcol, ldB, spec_off, cdmax, pad;
};

struct hs_comp { const double* ptr; long long ld; double scale; int off; int pad; };

struct hs_dev {
  front_args F;
  int* accepted; int* naccepted; int* sel;
  int do_accept, write_weights;
  hs_tab tab[HS_MAX_TAB];
  int ntab, nT, seg;  // table items in total; segments per item
  hs_syn syn;
  hs_comp comp[NH_MAX_COMP];
  int ncomp, nE;
  const double* conv; const double* flux; const double* elo; const double* ehi;
  const int* ul; const double* cl; const double* lp;
  nh_prior_pack pri;
  double* model_out; double* total;
  // LDS layout, offsets in doubles
  int o_w[NH_MAX_GRIDS], o_d[NH_MAX_GRIDS], o_lx[NH_MAX_GRIDS];
  int o_mkt, o_ig2, o_dig2, o_ig23, o_sq, o_amap, o_part_s, o_part_t, o_spec, nspec;
  int lds_doubles, threads;
  long long* dbg;  // NH_HS_DEBUG=1: shader-clock stamps of the first 8 workgroups, [8][16]
};

struct nh_halfstep_plan {
  hs_dev* dev;       // device copy of the descriptor
  size_t lds_bytes;
  int threads, blocks;
  long long* dbg;
};

// ints at the head of the LDS block (after qs/row/lg/acc)
enum { HI_ME = 0, HI_PA, HI_NA, HI_CD, HI_NS, HI_CNT };
#define HS_O_ROW 64
#define HS_O_LG 72
#define HS_O_ACC 76   // z, lnU, old logp, (pad)
#define HS_O_INT 80   // 16 ints
#define HS_O_FREE 88
#define HS_STAMP(k)                                                                  \
  do {                                                                                \
    if (D.dbg && tid == 0 && j < 8) D.dbg[j * 16 + (k)] = (long long)wall_clock64(); \
  } while (0)

__device__ __forceinline__ double hs_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// one table work item: columns [64 tile, 64 tile + 64) x segments [s0, s1) of table t for
// this workgroup's walker, whose w / dlw / lx live in LDS (wave-uniform reads)
template <bool SIGNED>
__device__ __forceinline__ double hs_table_item(const hs_tab& t, int nG, int tile, int s0, int s1,
                                                const double* ws, const double* ds,
                                                const double* lxs, int lane) {
  const int k = tile * 64 + lane;
  const unsigned kk = k < t.nK ? (unsigned)k : (unsigned)(t.nK - 1);
  const unsigned tbytes = (unsigned)nG * (unsigned)t.nK * 8u;
  const __amdgpu_buffer_rsrc_t rK =
      __builtin_amdgcn_make_buffer_rsrc((void*)t.Kt, 0, (int)tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rD =
      __builtin_amdgcn_make_buffer_rsrc((void*)t.dlnKt, 0, (int)tbytes, 0x00020000);
  const unsigned rowb = (unsigned)t.nK * 8u;
  unsigned ob = ((unsigned)s0 * (unsigned)t.nK + kk) * 8u;
  double acc = 0.0;
  double u1 = ws[s0] * nh_buf_f64(rK, ob);
  int s = s0;
  // eight segments per trip: sixteen table loads in flight per wave (one walker per
  // workgroup: the loop is bound by the L2 round trip, not by issue)
  for (; s + 8 <= s1; s += 8) {
    double K2[8], dK[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      K2[q] = nh_buf_f64(rK, ob + (q + 1) * rowb);
      dK[q] = nh_buf_f64(rD, ob + q * rowb);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double u2 = ws[s + q + 1] * K2[q];
      const double dl = ds[s + q] + dK[q];
      acc += SIGNED ? nh_seg_term<true>(u1, u2, dl, lxs[s + q])
                    : nh_seg_pos<false>(u1, u2, dl, lxs[s + q]);
      u1 = u2;
    }
    ob += 8 * rowb;
  }
  for (; s < s1; ++s) {
    const double K2 = nh_buf_f64(rK, ob + rowb);
    const double dK = nh_buf_f64(rD, ob);
    const double u2 = ws[s + 1] * K2;
    const double dl = ds[s] + dK;
    acc += SIGNED ? nh_seg_term<true>(u1, u2, dl, lxs[s]) : nh_seg_pos<false>(u1, u2, dl, lxs[s]);
    u1 = u2;
    ob += rowb;
  }
  return acc;
}

__global__ __launch_bounds__(1024) void k_half_step(const hs_dev* __restrict__ Dp) {
  extern __shared__ double sm[];
  const hs_dev& D = *Dp;
  const front_args& A = D.F;
  const pw_grids& G = A.G;
  const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = T >> 6;
  const int j = blockIdx.x;
  double* qs = sm;
  double* row = sm + HS_O_ROW;
  double* lg = sm + HS_O_LG;
  double* accs = sm + HS_O_ACC;
  int* hi = reinterpret_cast<int*>(sm + HS_O_INT);
  const bool has_syn = D.syn.grid >= 0;

  HS_STAMP(0);
  // ---- 0. everything that does not depend on the proposal is requested first -----------
  double nE_[NH_MAX_GRIDS], nE2_[NH_MAX_GRIDS], ngx_[NH_MAX_GRIDS], nlr_[NH_MAX_GRIDS],
      nln_[NH_MAX_GRIDS];
#pragma unroll
  for (int g = 0; g < NH_MAX_GRIDS; ++g) {
    nE_[g] = nE2_[g] = ngx_[g] = 1.0;
    nlr_[g] = nln_[g] = 0.0;
    if (g < G.n && tid < G.nG[g]) {
      const int nG = G.nG[g], i = tid;
      const bool last = i + 1 >= nG;
      nE_[g] = G.e[g][i];
      nE2_[g] = last ? nE_[g] : G.e[g][i + 1];
      ngx_[g] = G.xg[g][i];
      if (!last) nlr_[g] = G.lx[g] ? G.lx[g][i] : log(G.xg[g][i + 1] / ngx_[g]);
      nln_[g] = G.lne[g] ? G.lne[g][i] : log(nE_[g]);
    }
  }
  const int c = A.cursor[0];  // slice accepted last (-1: none yet in this block of moves)
  const int cn = c + 1;       // slice proposed, evaluated and accepted here
  const double* r = A.blk + (long long)cn * 3 * A.ns;
  const int* idx = reinterpret_cast<const int*>(r + 2 * A.ns);
  if (tid == 0) {
    hi[HI_CNT] = 0;
    hi[HI_NA] = 0;
    hi[HI_NS] = 0;
  }
  // lx of every grid, the single-row tables, the synchrotron grid's powers -> LDS
  for (int g = 0; g < G.n; ++g) {
    const int nG = G.nG[g];
    double* lxs = sm + D.o_lx[g];
    for (int i = tid; i < nG - 1; i += T)
      lxs[i] = G.lx[g] ? G.lx[g][i] : log(G.xg[g][i + 1] / G.xg[g][i]);
  }
  {
    int ko = D.o_mkt;
    for (int m = 0; m < A.nmom; ++m) {
      const int nG = G.nG[A.mom[m].grid];
      for (int i = tid; i < nG; i += T) {
        sm[ko + i] = A.mom[m].Kt[i];
        sm[ko + nG + i] = A.mom[m].dlnKt[i];
      }
      ko += 2 * nG;
    }
  }
  if (has_syn) {
    const int g = D.syn.grid, nG = G.nG[g];
    const double* gam = G.xg[g];
    for (int i = tid; i < nG; i += T) {
      const double gi = gam[i];
      const double v = 1.0 / (gi * gi);
      sm[D.o_ig2 + i] = v;
      sm[D.o_ig23 + i] = cbrt(v);
      double d = 0.0;
      if (i + 1 < nG) {
        const double rr = gi / gam[i + 1];  // 1/g2^2 - 1/g1^2 without cancellation
        d = v * (rr * rr - 1.0);
      }
      sm[D.o_dig2 + i] = d;
    }
  }
  // ---- chain history of the ensemble step that the previous launch closed ---------------
  // (hist->n counts the CLOSED steps -- the launch that accepts an odd slice increments it --
  // and the row of the step closed last is n - 1: nh_hist_append writes the same row, so a
  // row written by both is simply written twice)
  if (A.hist && c >= 1 && (c & 1)) {
    const long long rowh = A.hist->n - 1;
    if (A.hist->coords && rowh < A.hist->cap) {
      const long long N = 2LL * A.ns;
      double* hc = A.hist->coords + rowh * N * A.ndim;
      double* hl = A.hist->logp + rowh * N;
      const int* idx2 = reinterpret_cast<const int*>(A.blk + (long long)(cn ^ 1) * 3 * A.ns +
                                                     2 * A.ns);
      // rows of this launch's own walkers (before their accept) and of the complementary
      // half (nobody writes those); with fewer workgroups than walkers (sharded: the accept
      // is a later launch) every workgroup takes several
      for (int jj = j; jj < A.ns; jj += gridDim.x) {
        for (int h = 0; h < 2; ++h) {
          const int wr = h == 0 ? idx[jj] : idx2[jj];
          for (int t = tid; t < A.ndim; t += T)
            hc[(long long)wr * A.ndim + t] = A.coords[(long long)wr * A.ndim + t];
          if (tid == 0) hl[wr] = A.logp[wr];
        }
      }
    }
  }
  HS_STAMP(1);
  // ---- 1. proposal ---------------------------------------------------------------------
  if (tid < A.ndim) {
    const int g = A.lo + j;
    const double z = r[g];
    const int me = idx[g], pa = idx[A.ns + g];
    const double cj = A.coords[(long long)pa * A.ndim + tid];
    const double sj = A.coords[(long long)me * A.ndim + tid];
    const double q = cj - (cj - sj) * z;
    A.qT[(long long)tid * A.nloc + j] = q;
    qs[tid] = q;
    if (tid == 0) {
      A.factors[j] = (A.ndim - 1.0) * log(z);
      accs[0] = z;
      accs[1] = r[A.ns + g];
      accs[2] = A.logp[me];
      hi[HI_ME] = me;
      hi[HI_PA] = pa;
    }
  }
  __syncthreads();
  HS_STAMP(2);
  // ---- 2. parameter packs ----------------------------------------------------------------
  if (tid < A.npk * NH_MAX_LAZY) {
    const int q = tid / NH_MAX_LAZY, col = tid % NH_MAX_LAZY;
    if (col < A.pk[q].ncols) {
      const nh_lazy& z = A.pk[q].cols[col];
      double v = z.a;
      if (z.base) v = nh_lazy_apply(z, qs[(z.base - A.qT) / A.nloc]);
      A.pk[q].out[(long long)j * A.pk[q].ld + col] = v;
      if (A.pk[q].out == A.params) {
        row[col] = v;
        if (col == 1 || col == 3 || col == 5) lg[col >> 1] = v > 0.0 ? log(v) : 0.0;
      }
    }
  }
  __syncthreads();
  HS_STAMP(3);
  // ---- 3. particle weights on every grid (-> LDS); the synchrotron liveness search --------
  const pd_par p = {row[0], row[1], row[2], row[3], row[4], row[5], row[6]};
#pragma unroll
  for (int g = 0; g < NH_MAX_GRIDS; ++g) {
    if (g < G.n && tid < G.nG[g]) {
      const int nG = G.nG[g], i = tid;
      const bool last = i + 1 >= nG;
      double nn, dsh;
      pd_core(A.kind, p, nln_[g] - lg[0], nln_[g] - lg[1], lg[2] - lg[0], nE_[g] < p.eb,
              nE2_[g] < p.eb, nlr_[g], nn, dsh);
      nn *= G.scale[g];
      const double wv_ = ngx_[g] * nn, dv = last ? 0.0 : nlr_[g] + dsh;
      sm[D.o_w[g] + i] = wv_;
      sm[D.o_d[g] + i] = dv;
      if (D.write_weights) {
        G.w[g][(long long)j * nG + i] = wv_;
        G.dlw[g][(long long)j * nG + i] = dv;
      }
    }
  }
  for (int g = 0; g < G.n; ++g) {  // grids longer than the workgroup
    const int nG = G.nG[g];
    const double* e = G.e[g];
    const double* xg = G.xg[g];
    for (int i = tid + T; i < nG; i += T) {
      const bool last = i + 1 >= nG;
      const double E = e[i];
      const double E2 = last ? E : e[i + 1];
      const double gx = xg[i];
      double lr = 0.0;
      if (!last) lr = G.lx[g] ? G.lx[g][i] : log(xg[i + 1] / gx);
      const double lnE = G.lne[g] ? G.lne[g][i] : log(E);
      double nn, dsh;
      pd_core(A.kind, p, lnE - lg[0], lnE - lg[1], lg[2] - lg[0], E < p.eb, E2 < p.eb, lr, nn,
              dsh);
      nn *= G.scale[g];
      const double wv_ = gx * nn, dv = last ? 0.0 : lr + dsh;
      sm[D.o_w[g] + i] = wv_;
      sm[D.o_d[g] + i] = dv;
      if (D.write_weights) {
        G.w[g][(long long)j * nG + i] = wv_;
        G.dlw[g][(long long)j * nG + i] = dv;
      }
    }
  }
  double* spec = sm + D.o_spec;
  double Bw = 0.0, qfac = 0.0;
  if (has_syn) {
    Bw = D.syn.bcol >= 0 ? row[D.syn.bcol] : D.syn.B[(long long)j * D.syn.ldB];
    // x = E/Ec,  Ec = 3 e hbar B gamma^2 / (2 m_e c)         radiative.py:331-334
    qfac = NH_ERG_PER_EV * (2.0 * (NH_M_E_G * NH_C_CGS)) / (3.0 * NH_E_GAUSS * NH_HBAR_CGS * Bw);
    if (wv == nwv - 1) {
      // liveness of every photon energy: first node that can contribute (exp(-x) == 0 in
      // double beyond x = 746), compaction of the live ones, per-energy constants
      const int nG = G.nG[D.syn.grid], nseg = nG - 1, nEs = D.syn.nE;
      const double* ig2 = sm + D.o_ig2;
      int* amap = reinterpret_cast<int*>(sm + D.o_amap);
      int* ai0 = amap + nEs;
      double* sq = sm + D.o_sq;  // q | cbrt(q) | CS1 per live energy
      int base = 0;
      long long live_nodes = 0;
      for (int k0 = 0; k0 < nEs; k0 += 64) {
        const int k = k0 + lane;
        int i0 = nG;
        double q = 0.0;
        if (k < nEs) {
          q = D.syn.E_eV[k] * qfac;
          int lo = 0, hi2 = nG;
          while (lo < hi2) {
            const int mid = (lo + hi2) >> 1;
            if (q * ig2[mid] <= 746.0) hi2 = mid; else lo = mid + 1;
          }
          i0 = lo;
        }
        const bool live = i0 < nG;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
        if (live) {
          const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
          const int sb = max(i0 - 1, 0);
          amap[pos] = k;
          ai0[pos] = sb;
          const double E_erg = D.syn.E_eV[k] * NH_ERG_PER_EV;
          sq[pos] = q;
          sq[nEs + pos] = cbrt(q);
          // CS1 = sqrt(3) e^3 B / (2 pi m_e c^2 hbar E)          radiative.py:319-328
          sq[2 * nEs + pos] = (1.7320508075688772 * (NH_E_GAUSS * NH_E_GAUSS * NH_E_GAUSS) * Bw) /
                              (2.0 * NH_PI * NH_M_E_G * (NH_C_CGS * NH_C_CGS) * NH_HBAR_CGS * E_erg);
        }
        if (k < nEs && !live) spec[D.syn.spec_off + k] = 0.0;
        int ln = live ? nseg - max(i0 - 1, 0) : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ln += __shfl_down(ln, off, 64);
        live_nodes += __shfl(ln, 0, 64);
        base += __popcll(m);
      }
      if (lane == 0) {
        int Cd = 1, nS = 0;
        if (base > 0) {
          Cd = (int)((live_nodes / base + HS_SYN_NODES - 1) / HS_SYN_NODES);
          Cd = min(max(Cd, 1), D.syn.cdmax);
          nS = (base * Cd + 63) >> 6;
        }
        hi[HI_NA] = base;
        hi[HI_CD] = Cd;
        hi[HI_NS] = nS;
      }
    }
  }
  __syncthreads();
  HS_STAMP(4);
  // ---- 4. single-row reductions (We, Wp), one wave each ------------------------------------
  if (wv < A.nmom) {
    const nh_moment& m = A.mom[wv];
    const int g = m.grid, nG = G.nG[g];
    int ko = D.o_mkt;
    for (int q = 0; q < wv; ++q) ko += 2 * G.nG[A.mom[q].grid];
    const double* ws = sm + D.o_w[g];
    const double* ds = sm + D.o_d[g];
    const double* lxs = sm + D.o_lx[g];
    double acc = 0.0;
    for (int sgm = lane; sgm < nG - 1; sgm += 64) {
      const double u1 = ws[sgm] * sm[ko + sgm];
      const double u2 = ws[sgm + 1] * sm[ko + sgm + 1];
      const double dl = ds[sgm] + sm[ko + nG + sgm];
      acc += nh_seg_term(u1, u2, dl, lxs[sgm]);
    }
    acc = hs_wave_sum(acc);
    if (lane == 0) m.out[j] = acc;
  }
  HS_STAMP(5);
  // ---- 5. work items: table reductions and synchrotron nodes, pulled from one counter ------
  {
    const int nT = D.nT, nS = hi[HI_NS], nA = hi[HI_NA], Cd = hi[HI_CD];
    const int both = 2 * min(nT, nS), total = nT + nS;
    double* part_t = sm + D.o_part_t;
    double* part_s = sm + D.o_part_s;
    for (;;) {
      int it = 0;
      if (lane == 0) it = atomicAdd(&hi[HI_CNT], 1);
      it = __builtin_amdgcn_readfirstlane(it);
      if (it >= total) break;
      bool is_tab;
      int ix;
      if (it < both) {
        is_tab = (it & 1) == 0;
        ix = it >> 1;
      } else {
        is_tab = nT > nS;
        ix = it - (both >> 1);
      }
      if (is_tab) {
        int t = 0;
        while (t + 1 < D.ntab && ix >= D.tab[t + 1].item0) ++t;
        const hs_tab& tb = D.tab[t];
        const int loc = ix - tb.item0;
        const int tile = loc % tb.tiles, chunk = loc / tb.tiles;
        const int nG = G.nG[tb.grid];
        const int s0 = chunk * D.seg, s1 = min(nG - 1, s0 + D.seg);
        const double* ws = sm + D.o_w[tb.grid];
        const double* ds = sm + D.o_d[tb.grid];
        const double* lxs = sm + D.o_lx[tb.grid];
        const double acc = tb.nonneg ? hs_table_item<false>(tb, nG, tile, s0, s1, ws, ds, lxs, lane)
                                     : hs_table_item<true>(tb, nG, tile, s0, s1, ws, ds, lxs, lane);
        part_t[ix * 64 + lane] = acc;
      } else {
        // 64 (live energy, chunk) pairs of the synchrotron integrand
        const int vt = ix * 64 + lane;
        const int a = vt % nA, ch = vt / nA;
        if (ch < Cd) {
          const int g = D.syn.grid, nG = G.nG[g], nseg = nG - 1, nEs = D.syn.nE;
          const int* amap = reinterpret_cast<const int*>(sm + D.o_amap);
          const int* ai0 = amap + nEs;
          const double* ig2 = sm + D.o_ig2;
          const double* dig2 = sm + D.o_dig2;
          const double* ig23 = sm + D.o_ig23;
          const double* wr = sm + D.o_w[g];
          const double* dwr = sm + D.o_d[g];
          const double* lxs = sm + D.o_lx[g];
          const double* sq = sm + D.o_sq;
          const int sbeg = ai0[a];
          const int per = (nseg - sbeg + Cd - 1) / Cd;
          const int s0 = sbeg + ch * per;
          const int s1 = min(nseg, s0 + per);
          const double q = sq[a], cbq = sq[nEs + a];
          double acc = 0.0;
          if (s0 < s1) {
            double u1 = 0.0, P1 = 1.0;
            {
              const double x = q * ig2[s0];
              if (x <= 746.0) {
                P1 = syn_P(cbq * ig23[s0]);
                u1 = wr[s0] * (P1 * nh_exp_neg(x));  // gamma nelec dNdE / CS1, :335-338
              }
            }
            for (int s = s0; s < s1; ++s) {
              const double x = q * ig2[s + 1];
              double u2 = 0.0, P2 = 1.0;
              if (x <= 746.0) {
                P2 = syn_P(cbq * ig23[s + 1]);
                u2 = wr[s + 1] * (P2 * nh_exp_neg(x));
              }
              // ln|u2/u1| = ln(w2/w1) + ln(P2/P1) - (x2 - x1); unused when a node is 0
              const double dl = dwr[s] + syn_dlnP(P1, P2) - q * dig2[s];
              acc += nh_seg_term<false>(u1, u2, dl, lxs[s]);
              u1 = u2;
              P1 = P2;
            }
          }
          part_s[ch * nEs + a] = acc * sq[2 * nEs + a];  // linear in u: CS1 once per thread
        }
      }
    }
  }
  HS_STAMP(6);
  __syncthreads();
  HS_STAMP(7);
  // ---- 6. the walker's spectra meet in LDS (and go out to HBM for whoever reads them) -----
  {
    for (int t = 0; t < D.ntab; ++t) {
      const hs_tab& tb = D.tab[t];
      for (int k = tid; k < tb.nK; k += T) {
        const int tile = k >> 6, ln = k & 63;
        double sum = 0.0;
        for (int cidx = 0; cidx < tb.chunks; ++cidx)
          sum += sm[D.o_part_t + (tb.item0 + cidx * tb.tiles + tile) * 64 + ln];
        if (tb.scale) sum *= tb.scale[k];
        spec[tb.spec_off + k] = sum;
        tb.out[(long long)j * tb.ldo + k] = sum;
      }
    }
    if (has_syn) {
      const int nA = hi[HI_NA], Cd = hi[HI_CD], nEs = D.syn.nE;
      const int* amap = reinterpret_cast<const int*>(sm + D.o_amap);
      for (int a = tid; a < nA; a += T) {
        double sum = 0.0;
        for (int cidx = 0; cidx < Cd; ++cidx) sum += sm[D.o_part_s + cidx * nEs + a];
        sum *= NH_ERG_PER_EV;  // 1/(s erg) -> 1/(s eV), :340
        spec[D.syn.spec_off + amap[a]] = sum;
      }
    }
  }
  __syncthreads();
  if (has_syn)
    for (int k = tid; k < D.syn.nE; k += T)
      D.syn.out[(long long)j * D.syn.ldo + k] = spec[D.syn.spec_off + k];
  HS_STAMP(8);
  // ---- 7. likelihood + priors (core.py:64-121) and the accept, one wave ---------------------
  if (wv == 0) {
    const int nE = D.nE;
    double prior = 0.0;
    const bool has_prior = D.lp || D.pri.n > 0;
    if (has_prior && lane == 0) {
      prior = D.lp ? D.lp[j] : 0.0;
      for (int t = 0; t < D.pri.n; ++t) {
        const nh_lazy& z = D.pri.t[t].x;
        double v = z.a;
        if (z.base) {
          const long long d = z.base - A.qT;
          // a term on one of this walker's proposed coordinates: taken from LDS
          v = (d >= 0 && d < (long long)A.ndim * A.nloc && d % A.nloc == 0 && z.stride == 1)
                  ? nh_lazy_apply(z, qs[d / A.nloc])
                  : nh_lazy_apply(z, z.base[(long long)j * z.stride]);
        }
        const double p0 = D.pri.t[t].p0, p1 = D.pri.t[t].p1;
        double rr;
        switch (D.pri.t[t].kind) {
          case NH_PRIOR_UNIFORM: rr = (p0 <= v && v <= p1) ? 0.0 : -INFINITY; break;
          case NH_PRIOR_NORMAL: rr = -0.5 * (2.0 * NH_PI * p1) - (v - p0) * (v - p0) / (2.0 * p1); break;
          case NH_PRIOR_LOGUNIFORM: rr = (v > 0.0 && v >= p0 && v <= p1) ? 1.0 / v : -INFINITY; break;
          default: rr = v; break;
        }
        prior += rr;
      }
    }
    double acc = 0.0;
    int nviol = 0, nul = 0;
    for (int k = lane; k < nE; k += 64) {
      double m = 0.0;
      for (int q = 0; q < D.ncomp; ++q) {
        const double v = D.comp[q].off >= 0 ? spec[D.comp[q].off + k]
                                            : D.comp[q].ptr[(long long)j * D.comp[q].ld + k];
        m += D.comp[q].scale * v;
      }
      if (D.model_out) D.model_out[(long long)j * nE + k] = m;
      const double mc = m * D.conv[k];
      const double f = D.flux[k];
      if (D.ul[k]) {
        nul += 1;
        nviol += (mc > f) ? 1 : 0;
      } else {
        const double d = mc - f;
        const double sg = (d > 0.0) ? D.ehi[k] : D.elo[k];
        acc += -(d * d) / (2.0 * (sg * sg));
      }
    }
    int cnt = nviol | (nul << 16);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double a2 = __shfl_down(acc, off, 64);
      const int c2 = __shfl_down(cnt, off, 64);
      acc += a2;
      cnt += c2;
    }
    if (lane == 0) {
      nviol = cnt & 0xffff;
      nul = cnt >> 16;
      // quirk kept from core.py:89-92: cl is indexed by the violation count
      if (nul > 0) acc += (double)nviol * log(1.0 - D.cl[nviol]);
      if (has_prior) acc = isinf(prior) ? prior : acc + prior;  // core.py:115-119
      D.total[j] = acc;
    }
    if (D.do_accept) {  // emcee RedBlueMove.propose for this walker
      acc = __shfl(acc, 0, 64);
      const double z = accs[0];
      const double d = (A.ndim - 1.0) * log(z) + acc - accs[2];
      const bool ok = accs[1] < d;  // NaN compares false, as numpy
      const int me = hi[HI_ME];
      if (ok)
        for (int t = lane; t < A.ndim; t += 64)
          const_cast<double*>(A.coords)[(long long)me * A.ndim + t] = qs[t];
      if (lane == 0) {
        const int g = A.lo + j;
        if (ok) {
          const_cast<double*>(A.logp)[me] = acc;
          if (D.naccepted) D.naccepted[me] += 1;
        }
        D.accepted[g] = ok ? 1 : 0;
        if (D.sel) D.sel[g] = me;
      }
    }
  }
  HS_STAMP(9);
  // ---- 8. the last workgroup to finish moves the cursor on ---------------------------------
  __syncthreads();
  HS_STAMP(10);
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(A.done, 1) == (int)gridDim.x - 1) {
      *A.done = 0;
      A.cursor[0] = cn;
      if (A.hist && (cn & 1) && A.hist->coords) A.hist->n += 1;  // this launch closed a step
    }
  }
  HS_STAMP(11);
}

// history row `row` := the CURRENT ensemble (after the last half-step of a block of moves, at
// the end of a run, before anybody reads the chain: no later launch would have written it)
__global__ void k_hist_append(const double* __restrict__ coords, const double* __restrict__ logp,
                              long long N, int ndim, const nh_hist* hist, long long row) {
  if (row < 0) row = hist->n - 1;  // the step closed last
  if (!hist->coords || row < 0 || row >= hist->cap) return;
  const long long nc = N * ndim;
  double* hc = hist->coords + row * nc;
  double* hl = hist->logp + row * N;
  for (long long t = threadIdx.x; t < nc; t += blockDim.x) hc[t] = coords[t];
  for (long long t = threadIdx.x; t < N; t += blockDim.x) hl[t] = logp[t];
}

extern "C" int nh_hist_append(nh_ctx* c, const double* coords, const double* logp, long long N,
                              int ndim, const nh_hist* hist, long long row) {
  NH_REQUIRE(c && coords && logp && hist && N >= 1 && ndim >= 1, "bad argument");
  nh_prof_scope ps(c, NH_K_GLUE);
  hipLaunchKernelGGL(k_hist_append, dim3(1), dim3(1024), 0, c->stream, coords, logp, N, ndim, hist,
                     row);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_half_step_create(nh_ctx* c, const nh_hs_desc* d, nh_halfstep_plan** out) {
  NH_REQUIRE(c && d && out, "NULL pointer");
  NH_REQUIRE(d->coords && d->logp && d->blk && d->cursor && d->done && d->qT && d->factors &&
                 d->params && d->total, "NULL pointer in the descriptor");
  NH_REQUIRE(d->ns >= 1 && d->ndim >= 1 && d->ndim <= 64 && d->lo >= 0 && d->nloc >= 1 &&
                 d->lo + d->nloc <= d->ns, "bad proposal block");
  NH_REQUIRE(!d->do_accept || (d->lo == 0 && d->nloc == d->ns && d->accepted),
             "the in-kernel accept needs every walker of the slice in this launch");
  NH_REQUIRE(d->npacks >= 1 && d->npacks <= NH_MAX_PACK, "bad pack plan");
  NH_REQUIRE(d->kind >= NH_PD_POWERLAW && d->kind <= NH_PD_LOGPARABOLA, "unknown particle distribution kind");
  NH_REQUIRE(d->ngrids >= 1 && d->ngrids <= NH_MAX_GRIDS, "bad grid count");
  NH_REQUIRE(d->nmoms >= 0 && d->nmoms <= NH_MAX_MOMENT, "bad reductions");
  NH_REQUIRE(d->ntab >= 0 && d->ntab <= NH_HS_MAX_TAB && (d->ntab > 0 || d->syn.grid >= 0),
             "a half-step needs at least one emission component");
  NH_REQUIRE(d->ncomp >= 1 && d->ncomp <= NH_MAX_COMP && d->nE >= 1, "bad likelihood components");
  NH_REQUIRE(d->conv && d->flux && d->err_lo && d->err_hi && d->ul && d->cl, "NULL data column");
  NH_REQUIRE(d->nterms >= 0 && d->nterms <= NH_MAX_PRIOR, "bad prior terms");
  static_assert(NH_HS_MAX_TAB == HS_MAX_TAB, "table count");
  hs_dev H;
  memset(&H, 0, sizeof(H));
  front_args& A = H.F;
  A.coords = d->coords; A.logp = d->logp; A.blk = d->blk; A.cursor = d->cursor; A.done = d->done;
  A.ns = d->ns; A.ndim = d->ndim; A.lo = d->lo; A.nloc = d->nloc; A.qT = d->qT;
  A.factors = d->factors; A.hist = d->hist; A.npk = d->npacks; A.kind = d->kind;
  A.params = d->params; A.nmom = d->nmoms;
  bool have_params = false;
  for (int q = 0; q < d->npacks; ++q) {
    const nh_pack& pk = d->packs[q];
    NH_REQUIRE(pk.out && pk.ncols >= 1 && pk.ncols <= NH_MAX_LAZY && pk.ld >= pk.ncols,
               "bad pack request");
    for (int k = 0; k < pk.ncols; ++k) {
      const double* b = pk.cols[k].base;
      NH_REQUIRE(b == nullptr || (b >= d->qT && b < d->qT + (long long)d->ndim * d->nloc &&
                                  (b - d->qT) % d->nloc == 0 && pk.cols[k].stride == 1),
                 "a pack column must read one proposal coordinate (or be a constant)");
    }
    if (pk.out == d->params) {
      NH_REQUIRE(pk.ncols >= 7, "the particle rows need 7 columns");
      have_params = true;
    }
    A.pk[q] = pk;
  }
  NH_REQUIRE(have_params, "params must be the output of one of the packs");
  pw_grids& G = A.G;
  G.n = d->ngrids;
  G.off[0] = 0;
  int off = HS_O_FREE;
  for (int g = 0; g < d->ngrids; ++g) {
    const nh_grid& gr = d->grids[g];
    NH_REQUIRE(gr.e_eV && gr.xg && gr.nG >= 2 && (!d->write_weights || (gr.w && gr.dlw)),
               "bad grid descriptor");
    G.e[g] = gr.e_eV; G.xg[g] = gr.xg; G.w[g] = gr.w; G.dlw[g] = gr.dlw;
    G.lne[g] = gr.ln_e; G.lx[g] = gr.lx; G.scale[g] = gr.unit_scale; G.nG[g] = gr.nG;
    G.off[g + 1] = G.off[g] + gr.nG;
    A.mom_off[g] = -1;
    H.o_w[g] = off; off += gr.nG;
    H.o_d[g] = off; off += gr.nG;
    H.o_lx[g] = off; off += gr.nG;
  }
  H.o_mkt = off;
  for (int m = 0; m < d->nmoms; ++m) {
    NH_REQUIRE(d->moms[m].grid >= 0 && d->moms[m].grid < d->ngrids && d->moms[m].Kt &&
                   d->moms[m].dlnKt && d->moms[m].out, "bad reduction");
    A.mom[m] = d->moms[m];
    off += 2 * d->grids[d->moms[m].grid].nG;
  }
  H.accepted = d->accepted; H.naccepted = d->naccepted; H.sel = d->sel;
  H.do_accept = d->do_accept; H.write_weights = d->write_weights;
  // ---- synchrotron ----
  H.syn.grid = -1;
  int nspec = 0;
  if (d->syn.grid >= 0) {
    const nh_hs_syn& s = d->syn;
    NH_REQUIRE(s.grid < d->ngrids && s.E_eV && s.out && s.nE >= 1 && s.ldo >= s.nE &&
                   (s.bcol >= 0 ? s.bcol < NH_PD_NPAR : (s.B != nullptr && s.ldB >= 1)),
               "bad synchrotron component");
    const int nG = d->grids[s.grid].nG;
    H.syn.grid = s.grid; H.syn.E_eV = s.E_eV; H.syn.B = s.B; H.syn.out = s.out; H.syn.nE = s.nE;
    H.syn.ldo = s.ldo; H.syn.bcol = s.bcol; H.syn.ldB = s.ldB;
    H.o_ig2 = off; off += nG;
    H.o_dig2 = off; off += nG;
    H.o_ig23 = off; off += nG;
    H.o_sq = off; off += 3 * s.nE;
    H.o_amap = off; off += s.nE + 1;  // 2 nE ints
    int cdmax = 32;
    while (cdmax > 1 && (size_t)cdmax * s.nE * 8 > 40 * 1024) cdmax >>= 1;
    H.syn.cdmax = cdmax;
    H.o_part_s = off; off += cdmax * s.nE;
    H.syn.spec_off = nspec;
    nspec += s.nE;
  }
  // ---- table reductions ----
  H.ntab = d->ntab;
  int seg = 32;
  for (;;) {
    int nT = 0;
    for (int t = 0; t < d->ntab; ++t) {
      const int tiles = (d->tab[t].nK + 63) / 64;
      const int nseg = d->grids[d->tab[t].grid < 0 ? 0 : d->tab[t].grid].nG - 1;
      nT += tiles * ((nseg + seg - 1) / seg);
    }
    if (nT <= 96) break;
    seg *= 2;
  }
  H.seg = seg;
  int nT = 0;
  for (int t = 0; t < d->ntab; ++t) {
    const nh_hs_table& tb = d->tab[t];
    NH_REQUIRE(tb.grid >= 0 && tb.grid < d->ngrids && tb.Kt && tb.dlnKt && tb.out && tb.nK >= 1 &&
                   tb.ldo >= tb.nK, "bad table reduction");
    const int nG = d->grids[tb.grid].nG;
    NH_REQUIRE((long long)nG * tb.nK < (1LL << 28), "table too large for 32-bit offsets");
    hs_tab& o = H.tab[t];
    o.Kt = tb.Kt; o.dlnKt = tb.dlnKt; o.scale = tb.scale; o.out = tb.out; o.grid = tb.grid;
    o.nK = tb.nK; o.ldo = tb.ldo; o.nonneg = tb.nonnegative;
    o.tiles = (tb.nK + 63) / 64;
    o.chunks = (nG - 1 + seg - 1) / seg;
    o.item0 = nT;
    nT += o.tiles * o.chunks;
    o.spec_off = nspec;
    nspec += tb.nK;
  }
  H.nT = nT;
  H.o_part_t = off; off += nT * 64;
  H.o_spec = off; off += nspec;
  H.nspec = nspec;
  // ---- likelihood: where does each component of the model live? ----
  H.ncomp = d->ncomp; H.nE = d->nE;
  for (int q = 0; q < d->ncomp; ++q) {
    const nh_comp& cp = d->comps[q];
    NH_REQUIRE(cp.ptr && cp.ld >= d->nE, "bad component");
    hs_comp& o = H.comp[q];
    o.ptr = cp.ptr; o.ld = cp.ld; o.scale = cp.scale; o.off = -1;
    for (int t = 0; t < d->ntab && o.off < 0; ++t) {
      const long long dd = cp.ptr - d->tab[t].out;
      if (dd >= 0 && dd + d->nE <= d->tab[t].nK && cp.ld == d->tab[t].ldo)
        o.off = H.tab[t].spec_off + (int)dd;
    }
    if (o.off < 0 && d->syn.grid >= 0) {
      const long long dd = cp.ptr - d->syn.out;
      if (dd >= 0 && dd + d->nE <= d->syn.nE && cp.ld == d->syn.ldo) o.off = H.syn.spec_off + (int)dd;
    }
  }
  H.conv = d->conv; H.flux = d->flux; H.elo = d->err_lo; H.ehi = d->err_hi; H.ul = d->ul;
  H.cl = d->cl; H.lp = d->lp; H.model_out = d->model_out; H.total = d->total;
  H.pri.n = d->nterms;
  for (int t = 0; t < d->nterms; ++t) H.pri.t[t] = d->terms[t];
  H.lds_doubles = off;
  const size_t lds = (size_t)off * sizeof(double);
  NH_REQUIRE(lds <= 150 * 1024, "the model's grids and tables do not fit in LDS");
  // workgroup size: small reductions are bound by the dependent round trips, not by lanes
  int maxnG = 0;
  for (int g = 0; g < d->ngrids; ++g) maxnG = d->grids[g].nG > maxnG ? d->grids[g].nG : maxnG;
  long long work = (long long)nT * seg * 64 * 12;
  if (d->syn.grid >= 0) work += (long long)d->syn.nE * d->grids[d->syn.grid].nG * 110 / 3;
  int threads = work >= (1 << 20) ? 1024 : (work >= (1 << 18) ? 512 : 256);
  if (threads < 1024 && maxnG > threads) threads = maxnG > 512 ? 1024 : 512;
  if (const char* e = getenv("NH_HS_THREADS")) threads = atoi(e);
  NH_REQUIRE(threads >= 128 && threads <= 1024 && threads % 64 == 0, "bad workgroup size");
  NH_REQUIRE(threads / 64 > d->nmoms, "more single-row reductions than waves");
  H.threads = threads;
  H.dbg = nullptr;
  if (const char* e = getenv("NH_HS_DEBUG"))
    if (atoi(e) != 0) {
      NH_CHECK_HIP(hipMalloc(&H.dbg, 128 * sizeof(long long)));
      NH_CHECK_HIP(hipMemset(H.dbg, 0, 128 * sizeof(long long)));
    }
  nh_halfstep_plan* P = new nh_halfstep_plan();
  P->dbg = H.dbg;
  P->lds_bytes = lds;
  P->threads = threads;
  P->blocks = d->nloc;
  hipError_t e = hipMalloc(&P->dev, sizeof(hs_dev));
  if (e != hipSuccess) {
    delete P;
    return nh_set_error(NH_ENOMEM, "hipMalloc(descriptor): %s", hipGetErrorString(e));
  }
  e = hipMemcpy(P->dev, &H, sizeof(hs_dev), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(P->dev);
    delete P;
    return nh_set_error(NH_EHIP, "hipMemcpy(descriptor): %s", hipGetErrorString(e));
  }
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute((const void*)k_half_step, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
    if (e != hipSuccess) {
      (void)hipFree(P->dev);
      delete P;
      return nh_set_error(NH_EHIP, "hipFuncSetAttribute(LDS %zu): %s", lds, hipGetErrorString(e));
    }
  }
  *out = P;
  return NH_OK;
}

extern "C" int nh_half_step_launch(nh_ctx* c, nh_halfstep_plan* P) {
  NH_REQUIRE(c && P && P->dev, "bad argument");
  nh_prof_scope ps(c, NH_K_HALFSTEP);
  hipLaunchKernelGGL(k_half_step, dim3((unsigned)P->blocks), dim3(P->threads), P->lds_bytes,
                     c->stream, (const hs_dev*)P->dev);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_half_step_info(const nh_halfstep_plan* P, int* threads, int* blocks,
                                 long long* lds_bytes) {
  NH_REQUIRE(P, "bad argument");
  if (threads) *threads = P->threads;
  if (blocks) *blocks = P->blocks;
  if (lds_bytes) *lds_bytes = (long long)P->lds_bytes;
  return NH_OK;
}

// NH_HS_DEBUG=1: the phase stamps (100 MHz wall clock) of the first 8 workgroups of the
// last launch, out[8][16]; zeros when the plan was created without NH_HS_DEBUG
extern "C" int nh_half_step_stamps(nh_ctx* c, const nh_halfstep_plan* P, long long* out) {
  NH_REQUIRE(c && P && out, "bad argument");
  memset(out, 0, 128 * sizeof(long long));
  if (!P->dbg) return NH_OK;
  int rc = nh_sync(c);
  if (rc) return rc;
  NH_CHECK_HIP(hipMemcpy(out, P->dbg, 128 * sizeof(long long), hipMemcpyDeviceToHost));
  return NH_OK;
}

extern "C" int nh_half_step_destroy(nh_ctx* c, nh_halfstep_plan* P) {
  NH_REQUIRE(c, "ctx is NULL");
  if (!P) return NH_OK;
  int rc = nh_sync(c);
  if (P->dev) (void)hipFree(P->dev);
  if (P->dbg) (void)hipFree(P->dbg);
  delete P;
  return rc;
}
