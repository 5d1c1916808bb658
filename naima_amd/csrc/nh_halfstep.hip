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
#include "nh_hs.h"
#include "nh_syn2.h"

// The first arguments are what the proposal's chain of dependent reads starts from: scalar
// kernel arguments can be preloaded into SGPRs at dispatch (-amdgpu-kernarg-preload-count),
// so the chain does not begin with a trip to the kernel-argument segment.
// slice >= 0: the slice of the block of moves this launch works on (baked into a captured
// graph); slice < 0: derived from the plan's own launch counter.  flags: bit 0 = debug stamps.
//
// The head of the kernel is TWO round trips, written so that the compiler cannot make it more
// (the first version interleaved scalar loads of the 2.9 KB argument block, their waits and
// vector loads as the source happened to use them: seven dependent waits, 2.3 us before the
// first coordinate was asked for):
//   trip 1 (scalar):  the slice's (walker, partner, z, ln U) and hs_first, the block of every
//                     pointer and size the second trip needs -- one wait;
//   trip 2 (vector):  the coordinates (first in the queue: loads return in order), the pack
//                     descriptors, the grids' arrays -- all issued before anything is used.
// (input-only: the value has to BE there, and what follows keeps using the very value that
// was loaded -- a pointer that went through the asm as an in/out integer would come back without
// its address space and be dereferenced with flat loads)
typedef int hs_i16 __attribute__((ext_vector_type(16)));
typedef int hs_i8 __attribute__((ext_vector_type(8)));
typedef int hs_i2 __attribute__((ext_vector_type(2)));
typedef int hs_i4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ const T __attribute__((address_space(1))) * hs_gptr(int lo, int hi) {
  return (const T __attribute__((address_space(1)))*)(((unsigned long long)(unsigned)hi << 32) |
                                                      (unsigned)lo);
}
#define HS_KERNARG_H 48
static_assert(offsetof(hs_hot, F) == 0, "hs_first leads the argument block");
static_assert(sizeof(hs_hot) + HS_KERNARG_H <= 0xcc0, "the warm-up loads cover the argument block");
static_assert(sizeof(hs_hot) + HS_KERNARG_H >= 0xc84, "the last warm-up load (0xc80) stays inside the argument block");
static_assert(sizeof(hs_first) == 256 && offsetof(hs_first, qT) == 224 &&
                  offsetof(hs_first, syn_c) == 240 && offsetof(hs_first, nG) == 48 &&
                  offsetof(hs_first, e) == 64 && offsetof(hs_first, lne) == 128 &&
                  offsetof(hs_first, pkd) == 192, "hs_first: the layout the first trip reads");
// SYN = false: the instance for models without a synchrotron component (table-only: cfg1,
// cfg5) -- the synchrotron items' registers are not there to be allocated around
// S2: the synchrotron items in the log domain on the grid's comb (nh_syn2.h), as the resident loop
// runs them -- a log-uniform particle grid; its block in LDS behind H.o_s2 (doubles):
//   16 header { ilx, th, im, lml, 1/(2 lx), z0, {lm, P}, {nG, -} } | (P + 1) x 6 table | 1024 2^(j/1024) |
//   nG Lambda (ln gamma / 3 + ln scale) | nG + 2 GUARD Lambda ln w | 4 nE per-energy constants |
//   nE comb indices (ints); 1 / gamma^2 with its guards: where the direct form keeps dig2 | ig23
// the header, the table and the node constants behind the grid's three arrays in F.syn_c.
#define HS_S2_HDR 16
__device__ __attribute__((noinline)) double hs_log_ool(double x) { return log(x); }
template <bool SYN, bool S2 = false>
__global__ __launch_bounds__(1024) void k_half_step(const int* __restrict__ done_,
                                                    const double* __restrict__ blk_,
                                                    const double* coords_, int slice, int ns_,
                                                    int ndim_, int lo_, int flags_,
                                                    const hs_hot H, long long* clk_) {
  extern __shared__ double sm[];
  const hs_dev& D = H.C;
  const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = T >> 6;
  const int j = blockIdx.x;
  // K workgroups per walker (a launch of fewer walkers than the chip has CUs): each does the
  // whole prologue and its share of the work items; blocks (j, part) sit on one XCD whenever
  // the number of walkers is a multiple of 8 (linear block id = part gridDim.x + j)
  const int K = gridDim.y, part = blockIdx.y;
  double* qs = sm;
  double* row = sm + HS_O_ROW;
  double* lg = sm + HS_O_LG;
  double* accs = sm + HS_O_ACC;
  int* hi = reinterpret_cast<int*>(sm + HS_O_INT);
  const bool dbg_on = (flags_ & 1) != 0;
  HS_STAMP(0);
  if (dbg_on && tid == 0 && j < 1024) D.dbg[256 + j] = (long long)wall_clock64();
  // the span clock (nh_common.h; flags 2 / 4: this launch opens / closes a span -- a half-step of
  // one launch does both, the staged plan's first launch opens and its second closes)
  if ((flags_ & 2) && j == 0 && blockIdx.y == 0 && tid == 0) nh_clk_open(clk_);

  // ---- trip 1 ------------------------------------------------------------------------------
  // which slice?  `done` counts the workgroups that have finished since the current block of
  // moves was uploaded (every one adds 1 on its way out, nh_half_step_begin_block zeroes it):
  // launches completed = done / grid size, whatever this launch's own early finishers have
  // already added.  hbase = ensemble steps of the run completed before this block of moves.
  const int cn = slice >= 0 ? slice : done_[0] / (int)(gridDim.x * gridDim.y);  // the slice worked on here
  const int c = cn - 1;  // slice accepted by the previous launch
  const double* r = blk_ + (long long)cn * 3 * ns_;
  const int* idx = reinterpret_cast<const int*>(r + 2 * ns_);
  // ONE batch of scalar loads and ONE wait, spelled out: left to itself the compiler fetches
  // each field of the argument block where it is first used, with a wait each time.
  int me, pa;
  double mz, mlnu;
  hs_i16 fa, fb, fc, fd;
  {
    const int g = lo_ + j;  // (wave-uniform addresses)
    hs_i2 zb, ub;
    // hs_first sits at the head of H, H after the eight scalar arguments (44 bytes, aligned to
    // 8).  Not &H.F: taking the address of a by-value argument makes the compiler copy all of
    // it to scratch first.
    const char* kbase = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    const char* kfirst = kbase + HS_KERNARG_H;
    // One word of every other 64-byte line of the argument block (2.9 KB), by ONE wave of the
    // workgroup: the scalar cache starts every launch cold, and the fields the later phases
    // read -- each where it is first needed, each with a wait -- would cost a miss per line
    // and wave.  (The loads land in registers nobody reads; the wait of the batch below
    // covers them.)
    int w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (wv == nwv - 1) {
      asm volatile(
        "s_load_dword %[w0], %[ka], 0x140\n\t"
        "s_load_dword %[w1], %[ka], 0x180\n\t"
        "s_load_dword %[w2], %[ka], 0x1c0\n\t"
        "s_load_dword %[w3], %[ka], 0x200\n\t"
        "s_load_dword %[w0], %[ka], 0x240\n\t"
        "s_load_dword %[w1], %[ka], 0x280\n\t"
        "s_load_dword %[w2], %[ka], 0x2c0\n\t"
        "s_load_dword %[w3], %[ka], 0x300\n\t"
        "s_load_dword %[w0], %[ka], 0x340\n\t"
        "s_load_dword %[w1], %[ka], 0x380\n\t"
        "s_load_dword %[w2], %[ka], 0x3c0\n\t"
        "s_load_dword %[w3], %[ka], 0x400\n\t"
        "s_load_dword %[w0], %[ka], 0x440\n\t"
        "s_load_dword %[w1], %[ka], 0x480\n\t"
        "s_load_dword %[w2], %[ka], 0x4c0\n\t"
        "s_load_dword %[w3], %[ka], 0x500\n\t"
        "s_load_dword %[w0], %[ka], 0x540\n\t"
        "s_load_dword %[w1], %[ka], 0x580\n\t"
        "s_load_dword %[w2], %[ka], 0x5c0\n\t"
        "s_load_dword %[w3], %[ka], 0x600\n\t"
        "s_load_dword %[w0], %[ka], 0x640\n\t"
        "s_load_dword %[w1], %[ka], 0x680\n\t"
        "s_load_dword %[w2], %[ka], 0x6c0\n\t"
        "s_load_dword %[w3], %[ka], 0x700\n\t"
        "s_load_dword %[w0], %[ka], 0x740\n\t"
        "s_load_dword %[w1], %[ka], 0x780\n\t"
        "s_load_dword %[w2], %[ka], 0x7c0\n\t"
        "s_load_dword %[w3], %[ka], 0x800\n\t"
        "s_load_dword %[w0], %[ka], 0x840\n\t"
        "s_load_dword %[w1], %[ka], 0x880\n\t"
        "s_load_dword %[w2], %[ka], 0x8c0\n\t"
        "s_load_dword %[w3], %[ka], 0x900\n\t"
        "s_load_dword %[w0], %[ka], 0x940\n\t"
        "s_load_dword %[w1], %[ka], 0x980\n\t"
        "s_load_dword %[w2], %[ka], 0x9c0\n\t"
        "s_load_dword %[w3], %[ka], 0xa00\n\t"
        "s_load_dword %[w0], %[ka], 0xa40\n\t"
        "s_load_dword %[w1], %[ka], 0xa80\n\t"
        "s_load_dword %[w2], %[ka], 0xac0\n\t"
        "s_load_dword %[w3], %[ka], 0xb00\n\t"
        "s_load_dword %[w0], %[ka], 0xb40\n\t"
        "s_load_dword %[w1], %[ka], 0xb80\n\t"
        "s_load_dword %[w2], %[ka], 0xbc0\n\t"
        "s_load_dword %[w3], %[ka], 0xc00\n\t"
        "s_load_dword %[w0], %[ka], 0xc40\n\t"
        "s_load_dword %[w1], %[ka], 0xc80\n\t"
          : [w0] "+s"(w0), [w1] "+s"(w1), [w2] "+s"(w2), [w3] "+s"(w3)
          : [ka] "s"(kbase));
    }
    asm volatile(
        "s_load_dword %[me], %[pme], 0x0\n\t"
        "s_load_dword %[pa], %[ppa], 0x0\n\t"
        "s_load_dwordx2 %[zb], %[pz], 0x0\n\t"
        "s_load_dwordx2 %[ub], %[pu], 0x0\n\t"
        "s_load_dwordx16 %[fa], %[kf], 0x0\n\t"
        "s_load_dwordx16 %[fb], %[kf], 0x40\n\t"
        "s_load_dwordx16 %[fc], %[kf], 0x80\n\t"
        "s_load_dwordx16 %[fd], %[kf], 0xc0\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [me] "=&s"(me), [pa] "=&s"(pa), [zb] "=&s"(zb), [ub] "=&s"(ub), [fa] "=&s"(fa),
          [fb] "=&s"(fb), [fc] "=&s"(fc), [fd] "=&s"(fd)
        : [pme] "s"(idx + g), [ppa] "s"(idx + ns_ + g), [pz] "s"(r + g), [pu] "s"(r + ns_ + g),
          [kf] "s"(kfirst));
    // (w0..w3 stay reserved until here: a register handed to somebody else while one of the
    // warm-up loads is still on its way would be overwritten when it lands)
    asm volatile("" ::"s"(w0), "s"(w1), "s"(w2), "s"(w3));
    mz = __hiloint2double(zb.y, zb.x);
    mlnu = __hiloint2double(ub.y, ub.x);
  }
  struct {
    const nh_pack __attribute__((address_space(1))) * pk;
    const int __attribute__((address_space(1))) * hbase;
    const double __attribute__((address_space(1))) * logp;
    const nh_hist __attribute__((address_space(1))) * hist;
    int npk8, ngrids, nloc, ppk;
    double __attribute__((address_space(1))) * qT;
    double __attribute__((address_space(1))) * factors;
    const double __attribute__((address_space(1))) * syn_c;
    int syn_nG, broken;
    int nG[NH_MAX_GRIDS];
    const double __attribute__((address_space(1))) * e[NH_MAX_GRIDS];
    const double __attribute__((address_space(1))) * xg[NH_MAX_GRIDS];
    const double __attribute__((address_space(1))) * lne[NH_MAX_GRIDS];
    const double __attribute__((address_space(1))) * lx[NH_MAX_GRIDS];
    unsigned pkd[8];
  } F;
  F.pk = hs_gptr<nh_pack>(fa[0], fa[1]);
  F.hbase = hs_gptr<int>(fa[2], fa[3]);
  F.logp = hs_gptr<double>(fa[4], fa[5]);
  F.hist = hs_gptr<nh_hist>(fa[6], fa[7]);
  F.npk8 = fa[8]; F.ngrids = fa[9]; F.nloc = fa[10]; F.ppk = fa[11];
  F.qT = (double __attribute__((address_space(1)))*)hs_gptr<double>(fd[8], fd[9]);
  F.factors = (double __attribute__((address_space(1)))*)hs_gptr<double>(fd[10], fd[11]);
  F.syn_c = hs_gptr<double>(fd[12], fd[13]);
  F.syn_nG = fd[14];
  F.broken = fd[15];
#pragma unroll
  for (int g = 0; g < NH_MAX_GRIDS; ++g) {
    F.nG[g] = fa[12 + g];
    F.e[g] = hs_gptr<double>(fb[2 * g], fb[2 * g + 1]);
    F.xg[g] = hs_gptr<double>(fb[8 + 2 * g], fb[9 + 2 * g]);
    F.lne[g] = hs_gptr<double>(fc[2 * g], fc[2 * g + 1]);
    F.lx[g] = hs_gptr<double>(fc[8 + 2 * g], fc[9 + 2 * g]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) F.pkd[q] = (unsigned)fd[q];
  HS_STAMP(1);
  // ---- trip 2: everything is asked for before anything is used ------------------------------
  // the proposal's coordinates first
  // (no initial values: a merge of "0" and "what the load brings" at the end of each of these
  // conditionals would be a register copy -- and the wait for the load right here)
  double pcj, psj, pold, kcj, ksj;
  if (tid < ndim_) {
    pcj = coords_[(long long)pa * ndim_ + tid];
    psj = coords_[(long long)me * ndim_ + tid];
    if (tid == 0) pold = F.logp[me];
  }
  // A thread that evaluates a pack column (tid < 8 npacks) needs ONE proposed coordinate (which
  // one is walker-independent: a byte of F.pkd): it fetches that pair of coordinates itself,
  // and its column's descriptor in the same trip -- no barrier and no LDS hop between the
  // proposal and the packs.
  nh_lazy pkz;
  int pk_nc, pk_ld, pkd = -1;
  double* pk_out;
  if (tid < F.npk8) {
    unsigned word = F.pkd[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) word = (tid >> 2) == q ? F.pkd[q] : word;
    const int b = (int)((word >> (8 * (tid & 3))) & 0xFFu);
    if (b != 0xFF) {
      pkd = b;
      kcj = coords_[(long long)pa * ndim_ + pkd];
      ksj = coords_[(long long)me * ndim_ + pkd];
    }
    typedef const long long __attribute__((address_space(1))) * gw_t;
    static_assert(sizeof(nh_lazy) == 48 && offsetof(nh_pack, ncols) == 384 &&
                      offsetof(nh_pack, out) == 392, "nh_pack layout");
    const gw_t P = (gw_t)(F.pk + tid / NH_MAX_LAZY);
    const gw_t w = P + (tid % NH_MAX_LAZY) * 6;  // this thread's column (an nh_lazy)
    pkz.base = (const double*)w[0];  // (never dereferenced: pkd says which coordinate)
    pkz.stride = w[1];
    pkz.a = __longlong_as_double(w[2]);
    pkz.b = __longlong_as_double(w[3]);
    pkz.c = __longlong_as_double(w[4]);
    pkz.tf = (int)w[5];
    pkz.pad = 0;
    const long long shape = P[48];  // ncols | ld
    pk_nc = (int)shape;
    pk_ld = (int)(shape >> 32);
    pk_out = (double*)P[49];
  }
  // the grids' own arrays.  The nodes of all grids are dealt to the waves in units of 64
  // (unit u = 64 consecutive nodes of ONE grid: which grid is wave-uniform, its arrays are
  // scalar pointers); wave w holds units w and w + nwv in registers.  (One node of EVERY grid
  // per thread, the first layout, gave the front waves three nodes and the back ones none:
  // seven passes of the weights on one SIMD against four on another.)
  int ub[NH_MAX_GRIDS + 1];  // first unit of each grid
  ub[0] = 0;
#pragma unroll
  for (int g = 0; g < NH_MAX_GRIDS; ++g)
    ub[g + 1] = ub[g] + (g < F.ngrids ? (F.nG[g] + 63) >> 6 : 0);
  const int nunits = ub[NH_MAX_GRIDS];
  // ... to the waves that have nothing else to do before the second barrier: not the
  // likelihood wave (it evaluates the priors), not the tile waves at the back (the liveness
  // search of the synchrotron energies) -- with a unit on top they were the last to arrive
  const int tiles_ = SYN && H.syn_grid >= 0 ? (H.syn_nE + 63) >> 6 : 0;
  int nwork = nwv - 1 - tiles_, rank = wv == 0 ? 0 : wv - 1;
  bool worker = wv != 1 && wv < nwv - tiles_;
  if (nwork < 1) {  // (a workgroup of one or two waves: everybody)
    nwork = nwv;
    rank = wv;
    worker = true;
  }
  int ug[2], ui[2];  // slot -> grid (wave-uniform; -1: none), node of this lane
  double nE_[2], nE2_[2], ngx_[2], nlr_[2], nln_[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int u = rank + sl * nwork;
    int g = -1;
#pragma unroll
    for (int q = 0; q < NH_MAX_GRIDS; ++q)
      if (worker && u >= ub[q] && u < ub[q + 1]) g = q;
    ug[sl] = g;
    ui[sl] = 0;
    nE_[sl] = nE2_[sl] = ngx_[sl] = 1.0;
    nlr_[sl] = nln_[sl] = 0.0;
    if (g >= 0) {
      int nG = 0, u0 = 0;
      const double __attribute__((address_space(1)))* pe = nullptr;
      const double __attribute__((address_space(1)))* px = nullptr;
      const double __attribute__((address_space(1)))* pl = nullptr;
      const double __attribute__((address_space(1)))* pn = nullptr;
#pragma unroll
      for (int q = 0; q < NH_MAX_GRIDS; ++q)
        if (g == q) {
          nG = F.nG[q]; u0 = ub[q];
          pe = F.e[q]; px = F.xg[q]; pl = F.lx[q]; pn = F.lne[q];
        }
      const int i = (u - u0) * 64 + lane;
      ui[sl] = i;
      if (i < nG) {  // (clamped neighbour indices: nothing is selected before all loads are out)
        const int i1 = min(i + 1, nG - 1), il = min(i, nG - 2);
        if (F.broken) {  // (E itself is only compared with the break energy)
          nE_[sl] = pe[i];
          nE2_[sl] = pe[i1];
        }
        ngx_[sl] = px[i];
        nlr_[sl] = pl[il];
        nln_[sl] = pn[i];
      }
    }
  }
  // the synchrotron grid's walker-independent constants (made once, nh_half_step_create)
  double sc0 = 0.0, sc1 = 0.0, sc2 = 0.0;
  if (tid < F.syn_nG) {
    sc0 = F.syn_c[tid];
    sc1 = F.syn_c[F.syn_nG + tid];
    sc2 = F.syn_c[2 * F.syn_nG + tid];
  }
  const bool has_syn = SYN && H.syn_grid >= 0;
  // what the waves off the proposal's chain park in LDS (below): asked for in this same trip
  const bool fill_wave = wv >= 1 || nwv == 1;
  const int t0 = nwv == 1 ? tid : tid - 64, TT = nwv == 1 ? T : T - 64;
  const int npri = (int)(sizeof(nh_prior_pack) / sizeof(double));
  const double* psrc = reinterpret_cast<const double*>(&D.pri);
  const int kl = TT - 1 - t0;  // (the likelihood columns from the back)
  double vsc[HS_MAX_TAB], vpri, vse, vl[4];
  int vul;
  if (fill_wave) {
    if (t0 < npri) vpri = psrc[t0];
    if (has_syn && t0 < H.syn_nE) vse = H.syn_E[t0];
#pragma unroll
    for (int t = 0; t < HS_MAX_TAB; ++t)
      if (t < H.ntab && t0 < H.tnK[t]) vsc[t] = H.tscale[t] ? H.tscale[t][t0] : 1.0;
    if (kl < H.nE) {
      vl[0] = H.conv[kl];
      vl[1] = H.flux[kl];
      vl[2] = H.elo[kl];
      vl[3] = H.ehi[kl];
      vul = H.ul[kl];
    }
  }
  // history of the step the previous launch closed: its descriptor (the last wave writes it)
  const bool want_hist = F.hist != nullptr && c >= 1 && (c & 1) && part == 0;
  const bool hist_wave = wv == nwv - 1;
  double* hcoords = nullptr;
  double* hlogp = nullptr;
  long long hcap = 0;
  int stepbase = 0;
  if (want_hist && hist_wave) {
    hcoords = F.hist->coords;
    hlogp = F.hist->logp;
    hcap = F.hist->cap;
    stepbase = F.hbase[0];
  }
  const bool lik_wave = wv == (nwv > 1 ? 1 : 0);
  if (tid == 0) {
    hi[HI_CNT] = 0;
    hi[HI_LIVE] = 0;
    hi[HI_READY] = 0;
    hi[HI_NZ] = 0;
  }
  // ---- 1. proposal + 2. parameter packs (the first consumers of trip 2) -----------------------
  if (tid < ndim_) {
    const double q = pcj - (pcj - psj) * mz;
    F.qT[(long long)tid * F.nloc + j] = q;
    qs[tid] = q;
    if (tid == 0) {
      F.factors[j] = (ndim_ - 1.0) * log(mz);
      accs[0] = mz;
      accs[1] = mlnu;
      accs[2] = pold;
      hi[HI_ME] = me;
      hi[HI_PA] = pa;
    }
  }
  HS_STAMP(2);
  if (tid < F.npk8) {
    const int col = tid % NH_MAX_LAZY;
    if (col < pk_nc) {
      double v = pkz.a;
      const double qk = kcj - (kcj - ksj) * mz;  // (the same q as qs[pkd])
      if (pkd >= 0) v = nh_lazy_apply(pkz, qk);
      int ld = pk_ld;
      asm volatile("" : "+v"(ld));  // (its sign extension would be hoisted to the load: a wait)
      pk_out[(long long)j * ld + col] = v;
      if (tid / NH_MAX_LAZY == F.ppk) {  // the particle rows: also into LDS
        row[col] = v;
        if (col == 1 || col == 3 || col == 5) {
          // (a power of ten of a proposed coordinate: hs_ln_pow10, as the resident loop has it)
          const bool p10 = pkd >= 0 && pkz.tf == NH_TF_POW10;
          const double lv = log(fabs(p10 ? pkz.a : v));
          lg[col >> 1] = v > 0.0 ? (p10 ? hs_ln_pow10(lv, pkz.b, pkz.c, qk) : lv) : 0.0;
        }
      }
    }
  }
  if (dbg_on && j == 0 && lane == 0) D.dbg[224 + wv] = (long long)wall_clock64();
  // ---- grids' own arrays -> LDS ------------------------------------------------------------
  if (wv == nwv - 1) sm[HS_O_T64 + lane] = exp2((double)lane * 0.015625);
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
    if (ug[sl] >= 0) {
      const int g = ug[sl], i = ui[sl], nG = H.nG[g];
      if (i + 1 >= nG) nlr_[sl] = 0.0;  // (the last node of a grid has no segment to its right)
      if (i + 1 < nG) sm[H.o_lx[g] + i] = nlr_[sl];
    }
  for (int u = worker ? rank + 2 * nwork : nunits; u < nunits; u += nwork) {  // the rest: from memory
    int g = 0;
    while (u >= ub[g + 1]) ++g;
    const int i = (u - ub[g]) * 64 + lane;
    if (i + 1 < H.nG[g]) sm[H.o_lx[g] + i] = H.lx[g][i];
  }
  // the single-row tables (We, Wp), the likelihood's data columns, the prior terms ...: loaded
  // and parked in LDS by the waves that are NOT on the proposal's chain.  Every load of a
  // thread's first element of each array is issued before the first store (written as
  // load-then-store loops, one after the other, this cost the waves five round trips in a row
  // and the workgroup's first barrier waited for them 2.4 us after the proposal was ready).
  if (fill_wave) {
    // ---- ... and now the stores
    if (t0 < npri) sm[H.o_pri + t0] = vpri;
    for (int t = t0 + TT; t < npri; t += TT) sm[H.o_pri + t] = psrc[t];
    if (has_syn) {
      if (t0 < H.syn_nE) sm[H.o_synE + t0] = vse;
      for (int k = t0 + TT; k < H.syn_nE; k += TT) sm[H.o_synE + k] = H.syn_E[k];
    }
#pragma unroll
    for (int t = 0; t < HS_MAX_TAB; ++t)  // per-column factors of the reductions
      if (t < H.ntab) {
        if (t0 < H.tnK[t]) sm[H.o_scale + H.tspec[t] + t0] = vsc[t];
        for (int k = t0 + TT; k < H.tnK[t]; k += TT)
          sm[H.o_scale + H.tspec[t] + k] = H.tscale[t] ? H.tscale[t][k] : 1.0;
      }
    double* lik = sm + H.o_lik;  // conv | flux | elo | ehi | ul, nE each
    if (kl < H.nE) {
      lik[kl] = vl[0];
      lik[H.nE + kl] = vl[1];
      lik[2 * H.nE + kl] = vl[2];
      lik[3 * H.nE + kl] = vl[3];
      lik[4 * H.nE + kl] = (double)vul;
    }
    for (int k = kl + TT; k < H.nE; k += TT) {
      lik[k] = H.conv[k];
      lik[H.nE + k] = H.flux[k];
      lik[2 * H.nE + k] = H.elo[k];
      lik[3 * H.nE + k] = H.ehi[k];
      lik[4 * H.nE + k] = (double)H.ul[k];
    }
  }
  // (the log-domain instance reads 1 / gamma^2 and nothing else of these: the two arrays of the
  // direct form -- adjacent -- hold its guarded copy, o_s2ig below)
  if (tid < F.syn_nG) {
    sm[H.o_ig2 + tid] = sc0;
    if (!(SYN && S2)) {
      sm[H.o_ig23 + tid] = sc1;
      sm[H.o_dig2 + tid] = sc2;
    }
  }
  for (int i = tid + T; i < F.syn_nG; i += T) {  // a grid longer than the workgroup
    sm[H.o_ig2 + i] = F.syn_c[i];
    if (!(SYN && S2)) {
      sm[H.o_ig23 + i] = F.syn_c[F.syn_nG + i];
      sm[H.o_dig2 + i] = F.syn_c[2 * F.syn_nG + i];
    }
  }
  // (S2) the log-domain items' table and node constants: one more batch of loads, used after the
  // weights' barrier
  int s2P = 0, s2lm = 0, o_s2t = 0, o_s2lg = 0, o_s2ig = 0, o_s2lw = 0, o_s2q = 0, o_s2z = 0;
  if (SYN && S2) {
    const int nGs = F.syn_nG, o = H.o_s2;
    const auto* src = F.syn_c + 3 * nGs;  // (an address_space(1) pointer: see hs_gptr)
    const double lmPd = src[6];
    const hs_i2 lmP = {__double2loint(lmPd), __double2hiint(lmPd)};
    s2lm = __builtin_amdgcn_readfirstlane(lmP.x);
    s2P = __builtin_amdgcn_readfirstlane(lmP.y);
    const int ntb = HS_S2_HDR + (s2P + 1) * HS_S2_STRIDE;
    o_s2t = o + ntb;
    o_s2lg = o_s2t + HS_S2_TN;
    o_s2ig = H.o_dig2;  // (dig2, ig23: 2 nG >= nG + 2 guards doubles in a row, hs_s2_prepare's nG >= 2 guards)
    o_s2lw = o_s2lg + nGs;
    o_s2q = o_s2lw + nGs + 2 * HS_S2_GUARD;
    o_s2z = o_s2q + 4 * H.syn_nE;
    for (int i = tid; i < ntb; i += T) sm[o + i] = src[i];
    for (int i = tid; i < HS_S2_TN; i += T) sm[o_s2t + i] = exp2((double)i * (1.0 / HS_S2_TN));
    for (int i = tid; i < nGs; i += T) sm[o_s2lg + i] = src[ntb + i];
    for (int i = tid; i < nGs + 2 * HS_S2_GUARD; i += T) {  // (guards: the edge values, any finite number)
      const int ii = min(max(i - HS_S2_GUARD, 0), nGs - 1);
      sm[o_s2ig + i] = F.syn_c[ii];  // (1 / gamma^2)
      if (i < HS_S2_GUARD || i >= nGs + HS_S2_GUARD) sm[o_s2lw + i] = HS_S2_FLOOR;
    }
  }
  // ---- chain history rows (the step closed by the previous launch; hist->n is not used:
  // the row is the number of closed steps of this run - 1, what nh_hist_append is told too).
  // One wave's work, and not one of the proposal's chain.
  if (want_hist && hist_wave) {
    const long long rowh = (long long)stepbase + (cn >> 1) - 1;
    if (hcoords && rowh >= 0 && rowh < hcap) {
      const long long N = 2LL * H.ns;
      double* hc = hcoords + rowh * N * H.ndim;
      double* hl = hlogp + rowh * N;
      const int* idx2 = reinterpret_cast<const int*>(H.blk + (long long)(cn ^ 1) * 3 * H.ns +
                                                     2 * H.ns);
      // rows of this launch's own walkers (before their accept) and of the complementary
      // half (nobody writes those); with fewer workgroups than walkers (sharded: the accept
      // is a later launch) every workgroup takes several
      for (int jj = j; jj < H.ns; jj += gridDim.x) {
        for (int h = 0; h < 2; ++h) {
          const int wr = h == 0 ? idx[jj] : idx2[jj];
          for (int t = lane; t < H.ndim; t += 64)
            hc[(long long)wr * H.ndim + t] = H.coords[(long long)wr * H.ndim + t];
          if (lane == 0) hl[wr] = H.logp[wr];
          for (int b = 0; b < D.nblob; ++b) {  // the blobs that belong to these positions
            const nh_hs_blob& bl = D.blob[b];
            double* hb = bl.hist ? *bl.hist : nullptr;
            if (hb)
              for (int t = lane; t < bl.m; t += 64)
                hb[(rowh * N + wr) * bl.m + t] = bl.cur[(long long)wr * bl.m + t];
          }
        }
      }
    }
  }
  if (dbg_on && j == 0 && lane == 0) D.dbg[208 + wv] = (long long)wall_clock64();
  __syncthreads();
  HS_STAMP(3);
  if (tid == 0 && j == 0) H.cursor[0] = cn;  // for launches that follow the older slice protocol
  if (K > 1) {  // the work items of the other workgroups of this walker: their slots count as 0
    for (int t = tid; t < D.nT * 64; t += T) sm[H.o_part_t + t] = 0.0;
    if (SYN && H.syn_grid >= 0)
      for (int t = tid; t < D.syn_cdmax * H.syn_nE; t += T) sm[H.o_part_s + t] = 0.0;
  }
  // the single-row tables (We, Wp) -> LDS, by the waves at the back of the workgroup that are
  // not tile waves (those search the synchrotron energies' live ranges now): they hold at most
  // one unit of nodes below, and nobody reads these before the next barrier
  {
    const int Tb = (nwv - tiles_ > 0 ? nwv - tiles_ : nwv) * 64;
    int ko = H.o_mkt;
    for (int m = 0; m < H.nmom; ++m) {
      const int nG = H.nG[H.mgrid[m]];
      for (int i = Tb - 1 - tid; i >= 0 && i < nG; i += Tb) {
        sm[ko + i] = H.mKt[m][i];
        sm[ko + nG + i] = H.mdK[m][i];
      }
      ko += 2 * nG;
    }
  }
  // ---- the priors (core.py:34-58, 99-101), now: a proposal the prior forbids is never accepted
  // whatever its likelihood, so none of its integrals is evaluated (the reference evaluates
  // the model and throws it away, core.py:103-119).  Besides the time saved on such walkers
  // this keeps the launch time independent of how absurd they are: a negative B makes every
  // (energy, gamma) node of the synchrotron integrand "live" -- 4x the work of a normal walker.
  if (lik_wave) {
    const nh_prior_pack& PR = *reinterpret_cast<const nh_prior_pack*>(sm + H.o_pri);
    const int npri = PR.n;
    const bool has_prior = D.lp || npri > 0;
    double term = 0.0;
    if (lane < npri) {  // one term per lane
      const nh_prior& pt = PR.t[lane];
      const nh_lazy z = pt.x;
      double v = z.a;
      if (z.base) {
        const long long d = z.base - H.qT;
        // a term on one of this walker's proposed coordinates: taken from LDS
        v = (d >= 0 && d < (long long)H.ndim * H.nloc && d % H.nloc == 0 && z.stride == 1)
                ? nh_lazy_apply(z, qs[d / H.nloc])
                : nh_lazy_apply(z, z.base[(long long)j * z.stride]);
      }
      const double p0 = pt.p0, p1 = pt.p1;
      switch (pt.kind) {
        case NH_PRIOR_UNIFORM: term = (p0 <= v && v <= p1) ? 0.0 : -INFINITY; break;
        case NH_PRIOR_NORMAL: term = -0.5 * (2.0 * NH_PI * p1) - (v - p0) * (v - p0) / (2.0 * p1); break;
        case NH_PRIOR_LOGUNIFORM: term = (v > 0.0 && v >= p0 && v <= p1) ? 1.0 / v : -INFINITY; break;
        default: term = v; break;
      }
    }
    // (summed in term order, as the separate kernels do: lane 0 adds them up)
    double prior = 0.0;
    for (int t = 0; t < npri; ++t) prior += __shfl(term, t, 64);
    if (D.lp) prior += D.lp[j];
    if (lane == 0) {
      accs[3] = prior;
      hi[HI_DEAD] = (has_prior && isinf(prior)) ? 1 : 0;
    }
  }
  // ---- 3. particle weights on every grid (-> LDS); the synchrotron liveness search --------
  const pd_par p = {row[0], row[1], row[2], row[3], row[4], row[5], row[6]};
  double* spec = sm + H.o_spec;
  double Bw = 0.0, qfac = 0.0;
  if (has_syn) {
    Bw = D.syn_bcol >= 0 ? row[D.syn_bcol] : D.synB[(long long)j * D.syn_ldB];
    // x = E/Ec,  Ec = 3 e hbar B gamma^2 / (2 m_e c)         radiative.py:331-334
    qfac = NH_ERG_PER_EV * (2.0 * (NH_M_E_G * NH_C_CGS)) / (3.0 * NH_E_GAUSS * NH_HBAR_CGS * Bw);
  }
  // liveness of every photon energy: first node that can contribute (exp(-x) == 0 in double
  // beyond x = 746).  Tile t (64 energies) belongs to wave nwv-1-t (the waves at the back: the
  // front ones carry the longest grids' weight nodes); each keeps its lanes' answers in
  // registers and leaves the tile's count in LDS.  The live energies are compacted IN ORDER
  // after the barrier (neighbouring lanes of a synchrotron work item then walk ranges of
  // nearly the same length: compacted in arrival order the items ran 25 % longer).
  int lv_i0 = 0, lv_k = -1;
  double lv_q = 0.0, lv_E = 0.0;
  bool lv_live = false;
  int* tcnt = reinterpret_cast<int*>(sm + HS_O_INT) + 8;  // [<= 8] live energies per tile
  const int syn_tiles = tiles_;
  if (has_syn && nwv - 1 - wv < syn_tiles) {
    const int nG = H.nG[H.syn_grid];
    const double* ig2 = sm + H.o_ig2;
    const int t = nwv - 1 - wv;
    lv_k = t * 64 + lane;
    lv_i0 = nG;
    if (lv_k < H.syn_nE) {
      lv_E = sm[H.o_synE + lv_k];
      lv_q = lv_E * qfac;
      int lo = 0, hi2 = nG;  // first i with q*ig2[i] <= 746 (ig2 decreases with i)
      while (lo < hi2) {
        const int mid = (lo + hi2) >> 1;
        if (lv_q * ig2[mid] <= 746.0) hi2 = mid; else lo = mid + 1;
      }
      lv_i0 = lo;
    }
    lv_live = lv_i0 < nG;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(lv_live);
    int ln = lv_live ? (nG - 1) - lv_i0 : 0;  // segments this energy walks
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ln += __shfl_down(ln, off, 64);
    if (lane == 0) {
      tcnt[t] = __popcll(m);
      if (ln > 0) atomicAdd(&hi[HI_LIVE], ln);
    }
  }
  // A grid on which EVERY weight is zero (a walker far off in parameter space: cutoff energy
  // 10^-1000 TeV ...) contributes exact zeros to everything integrated over it (utils.py:347-348):
  // its work items are skipped.  Such walkers are the ones that get stuck on the likelihood's
  // zero-flux plateau; evaluated in full they cost 1.5x a normal walker and set the launch time.
  int nzmask = 0;
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
    if (ug[sl] >= 0) {  // (wave-uniform)
      const int g = ug[sl], i = ui[sl], nG = H.nG[g];
      if (i < nG) {
        const bool last = i + 1 >= nG;
        double nn, dsh;
        pd_core(D.kind, p, nln_[sl] - lg[0], nln_[sl] - lg[1], lg[2] - lg[0], nE_[sl] < p.eb,
                nE2_[sl] < p.eb, nlr_[sl], nn, dsh, sm + HS_O_T64);
        const double nraw = nn;
        nn *= H.scale[g];
        const double wv_ = ngx_[sl] * nn, dv = last ? 0.0 : nlr_[sl] + dsh;
        sm[H.o_w[g] + i] = wv_;
        sm[H.o_d[g] + i] = dv;
        if (SYN && S2 && g == H.syn_grid)  // Lambda ln|w| + the node's share of ln Gtilde (nh_syn2.h); a zero weight: the floor
          sm[o_s2lw + HS_S2_GUARD + i] = wv_ != 0.0 ? fma(HS_S2_LAMBDA, hs_log_ool(fabs(nraw)), sm[o_s2lg + i]) : HS_S2_FLOOR;
        if (H.o_dp[g] >= 0) {  // (wave-uniform) what the non-negative table items read
          const double il = last ? 0.0 : nh_rcp(nlr_[sl]);
          sm[H.o_dp[g] + i] = dv * il;
          sm[H.o_th[g] + i] = NH_SEG_SMALL_POS * il;
        }
        if (wv_ != 0.0) nzmask |= 1 << g;
        if (SYN && S2 && !isfinite(wv_)) nzmask |= 256 << g;  // (the log-domain items: see syn_nan)
        if (D.write_weights) {
          D.w[g][(long long)j * nG + i] = wv_;
          D.dlw[g][(long long)j * nG + i] = dv;
        }
      }
    }
  for (int u = worker ? rank + 2 * nwork : nunits; u < nunits; u += nwork) {  // the rest: from memory
    int g = 0;
    while (u >= ub[g + 1]) ++g;
    const int nG = H.nG[g], i = (u - ub[g]) * 64 + lane;
    if (i < nG) {
      const bool last = i + 1 >= nG;
      const double E = H.e[g][i];
      const double E2 = last ? E : H.e[g][i + 1];
      const double gx = H.xg[g][i];
      double lr = 0.0;
      if (!last) lr = H.lx[g][i];
      const double lnE = H.lne[g][i];
      double nn, dsh;
      pd_core(D.kind, p, lnE - lg[0], lnE - lg[1], lg[2] - lg[0], E < p.eb, E2 < p.eb, lr, nn,
              dsh, sm + HS_O_T64);
      const double nraw = nn;
      nn *= H.scale[g];
      const double wv_ = gx * nn, dv = last ? 0.0 : lr + dsh;
      sm[H.o_w[g] + i] = wv_;
      sm[H.o_d[g] + i] = dv;
      if (SYN && S2 && g == H.syn_grid)
        sm[o_s2lw + HS_S2_GUARD + i] = wv_ != 0.0 ? fma(HS_S2_LAMBDA, hs_log_ool(fabs(nraw)), sm[o_s2lg + i]) : HS_S2_FLOOR;
      if (H.o_dp[g] >= 0) {
        const double il = last ? 0.0 : nh_rcp(lr);
        sm[H.o_dp[g] + i] = dv * il;
        sm[H.o_th[g] + i] = NH_SEG_SMALL_POS * il;
      }
      if (wv_ != 0.0) nzmask |= 1 << g;
      if (SYN && S2 && !isfinite(wv_)) nzmask |= 256 << g;
      if (D.write_weights) {
        D.w[g][(long long)j * nG + i] = wv_;
        D.dlw[g][(long long)j * nG + i] = dv;
      }
    }
  }
  {
    int any = nzmask;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) any |= __shfl_xor(any, off, 64);
    if (lane == 0 && any) atomicOr(&hi[HI_NZ], any);
  }
  if (dbg_on && j == 0 && lane == 0) D.dbg[192 + wv] = (long long)wall_clock64();
  __syncthreads();
  HS_STAMP(4);
  const int nz = hi[HI_DEAD] ? 0 : hi[HI_NZ];  // (forbidden by the prior: nothing is integrated)
  // chunks per live energy and the number of synchrotron work items (every thread computes the
  // same); the tile waves then compact the live energies IN ORDER and leave their constants
  // in LDS -- nobody waits for that at a barrier: whoever pulls the first synchrotron item
  // (the first pulls are table items) checks the ready count
  int nA = 0, Cd = 1, nS = 0;
  if (has_syn) {
    const int nG = H.nG[H.syn_grid], nEs = H.syn_nE;
    for (int q = 0; q < syn_tiles; ++q) nA += tcnt[q];
    // (the log-domain items take ln q and ln w: a magnetic field that is not positive, or a weight
    // that is not finite, makes the reference's spectrum NaN -- x < 0 overflows exp(-x), inf x 0 --
    // and this one with it; k_half_step_run the same)
    const bool syn_nan = S2 && !hi[HI_DEAD] &&
                         (!(Bw > 0.0) || !(Bw < INFINITY) || (nz >> (8 + H.syn_grid) & 1) != 0);
    const bool syn_zero = !(nz >> H.syn_grid & 1) || syn_nan;  // nothing to integrate
    if (nA > 0 && !syn_zero) {
      Cd = (hi[HI_LIVE] / nA + D.syn_nodes - 1) / D.syn_nodes;
      Cd = min(max(Cd, 1), D.syn_cdmax);
      nS = (nA * Cd + 63) >> 6;
    }
    if (syn_zero) {
      for (int k = tid; k < nEs; k += T) spec[H.syn_spec_off + k] = syn_nan ? NAN : 0.0;
      nA = 0;
    }
    if (lv_k >= 0 && !syn_zero) {
      int* amap = reinterpret_cast<int*>(sm + H.o_amap);
      int* ai0 = amap + nEs;
      double* sq = sm + H.o_sq;  // q | cbrt(q) | CS1 per live energy
      const int t = nwv - 1 - wv;
      int base = 0;
      for (int q = 0; q < t; ++q) base += tcnt[q];
      const unsigned long long m = __builtin_amdgcn_ballot_w64(lv_live);
      if (lv_live) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        amap[pos] = lv_k;
        // from the first LIVE node on: the segment to its left has a zero node and contributes
        // an exact 0 (utils.py:347-348) -- no dead node inside a range, no zero test per segment
        ai0[pos] = lv_i0;
        sq[pos] = lv_q;
        sq[nEs + pos] = cbrt(lv_q);
        // CS1 = sqrt(3) e^3 B / (2 pi m_e c^2 hbar E)          radiative.py:319-328
        const double cs1 = (1.7320508075688772 * (NH_E_GAUSS * NH_E_GAUSS * NH_E_GAUSS) * Bw) /
                           (2.0 * NH_PI * NH_M_E_G * (NH_C_CGS * NH_C_CGS) * NH_HBAR_CGS *
                            (lv_E * NH_ERG_PER_EV));
        sq[2 * nEs + pos] = cs1;
        if (S2) {
          // where this energy's nodes sit on the comb (nh_syn2.h): node i at z + i steps below
          // T_top, z = Z + f; ln Gtilde's t / 3 + ln 1.808 rides with the energy; the sign of the
          // amplitude in front of the integral
          const double* hdr = sm + H.o_s2;
          const double lnq = hs_log_ool(lv_q);
          const double z = fma(-lnq, hdr[4], hdr[5]);
          const double Zf = floor(z);
          double* s2q = sm + o_s2q;
          s2q[pos] = lv_q;
          s2q[nEs + pos] = (HS_S2_LAMBDA / 3.0) * lnq;  // (ln 1.808 rides in the table)
          s2q[2 * nEs + pos] = (z - Zf) * hdr[2];
          s2q[3 * nEs + pos] = p.A < 0.0 ? -cs1 : cs1;
          reinterpret_cast<int*>(sm + o_s2z)[pos] = (int)Zf;
        }
      } else if (lv_k < nEs) {
        spec[H.syn_spec_off + lv_k] = 0.0;
      }
      (void)nG;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) atomicAdd(&hi[HI_READY], 1);
    }
  }
  // ---- 4. single-row reductions (We, Wp), one wave each (from the back) ------------------
  if (nwv - 1 - wv < H.nmom) {
    const int m = nwv - 1 - wv;
    const int g = H.mgrid[m], nG = H.nG[g];
    int ko = H.o_mkt;
    for (int q = 0; q < m; ++q) ko += 2 * H.nG[H.mgrid[q]];
    const double* ws = sm + H.o_w[g];
    const double* ds = sm + H.o_d[g];
    const double* lxs = sm + H.o_lx[g];
    double acc = 0.0;
    for (int sgm = lane; sgm < nG - 1; sgm += 64) {
      const double u1 = ws[sgm] * sm[ko + sgm];
      const double u2 = ws[sgm + 1] * sm[ko + sgm + 1];
      const double dl = ds[sgm] + sm[ko + nG + sgm];
      acc += nh_seg_term(u1, u2, dl, lxs[sgm]);
    }
    acc = hs_wave_sum(acc);
    if (lane == 0) {
      D.mom_out[m][j] = acc;
      sm[D.o_mrow + H.nE + m] = acc;
    }
  }
  HS_STAMP(5);
  // ---- 5. work items: table reductions and synchrotron nodes, pulled from one counter ------
  {
    const int nT = D.nT;
    // pull order: a first round of table items (one per wave: the synchrotron constants are
    // still being written), then the two kinds alternate, synchrotron (the longer items) first
    const int F = min(nT, nwv);
    const int both = 2 * min(nT - F, nS), total = nT + nS;
    bool syn_ready = !has_syn;
    double* part_t = sm + H.o_part_t;
    double* part_s = sm + H.o_part_s;
    int dbg_nt = 0, dbg_ns = 0;
    if (dbg_on && j == 0 && lane == 0) D.dbg[176 + wv] = (long long)wall_clock64();
    for (;;) {
      int it = 0;
      if (lane == 0) it = atomicAdd(&hi[HI_CNT], 1);
      it = __builtin_amdgcn_readfirstlane(it);
      if (K > 1) {  // this workgroup's share: one of every K items, rotating (the two kinds
        if (it * K >= total) break;  // alternate: a fixed residue would take one kind only)
        it = it * K + ((part + it) & (K - 1));
        if (it >= total) continue;
      }
      if (it >= total) break;
      bool is_tab;
      int ix;
#if HS_ORDER == 1
      is_tab = it < nT;
      ix = is_tab ? it : it - nT;
      (void)both;
#elif HS_ORDER == 2
      is_tab = it >= nS;
      ix = is_tab ? it - nS : it;
      (void)both;
#else
      if (it < F) {
        is_tab = true;
        ix = it;
      } else if (it - F < both) {
        is_tab = ((it - F) & 1) != 0;
        ix = is_tab ? F + ((it - F) >> 1) : (it - F) >> 1;
      } else {
        is_tab = nT - F > nS;
        ix = is_tab ? it - nS : it - nT;
      }
#endif
#ifdef HS_SKIP_TAB
      if (is_tab) { part_t[ix * 64 + lane] = 0.0; continue; }
#endif
#ifdef HS_SKIP_SYN
      if (!is_tab) { const int vt0 = ix * 64 + lane; if (vt0 / nA < Cd) part_s[(vt0 / nA) * H.syn_nE + vt0 % nA] = 0.0; continue; }
#endif
      if (is_tab) ++dbg_nt; else ++dbg_ns;
      if (is_tab) {
        // (workgroups of one XCD start at different items: what one has fetched into the L2
        // the others find there)
        ix = (ix + (j >> 3) * 5) % nT;
        int t = 0;
        while (t + 1 < D.ntab && ix >= D.tab[t + 1].item0) ++t;
        const hs_tab& tb = D.tab[t];
        const int loc = ix - tb.item0;
        const int tile = loc % tb.tiles, chunk = loc / tb.tiles;
        const int tg = __builtin_amdgcn_readfirstlane(tb.grid);
        const int nG = __builtin_amdgcn_readfirstlane(H.nG[tg]);
        int s0, s1;
        hs_chunk_range(tb.chunks, chunk, D.seg, nG - 1, s0, s1);
        const double* ws = sm + __builtin_amdgcn_readfirstlane(H.o_w[tg]);
        // (a non-negative table: the pre-divided log-ratios and the series thresholds)
        const bool pre = __builtin_amdgcn_readfirstlane(tb.nonneg) != 0;
        const double* ds = sm + __builtin_amdgcn_readfirstlane(pre ? H.o_dp[tg] : H.o_d[tg]);
        const double* lxs = sm + __builtin_amdgcn_readfirstlane(pre ? H.o_th[tg] : H.o_lx[tg]);
        double acc;
        if (!(nz >> tg & 1))
          acc = 0.0;
        else if (__builtin_amdgcn_readfirstlane(tb.sub) > 1)
          acc = tb.nonneg ? hs_table_item_packed<false, SYN ? 4 : 6>(tb, nG, s0, s1, ws, ds, lxs, lane)
                          : hs_table_item_packed<true, SYN ? 4 : 6>(tb, nG, s0, s1, ws, ds, lxs, lane);
        else
          acc = tb.nonneg ? hs_table_item<false>(tb, nG, tile, s0, s1, ws, ds, lxs, lane)
                          : hs_table_item<true>(tb, nG, tile, s0, s1, ws, ds, lxs, lane);
        part_t[ix * 64 + lane] = acc;
      } else if (SYN) {
        // 64 (live energy, chunk) pairs of the synchrotron integrand
        if (!syn_ready) {  // (wave-uniform) the tile waves' constants must have landed
          while (__atomic_load_n(&hi[HI_READY], __ATOMIC_RELAXED) < syn_tiles)
            __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          syn_ready = true;
        }
        if (S2) {
          const int nEs = H.syn_nE;
          const double* hdr = sm + H.o_s2;
          hs_syn2_par S;
          S.lm = s2lm; S.P = s2P; S.nG = H.nG[H.syn_grid]; S.pad = 0;
          S.ilx = hdr[0]; S.th = hdr[1]; S.im = hdr[2]; S.lml = hdr[3];
          hs_syn2_item(ix, lane, nA, Cd, nEs, S, reinterpret_cast<const int*>(sm + H.o_amap) + nEs,
                       reinterpret_cast<const int*>(sm + o_s2z), sm + o_s2q,
                       hs_lds_addr(sm + o_s2lw + HS_S2_GUARD), hs_lds_addr(sm + o_s2ig + HS_S2_GUARD),
                       hs_lds_addr(sm + H.o_s2 + HS_S2_HDR), hs_lds_addr(sm + o_s2t), part_s);
        } else {
          const int g = H.syn_grid, nEs = H.syn_nE;
          const hs_syn_lds L = {reinterpret_cast<const int*>(sm + H.o_amap), sm + H.o_ig2,
                                sm + H.o_dig2, sm + H.o_ig23, sm + H.o_w[g], sm + H.o_d[g],
                                sm + H.o_lx[g], sm + H.o_sq, sm + HS_O_T64};
          hs_syn_item(ix, lane, nA, Cd, H.nG[g], nEs, L, part_s);
        }
      }
    }
    if (dbg_on && j == 0 && lane == 0) {
      D.dbg[128 + wv] = (long long)wall_clock64();
      D.dbg[144 + wv] = dbg_nt;
      D.dbg[160 + wv] = dbg_ns;
    }
    if (dbg_on && j < 1024 && lane == 0) {
      D.dbg[18688 + (j * 16 + wv) * 3] = (long long)wall_clock64();
      D.dbg[18688 + (j * 16 + wv) * 3 + 1] = dbg_nt;
      D.dbg[18688 + (j * 16 + wv) * 3 + 2] = dbg_ns | (nA << 8) | (Cd << 20);
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
        const double* pp = sm + H.o_part_t + (tb.item0 + tile) * 64 + ln;
        const int stride = tb.tiles * 64, chunks = HS_CHUNKS(tb.chunks);
        double sum = 0.0;
        for (int c0 = 0; c0 < chunks; c0 += 8) {  // eight partial sums in flight, fixed order
          double v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = c0 + q < chunks ? pp[(c0 + q) * stride] : 0.0;
          sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        sum *= sm[H.o_scale + tb.spec_off + k];
        spec[tb.spec_off + k] = sum;
        if (K == 1) tb.out[(long long)j * tb.ldo + k] = sum;
      }
    }
    if (has_syn) {
      const int nEs = H.syn_nE;
      const int* amap = reinterpret_cast<const int*>(sm + H.o_amap);
      for (int a = T - 1 - tid; a < nA; a += T) {  // (from the back: the tables took the front)
        const double* pp = sm + H.o_part_s + a;
        double sum = 0.0;
        for (int c0 = 0; c0 < Cd; c0 += 8) {
          double v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = c0 + q < Cd ? pp[(c0 + q) * nEs] : 0.0;
          sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        sum *= NH_ERG_PER_EV;  // 1/(s erg) -> 1/(s eV), :340
        spec[H.syn_spec_off + amap[a]] = sum;
      }
    }
  }
  __syncthreads();
  HS_STAMP(8);
  if (K > 1) {
    // ---- 6b. the K workgroups of this walker meet: partial spectra out (write-through), one
    // ticket each; whoever draws the last one sums all K partials in the order of their index
    // (the result does not depend on who arrived when) and carries on.  Nobody waits.
    // (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores, drained, before the ticket;
    // sc1 loads behind it.  Nobody has read these lines earlier in this launch.)
    unsigned long long* xs = reinterpret_cast<unsigned long long*>(D.xspec) +
                             ((long long)j * K + part) * D.nspec;
    for (int k = tid; k < D.nspec; k += T)
      __hip_atomic_store(xs + k, (unsigned long long)__double_as_longlong(spec[k]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      hi[HI_TICK] = __hip_atomic_fetch_add(D.tick + j, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (hi[HI_TICK] != K - 1) {  // (the whole workgroup)
      if (tid == 0) {
        const int through = atomicAdd(H.done, 1) + 1;
        if ((flags_ & 4) && through % (int)(gridDim.x * gridDim.y) == 0) nh_clk_close(clk_);
      }
      return;
    }
    if (tid == 0) D.tick[j] = 0;  // for the next launch
    const unsigned long long* xa = reinterpret_cast<const unsigned long long*>(D.xspec) +
                                   (long long)j * K * D.nspec;
    for (int k = tid; k < D.nspec; k += T) {
      double sum = 0.0;
      for (int q = 0; q < K; ++q)
        sum += __longlong_as_double((long long)__hip_atomic_load(
            xa + (long long)q * D.nspec + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      spec[k] = sum;
    }
    __syncthreads();
    for (int t = 0; t < D.ntab; ++t) {
      const hs_tab& tb = D.tab[t];
      for (int k = tid; k < tb.nK; k += T) tb.out[(long long)j * tb.ldo + k] = spec[tb.spec_off + k];
    }
  }
  if (has_syn) {
    // (energies [n1, nE) of a component that stands for two Synchrotron.flux calls go to a second
    // array that follows the first: nh_hs.h, syn_ldo)
    // (gridDim.x = the launch's walkers: H.nloc, read HERE, made the compiler copy the whole
    // descriptor to scratch -- scripts/hs_usage.sh)
    const int ldo = D.syn_ldo & 0xFFFFF, n1 = D.syn_ldo >> 20;
    const long long second = (long long)gridDim.x * ldo + (long long)j * (H.syn_nE - n1) - n1;
    for (int k = tid; k < H.syn_nE; k += T) {
      const long long at = (n1 > 0 && k >= n1) ? second + k : (long long)j * ldo + k;
      D.syn_out[at] = spec[H.syn_spec_off + k];
    }
  }
  // ---- 7. likelihood + priors (core.py:64-121) and the accept, one wave ---------------------
  if (lik_wave) {
    const int nE = H.nE;
    const bool has_prior = D.lp || reinterpret_cast<const nh_prior_pack*>(sm + H.o_pri)->n > 0;
    const double prior = accs[3];  // (evaluated beside the weights, see above)
    double acc = 0.0;
    int nviol = 0, nul = 0;
    auto column = [&](int k, double conv, double f, double elo, double ehi, int ul) {
      double m = 0.0;
      for (int q = 0; q < D.ncomp; ++q) {
        const double v = D.comp[q].off >= 0 ? spec[D.comp[q].off + k]
                                            : D.comp[q].ptr[(long long)j * D.comp[q].ld + k];
        m += D.comp[q].scale * v;
      }
      if (D.model_out) D.model_out[(long long)j * nE + k] = m;
      if (D.nblob) sm[D.o_mrow + k] = m;
      const double mc = m * conv;
      if (ul) {
        nul += 1;
        nviol += (mc > f) ? 1 : 0;
      } else {
        const double d = mc - f;
        const double sg = (d > 0.0) ? ehi : elo;
        acc += -(d * d) / (2.0 * (sg * sg));
      }
    };
    {
      const double* lik = sm + H.o_lik;
      for (int k = lane; k < nE; k += 64)
        column(k, lik[k], lik[nE + k], lik[2 * nE + k], lik[3 * nE + k], lik[4 * nE + k] != 0.0);
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
      if (nul > 0) acc += (double)nviol * log(1.0 - H.cl[nviol]);
      if (has_prior) acc = isinf(prior) ? prior : acc + prior;  // core.py:115-119
      D.total[(long long)j * (D.sendw > 0 ? D.sendw : 1)] = acc;
    }
    if (D.sendw > 0) {  // sharded loop: the blobs travel with the log-probability
      double* rowp = D.total + (long long)j * D.sendw + 1;
      for (int b = 0; b < D.nblob; ++b) {
        const nh_hs_blob& bl = D.blob[b];
        if (bl.kind == 0) {
          for (int t = lane; t < bl.m; t += 64) rowp[t] = sm[D.o_mrow + t];
        } else if (lane == 0) {
          rowp[0] = nh_lazy_apply(bl.lazy, sm[D.o_mrow + H.nE + bl.mom]);
        }
        rowp += bl.m;
      }
    }
    if (D.do_accept) {  // emcee RedBlueMove.propose for this walker
      acc = __shfl(acc, 0, 64);
      const double z = accs[0];
      const double d = (H.ndim - 1.0) * log(z) + acc - accs[2];
      const bool ok = accs[1] < d;  // NaN compares false, as numpy
      const int me2 = hi[HI_ME];
      if (ok) {
        for (int t = lane; t < H.ndim; t += 64)
          const_cast<double*>(H.coords)[(long long)me2 * H.ndim + t] = qs[t];
        for (int b = 0; b < D.nblob; ++b) {  // the accepted position's blobs
          const nh_hs_blob& bl = D.blob[b];
          if (bl.kind == 0) {
            for (int t = lane; t < bl.m; t += 64) bl.cur[(long long)me2 * bl.m + t] = sm[D.o_mrow + t];
          } else if (lane == 0) {
            bl.cur[me2] = nh_lazy_apply(bl.lazy, sm[D.o_mrow + H.nE + bl.mom]);
          }
        }
      }
      if (lane == 0) {
        const int g = H.lo + j;
        if (ok) {
          const_cast<double*>(H.logp)[me2] = acc;
          if (D.naccepted) D.naccepted[me2] += 1;
        }
        D.accepted[g] = ok ? 1 : 0;
        if (D.sel) D.sel[g] = me2;
        // emcee raises "Probability function returned NaN" here; the launch cannot, it counts
      }
    }
    // ---- 8. one more workgroup is through (nobody waits for the answer) -------------------
    if (lane == 0) {
      const int through = atomicAdd(H.done, 1) + 1;
      if ((flags_ & 4) && through % (int)(gridDim.x * gridDim.y) == 0) nh_clk_close(clk_);
      // emcee raises "Probability function returned NaN" here; the launch cannot, it counts
      if (acc != acc) atomicAdd(H.done + 2, 1);
      if (hi[HI_DEAD]) atomicAdd(H.done + 3, 1);  // (forbidden by the prior: nothing was integrated)
    }
    if (dbg_on && lane == 0 && j < 1024) D.dbg[256 + 1024 + j] = (long long)wall_clock64();
  }
  HS_STAMP(9);
}

// blob history row `row` := the current blobs of the whole ensemble
__global__ void k_blob_hist_append(hs_hot H, long long row) {
  const hs_dev& D = H.C;
  const long long N = 2LL * H.ns;
  for (int b = 0; b < D.nblob; ++b) {
    const nh_hs_blob& bl = D.blob[b];
    double* hb = bl.hist ? *bl.hist : nullptr;
    if (!hb || row < 0) continue;
    const long long n = N * bl.m;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (long long)gridDim.x * blockDim.x)
      hb[row * n + t] = bl.cur[t];
  }
}

extern "C" int nh_half_step_append_blobs(nh_ctx* c, const nh_halfstep_plan* P, long long row) {
  NH_REQUIRE(c && P, "bad argument");
  if (P->hot.C.nblob == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_GLUE);
  hipLaunchKernelGGL(k_blob_hist_append, dim3(64), dim3(256), 0, c->stream, P->hot, row);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// KD[i][k] = {Kt[i][k], dlnKt[i][k]}: the layout the half-step kernel streams (one 16-byte load
// per node).  Where the table changes sign between nodes i and i+1 the log-ratio is stored as
// NaN: the reference's b = log10(y2/y1)/log10(x2/x1) is NaN there and the segment takes its log
// branch (utils.py:336-345); the sign pattern belongs to the table, not to the walker (the
// weights multiply both nodes by numbers of one sign), so it is settled here, once.
__global__ void k_table_interleave(const double* __restrict__ Kt, const double* __restrict__ dlnKt,
                                   const double* __restrict__ lx, int nG, int nK,
                                   double* __restrict__ KD) {
  const long long n = (long long)nG * nK;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double k1 = Kt[i];
    double d = dlnKt[i];
    if (i + nK < n) {
      const double k2 = Kt[i + nK];
      if (k1 != 0.0 && k2 != 0.0 && ((__double2hiint(k1) ^ __double2hiint(k2)) < 0)) d = NAN;
      // a non-negative table: log-ratios in units of the segment's lx (hs_seg_pre); the zero
      // marker NH_DL_ZERO stays what it is (its reciprocal has to underflow to 0)
      if (lx && d < NH_DL_ZERO) d /= lx[i / nK];
    }
    KD[2 * i] = k1;
    KD[2 * i + 1] = d;
  }
}

extern "C" int nh_table_interleave(nh_ctx* c, const double* Kt, const double* dlnKt,
                                   const double* lx, int nG, int nK, double* KD) {
  NH_REQUIRE(c && Kt && dlnKt && KD && nG >= 0 && nK >= 0, "bad argument");
  const long long n = (long long)nG * nK;
  if (n == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_TABLES);
  hipLaunchKernelGGL(k_table_interleave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                     Kt, dlnKt, lx, nG, nK, KD);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// the synchrotron grid's constants: 1/gamma^2, its cube root (x^(1/3) = q^(1/3) ig23), and
// 1/g2^2 - 1/g1^2 = (1/g1^2) (exp(-2 ln(g2/g1)) - 1) without cancellation -- the same for every
// walker and every launch: computed once per plan (the kernel used to spend 1.3 us of every
// launch on them, all SIMDs busy, before its first barrier)
__global__ void k_syn_consts(const double* __restrict__ gam, const double* __restrict__ lx, int nG,
                             double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nG) return;
  const double gi = gam[i];
  const double v = 1.0 / (gi * gi);
  out[i] = v;
  out[nG + i] = cbrt(v);
  out[2 * nG + i] = i + 1 < nG ? v * expm1(-2.0 * lx[i]) : 0.0;
}

// word := value, stream-ordered (the slice bookkeeping of nh_half_step_begin_block)
__global__ void k_set_word(int* p, int v) { *p = v; }
__global__ void k_set_word2(int* p, int v0, int v1) {
  p[0] = v0;
  p[1] = v1;
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

static int hs_create(nh_ctx* c, const nh_hs_desc* d, nh_halfstep_plan** out, int kmax,
                     int segscale, bool* lds_overflow) {
  NH_REQUIRE(c && d && out, "NULL pointer");
  NH_REQUIRE(d->coords && d->logp && d->blk && d->cursor && d->qT && d->factors && d->params &&
                 d->total, "NULL pointer in the descriptor");
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
  static_assert(sizeof(hs_hot) <= 3800, "the by-value kernel argument must fit the 4 KB segment");
  hs_hot H;
  memset(&H, 0, sizeof(H));
  hs_dev& C = H.C;
  nh_pack packs_host[NH_MAX_PACK];
  memset(packs_host, 0, sizeof(packs_host));
  H.coords = d->coords; H.logp = d->logp; H.blk = d->blk; H.cursor = d->cursor;
  H.qT = d->qT; H.factors = d->factors; H.hist = d->hist;
  H.ns = d->ns; H.ndim = d->ndim; H.lo = d->lo; H.nloc = d->nloc;
  C.npk = d->npacks; C.kind = d->kind; C.params = d->params;
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
    packs_host[q] = pk;
  }
  NH_REQUIRE(have_params, "params must be the output of one of the packs");
  NH_REQUIRE(d->ndim < 255, "at most 254 fit parameters");
  for (int t = 0; t < d->ntab; ++t) {  // (checked before anything indexes grids[] with it)
    const nh_hs_table& tb = d->tab[t];
    NH_REQUIRE(tb.grid >= 0 && tb.grid < d->ngrids && tb.KD && tb.out && tb.nK >= 1 &&
                   tb.ldo >= tb.nK, "bad table reduction");
  }
  if (getenv("NH_HS_TEST_REJECT"))  // (test hook: a plan this function turns down)
    return nh_set_error(NH_EINVAL, "half-step plan rejected (NH_HS_TEST_REJECT)");
  H.ngrids = d->ngrids;
  int off = HS_O_FREE;
  for (int g = 0; g < d->ngrids; ++g) {
    const nh_grid& gr = d->grids[g];
    NH_REQUIRE(gr.e_eV && gr.xg && gr.nG >= 2 && (!d->write_weights || (gr.w && gr.dlw)),
               "bad grid descriptor");
    NH_REQUIRE(gr.ln_e && gr.lx, "the half-step kernel needs the grids' ln_e and lx arrays");
    H.e[g] = gr.e_eV; H.xg[g] = gr.xg; H.lne[g] = gr.ln_e; H.lx[g] = gr.lx;
    H.scale[g] = gr.unit_scale; H.nG[g] = gr.nG;
    C.w[g] = gr.w; C.dlw[g] = gr.dlw;
    H.o_w[g] = off; off += gr.nG;
    H.o_d[g] = off; off += gr.nG;
    H.o_lx[g] = off; off += gr.nG;
    H.o_dp[g] = H.o_th[g] = -1;
    for (int t = 0; t < d->ntab; ++t)
      if (d->tab[t].grid == g && d->tab[t].nonnegative) {
        H.o_dp[g] = off; off += gr.nG;
        H.o_th[g] = off; off += gr.nG;
        break;
      }
  }
  H.o_mkt = off;
  H.nmom = d->nmoms;
  for (int m = 0; m < d->nmoms; ++m) {
    NH_REQUIRE(d->moms[m].grid >= 0 && d->moms[m].grid < d->ngrids && d->moms[m].Kt &&
                   d->moms[m].dlnKt && d->moms[m].out, "bad reduction");
    H.mKt[m] = d->moms[m].Kt; H.mdK[m] = d->moms[m].dlnKt; H.mgrid[m] = d->moms[m].grid;
    C.mom_out[m] = d->moms[m].out;
    off += 2 * d->grids[d->moms[m].grid].nG;
  }
  C.accepted = d->accepted; C.naccepted = d->naccepted; C.sel = d->sel;
  C.do_accept = d->do_accept; C.write_weights = d->write_weights;
  // ---- synchrotron ----
  H.syn_grid = -1;
  int nspec = 0;
  int syn_nE = 0;
  if (d->syn.grid >= 0) {
    const nh_hs_syn& s = d->syn;
    NH_REQUIRE(s.grid < d->ngrids && s.E_eV && s.out && s.nE >= 1 && s.ldo >= (s.out2 ? s.n1 : s.nE) &&
                   (s.bcol >= 0 ? s.bcol < NH_PD_NPAR : (s.B != nullptr && s.ldB >= 1)),
               "bad synchrotron component");
    const int nG = d->grids[s.grid].nG;
    NH_REQUIRE(s.nE <= 512, "at most 512 photon energies for the synchrotron component");
    H.syn_grid = s.grid; H.syn_E = s.E_eV; H.syn_nE = syn_nE = s.nE;
    C.synB = s.B; C.syn_out = s.out; C.syn_ldo = s.ldo; C.syn_bcol = s.bcol; C.syn_ldB = s.ldB;
    NH_REQUIRE(s.out2 == nullptr || (s.n1 >= 1 && s.n1 < s.nE && s.ldo >= s.n1 && s.ldo2 >= s.nE - s.n1),
               "bad split of the synchrotron component's output");
    NH_REQUIRE(s.out2 == nullptr || (s.out2 == s.out + (long long)d->nloc * s.ldo && s.ldo2 == s.nE - s.n1 &&
                                     s.ldo < (1 << 20) && s.n1 < (1 << 11)),
               "the synchrotron component's second array must follow its first (out + nloc * ldo), rows of nE - n1");
    if (s.out2) C.syn_ldo = s.ldo | (s.n1 << 20);
    H.o_ig2 = off; off += nG;
    H.o_dig2 = off; off += nG;
    H.o_ig23 = off; off += nG;
    H.o_sq = off; off += 3 * s.nE;
    H.o_amap = off; off += s.nE + 1;  // 2 nE ints
    int cdmax = (int)(((size_t)nh_env_int("NH_HS_PART_KB", 40) * 1024) / ((size_t)s.nE * 8));  // (partial sums: 40 KB of LDS at most)
    cdmax = cdmax > 32 ? 32 : (cdmax < 1 ? 1 : cdmax);
    C.syn_cdmax = cdmax;
    H.o_part_s = off; off += cdmax * s.nE;
    H.syn_spec_off = nspec;
    nspec += s.nE;
  }
  // workgroup size: small reductions are bound by the dependent round trips, not by lanes
  int maxnG = 0;
  for (int g = 0; g < d->ngrids; ++g) maxnG = d->grids[g].nG > maxnG ? d->grids[g].nG : maxnG;
  long long work = 0;
  for (int t = 0; t < d->ntab; ++t)
    work += (long long)((d->tab[t].nK + 63) / 64) * 64 * d->grids[d->tab[t].grid].nG * 12;
  if (d->syn.grid >= 0) work += (long long)d->syn.nE * d->grids[d->syn.grid].nG * 110 / 3;
  int threads = work >= (1 << 20) ? 1024 : (work >= (1 << 18) ? 512 : 256);
  if (threads < 1024 && maxnG > threads) threads = maxnG > 512 ? 1024 : 512;
  int ncu = 256;
  {
    hipDeviceProp_t prop;
    int devid = 0;
    if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess &&
        prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
  }
  // (several ranks rehearsing a multi-GPU run on ONE device -- bench.py --gpus N under
  // NAIMA_AMD_DEVICE -- each plan for their share of its CUs: resident workgroups of all ranks
  // have to fit the chip together)
  {
    const int share = nh_env_int("NAIMA_AMD_CU_SHARE", 1);
    if (share > 1) ncu = ncu / share > 1 ? ncu / share : 1;
  }
  // A table-only model's launch is half prologue and tail (dependent round trips, single-wave
  // phases: 9 of cfg5's 16 us) with the CU's other waves idle.  When a launch holds more
  // walkers than the chip has CUs, smaller workgroups put several walkers on a CU at once and
  // one walker's prologue runs beside another's items: cfg5 at 1024 walkers per launch 66.8 us
  // (1024 threads, four rounds) -> 51.6 (512) -> 37.0 (256 threads, four workgroups per CU).
  // Launches of at most one walker per CU keep the large workgroup (256 walkers: 15.6 us at
  // 1024 threads, 16.7 at 512, 24.5 at 256).
  if (d->syn.grid < 0)
    while (threads > 256 && (long long)d->nloc * threads > 1024LL * ncu) threads >>= 1;
  if (const char* e = getenv("NH_HS_THREADS")) threads = atoi(e);
  NH_REQUIRE(threads >= 128 && threads <= 1024 && threads % 64 == 0, "bad workgroup size");
  if (d->syn.grid >= 0 && threads / 64 < (d->syn.nE + 63) / 64 + 2) threads = 1024;
  while (threads < 1024 && threads / 64 <= d->nmoms + 1) threads <<= 1;  // (a wave per single-row reduction)
  NH_REQUIRE(threads / 64 > d->nmoms + 1, "more single-row reductions than waves");
  // ---- a launch of fewer walkers than the chip has CUs: K workgroups per walker ----
  // (each repeats the prologue -- on CUs that would otherwise idle -- and takes every K-th
  // work item; the items are cut finer so that every wave of every workgroup still gets some)
  int split = 1;
  {
    // (worth it where the work items are most of a launch: measured on cfg2 / cfg3 at 128
    // walkers per launch 42.3 -> 32.7 us and 33.8 -> 28.0 us; the hand-off costs ~3.5 us, which
    // the table-only models cfg1 / cfg5 -- 7 us of items in a 17 us launch -- do not get back)
    if (work >= nh_env_int("NH_HS_SPLIT_MIN_WORK", 1 << 20))
      while (split * 2 <= kmax && (long long)d->nloc * split * 2 <= ncu) split *= 2;
  }
  // ---- a table-only model with ONE table, at most half as many walkers per launch as
  // CUs (BASELINE's PionDecay fit at one GPU's share of its 2048 walkers): two workgroups per
  // walker that split the grid's ROWS -- each forms the weights of its half, reduces its half of
  // the table (from registers: half the rows fit a lane's) and of the single-row reductions, and
  // the second hands its partial spectrum to the first (the resident loop: hs_run.rowsplit; the
  // per-launch kernel runs the same plan with its items interleaved, as any split launch)
  bool rowsplit = false;
  if (split == 1 && d->syn.grid < 0 && d->ntab == 1 && kmax >= 2 &&
      (long long)d->nloc * 2 <= ncu && nh_env_int("NH_HS_ROWSPLIT", 1) != 0) {
    split = 2;
    rowsplit = true;
  }
  // ---- a table-only model whose items' rows fit a lane's registers: workgroups of 512 threads
  // (256 vector registers per lane), for the resident loop's register-resident items (nh_hs.h:
  // hs_rt_item; k_half_step_run<false, ., false, RT>).  The rows an item walks start at the first
  // one in which a column is non-zero (the resident loop's sorted copies: at least that far up).
  bool rt_plan = false;
  // (launches of at most one workgroup per CU: a workgroup of this instance has a CU to itself)
  if (d->syn.grid < 0 && d->ntab > 0 && threads >= 256 && (long long)d->nloc * split <= ncu &&
      nh_env_int("NH_HS_RT", 1) != 0) {
    bool fits = true;
    int tiles = 0;
    for (int t = 0; t < d->ntab; ++t) {
      tiles += (d->tab[t].nK + 63) / 64;
    }
    const int t_rt = threads > 512 ? 512 : threads;
    const int free_waves = (t_rt / 64 - d->nmoms) * split;
    fits = fits && free_waves >= 1 && tiles <= free_waves;
    const int per_tile = fits ? free_waves / tiles : 1;
    if (fits && nh_sync(c) != NH_OK) fits = false;
    for (int t = 0; t < d->ntab && fits; ++t) {
      const nh_hs_table& tb = d->tab[t];
      const int nG = d->grids[tb.grid].nG, nK = tb.nK;
      if ((size_t)nG * nK * 16 > ((size_t)64 << 20)) {  // (nothing of that size fits a lane's registers)
        fits = false;
        break;
      }
      std::vector<double> kd((size_t)nG * nK * 2);
      if (hipMemcpy(kd.data(), tb.KD, kd.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        fits = false;
        break;
      }
      int r0 = nG - 1;
      for (int i = 0; i < nG - 1 && r0 == nG - 1; ++i)
        for (int k = 0; k < nK; ++k)
          if (kd[((size_t)i * nK + k) * 2] != 0.0) { r0 = i; break; }
      const int sub = nK <= 32 ? (nK <= 16 ? (nK <= 8 ? (nK <= 4 ? (nK <= 2 ? (nK <= 1 ? 64 : 32) : 16) : 8) : 4) : 2) : 1;
      const int per = (nG - 1 - r0 + per_tile - 1) / per_tile;
      fits = (per + sub - 1) / sub + 1 <= HS_RT_NODES;
      if (getenv("NH_HS_RT_DEBUG"))
        fprintf(stderr, "RT: table %d nG %d nK %d r0 %d per_tile %d per %d sub %d nodes %d fits %d\n", t, nG, nK, r0,
                per_tile, per, sub, (per + sub - 1) / sub + 1, (int)fits);
    }
    if (getenv("NH_HS_RT_DEBUG")) fprintf(stderr, "RT: fits %d threads %d nmoms %d split %d\n", (int)fits, threads, d->nmoms, split);
    if (fits) {
      threads = t_rt;
      rt_plan = true;
    }
  }
  // ---- table reductions: work items of `seg` segments x 64 columns ----
  // With a synchrotron component the items interleave with its (issue-bound) items and 32
  // segments keep the waves evenly loaded; without one, ONE round of equal items over the
  // waves that are free from the start (not on a single-row reduction) ends soonest.
  C.ntab = d->ntab;
  // (48 segments per item where a walker has one workgroup: with the synchrotron items in the log
  // domain -- half the instructions they were -- the table items are the larger share of the
  // resident loop's work, and a third fewer of them means a third less set-up: cfg3 10.6 -> 11.0 M
  // walker-steps/s together with 48-node synchrotron items; 64: 10.9, 96: 10.7, 128: 10.3.  The
  // per-launch kernel loses 2 % to it.)
  int seg = split >= 4 ? 8 : (split == 2 ? 16 : (d->syn.grid >= 0 ? 48 : 32));
  if (split > 1) seg *= segscale;  // (coarser items: fewer partial sums in LDS)
  C.syn_nodes = HS_SYN_NODES / split > 3 ? HS_SYN_NODES / split : 3;
  if (const char* e = getenv("NH_HS_SEG")) seg = atoi(e) >= 4 ? atoi(e) : seg;  // (tuning experiments)
  if (const char* e = getenv("NH_HS_SYN_NODES")) C.syn_nodes = atoi(e) >= 1 ? atoi(e) : C.syn_nodes;
  if (d->syn.grid < 0 && d->ntab > 0) {
    int tiles = 0, maxseg = 0;
    for (int t = 0; t < d->ntab; ++t) {
      tiles += (d->tab[t].nK + 63) / 64;
      const int nseg = d->grids[d->tab[t].grid].nG - 1;
      maxseg = nseg > maxseg ? nseg : maxseg;
    }
    const int free_waves = (threads / 64 - d->nmoms) * split;
    int per_tile = free_waves / tiles > 1 ? free_waves / tiles : 1;
    per_tile *= nh_env_int("NH_HS_TAB_ROUNDS", 1);  // (tuning experiments: that many rounds of shorter items)
    seg = (maxseg + per_tile - 1) / per_tile;
    if (seg < 8) seg = 8;
  }
  // NH_HS_GRADE=1: the last third of a table's segments in chunks of half the length
  const bool grade = nh_env_int("NH_HS_GRADE", 0) != 0 && d->syn.grid >= 0;
  auto chunking = [&](int nseg, int sg, int& nfull, int& seg2) {
    nfull = (nseg + sg - 1) / sg;
    seg2 = sg;
    if (grade && sg >= 8) {
      nfull = (2 * nseg / 3) / sg;
      seg2 = sg / 2;
    }
    const int rest = nseg - nfull * sg;
    return nfull + (rest > 0 ? (rest + seg2 - 1) / seg2 : 0);
  };
  for (;;) {
    int nT = 0;
    for (int t = 0; t < d->ntab; ++t) {
      const int tiles = (d->tab[t].nK + 63) / 64;
      const int nseg = d->grids[d->tab[t].grid].nG - 1;
      int nf, s2;
      nT += tiles * chunking(nseg, seg, nf, s2);
    }
    if (nT <= (split > 1 ? 160 : (grade ? 128 : 96))) break;
    seg *= 2;
  }
  C.seg = seg;
  int nT = 0;
  for (int t = 0; t < d->ntab; ++t) {
    const nh_hs_table& tb = d->tab[t];
    const int nG = d->grids[tb.grid].nG;
    NH_REQUIRE((long long)nG * tb.nK < (1LL << 27), "table too large for 32-bit offsets");
    hs_tab& o = C.tab[t];
    o.KD = tb.KD; o.scale = tb.scale;
    H.tscale[t] = tb.scale; H.tnK[t] = tb.nK; o.out = tb.out; o.grid = tb.grid;
    o.nK = tb.nK; o.ldo = tb.ldo; o.nonneg = tb.nonnegative;
    o.tiles = (tb.nK + 63) / 64;
    o.nKp = 64;
    o.sub = 1;
    if (tb.nK <= 32) {
      o.nKp = 32;
      while (o.nKp / 2 >= tb.nK && o.nKp > 1) o.nKp /= 2;
      o.sub = 64 / o.nKp;
    }
    int nfull = 0, seg2 = 0;
    const int nchunks = chunking(nG - 1, seg, nfull, seg2);
    NH_REQUIRE(nchunks < 256 && nfull < 256 && seg2 < 32768, "table cut into too many chunks");
    o.chunks = nchunks | (nfull << 8) | (seg2 << 16);
    o.item0 = nT;
    nT += o.tiles * nchunks;
    o.spec_off = nspec;
    H.tspec[t] = nspec;
    nspec += tb.nK;
  }
  C.nT = nT;
  H.o_part_t = off; off += nT * 64;
  H.o_spec = off; off += nspec;
  C.nspec = nspec;
  H.o_lik = off; off += 5 * d->nE;
  H.o_scale = off; off += nspec;
  H.o_synE = off; off += syn_nE;
  H.o_pri = off; off += (int)(sizeof(nh_prior_pack) / sizeof(double)) + 1;
  H.ntab = d->ntab;
  NH_REQUIRE(d->nblobs >= 0 && d->nblobs <= NH_HS_MAX_BLOB, "bad blob count");
  C.nblob = d->nblobs;
  C.sendw = d->do_accept ? 0 : d->send_width;
  if (C.sendw > 0) {
    int wsum = 1;
    for (int b = 0; b < d->nblobs; ++b) wsum += d->blobs[b].m;
    NH_REQUIRE(C.sendw >= wsum, "send_width smaller than 1 + the blobs' lengths");
  }
  C.o_mrow = off; off += d->nE + NH_MAX_MOMENT;
  for (int b = 0; b < d->nblobs; ++b) {
    const nh_hs_blob& bl = d->blobs[b];
    NH_REQUIRE(d->do_accept || d->send_width > 0,
               "blobs in the launch need the in-launch accept or an exchange row");
    NH_REQUIRE(bl.cur && ((bl.kind == 0 && bl.m == d->nE) ||
                          (bl.kind == 1 && bl.m == 1 && bl.mom >= 0 && bl.mom < d->nmoms)),
               "bad blob");
    C.blob[b] = bl;
  }
  // ---- likelihood: where does each component of the model live? ----
  C.ncomp = d->ncomp; H.nE = d->nE;
  for (int q = 0; q < d->ncomp; ++q) {
    const nh_comp& cp = d->comps[q];
    NH_REQUIRE(cp.ptr && cp.ld >= d->nE, "bad component");
    hs_comp& o = C.comp[q];
    o.ptr = cp.ptr; o.ld = cp.ld; o.scale = cp.scale; o.off = -1;
    for (int t = 0; t < d->ntab && o.off < 0; ++t) {
      const long long dd = cp.ptr - d->tab[t].out;
      if (dd >= 0 && dd + d->nE <= d->tab[t].nK && cp.ld == d->tab[t].ldo)
        o.off = C.tab[t].spec_off + (int)dd;
    }
    if (o.off < 0 && d->syn.grid >= 0) {
      const long long dd = cp.ptr - d->syn.out;
      if (dd >= 0 && dd + d->nE <= syn_nE && cp.ld == d->syn.ldo) o.off = H.syn_spec_off + (int)dd;
    }
  }
  H.conv = d->conv; H.flux = d->flux; H.elo = d->err_lo; H.ehi = d->err_hi; H.ul = d->ul;
  H.cl = d->cl;
  C.lp = d->lp; C.model_out = d->model_out; C.total = d->total;
  C.pri.n = d->nterms;
  for (int t = 0; t < d->nterms; ++t) C.pri.t[t] = d->terms[t];
  // ---- the synchrotron items in the log domain (nh_syn2.h), as the resident loop runs them: a
  // log-uniform grid (np.logspace, radiative.py:147-154), its table and node constants behind the
  // grid's three arrays in syn_c, their block in LDS behind o_s2 (k_half_step<true, true>)
  hs_s2_host s2h;
  H.o_s2 = 0;
  const size_t lds_core = (size_t)off * sizeof(double);  // (what the resident loop builds on: it has a block of its own)
  if (d->syn.grid >= 0 && nh_env_int("NH_HS_SYN2", 1) != 0) {
    const int sg = d->syn.grid, nGs = d->grids[sg].nG;
    std::vector<double> gam((size_t)nGs);
    bool ok = nh_sync(c) == NH_OK;
    if (ok && hipMemcpy(gam.data(), d->grids[sg].xg, (size_t)nGs * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
      (void)hipGetLastError();
      ok = false;
    }
    ok = ok && hs_s2_prepare(gam.data(), nGs, d->grids[sg].unit_scale, s2h);
    if (getenv("NH_HS_RT_DEBUG"))
      fprintf(stderr, "S2: grid %d nG %d nE %d log-uniform %d off %d (%.1f KB) split %d\n", sg, nGs, d->syn.nE, (int)ok, off,
              off * 8.0 / 1024, split);
    if (ok) {
      const int o = off + (off & 1);  // (the pieces are read as ds_read_b128)
      const int need = HS_S2_HDR + (s2h.par.P + 1) * HS_S2_STRIDE + HS_S2_TN + nGs + (nGs + 2 * HS_S2_GUARD) +
                       4 * d->syn.nE + (d->syn.nE + 1) / 2 + 1;
      if ((size_t)(o + need) * sizeof(double) <= 160 * 1024) {  // (a CU's LDS; the plan's own layout keeps to 150 KB)
        H.o_s2 = o;
        off = o + need;
        // (items that start on a piece boundary with a node and six coefficients of their own:
        // short ones pay for that -- the resident loop's lengths)
        // (cfg3 / 512, one launch per half-step: 30.6 us at 16 or 24 nodes per item, 31.1 at 32,
        // 33.9 at 48, 32.7 in the direct form)
        if (!getenv("NH_HS_SYN_NODES")) C.syn_nodes = split == 1 ? 24 : (C.syn_nodes < 16 ? 16 : C.syn_nodes);
      }
    }
  }
  const size_t lds = (size_t)off * sizeof(double);
  if (lds_core > 150 * 1024) *lds_overflow = true;
  NH_REQUIRE(lds_core <= 150 * 1024, "the model's grids and tables do not fit in LDS");
  C.dbg = nullptr;
  if (const char* e = getenv("NH_HS_DEBUG"))
    if (atoi(e) != 0) {
      NH_CHECK_HIP(hipMalloc(&C.dbg, 67840 * sizeof(long long)));
      NH_CHECK_HIP(nh_fill_now(c, C.dbg, 0, 67840 * sizeof(long long)));
    }
  nh_halfstep_plan* P = new nh_halfstep_plan();
  P->dbg = C.dbg;
  P->span = 3;
  P->lds_bytes = lds;
  P->lds_core = lds_core;
  P->threads = threads;
  P->blocks = d->nloc;
  P->split = split;
  P->rowsplit = rowsplit ? 1 : 0;
  P->rt = rt_plan ? 1 : 0;
  P->dev = nullptr;
  P->words = nullptr;
  P->syn_c = nullptr;
  P->xspec = nullptr;
  P->tick = nullptr;
  hipError_t e = hipMalloc(&P->dev, sizeof(packs_host));
  if (e == hipSuccess && H.syn_grid >= 0) {
    const int nGs = H.nG[H.syn_grid];
    const size_t ns2 = H.o_s2 ? HS_S2_HDR + s2h.data.size() : 0;
    e = hipMalloc(&P->syn_c, (3 * (size_t)nGs + ns2) * sizeof(double));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_syn_consts, dim3((unsigned)((nGs + 255) / 256)), dim3(256), 0, c->stream,
                         H.xg[H.syn_grid], H.lx[H.syn_grid], nGs, P->syn_c);
      e = hipGetLastError();
    }
    if (e == hipSuccess && ns2) {
      std::vector<double> up(ns2, 0.0);
      up[0] = s2h.par.ilx; up[1] = s2h.par.th; up[2] = s2h.par.im; up[3] = s2h.par.lml;
      up[4] = s2h.invd; up[5] = s2h.z0;
      int* iv = reinterpret_cast<int*>(&up[6]);
      iv[0] = s2h.par.lm; iv[1] = s2h.par.P; iv[2] = s2h.par.nG; iv[3] = 0;
      for (size_t q = 0; q < s2h.data.size(); ++q) up[HS_S2_HDR + q] = s2h.data[q];
      e = nh_put_now(c, P->syn_c + 3 * (size_t)nGs, up.data(), ns2 * sizeof(double));
    }
  }
  if (e == hipSuccess && split > 1) {
    e = hipMalloc(&P->xspec, (size_t)d->nloc * split * nspec * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&P->tick, (size_t)d->nloc * sizeof(int));
    if (e == hipSuccess) e = nh_fill_now(c, P->tick, 0, (size_t)d->nloc * sizeof(int));
  }
  if (e == hipSuccess) e = hipMalloc(&P->words, 4 * sizeof(int));  // done | hbase | NaN proposals | -
  if (e == hipSuccess) e = nh_put_now(c, P->dev, packs_host, sizeof(packs_host));
  if (e == hipSuccess) e = nh_fill_now(c, P->words, 0, 4 * sizeof(int));
  if (e == hipSuccess && lds > 64 * 1024)
    e = H.syn_grid >= 0
            ? (H.o_s2 ? hipFuncSetAttribute((const void*)k_half_step<true, true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                      : hipFuncSetAttribute((const void*)k_half_step<true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            : hipFuncSetAttribute((const void*)k_half_step<false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    if (P->dev) (void)hipFree(P->dev);
    if (P->words) (void)hipFree(P->words);
    if (P->syn_c) (void)hipFree(P->syn_c);
    if (P->xspec) (void)hipFree(P->xspec);
    if (P->tick) (void)hipFree(P->tick);
    if (P->dbg) (void)hipFree(P->dbg);
    delete P;
    return nh_set_error(NH_EHIP, "half-step plan: %s", hipGetErrorString(e));
  }
  H.C.pk = P->dev;
  H.C.xspec = P->xspec;
  H.C.tick = P->tick;
  H.done = P->words;
  H.hbase = P->words + 1;
  {  // the first scalar trip's block (hs_first)
    hs_first& F = H.F;
    memset(&F, 0, sizeof(F));
    F.pk = P->dev; F.hbase = H.hbase; F.logp = H.logp; F.hist = H.hist;
    F.npk8 = d->npacks * NH_MAX_LAZY; F.ngrids = H.ngrids; F.nloc = H.nloc;
    F.qT = H.qT; F.factors = H.factors;
    F.syn_c = P->syn_c; F.syn_nG = H.syn_grid >= 0 ? H.nG[H.syn_grid] : 0;
    F.broken = (d->kind == NH_PD_BROKENPL || d->kind == NH_PD_ECBPL) ? 1 : 0;
    F.ppk = -1;
    for (int q = 0; q < d->npacks; ++q)
      if (packs_host[q].out == d->params) F.ppk = q;
    for (int g = 0; g < H.ngrids; ++g) {
      F.nG[g] = H.nG[g];
      F.e[g] = H.e[g]; F.xg[g] = H.xg[g]; F.lne[g] = H.lne[g]; F.lx[g] = H.lx[g];
    }
    for (int t = 0; t < 32; ++t) {
      unsigned b = 0xFFu;
      const int q = t / NH_MAX_LAZY, col = t % NH_MAX_LAZY;
      if (q < d->npacks && col < packs_host[q].ncols && packs_host[q].cols[col].base) {
        const long long off = packs_host[q].cols[col].base - H.qT;
        b = (unsigned)(off / H.nloc);  // (a proposal coordinate: checked above)
      }
      F.pkd[t >> 2] |= b << (8 * (t & 3));
    }
  }
  P->hot = H;
  *out = P;
  return NH_OK;
}

// (a split launch cuts the work items finer and needs more LDS for their partial sums: where
// that does not fit, fewer workgroups per walker)
extern "C" int nh_half_step_create(nh_ctx* c, const nh_hs_desc* d, nh_halfstep_plan** out) {
  int kmax = 8;
  if (const char* e = getenv("NH_HS_SPLIT")) kmax = atoi(e) > 0 ? atoi(e) : 1;
  for (;; kmax >>= 1) {
    for (int segscale = 1; segscale <= 4; segscale *= 2) {  // coarser items before fewer workgroups
      bool lds_overflow = false;
      const int rc = hs_create(c, d, out, kmax, segscale, &lds_overflow);
      if (rc == NH_OK || !lds_overflow) return rc;
      if (kmax <= 1) return rc;
    }
  }
}

// A new block of moves has been uploaded to `blk`: the next launch proposes its slice
// `first_slice` (0, unless the caller has already worked through the first slices of the block
// by other means).  steps_before = ensemble steps of this run completed before this block of
// moves (the history row of the block's first step).  Stream-ordered, no host synchronisation.
extern "C" int nh_half_step_begin_block(nh_ctx* c, nh_halfstep_plan* P, int first_slice,
                                        int steps_before) {
  NH_REQUIRE(c && P && first_slice >= 0 && steps_before >= 0, "bad argument");
  hipLaunchKernelGGL(k_set_word2, dim3(1), dim3(1), 0, c->stream, P->words,
                     first_slice * P->blocks * P->split, steps_before);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_half_step_launch(nh_ctx* c, nh_halfstep_plan* P, int slice) {
  NH_REQUIRE(c && P && P->dev, "bad argument");
  nh_prof_scope ps(c, NH_K_HALFSTEP);
  const hs_hot& H = P->hot;
  const dim3 grid((unsigned)P->blocks, (unsigned)P->split);
  if (H.syn_grid >= 0 && H.o_s2)
    hipLaunchKernelGGL((k_half_step<true, true>), grid, dim3(P->threads), P->lds_bytes, c->stream,
                       (const int*)H.done, H.blk, H.coords, slice, H.ns, H.ndim, H.lo,
                       (P->dbg ? 1 : 0) | (P->span << 1), H, c->clk);
  else if (H.syn_grid >= 0)
    hipLaunchKernelGGL(k_half_step<true>, grid, dim3(P->threads), P->lds_bytes, c->stream,
                       (const int*)H.done, H.blk, H.coords, slice, H.ns, H.ndim, H.lo,
                       (P->dbg ? 1 : 0) | (P->span << 1), H, c->clk);
  else
    hipLaunchKernelGGL(k_half_step<false>, grid, dim3(P->threads), P->lds_bytes, c->stream,
                       (const int*)H.done, H.blk, H.coords, slice, H.ns, H.ndim, H.lo,
                       (P->dbg ? 1 : 0) | (P->span << 1), H, c->clk);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// Which launches of a half-step open and close a span of the context's clock (nh_clock_read).
// The default: a launch of the plan is a span by itself.  A half-step of several launches -- the
// staged plan of a model that asks for its synchrotron spectrum twice, with the SSC seed
// integral's kernels in between -- opens with its first launch (open = 1, close = 0) and closes
// with its last (0, 1): the span then covers the kernels in between as well.
extern "C" int nh_half_step_span(nh_halfstep_plan* P, int open, int close) {
  NH_REQUIRE(P, "bad argument");
  P->span = (open ? 1 : 0) | (close ? 2 : 0);
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

extern "C" int nh_half_step_syn_form(const nh_halfstep_plan* P, int* form) {
  NH_REQUIRE(P && form, "bad argument");
  *form = P->hot.syn_grid < 0 ? 0 : (P->hot.o_s2 ? 2 : 1);
  return NH_OK;
}

extern "C" int nh_half_step_split(const nh_halfstep_plan* P, int* split) {
  NH_REQUIRE(P && split, "bad argument");
  *split = P->split;
  return NH_OK;
}

// NH_HS_DEBUG=1: the phase stamps (100 MHz wall clock) of the first 8 workgroups of the
// last launch, out[8][16], then for workgroup 0 per wave: end of its work items [16], table
// items taken [16], synchrotron items taken [16], start of its first item [16]
extern "C" int nh_half_step_stamps(nh_ctx* c, const nh_halfstep_plan* P, long long* out) {
  NH_REQUIRE(c && P && out, "bad argument");
  memset(out, 0, 67840 * sizeof(long long));
  if (!P->dbg) return NH_OK;
  int rc = nh_sync(c);
  if (rc) return rc;
  NH_CHECK_HIP(hipMemcpy(out, P->dbg, 67840 * sizeof(long long), hipMemcpyDeviceToHost));
  return NH_OK;
}

// proposals whose log-probability came out NaN since the plan was made (or since the last
// reset): emcee's EnsembleSampler.compute_log_prob raises ValueError("Probability function
// returned NaN") on the first one; a launch rejects the proposal (NaN compares false) and counts.
// Synchronises the stream.
extern "C" int nh_half_step_nan_count(nh_ctx* c, nh_halfstep_plan* P, int reset, int* count) {
  NH_REQUIRE(c && P && count, "bad argument");
  int rc = nh_sync(c);
  if (rc) return rc;
  NH_CHECK_HIP(hipMemcpy(count, P->words + 2, sizeof(int), hipMemcpyDeviceToHost));
  if (reset && *count) NH_CHECK_HIP(nh_fill_now(c, P->words + 2, 0, sizeof(int)));
  return NH_OK;
}

// the same, and beside it the proposals the prior forbade (core.py:99-101, 115-119: -inf; the
// launch evaluates none of their integrals, the reference evaluates the model and discards it).
// A negative *nan / *forbidden on entry SETS that counter to -value - 1 first (a replayed block of
// moves must not count its proposals twice).
extern "C" int nh_half_step_counts(nh_ctx* c, nh_halfstep_plan* P, int reset, int* nan, int* forbidden) {
  NH_REQUIRE(c && P && nan && forbidden, "bad argument");
  int rc = nh_sync(c);
  if (rc) return rc;
  int w[2];
  NH_CHECK_HIP(hipMemcpy(w, P->words + 2, sizeof(w), hipMemcpyDeviceToHost));
  if (*nan < 0 || *forbidden < 0) {
    if (*nan < 0) w[0] = -*nan - 1;
    if (*forbidden < 0) w[1] = -*forbidden - 1;
    NH_CHECK_HIP(nh_put_now(c, P->words + 2, w, sizeof(w)));
  }
  *nan = w[0];
  *forbidden = w[1];
  if (reset && (w[0] || w[1])) NH_CHECK_HIP(nh_fill_now(c, P->words + 2, 0, sizeof(w)));
  return NH_OK;
}

extern "C" int nh_half_step_destroy(nh_ctx* c, nh_halfstep_plan* P) {
  NH_REQUIRE(c, "ctx is NULL");
  if (!P) return NH_OK;
  int rc = nh_sync(c);
  if (P->dev) (void)hipFree(P->dev);
  if (P->syn_c) (void)hipFree(P->syn_c);
  if (P->xspec) (void)hipFree(P->xspec);
  if (P->tick) (void)hipFree(P->tick);
  if (P->words) (void)hipFree(P->words);
  if (P->dbg) (void)hipFree(P->dbg);
  delete P;
  return rc;
}
