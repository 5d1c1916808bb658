// nh_persist.hip -- a whole block of moves in ONE launch: the half-step kernel with its
// workgroups resident, walkers handed from half-step to half-step by per-walker records
// instead of by the kernel boundary.
//
// Why.  k_half_step (nh_halfstep.hip) carries one walker from the stretch-move proposal to
// the accept in one launch, and the launch boundary is the ensemble-wide barrier between
// half-steps (emcee RedBlueMove: the second half moves against the first half's NEW
// positions; reference call sites core.py:128, 450-457).  But a walker of half-step h + 1
// needs exactly TWO records -- its own and its partner's -- not the whole launch.  Paying
// the boundary meant, per half-step of cfg3 (256 workgroups, round-2 phase stamps): 1.45 us
// of boundary, two cold round trips (argument block, grids' arrays: 2.5 us) and 2 us of LDS
// refills before the first useful instruction, the SLOWEST of 256 workgroups (35.1 us against
// a median of 33.2) setting the pace of all, every XCD's L2 cold again (each pulled its own
// copy of the 1.1 MB emission table from the Infinity Cache: 80x the algorithmic traffic).
//
// Here gridDim.x workgroups (all co-resident: one per CU at 1024 threads) loop over the
// slices of the block of moves.  What does not depend on the walker is fetched ONCE per
// launch and stays in LDS: the grids' nodes, ln E and segment widths, the synchrotron grid's
// constants, the exp table, the likelihood's data columns, the prior terms, the parameter
// packs' descriptors.  The ensemble lives in a ring of rows, row k = the state after step
// k - 1 of the launch, one record per walker: 2 (ndim + 1) granules of 8 bytes
// { 32 bits of payload | 32-bit tag }, tag = (launch sequence number, row).  A granule is one
// naturally aligned 8-byte write-through (sc1) store, so whoever reads it sees all of it or
// none of it, and a stale or not-yet-written granule is RECOGNISED by its tag, never consumed
// (MI355X_MICROARCH.md, hand-off price list: data-tagged granules, ~1 us producer to
// consumer, no fence).  Every record is written once per launch, by the workgroup that moved
// the walker in that step; a consumer polls the 2 x 2 (ndim + 1) granules it needs with sc1
// loads from one wave.  There is no other synchronisation: dependencies point strictly
// backwards in (slice, walker) order and every workgroup works through its walkers in that
// order, so the launch cannot deadlock as long as all workgroups are resident (the host sizes
// the grid by the occupancy query); every poll loop is bounded and a time-out aborts the
// whole launch with a status word instead of hanging the device.
//
// Inside a workgroup wave 0 runs ahead: as soon as the spectra of slice h are summed it polls
// for the records of slice h + 1, proposes and evaluates the parameter packs, while wave 1
// finishes slice h (likelihood, accept, publish, history).  The small per-walker block of LDS
// (proposal, particle row, counters) is double-buffered for that overlap.
//
// Outputs: the ring (its last row is converted back to the flat coords / logp arrays by
// k_run_epilogue), the chain history rows (each entry written once, by the walker's mover),
// the accept flags accw[step][walker], blobs of ACCEPTED proposals (history row, or the
// current-blob array when no history is kept); k_run_epilogue fills the blob rows of rejected
// proposals forward from the previous step and adds the accept flags to the acceptance
// counters.
#include "nh_hs.h"
#include "nh_syn2.h"
#include <vector>
#include <chrono>
#include <unistd.h>

#define HS_RUN_MAX_STEPS 32  // steps per launch (one block of moves); ring rows = this + 1
#define HS_RUN_ERR_TIMEOUT 1
#define HS_RUN_ERR_PEER 2
#define HS_RUN_ERR_LAST_ROW 3  // (epilogue of a shared ensemble: a rank's last records never came)
#define HS_RUN_MAX_RANKS 8     // GPUs of one node that may share an ensemble
#define HS_RUN_HEAD 512        // granules ahead of the rings in a shared allocation (probe slots)
#define HS_O_HX 56           // two ints of the small block: HX_PRE, HX_SPECRD (two walkers in flight)
enum { HX_PRE = 0, HX_SPECRD = 1 };
#define HS_O_LNA 48          // in the small per-walker block (proposal coordinates: <= 15 of its first 64 doubles)
#define HS_RUN_TRAIL 16        // ints of first-row entries per table in LDS (>= tiles of any table)
#define HS_RUN_PKW 8           // doubles per parameter-pack column in LDS
#ifndef HS_RUN_PK
#define HS_RUN_PK 5      // nodes per trip of a narrow table's items in the table-only instances (nh_hs.h): 6 spilled 8 VGPRs (cfg5 9.12 M walker-steps/s, 9.77 M at 5; cfg1 1.49 -> 1.46 M)
#endif

// the weights' unit table (hs_run.o_ut): ints per unit
enum { HSU_FL = 0, HSU_NL, HSU_G, HSU_W, HSU_D, HSU_DP, HSU_IL, HSU_TH, HSU_LX, HSU_LNE, HSU_GX, HSU_GE,
       HSU_S2LW, HSU_S2LG, HSU_SC_LO, HSU_SC_HI, HSU_N };
// HSU_FL: bits 0-1 = 0 plain | 1 the log-domain synchrotron items read ln w of this grid too | 2 and
// nothing else of it; bit 2: the non-negative table items' pre-divided log-ratios are kept; bit 3:
// ... with 1 / lx and the thresholds in LDS since the launch began; bit 4: the unit is its grid's last
static_assert(HSU_N == 16, "a unit's descriptor is four 16-byte reads");

// the table items' descriptors (hs_run.o_it): HS_MAX_TAB x 16 ints per TABLE (HST_*), then 4 ints
// per ITEM { table | tile << 4, first row, end row, the same two ignoring the leading zero rows (16
// bits each) }
enum { HST_KD_LO = 0, HST_KD_HI, HST_NK, HST_NG, HST_AW, HST_AD, HST_AL, HST_FLAGS, HST_SUB, HST_NKP,
       HST_N = 16 };
#define HSI_N 4

struct hs_run {
  unsigned long long* ring;  // [HS_RUN_MAX_STEPS + 1][N][gr] granules
  int* status;               // device word: 0, or HS_RUN_ERR_* of the first workgroup that gave up
  int* accw;                 // [HS_RUN_MAX_STEPS][N]: 1 where the step's proposal was accepted
  double* hcoords; double* hlogp;  // chain history of this call (or NULL) ...
  double* hblob[NH_HS_MAX_BLOB];   // ... and the blobs' (or NULL)
  long long hrow0, hcap;     // history row of the launch's first step; rows the history holds
  long long* dbg;            // NH_HS_DEBUG: [256][64][8] wall-clock stamps, or NULL
  // gridDim.y = K > 1 workgroups share a walker (a half-step of fewer walkers than CUs, as in
  // k_half_step): partial spectra of slice s at xspec[s][walker][K][nspec], arrival tickets at
  // tick[s][walker] (zeroed before the launch) -- per slice, because nothing stops one of a
  // walker's workgroups from running slices ahead of another
  double* xspec; int* tick;
  int slice0, nslices;       // slices [slice0, slice0 + nslices) of the block of moves; slice0 even
  unsigned seq;              // launch sequence number (tags)
  int gr, N;                 // granules per record (a multiple of 16: one record = whole lines)
  int o_gx[NH_MAX_GRIDS], o_lne[NH_MAX_GRIDS], o_ge[NH_MAX_GRIDS];  // LDS: grid nodes, ln E, E
  int o_pk, o_small1, o_olds;  // LDS: pack descriptors, the second small block, old coordinates
  int o_lcl;                   // LDS: ln(1 - cl[n]), n <= nE
  int o_trail;                 // LDS: the tables' trailers (ints)
  // the loop's own copies of the plan's tables with their columns sorted (nh_hs.h; NULL: the
  // plan's table as it is)
  const double* kds[HS_MAX_TAB];
  // order: 2 = synchrotron items (twice as long as table items) first, then the tables -- the
  // items pulled last decide how far apart the waves reach the barrier (cfg3: 18.9 -> 18.6 us of
  // items + wait); 0 = the two kinds alternate as in k_half_step
  int spin_limit, order;
  // synchrotron nodes per thread and work item.  The plan's 10 suit a launch per half-step and
  // workgroups that share a walker; one resident workgroup per walker runs fastest on items
  // three times that long (cfg3, us per 40 half-steps: 1 015 at 10, 981 at 16, 975 at 24, 959
  // at 30 and at 40; two workgroups per walker -- cfg2 -- 901 at 5, 885 at 10, 914 at 16)
  int syn_nodes;
  int rebalance;  // a tile's chunks divide the rows above its first non-zero one (table-only models)
  // what the host learns about a launch without a round trip of its own: k_run_epilogue stores
  // { status, NaN proposals, forbidden proposals, launch number } (the number last) into one of
  // two slots of page-locked host memory (nh_half_step_run_report)
  int* report;
  int report_launch;
  // NaN / forbidden proposals THIS launch has met (two device ints, a pair per launch parity):
  // k_run_epilogue adds them to the plan's counters only when the launch ended well, so that a
  // launch that gave up part of the way leaves the counters as it found them whoever of its
  // workgroups had already counted (what a replay of its block of moves starts from)
  int* lcnt;
  int o_il[NH_MAX_GRIDS];  // LDS: 1 / lx per node of a grid a non-negative table is reduced over (-1: none)
  // LDS: the weights' units -- 64 consecutive nodes of one grid each -- as 16 ints per unit
  // (HSU_*: grid, first node, node count, where the node's inputs and outputs sit in LDS, the
  // grid's scale), built once per launch; -1: the grids' nodes are not in LDS (small workgroups)
  int o_ut;
  // LDS: the table work items -- which table, column tile and rows, where the walker's arrays
  // sit -- as 16 ints per item (HSI_*), built once per launch: a work item read its table's
  // descriptor out of the kernel-argument segment field by field (dependent vector loads: the
  // table index is not a constant) and re-derived its rows -- ~300 vector instructions before its
  // first segment, as many as the segments of a 32-row item cost; -1: not used
  int o_it;
  // LDS: per column of the tables' (sorted) spectra 4 ints { first partial sum | stride | chunks |
  // where it goes in spec } + its scale's index rides with `where`; and the model's components as
  // NH_MAX_COMP x { offset in spec (int, as a double's low word) | scale }: the phases behind
  // barrier 3 -- one or a few waves, everybody else waiting -- took these out of the
  // kernel-argument segment one dependent scalar load at a time (four round trips per component)
  int o_sum, o_cmp;
  int o_pci;  // LDS: per prior term the proposed coordinate it reads, or -1 (ints)
  int o_mt;     // LDS: per single-row reduction { w, dlw, lx, K | dlnK (LDS byte addresses), nG, first row, last row } (ints)
  int o_lnt;    // LDS: 128 x { 1 / c_j, ln c_j }, c_j the centres of [1/2, 1)'s 128 bins (hsr_ln_tab; log-domain instances)
  int o_synce;  // LDS: CS1 / B per photon energy (walker-independent: a division per live energy and slice otherwise)
  // K workgroups of a table-only walker split the grid's ROWS (nh_halfstep.hip: the plan's
  // rowsplit): workgroup `part` owns the nodes [b_part, b_part+1] -- the boundaries sit between
  // two of the table's chunks -- forms the weights there, reduces its chunks and its part of the
  // single-row reductions, and hands its partial sums to workgroup 0 of the walker
  int rowsplit;
  int tcompact;  // K > 1: a table item's partial sums in the slot of its group (hs_run_create)
  int o_rs;   // LDS: this workgroup's nodes per grid and its units (rowsplit)
  int nxmax;  // doubles a workgroup hands over at most (spectrum + single-row reductions)
  int sum_cols;  // columns of all the tables together (the sum phase's loop bound)
  int dbg_skip;  // -DNH_LAB builds only: NH_RUN_DEBUG_SKIP (experiments: instruction counts by kind): 1 no synchrotron items, 2 no table items
  // ---- an ensemble shared by several GPUs (nrank > 1; see "The ensemble across GPUs" below):
  // `ring` is this launch's ring in THIS rank's memory, peer[p] the same ring in rank p's
  // (peer[rank] == ring); a mover stores its walker's record into every one of them
  int nrank, rank;
  unsigned long long* peer[HS_RUN_MAX_RANKS];
  int* nacc_own;   // acceptance counters of the moves THIS rank made (the plan's are replicated)
  int* hacc;       // with a history: [hcap][N] -1 | 0 | 1 = not moved by this rank | rejected | accepted
  int* curstamp;   // with blobs: [N] stamp of the last move this rank accepted for the walker
  int stamp0;         // stamp of the launch's first step (counts the steps of all launches)
  // NH_RUN_PUBLISH_DELAY (100 MHz ticks; experiments): a mover waits this long before it stores a
  // record -- every hand-off of the shared loop then pays what a slower link would add
  int publish_delay;
  // ---- the synchrotron items in the log domain, on the grid's comb (nh_syn2.h; syn2 != 0) ----
  int syn2, s2_own;  // s2_own: nobody else reads the synchrotron grid's w / dlw (its LDS is reused)
  int o_s2tab, o_s2lw, o_s2ig, o_s2lg, o_s2q, o_s2z, o_s2t;  // LDS: table | Lambda ln w (guards either side) |
                                                      // 1/gamma^2 (guards) | Lambda (ln gamma / 3 +
                                                      // ln scale) | per live energy 4 doubles | comb index | 2^(j/128)
  const double* s2_dev;  // device: the table's (P + 1) x 6 doubles, then the grid's nG values of the above
  hs_syn2_par s2;
  double s2_z0, s2_invd;  // z = s2_z0 - ln(q) s2_invd: where node 0 sits on the comb below T_top
  double s2_r746;         // (T_top - ln 746) s2_invd - s2_z0: the first live node is ceil(ln(q) s2_invd + this)
  double s2_lnw0;         // ln |n| + ln(E / eV) above this: the weight gamma n scale is not 0 in double
  int pipeline;           // two walkers in flight (a workgroup with several walkers of a slice): NH_RUN_PIPELINE
  long long* clk;         // the context's span clock (nh_common.h): this launch opens a span, k_run_epilogue closes it
};

static_assert(sizeof(hs_hot) + sizeof(hs_run) <= 4000, "both argument blocks fit the kernarg segment");

__device__ __forceinline__ unsigned long long hs_ld_sc1(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hs_st_sc1(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// system scope (sc0 sc1): records that another GPU writes into / reads from fine-grained memory
__device__ __forceinline__ unsigned long long hs_ld_sys(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void hs_st_sys(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned hs_tag(unsigned seq, int row) {
  return (seq << 8) | (unsigned)(row + 1);  // (never 0: a zeroed ring matches nothing)
}
// granule L of the record of the doubles v[0 .. n): L = 2 d + {0: low, 1: high word}
__device__ __forceinline__ unsigned long long hs_granule(double v, int L, unsigned tag) {
  const unsigned w = (L & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
  return ((unsigned long long)tag << 32) | w;
}

// The library-math phases of a slice (parameter packs: exp10 / log ...; particle weights:
// exp, expm1; the likelihood's log) run ONCE per slice and thread, but their polynomial
// coefficients are loop invariants of the slice loop: inlined, the compiler parks ~70 of them
// in registers across the whole loop and spills them (190 dwords of scratch, reloaded on the
// serial path of every slice).  As real calls they are materialised where they are used.
__device__ __attribute__((noinline)) double hsr_lazy_apply(double a, double b, double c, int tf,
                                                           double raw) {
  nh_lazy z;
  z.base = nullptr; z.stride = 1; z.a = a; z.b = b; z.c = c; z.tf = tf; z.pad = 0;
  return nh_lazy_apply(z, raw);
}
__device__ __attribute__((noinline)) double hsr_log(double x) { return log(x); }
__device__ __attribute__((noinline)) double hsr_exp10(double x) { return exp10(x); }
__device__ __attribute__((noinline)) double hsr_cbrt(double x) { return cbrt(x); }
// ln x by a 128-bin table of [1/2, 1) and a degree-6 series of the remainder (|r| <= 2^-8:
// r^7 / 7 < 2e-18) -- 20 instructions where the library's is ~70 behind a call: the liveness waves
// take the logarithm of q = E / E_c for every photon energy of every slice, and with three of them
// (cfg2's 179 energies) they reached barrier 2 a microsecond behind the waves that form the weights.
// About one unit in the last place of a result of order 10; anything that is not a positive normal
// number goes to the library (wave-uniform branch).  a_t: LDS byte address of { 1 / c_j, ln c_j }.
typedef double hsr_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const hsr_d2 hsr_lcd2;
__device__ __attribute__((noinline)) double hsr_ln_tab(double x, unsigned a_t) {  // (a call: inlined, its coefficients are parked across the slice loop)
  const bool plain = x >= 2.2250738585072014e-308 && x < INFINITY;
  if (__builtin_amdgcn_ballot_w64(!plain) != 0ull) {
    asm volatile("" ::: "memory");  // (keep the call behind the branch)
    return hsr_log(x);
  }
  const double m = __builtin_amdgcn_frexp_mant(x);  // [1/2, 1)
  const int e = __builtin_amdgcn_frexp_exp(x);
  unsigned j = ((unsigned)__double2hiint(m) >> 13) & 127u;  // the seven mantissa bits below the leading one
  asm("" : "+v"(j));
  const hsr_d2 t = *(hsr_lcd2*)(unsigned long long)(a_t + (j << 4));
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, -1.6666666666666666e-01, 0.2);
  p = fma(p, r, -0.25);
  p = fma(p, r, 3.3333333333333331e-01);
  p = fma(p, r, -0.5);
  p = fma(p * r, r, r);  // r - r^2/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6
  const double ed = (double)e;
  // e ln 2 in two pieces (the first exact for |e| < 2^11), the small ones first
  return fma(ed, 6.93147180369123816490e-01, t.y) + fma(ed, 1.90821492927058770002e-10, p);
}
struct hsr_node { double n, dsh, ex; };  // ex = ln(n / A)
struct hsr_node2 { double n0, dsh0, n1, dsh1, ex0, ex1; };
// b12 / b0 / b1: bit 0, 1 = this node / the next lie below the break; bit 2 = ln(n / A) only
// t64: LDS byte address of 2^(j/64)
__device__ __attribute__((noinline)) hsr_node hsr_pd_core(int kind, double A, double al, double be,
                                                          double a2, double lxx, double lxc,
                                                          double lkb, int b12, double lr,
                                                          unsigned t64) {
  pd_par p;
  p.A = A; p.e0 = 0.0; p.al = al; p.ec = 0.0; p.be = be; p.eb = 0.0; p.a2 = a2;
  hsr_node r;
  pd_core(kind, p, lxx, lxc, lkb, (b12 & 1) != 0, (b12 & 2) != 0, lr, r.n, r.dsh, nullptr, &r.ex,
          (b12 & 4) == 0, t64);
  return r;
}

// two nodes of (possibly) different grids in one go: the two dependent chains of ~100
// instructions interleave -- a wave that holds two units of nodes is not twice as late
__device__ __attribute__((noinline)) hsr_node2 hsr_pd_core2(int kind, double A, double al,
                                                            double be, double a2, double lkb,
                                                            double lxx0, double lxc0, int b0,
                                                            double lr0, double lxx1, double lxc1,
                                                            int b1, double lr1, unsigned t64) {
  pd_par p;
  p.A = A; p.e0 = 0.0; p.al = al; p.ec = 0.0; p.be = be; p.eb = 0.0; p.a2 = a2;
  hsr_node2 r;
  pd_core(kind, p, lxx0, lxc0, lkb, (b0 & 1) != 0, (b0 & 2) != 0, lr0, r.n0, r.dsh0, nullptr, &r.ex0,
          (b0 & 4) == 0, t64);
  pd_core(kind, p, lxx1, lxc1, lkb, (b1 & 1) != 0, (b1 & 2) != 0, lr1, r.n1, r.dsh1, nullptr, &r.ex1,
          (b1 & 4) == 0, t64);
  return r;
}

// ---- the particle weights of a slice: a wave's units, from LDS to LDS ---------------------------
// One call per wave and slice, specialised by the kind of distribution.  What it replaced: the unit
// loop inline in the kernel with hsr_pd_core / hsr_pd_core2 called per pair of units -- 350 vector
// instructions for a wave with two units of which ~140 were the nodes' arithmetic: the callee took
// the kind and the break flags in vector registers and branched on them lane by lane (every case of
// the switch an exec-mask region), 25 moves marshalled each call, 16 v_readfirstlane per unit turned
// the unit's descriptor into scalars.  Here the descriptor's addresses stay in vector registers
// (uniform contents: an address is only ever added to the lane's offset), one flag word per unit is
// made scalar, the kind is a template argument.
// All arguments are wave-uniform; LDS byte addresses unless said otherwise.
typedef __attribute__((address_space(3))) double hsr_ld;
typedef __attribute__((address_space(3))) const int hsr_lci;
typedef int hsr_i4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const hsr_i4 hsr_lci4;
__device__ __forceinline__ double hsr_rd(unsigned a) { return *(const hsr_ld*)(unsigned long long)a; }
__device__ __forceinline__ void hsr_wr(unsigned a, double v) { *(hsr_ld*)(unsigned long long)a = v; }

struct hsr_unit {  // a unit's descriptor as it is read (HSU_*), and this lane's inputs
  unsigned a_w, a_d, a_dp, a_il, a_th, a_s2lw, a_s2lg;
  int fl, nl, g, ic8;
  bool on, last;
  double sc, lr, lne, gx, il, s2lg;
  int b;
};

template <bool BROKEN>
__device__ __forceinline__ void hsr_unit_read(hsr_unit& U, unsigned a_ut, int uu, int lane, double eb) {
  const unsigned a = a_ut + (unsigned)uu * (HSU_N * 4);
  const hsr_i4 d0 = *(hsr_lci4*)(unsigned long long)a;
  const hsr_i4 d1 = *(hsr_lci4*)(unsigned long long)(a + 16);
  const hsr_i4 d2 = *(hsr_lci4*)(unsigned long long)(a + 32);
  const hsr_i4 d3 = *(hsr_lci4*)(unsigned long long)(a + 48);
  U.fl = __builtin_amdgcn_readfirstlane(d0.x);
  U.nl = d0.y; U.g = d0.z; U.a_w = (unsigned)d0.w;
  U.a_d = (unsigned)d1.x; U.a_dp = (unsigned)d1.y; U.a_il = (unsigned)d1.z; U.a_th = (unsigned)d1.w;
  U.a_s2lw = (unsigned)d3.x; U.a_s2lg = (unsigned)d3.y;
  U.sc = __hiloint2double(d3.w, d3.z);
  U.on = lane < U.nl;
  const int ic = min(lane, U.nl - 1);  // (lanes past the grid's end: its last node, discarded)
  U.last = ic + 1 >= U.nl;
  U.ic8 = ic << 3;
  U.lr = hsr_rd((unsigned)d2.x + U.ic8);  // (0 at the last node)
  U.lne = hsr_rd((unsigned)d2.y + U.ic8);
  U.gx = hsr_rd((unsigned)d2.z + U.ic8);
  U.il = (U.fl & 8) ? hsr_rd(U.a_il + U.ic8) : 0.0;
  U.s2lg = (U.fl & 3) ? hsr_rd(U.a_s2lg + U.ic8) : 0.0;
  U.b = 0;
  if (BROKEN) {
    const double E = hsr_rd((unsigned)d2.w + U.ic8);
    const double E2 = hsr_rd((unsigned)d2.w + (U.last ? U.ic8 : U.ic8 + 8));
    U.b = (E < eb ? 1 : 0) | (E2 < eb ? 2 : 0);
  }
}

// (one node's results -> LDS and the flags: grids with a non-zero weight | << 8: with one that is
// not finite)
template <bool S2SYN>
__device__ __forceinline__ int hsr_unit_write(const hsr_unit& U, double n, double dsh, double ex, double lnA,
                                              double s2_lnw0) {
  const double nn = n * U.sc;
  double wv_ = U.gx * nn;
  const double dv = U.last ? 0.0 : U.lr + dsh;
  bool plain = true;  // (wave-uniform)
  if (S2SYN && (U.fl & 3) != 0) {
    // the log-domain items' Lambda ln|w| + Lambda ln cbrt(1/gamma^2) (nh_syn2.h); a zero weight (or
    // amplitude) is the floor: an exact 0, and exact zeros for its segments
    const double lna = lnA + ex;  // ln |n|
    const double lw = fma(HS_S2_LAMBDA, lna, U.s2lg);
    // (w = gamma n scale underflows to an exact 0 in the reference below ln w = -744.44: the floor
    // -- an exact zero node -- from there on)
    const bool nonzero = lna + U.lne > s2_lnw0;
    if (U.on) hsr_wr(U.a_s2lw + U.ic8, nonzero ? lw : HS_S2_FLOOR);
    if ((U.fl & 3) == 2) {  // (nobody reads this grid's w / dlw, and gx holds another array)
      plain = false;
      wv_ = !(lw < INFINITY) ? lw : (nonzero ? 1.0 : 0.0);  // (what the flags below look at)
    }
  }
  if (plain && U.on) {
    hsr_wr(U.a_w + U.ic8, wv_);
    hsr_wr(U.a_d + U.ic8, dv);
    if (U.fl & 4) {  // what the non-negative table items read
      if (U.fl & 8) {  // (1 / lx and the threshold: in LDS since the launch began)
        hsr_wr(U.a_dp + U.ic8, dv * U.il);
      } else {
        const double il = U.last ? 0.0 : nh_rcp(U.lr);
        hsr_wr(U.a_dp + U.ic8, dv * il);
        hsr_wr(U.a_th + U.ic8, NH_SEG_SMALL_POS * il);
      }
    }
  }
  // (a weight that is not finite -- a far-off walker whose distribution overflows -- makes
  // 0 x inf = NaN of a zero table entry, as in the reference: no row of such a walker's tables is
  // skipped)
  int fl = 0;
  if (__builtin_amdgcn_ballot_w64(U.on && wv_ != 0.0) != 0ull) fl |= 1 << U.g;
  if (__builtin_amdgcn_ballot_w64(U.on && !isfinite(wv_)) != 0ull) fl |= 256 << U.g;
  return fl;
}

// a_ul != 0: the units are taken from that list of unit numbers (rows split between a walker's
// workgroups: the units that hold this one's nodes); units u0 and u0 + ustep (if < uend), their
// chains of ~70 dependent instructions interleaved
template <int KIND, bool S2SYN>
__device__ __attribute__((noinline)) int hsr_weights(unsigned a_ut, unsigned a_ul, int u0, int ustep,
                                                     int uend, unsigned a_row, unsigned a_lg,
                                                     unsigned a_lna, unsigned t64, double s2_lnw0) {
  a_ut = __builtin_amdgcn_readfirstlane(a_ut);
  a_ul = __builtin_amdgcn_readfirstlane(a_ul);
  u0 = __builtin_amdgcn_readfirstlane(u0);
  ustep = __builtin_amdgcn_readfirstlane(ustep);
  uend = __builtin_amdgcn_readfirstlane(uend);
  t64 = __builtin_amdgcn_readfirstlane(t64);
  __builtin_assume(t64 != 0u);  // (pd_core: the exponentials through the table in LDS, no other form compiled in)
  const int lane = (int)__lane_id();
  constexpr bool BROKEN = KIND == NH_PD_BROKENPL || KIND == NH_PD_ECBPL;
  pd_par p;  // (the particle row of the slice: {A, e_0, alpha, e_cutoff, beta, e_break, alpha_2})
  p.A = hsr_rd(a_row); p.e0 = 0.0; p.al = hsr_rd(a_row + 16); p.ec = 0.0; p.be = hsr_rd(a_row + 32);
  p.eb = BROKEN ? hsr_rd(a_row + 40) : 0.0; p.a2 = BROKEN ? hsr_rd(a_row + 48) : 0.0;
  const double lg0 = hsr_rd(a_lg), lg1 = hsr_rd(a_lg + 8);
  const double lkb = BROKEN ? hsr_rd(a_lg + 16) - lg0 : 0.0;
  const double lnA = S2SYN ? hsr_rd(a_lna) : 0.0;
  int fl = 0;
  // (ONE pair of units per call -- the caller loops, once for all but the largest grids: with the
  // loop in here the compiler parks the exponentials' coefficients in ~60 vector registers across
  // it, callee-saved ones that it then saves to scratch on the way in)
  {
    const int u = u0;
    const bool two = u + ustep < uend;  // (wave-uniform)
    int uu0 = u, uu1 = two ? u + ustep : u;
    if (a_ul) {
      uu0 = __builtin_amdgcn_readfirstlane(*(hsr_lci*)(unsigned long long)(a_ul + 4u * (unsigned)uu0));
      uu1 = __builtin_amdgcn_readfirstlane(*(hsr_lci*)(unsigned long long)(a_ul + 4u * (unsigned)uu1));
    }
    hsr_unit U0, U1;
    hsr_unit_read<BROKEN>(U0, a_ut, uu0, lane, p.eb);
    if (two) hsr_unit_read<BROKEN>(U1, a_ut, uu1, lane, p.eb);
    // (the log-domain synchrotron items read ln w and nothing else of a grid: no exponential, no
    // expm1 for its nodes -- wave-uniform, a unit is one grid's)
    double n0, d0, e0, n1 = 0.0, d1 = 0.0, e1 = 0.0;
    if (two) {
      pd_core(KIND, p, U0.lne - lg0, U0.lne - lg1, lkb, (U0.b & 1) != 0, (U0.b & 2) != 0, U0.lr, n0, d0,
              nullptr, &e0, (U0.fl & 3) != 2, t64);
      pd_core(KIND, p, U1.lne - lg0, U1.lne - lg1, lkb, (U1.b & 1) != 0, (U1.b & 2) != 0, U1.lr, n1, d1,
              nullptr, &e1, (U1.fl & 3) != 2, t64);
      fl |= hsr_unit_write<S2SYN>(U0, n0, d0, e0, lnA, s2_lnw0);
      fl |= hsr_unit_write<S2SYN>(U1, n1, d1, e1, lnA, s2_lnw0);
    } else {
      pd_core(KIND, p, U0.lne - lg0, U0.lne - lg1, lkb, (U0.b & 1) != 0, (U0.b & 2) != 0, U0.lr, n0, d0,
              nullptr, &e0, (U0.fl & 3) != 2, t64);
      fl |= hsr_unit_write<S2SYN>(U0, n0, d0, e0, lnA, s2_lnw0);
    }
  }
  return fl;
}

template <bool S2SYN>
__device__ __forceinline__ int hsr_weights_kind(int kind, unsigned a_ut, unsigned a_ul, int u0, int ustep,
                                                int uend, unsigned a_row, unsigned a_lg, unsigned a_lna,
                                                unsigned t64, double s2_lnw0) {
  switch (kind) {  // (a kernel argument: a scalar branch)
    case NH_PD_POWERLAW:
      return hsr_weights<NH_PD_POWERLAW, S2SYN>(a_ut, a_ul, u0, ustep, uend, a_row, a_lg, a_lna, t64, s2_lnw0);
    case NH_PD_ECPL:
      return hsr_weights<NH_PD_ECPL, S2SYN>(a_ut, a_ul, u0, ustep, uend, a_row, a_lg, a_lna, t64, s2_lnw0);
    case NH_PD_BROKENPL:
      return hsr_weights<NH_PD_BROKENPL, S2SYN>(a_ut, a_ul, u0, ustep, uend, a_row, a_lg, a_lna, t64, s2_lnw0);
    case NH_PD_ECBPL:
      return hsr_weights<NH_PD_ECBPL, S2SYN>(a_ut, a_ul, u0, ustep, uend, a_row, a_lg, a_lna, t64, s2_lnw0);
    default:
      return hsr_weights<NH_PD_LOGPARABOLA, S2SYN>(a_ut, a_ul, u0, ustep, uend, a_row, a_lg, a_lna, t64, s2_lnw0);
  }
}

// workgroup 0 only: when does each WAVE reach barrier 1 / 2 / 3 and finish its last item
#ifndef HSR_FINE_PART
#define HSR_FINE_PART 0  // (which of a walker's workgroups writes the fine stamps)
#endif
#ifdef HSR_FINE
#define HSR_WSTAMP(k) do {} while (0)
#else
#define HSR_WSTAMP(k)                                                                  \
  do {                                                                                  \
    if (R.dbg && lane == 0 && blockIdx.x == 0 && it < 64)                               \
      R.dbg[256 * 64 * 8 + (it * 4 + (k)) * 16 + wv] = (long long)wall_clock64();       \
  } while (0)
#endif
// -DHSR_FINE (experiments: scripts/r5_fine.py): workgroup 0, iterations 32 .. 39, eight stamps per
// wave inside a slice -- where does a phase's time go, wave by wave
#ifdef HSR_FINE
#define HSR_FSTAMP(k)                                                                   \
  do {                                                                                   \
    if (R.dbg && lane == 0 && blockIdx.x == 0 && blockIdx.y == HSR_FINE_PART && it >= 32 && it < 40) \
      R.dbg[256 * 64 * 8 + (((it & 7) * 8 + (k)) * 16) + wv] = (long long)wall_clock64(); \
  } while (0)
#else
#define HSR_FSTAMP(k) do {} while (0)
#endif
#if defined(HSR_FINE) && HSR_FINE == 2  // (the same slots, for the tail of a slice instead)
#undef HSR_FSTAMP
#define HSR_FSTAMP(k) do {} while (0)
#define HSR_GSTAMP(k)                                                                   \
  do {                                                                                   \
    if (R.dbg && lane == 0 && blockIdx.x == 0 && blockIdx.y == HSR_FINE_PART && it >= 32 && it < 40) \
      R.dbg[256 * 64 * 8 + (((it & 7) * 8 + (k)) * 16) + wv] = (long long)wall_clock64(); \
  } while (0)
#else
#define HSR_GSTAMP(k) do {} while (0)
#endif
#define HSR_STAMP(k)                                                                   \
  do {                                                                                  \
    if (R.dbg && tid == 0 && blockIdx.x < 256 && it < 64)                               \
      R.dbg[((long long)blockIdx.x * 64 + it) * 8 + (k)] = (long long)wall_clock64();   \
  } while (0)

// S2: the synchrotron items in the log domain (nh_syn2.h) -- an instance of its own, so that
// neither form carries the other's registers and code
// RT > 0: a table-only model in workgroups of at most 512 threads whose table items stay in
// registers for the launch, RT nodes per lane at most (nh_hs.h: hs_rt_item)
struct hsr_a_out { double pval; int pcol, bad, ok; };  // what phase A leaves in wave 0's registers
// DEEP: two walkers in flight -- an instance of its own (launches whose workgroups have several
// walkers of a slice to themselves): the other instances' code is the code it was, to the register
template <bool SYN, bool MULTI, bool S2, int RT = 0, bool DEEP = false>
__global__ __launch_bounds__(RT > 0 ? 512 : 1024) void k_half_step_run(const hs_hot H, const hs_run R) {
  extern __shared__ double sm[];
  const hs_dev& D = H.C;
  const int T = blockDim.x, tid0 = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6), nwv = T >> 6;
  int tid = tid0, lane = tid0 & 63;
  const int ns = H.ns, ndim = H.ndim, N = R.N;
  const int K = gridDim.y, part = blockIdx.y;
  const bool has_syn = SYN && H.syn_grid >= 0;
  const bool broken = H.F.broken != 0;
  const int GRn = 2 * (ndim + 1);  // granules of a record that carry data (<= 32)
  const bool lik_wave = wv == (nwv > 1 ? 1 : 0);
  const int npk8 = H.F.npk8;
  constexpr bool multi = MULTI;  // the ensemble is shared with other GPUs (an instance of its
                                 // own: the one-GPU kernel is the code it was)

  // =========================== once per launch ==============================================
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid0 == 0) nh_clk_open(R.clk);
  if (wv == nwv - 1) sm[HS_O_T64 + lane] = exp2((double)lane * 0.015625);
  if (DEEP && tid0 < 2) {  // (two walkers in flight: nothing made ahead, nothing read yet)
    int* hx0 = reinterpret_cast<int*>((tid0 ? sm + R.o_small1 : sm) + HS_O_HX);
    hx0[HX_PRE] = 0;
    hx0[HX_SPECRD] = 0;
  }
  for (int g = 0; g < H.ngrids; ++g) {
    const int nG = H.nG[g];
    for (int i = tid; i < nG; i += T) {
      sm[H.o_lx[g] + i] = i + 1 < nG ? H.lx[g][i] : 0.0;
      if (R.o_il[g] >= 0) {
        // 1 / lx and the series threshold 2^-10 / lx of a table grid's segments do not depend on
        // the walker: once per launch (they were a reciprocal and two stores per node and slice)
        const double il = i + 1 < nG ? nh_rcp(H.lx[g][i]) : 0.0;
        sm[R.o_il[g] + i] = il;
        sm[H.o_th[g] + i] = NH_SEG_SMALL_POS * il;
      }
      if (R.o_gx[0] >= 0) {  // (small workgroups leave the nodes in L2: LDS decides how many fit a CU)
        if (!(SYN && S2 && R.s2_own && g == H.syn_grid)) sm[R.o_gx[g] + i] = H.xg[g][i];
        sm[R.o_lne[g] + i] = H.lne[g][i];
        if (broken) sm[R.o_ge[g] + i] = H.e[g][i];
      }
    }
  }
  {
    const int npri = (int)(sizeof(nh_prior_pack) / sizeof(double));
    const double* psrc = reinterpret_cast<const double*>(&D.pri);
    for (int t = tid; t < npri; t += T) sm[H.o_pri + t] = psrc[t];
    // which proposed coordinate a prior term reads (-1: none of this walker's -- a device array of
    // its own): decided once per launch, it was two 64-bit divisions per term and slice
    for (int t = tid; t < NH_MAX_PRIOR; t += T) {
      int ci = -1;
      if (t < D.pri.n && D.pri.t[t].x.base) {
        const long long d = D.pri.t[t].x.base - H.qT;
        if (d >= 0 && d < (long long)H.ndim * H.nloc && d % H.nloc == 0 && D.pri.t[t].x.stride == 1)
          ci = (int)(d / H.nloc);
      }
      reinterpret_cast<int*>(sm + R.o_pci)[t] = ci;
    }
    if (has_syn) {
      for (int k = tid; k < H.syn_nE; k += T) {
        const double E = H.syn_E[k];
        sm[H.o_synE + k] = E;
        // CS1 / B = sqrt(3) e^3 / (2 pi m_e c^2 hbar E)          radiative.py:319-328
        sm[R.o_synce + k] = (1.7320508075688772 * (NH_E_GAUSS * NH_E_GAUSS * NH_E_GAUSS)) /
                            (2.0 * NH_PI * NH_M_E_G * (NH_C_CGS * NH_C_CGS) * NH_HBAR_CGS * (E * NH_ERG_PER_EV));
      }
      const int nGs = H.F.syn_nG;
      for (int i = tid; i < nGs; i += T) {
        sm[H.o_ig2 + i] = H.F.syn_c[i];
        if (!(S2 && R.s2_own)) {  // (the log-domain items keep their own arrays there)
          sm[H.o_ig23 + i] = H.F.syn_c[nGs + i];
          sm[H.o_dig2 + i] = H.F.syn_c[2 * nGs + i];
        }
      }
    }
    if (SYN && S2) {
      const int nGs = H.F.syn_nG, ntb = (R.s2.P + 1) * HS_S2_STRIDE;
      for (int i = tid; i < ntb; i += T) sm[R.o_s2tab + i] = R.s2_dev[i];
      for (int i = tid; i < HS_S2_TN; i += T) sm[R.o_s2t + i] = exp2((double)i * (1.0 / HS_S2_TN));
      if (tid < 128) {  // hsr_ln_tab's table
        const double c = 0.5 + ((double)tid + 0.5) * (1.0 / 256.0);
        sm[R.o_lnt + 2 * tid] = 1.0 / c;
        sm[R.o_lnt + 2 * tid + 1] = log(c);
      }
      for (int i = tid; i < nGs; i += T) sm[R.o_s2lg + i] = R.s2_dev[ntb + i];
      for (int i = tid; i < nGs + 2 * HS_S2_GUARD; i += T) {  // (guards: the edge values, any finite number)
        const int ii = min(max(i - HS_S2_GUARD, 0), nGs - 1);
        sm[R.o_s2ig + i] = H.F.syn_c[ii];  // (1 / gamma^2)
        // ln w of the guard nodes: zeros of their own (the walker's nodes are written every slice)
        if (i < HS_S2_GUARD || i >= nGs + HS_S2_GUARD) sm[R.o_s2lw + i] = HS_S2_FLOOR;
      }
    }
    for (int t = 0; t < H.ntab; ++t)
      for (int k = tid; k < H.tnK[t]; k += T)
        sm[H.o_scale + H.tspec[t] + k] = H.tscale[t] ? H.tscale[t][k] : 1.0;
    double* lik = sm + H.o_lik;  // conv | flux | elo | ehi | ul, nE each
    // (ln(1 - cl[n]) for every violation count the likelihood can meet, core.py:89-92)
    for (int k = tid; k <= H.nE; k += T) sm[R.o_lcl + k] = k < H.nE ? log(1.0 - H.cl[k]) : 0.0;
    for (int k = tid; k < H.nE; k += T) {
      lik[k] = H.conv[k];
      lik[H.nE + k] = H.flux[k];
      // (-1 / (2 sigma^2), core.py:79-87: the division once per launch, not on the tail of every
      // slice -- the likelihood wave works alone there, and a double division is a chain of
      // thirty dependent instructions)
      lik[2 * H.nE + k] = -0.5 / (H.elo[k] * H.elo[k]);
      lik[3 * H.nE + k] = -0.5 / (H.ehi[k] * H.ehi[k]);
      lik[4 * H.nE + k] = (double)H.ul[k];
    }
    int ko = H.o_mkt;
    for (int m = 0; m < H.nmom; ++m) {  // the single-row tables (We, Wp)
      const int nG = H.nG[H.mgrid[m]];
      for (int i = tid; i < nG; i += T) {
        sm[ko + i] = H.mKt[m][i];
        sm[ko + nG + i] = H.mdK[m][i];
      }
      ko += 2 * nG;
    }
    // the tables' trailers (nh_hs.h: first non-zero row per tile, column order): ints
    // { row0[HS_MAX_TAB][HS_RUN_TRAIL] | perm[nspec] }, identity for a table without one
    {
      int* tr = reinterpret_cast<int*>(sm + R.o_trail);
      for (int t = 0; t < H.ntab; ++t) {
        const hs_tab& tb = D.tab[t];
        const double* kds = t == 0 ? R.kds[0] : t == 1 ? R.kds[1] : t == 2 ? R.kds[2] : R.kds[3];
        const bool has = kds != nullptr;
        const int* src = hs_tab_trailer(has ? kds : tb.KD, H.nG[tb.grid], tb.nK);
        for (int q = tid; q < HS_RUN_TRAIL; q += T) tr[t * HS_RUN_TRAIL + q] = has && q < HS_TRAIL_TILES ? src[q] : 0;
        for (int k = tid; k < tb.nK; k += T)
          tr[HS_RUN_TRAIL * HS_MAX_TAB + tb.spec_off + k] = has ? src[HS_TRAIL_TILES + k] : k;
      }
    }
    // the parameter packs' columns, one per thread of wave 0: a | b | c | tf, ncols | ld | out | ln|a|
    if (tid < npk8) {
      const nh_pack& P = D.pk[tid / NH_MAX_LAZY];
      const nh_lazy& z = P.cols[tid % NH_MAX_LAZY];
      double* o = sm + R.o_pk + tid * HS_RUN_PKW;
      o[6] = log(fabs(z.a));
      {  // (which proposed coordinate the column reads: -1 a constant)
        const unsigned word = H.F.pkd[tid >> 2];
        const int b = (int)((word >> (8 * (tid & 3))) & 0xFFu);
        reinterpret_cast<int*>(o + 7)[0] = b != 0xFF ? b : -1;
        reinterpret_cast<int*>(o + 7)[1] = P.ncols > (int)(tid % NH_MAX_LAZY) ? 1 : 0;
      }
      o[0] = z.a;
      o[1] = z.b;
      o[2] = z.c;
      reinterpret_cast<int*>(o + 3)[0] = z.tf;
      reinterpret_cast<int*>(o + 3)[1] = P.ncols;
      reinterpret_cast<long long*>(o + 4)[0] = (long long)P.ld;
      reinterpret_cast<double**>(o + 5)[0] = P.out;
    }
  }
  // the weights' units: what the lanes of a unit read and write, resolved to LDS offsets ONCE --
  // in the slice loop these came out of the kernel-argument segment one scalar load and one wait
  // at a time (the fine stamps of round 5: 0.8 us of a unit before its first node's arithmetic,
  // 0.65 us behind it)
  if (R.o_ut >= 0) {
    int* ut = reinterpret_cast<int*>(sm + R.o_ut);
    int u0 = 0;
    for (int g = 0; g < H.ngrids; ++g) {
      const int nu = (H.nG[g] + 63) >> 6;
      const bool s2g = SYN && S2 && g == H.syn_grid;
      for (int q = tid; q < nu; q += T) {
        int* d = ut + (u0 + q) * HSU_N;
        const int i0 = q * 64;
        const bool dp = H.o_dp[g] >= 0, il = R.o_il[g] >= 0;
        auto at = [&](int off) { return (int)hs_lds_addr(sm + off + i0); };
        d[HSU_FL] = (s2g ? (R.s2_own ? 2 : 1) : 0) | (dp ? 4 : 0) | (dp && il ? 8 : 0);
        d[HSU_NL] = H.nG[g] - i0;  // nodes of the grid from this unit's first on
        d[HSU_G] = g;
        d[HSU_W] = at(H.o_w[g]);
        d[HSU_D] = at(H.o_d[g]);
        d[HSU_DP] = dp ? at(H.o_dp[g]) : 0;
        d[HSU_IL] = il ? at(R.o_il[g]) : 0;
        d[HSU_TH] = H.o_th[g] >= 0 ? at(H.o_th[g]) : 0;
        d[HSU_LX] = at(H.o_lx[g]);
        d[HSU_LNE] = at(R.o_lne[g]);
        d[HSU_GX] = at(R.o_gx[g]);
        d[HSU_GE] = at(R.o_ge[g]);
        d[HSU_S2LW] = s2g ? at(R.o_s2lw + HS_S2_GUARD) : 0;
        d[HSU_S2LG] = s2g ? at(R.o_s2lg) : 0;
        d[HSU_SC_LO] = __double2loint(H.scale[g]);
        d[HSU_SC_HI] = __double2hiint(H.scale[g]);
      }
      u0 += nu;
    }
  }
  {
    int* st = reinterpret_cast<int*>(sm + R.o_sum);
    int c0 = 0;  // (the tables' columns one after the other)
    for (int t = 0; t < D.ntab; ++t) {
      const hs_tab& tb = D.tab[t];
      const double* kds = t == 0 ? R.kds[0] : t == 1 ? R.kds[1] : t == 2 ? R.kds[2] : R.kds[3];
      const int* perm = kds ? hs_tab_trailer(kds, H.nG[tb.grid], tb.nK) + HS_TRAIL_TILES : nullptr;
      for (int k = tid; k < tb.nK; k += T) {
        int* d = st + 2 * (c0 + k);  // { first partial sum | stride << 16, chunks | where it goes << 8 }
        d[0] = (H.o_part_t + (tb.item0 + (k >> 6)) * 64 + (k & 63)) | ((tb.tiles * 64) << 16);
        // (K > 1, compact slots: { the column's first item | the tile's lane << 10 | items per chunk << 16 })
        if (K > 1 && R.tcompact) d[0] = (tb.item0 + (k >> 6)) | ((k & 63) << 10) | (tb.tiles << 16);
        d[1] = HS_CHUNKS(tb.chunks) | ((tb.spec_off + (perm ? perm[k] : k)) << 8);  // (sorted columns: back in order)
      }
      c0 += tb.nK;
    }
    double* cm = sm + R.o_cmp;
    for (int q = tid; q < NH_MAX_COMP; q += T) {  // (LDS offsets of the components' spectra)
      cm[2 * q] = q < D.ncomp ? (double)(H.o_spec + D.comp[q].off) : (double)(R.o_cmp + 2 * NH_MAX_COMP);
      cm[2 * q + 1] = q < D.ncomp ? D.comp[q].scale : 0.0;
    }
    if (tid == 0) cm[2 * NH_MAX_COMP] = 0.0;  // (what a component the model does not have reads)
  }
  if (R.o_it >= 0) {
    int* ttab = reinterpret_cast<int*>(sm + R.o_it);
    int* itab = ttab + HS_MAX_TAB * HST_N;
    for (int t = tid; t < D.ntab; t += T) {
      const hs_tab& tb = D.tab[t];
      const int tg = tb.grid;
      const double* kds = t == 0 ? R.kds[0] : t == 1 ? R.kds[1] : t == 2 ? R.kds[2] : R.kds[3];
      const double* kd = kds ? kds : tb.KD;
      const bool pre = tb.nonneg != 0;
      int* d = ttab + t * HST_N;
      d[HST_KD_LO] = (int)(unsigned)(unsigned long long)kd;
      d[HST_KD_HI] = (int)(unsigned)((unsigned long long)kd >> 32);
      d[HST_NK] = tb.nK;
      d[HST_NG] = H.nG[tg];
      d[HST_AW] = (int)hs_lds_addr(sm + H.o_w[tg]);
      d[HST_AD] = (int)hs_lds_addr(sm + (pre ? H.o_dp[tg] : H.o_d[tg]));
      d[HST_AL] = (int)hs_lds_addr(sm + (pre ? H.o_th[tg] : H.o_lx[tg]));
      d[HST_FLAGS] = tg | (pre ? 16 : 0);
      d[HST_SUB] = tb.sub;
      d[HST_NKP] = tb.nKp;
    }
    for (int ix = tid; ix < D.nT; ix += T) {
      int t = 0;
      while (t + 1 < D.ntab && ix >= D.tab[t + 1].item0) ++t;
      const hs_tab& tb = D.tab[t];
      const int loc = ix - tb.item0;
      const int tile = loc % tb.tiles, chunk = loc / tb.tiles;
      const int tg = tb.grid, nG = H.nG[tg];
      const double* kds = t == 0 ? R.kds[0] : t == 1 ? R.kds[1] : t == 2 ? R.kds[2] : R.kds[3];
      int s0f, s1f;
      hs_chunk_range(tb.chunks, chunk, D.seg, nG - 1, s0f, s1f);
      // (rows below the tile's first non-zero one contribute exact zeros: not walked -- unless the
      // walker has a weight that is not finite, 0 x inf: the item then takes s0f, s1f)
      const int r0 = kds ? hs_tab_trailer(kds, nG, tb.nK)[min(tile, HS_TRAIL_TILES - 1)] : 0;
      int s0 = max(s0f, r0), s1 = s1f;
      if (R.rebalance) {
        const int nch = HS_CHUNKS(tb.chunks);
        const int per = (max(nG - 1 - r0, 0) + nch - 1) / nch;
        s0 = r0 + chunk * per;
        s1 = min(nG - 1, s0 + per);
      }
      int* d = itab + ix * HSI_N;
      d[0] = t | (tile << 4);
      d[1] = s0;
      d[2] = s1;
      // (what the item walks when a weight is not finite -- 0 x inf = NaN for the zero rows, as the
      // reference: every row.  Rebalanced chunks keep their ranges, the tile's first one reaching
      // down to row 0: the unbalanced ranges would reach, for the second of two workgroups that
      // split the rows, below the nodes it formed -- ADVICE r5)
      d[3] = R.rebalance ? ((chunk == 0 ? 0 : s0) | (s1 << 16)) : (s0f | (s1f << 16));
    }
  }
  if (R.rowsplit) {
    // This workgroup's nodes of every grid, rs[2 g], rs[2 g + 1] = first and last, and the units
    // that hold them: rs[8] of them, listed from rs[9] on.  The table's grid is cut where a group
    // of its chunks ends (one table: its chunks divide the rows from the first non-zero one on,
    // K equal groups); a grid that only a single-row reduction reads, in K runs of whole units.
    if (tid == 0) {
      const hs_tab& tb = D.tab[0];
      int* rs = reinterpret_cast<int*>(sm + R.o_rs);
      int nmy = 0, u0 = 0;
      for (int g = 0; g < H.ngrids; ++g) {
        const int nG = H.nG[g], nu = (nG + 63) >> 6;
        int lo, hi2;
        if (g == tb.grid) {
          const int nch = HS_CHUNKS(tb.chunks);
          const int r0 = R.kds[0] ? hs_tab_trailer(R.kds[0], nG, tb.nK)[0] : 0;
          const int per = (max(nG - 1 - r0, 0) + nch - 1) / nch;
          const int cpp = nch / K;  // chunks per workgroup
          lo = part == 0 ? 0 : min(nG - 1, r0 + part * cpp * per);
          hi2 = part == K - 1 ? nG - 1 : min(nG - 1, r0 + (part + 1) * cpp * per);
        } else {
          lo = min(nG - 1, ((nu * part) / K) * 64);
          hi2 = part == K - 1 ? nG - 1 : min(nG - 1, ((nu * (part + 1)) / K) * 64);
        }
        rs[2 * g] = lo;
        rs[2 * g + 1] = hi2;
        for (int u = lo >> 6; u <= hi2 >> 6; ++u) rs[9 + nmy++] = u0 + u;
        u0 += nu;
      }
      rs[8] = nmy;
      if (R.dbg && blockIdx.x == 0 && part < 4) {  // (NH_HS_DEBUG: this part's units | its rows of every grid)
        R.dbg[256 * 64 * 8 + 3 * 16 + 4 + part] = nmy;
        for (int g = 0; g < H.ngrids && g < 2; ++g)
          R.dbg[256 * 64 * 8 + 7 * 16 + part * 4 + 2 * g] = rs[2 * g] | ((long long)rs[2 * g + 1] << 32);
        R.dbg[256 * 64 * 8 + 7 * 16 + part * 4 + 1] |= (long long)(R.kds[0] ? hs_tab_trailer(R.kds[0], H.nG[D.tab[0].grid], D.tab[0].nK)[0] : -1) << 48;
      }
    }
  }
  // row 0 of the ring: the ensemble as the flat arrays hold it (written by earlier launches or
  // by the host: the kernel boundary has made it visible).  Walker w by workgroup w mod grid.
  if (wv == 0) {
    const unsigned tag0 = hs_tag(R.seq, 0);
    for (int w = blockIdx.y * gridDim.x + blockIdx.x; w < N; w += gridDim.x * gridDim.y) {
      if (lane < GRn) {
        const int d = lane >> 1;
        const double v = d < ndim ? H.coords[(long long)w * ndim + d] : H.logp[w];
        unsigned long long* dst = R.ring + (long long)w * R.gr + lane;
        if (multi) hs_st_sys(dst, hs_granule(v, lane, tag0));  // (every rank: its own copy)
        else hs_st_sc1(dst, hs_granule(v, lane, tag0));
      }
    }
  }
  __syncthreads();
  // the single-row reductions (We, Wp): what their wave needs, resolved once -- in the slice loop
  // it looked the grid up by a wave-dependent index in the by-value argument block (H.mgrid[m],
  // H.nG[g], H.o_w[g]: dependent loads from the kernel-argument segment at the head of that wave's
  // items phase, every slice)
  if (RT > 0 && tid < H.nmom) {
    const int m = tid, g = H.mgrid[m], nG = H.nG[g];
    int ko = H.o_mkt;
    for (int q = 0; q < m; ++q) ko += 2 * H.nG[H.mgrid[q]];
    int sg0 = 0, sg1 = nG - 1;
    if (R.rowsplit) {  // (this workgroup's rows: the walker's first workgroup adds the parts)
      const int* rs = reinterpret_cast<const int*>(sm + R.o_rs);
      sg0 = rs[2 * g];
      sg1 = rs[2 * g + 1];
    }
    int* d = reinterpret_cast<int*>(sm + R.o_mt) + 8 * m;
    d[0] = (int)hs_lds_addr(sm + H.o_w[g]);
    d[1] = (int)hs_lds_addr(sm + H.o_d[g]);
    d[2] = (int)hs_lds_addr(sm + H.o_lx[g]);
    d[3] = (int)hs_lds_addr(sm + ko);
    d[4] = nG;
    d[5] = sg0;
    d[6] = sg1;
    d[7] = 0;
  }

  // which grid does unit u (64 consecutive nodes of one grid) belong to
  int ub[NH_MAX_GRIDS + 1];
  ub[0] = 0;
#pragma unroll
  for (int g = 0; g < NH_MAX_GRIDS; ++g)
    ub[g + 1] = ub[g] + (g < H.ngrids ? (H.nG[g] + 63) >> 6 : 0);
  const int nunits = ub[NH_MAX_GRIDS];
  const int tiles_ = has_syn ? (H.syn_nE + 63) >> 6 : 0;
  // the weights' nodes go to the waves that have nothing else to do before the second
  // barrier: not the likelihood wave (priors), not the tile waves (liveness search)
  // (tried: a unit for the tile waves and the likelihood wave after their own duty -- the
  // likelihood wave, priors and a logarithm first, then became the last to arrive: 3.1 -> 3.7 us)
  int nwork = nwv - 1 - tiles_, rank = wv == 0 ? 0 : wv - 1;
  bool worker = wv != 1 && wv < nwv - tiles_;
  if (nwork < 1) {
    nwork = nwv;
    rank = wv;
    worker = true;
  }

  // ---- RT: this wave's table item, once: its rows into registers ----------------------------
  // (the item a wave takes is a function of its index -- item = wave, rotated over the K
  // workgroups of a walker as the pulls are -- and its rows of the walker, through the tile's
  // first non-zero row alone: the same every slice)
  hs_rt_item<(RT > 0 ? RT : 1)> rt;
  rt.ix = -1;
  if constexpr (RT > 0) {
    const int total = D.nT;
    int item = wv;
    bool have = item < total;
    if (K > 1 && R.rowsplit) {  // (this part's chunks: one after the other; K is a power of two)
      const int per_part = total >> __builtin_ctz(K);
      have = item < per_part;
      item += part * per_part;
    } else if (K > 1) {
      have = item * K < total;
      item = item * K + ((part + item) & (K - 1));
      have = have && item < total;
    }
    if (have) {
      int t = 0;
      while (t + 1 < D.ntab && item >= D.tab[t + 1].item0) ++t;
      const hs_tab& tb = D.tab[t];
      const int loc = item - tb.item0;
      const int tile = loc % tb.tiles, chunk = loc / tb.tiles;
      const int tg = __builtin_amdgcn_readfirstlane(tb.grid);
      const int nG = __builtin_amdgcn_readfirstlane(H.nG[tg]);
      const int r0 = __builtin_amdgcn_readfirstlane(
          reinterpret_cast<const int*>(sm + R.o_trail)[t * HS_RUN_TRAIL + tile]);
      int s0, s1;
      hs_chunk_range(tb.chunks, chunk, D.seg, nG - 1, s0, s1);
      if (R.rebalance) {
        const int nch = HS_CHUNKS(tb.chunks);
        const int per = (max(nG - 1 - r0, 0) + nch - 1) / nch;
        s0 = r0 + chunk * per;
        s1 = min(nG - 1, s0 + per);
      } else {
        s0 = max(s0, r0);
      }
      const double* kds = t == 0 ? R.kds[0] : t == 1 ? R.kds[1] : t == 2 ? R.kds[2] : R.kds[3];
      const bool pre = __builtin_amdgcn_readfirstlane(tb.nonneg) != 0;
      if (s0 < s1 && (!pre || H.o_dp[tg] >= 0) &&
          hs_rt_load<(RT > 0 ? RT : 1)>(rt, tb, kds ? kds : tb.KD, nG, tile, s0, s1, tid0 & 63, sm + H.o_w[tg],
                                        sm + (pre ? H.o_dp[tg] : H.o_d[tg]),
                                        sm + (pre ? H.o_th[tg] : H.o_lx[tg]))) {
        rt.ix = item;
        rt.slot = (K > 1 && R.tcompact) ? wv : item;
        rt.t = t;
        rt.tile = tg;  // (the grid: what the slice's non-zero mask is indexed by)
        rt.pre = pre ? 1 : 0;
      }
    }
  }

  // (the register-resident instances' likelihood wave: the components' places and factors -- the
  // table above -- in registers for the launch: one LDS round trip less between the last barrier of
  // a slice and the record's store)
  int lk_coff[NH_MAX_COMP];
  double lk_cscl[NH_MAX_COMP];
#pragma unroll
  for (int q = 0; q < NH_MAX_COMP; ++q) {
    lk_coff[q] = 0;
    lk_cscl[q] = 0.0;
    if (RT > 0 && lik_wave) {
      lk_coff[q] = (int)sm[R.o_cmp + 2 * q];
      lk_cscl[q] = sm[R.o_cmp + 2 * q + 1];
    }
  }

  // (two walkers in flight: only where a workgroup has the walker to itself and more than one of
  // a slice -- the 1024-thread instances on ensembles of more walkers than twice the CUs)
  // =========================== the slices ====================================================
  int it = 0;
  for (int s = 0; s < R.nslices; ++s) {
    const int h = R.slice0 + s, tl = s >> 1, half = s & 1;  // tl: step of the launch
    const double* r = H.blk + (long long)h * 3 * ns;
    const int* idx = reinterpret_cast<const int*>(r + 2 * ns);
    for (int j = blockIdx.x; j < H.nloc; j += gridDim.x, ++it) {
      // (the thread index is made opaque once per slice: everything derived from it -- LDS
      // addresses, lane roles of every phase -- would otherwise be hoisted out of the slice loop
      // and kept in registers across all phases: 190 registers of spill)
      tid = tid0;
      asm volatile("" : "+v"(tid));
      lane = tid & 63;
      const int par = it & 1;
      double* qs = par ? sm + R.o_small1 : sm;
      double* row = qs + HS_O_ROW;
      double* lg = qs + HS_O_LG;
      double* accs = qs + HS_O_ACC;
      int* hi = reinterpret_cast<int*>(qs + HS_O_INT);
      double* olds = sm + R.o_olds + par * 64;
      int* hx_cur = reinterpret_cast<int*>(qs + HS_O_HX);
      HSR_STAMP(0);
      double pval = 0.0;  // (wave 0) this lane's parameter-pack column, once evaluated ...
      int pcol = -1;      // ... and which one it is (-1: none)
      int slice_bad = 0;  // (wave 0) the records never came
      // ---- A. wave 0: the two records, the proposal, the parameter packs ---------------------
      // (a function of the walker it is made for: wave 0 makes the NEXT walker's, too, ahead of its
      // own work items when this workgroup has more walkers to come -- `ahead`: one look at the
      // records, no waiting; see "two walkers in flight" below)
      auto phase_a = [&](int s_, int j_, int par_) -> hsr_a_out {
        double pval = 0.0;
        int pcol = -1, slice_bad = 0;
        const int tl = s_ >> 1, half = s_ & 1;
        const double* r = H.blk + (long long)(R.slice0 + s_) * 3 * ns;
        const int* idx = reinterpret_cast<const int*>(r + 2 * ns);
        const int j = j_;
        double* qs = par_ ? sm + R.o_small1 : sm;
        double* row = qs + HS_O_ROW;
        double* lg = qs + HS_O_LG;
        double* accs = qs + HS_O_ACC;
        int* hi = reinterpret_cast<int*>(qs + HS_O_INT);
        int* hx = reinterpret_cast<int*>(qs + HS_O_HX);
        double* olds = sm + R.o_olds + par_ * 64;
        {
#define HSR_A_AHEAD true
#define HSR_A_NOT_YET return hsr_a_out{0.0, -1, 0, 0}
#include "nh_persist_phase_a.inc"
#undef HSR_A_AHEAD
#undef HSR_A_NOT_YET
        }
        {
          // (... and the turn's priors with it: the likelihood wave is still in the previous walker's
          // tail when that turn opens, and was the last to reach its barrier 2 by a microsecond)
#include "nh_persist_priors.inc"
        }
        return hsr_a_out{pval, pcol, slice_bad, 1};
      };
      // ---- two walkers in flight (round 6) ---------------------------------------------------
      // A workgroup with several walkers of a slice took them strictly in turn: records, proposal,
      // packs (wave 0 alone, the other fifteen at barrier 1) -> weights -> items -> sums -> likelihood,
      // accept, record (one wave).  The next walker's phase A depends on nothing this walker
      // computes -- its records belong to the slice before (the FIRST walker of the next slice may
      // need this very walker's: `ahead` looks once and never waits) -- so wave 0 makes it at the
      // head of this walker's items phase, into the other small block; the next turn then opens
      // without phase A and WITHOUT barrier 1: the waves go straight from this walker's sums into
      // the next one's weights while the likelihood wave is still in this walker's tail.  What that
      // tail reads and the next walker's phases ahead of barrier 2 write is disjoint but for `spec`
      // (a tile wave zeroes the dead energies' entries): the tile waves wait for HX_SPECRD of the
      // previous turn's block, set by the likelihood wave once it holds its columns.
      bool pre_a = false;
      if constexpr (DEEP) pre_a = hx_cur[HX_PRE] != 0;  // (workgroup-uniform: written ahead of barrier 3 of the turn before)
      // (tried: barrier 1 kept for a slice's FIRST walker, so that the record of the slice's last
      // one -- what other workgroups' next turns wait for -- is made on a SIMD it has to itself:
      // cfg3 / 2048 15.28 -> 15.07 M, cfg2 / 1024 13.72 -> 13.18 M)
      const bool skip1 = pre_a;
      if (wv == 0 && !pre_a) {
        int* hx = hx_cur;
        (void)hx;
#define HSR_A_AHEAD false
#define HSR_A_NOT_YET do {} while (0)
#include "nh_persist_phase_a.inc"
#undef HSR_A_AHEAD
#undef HSR_A_NOT_YET
      }
      HSR_WSTAMP(0);
      if (!(DEEP && skip1)) __syncthreads();  // -------------------------------------------- #1
      HSR_STAMP(2);
      // (a record that never came -- hi[HI_TICK]: the whole workgroup leaves.  The table-only
      // instances with their rows in registers look at it behind barrier 2, with the words every wave
      // reads there anyway: here it is one more LDS round trip ahead of everybody's weights -- cfg5 /
      // 256 11.57 -> 11.72 M, cfg1 1.73 -> 1.77 M; what a slice without its records computes until
      // then is garbage nobody keeps.  The 1024-thread instances measured 0.2 % slower that way.)
      if (RT == 0 && hi[HI_TICK] != 0) return;
      if (wv == 0 && pcol >= 0 && !slice_bad) {
        // the packs' rows in HBM, for whoever reads them outside this launch: behind the barrier
        // everybody else was waiting at (made ahead: pcol stayed -1 here, stored where it was made)
        const double* o = sm + R.o_pk + lane * HS_RUN_PKW;
        const long long ld = reinterpret_cast<const long long*>(o + 4)[0];
        double* out = reinterpret_cast<double* const*>(o + 5)[0];
        out[(long long)j * ld + pcol] = pval;
      }
      HSR_FSTAMP(0);
      if (K > 1) {  // the work items of the walker's other workgroups: their slots count as 0
        if (!R.tcompact)  // (compact slots: a slot is read only where this workgroup wrote it)
          for (int t = tid; t < D.nT * 64; t += T) sm[H.o_part_t + t] = 0.0;
        if (has_syn)
          for (int t = tid; t < D.syn_cdmax * H.syn_nE; t += T) sm[H.o_part_s + t] = 0.0;
      }
      // ---- B. priors (core.py:34-58, 99-101): a proposal the prior forbids is never accepted,
      // so none of its integrals is evaluated (the reference evaluates and discards,
      // core.py:103-119)
      if (lik_wave && !(DEEP && pre_a)) {
#include "nh_persist_priors.inc"
      }
      // ---- particle weights on every grid (-> LDS); the synchrotron liveness search ----------
      const pd_par p = {row[0], row[1], row[2], row[3], row[4], row[5], row[6]};
      double* spec = sm + H.o_spec;
      double Bw = 0.0, qfac = 0.0;
      if (has_syn) Bw = D.syn_bcol >= 0 ? row[D.syn_bcol] : D.synB[(long long)j * D.syn_ldB];
      int lv_i0 = 0, lv_k = -1;
      double lv_q = 0.0, lv_E = 0.0;
      bool lv_live = false;
      int* tcnt = hi + 8;  // [<= 8] live energies per tile
      const int syn_tiles = tiles_;
      if (has_syn && nwv - 1 - wv < syn_tiles) {
        // x = E/Ec,  Ec = 3 e hbar B gamma^2 / (2 m_e c)         radiative.py:331-334
        // (the division: on the tile waves only -- thirty dependent instructions at the head of
        // every other wave's weights otherwise)
        // (1 / B by v_rcp_f64 and two Newton steps: a last-place difference in q, nothing a spectrum
        // sees; a field that is not positive and finite makes the spectrum NaN whatever q is, below)
        qfac = (NH_ERG_PER_EV * (2.0 * (NH_M_E_G * NH_C_CGS)) / (3.0 * NH_E_GAUSS * NH_HBAR_CGS)) * nh_rcp(Bw);
        const int nG = H.nG[H.syn_grid];
        const double* ig2 = sm + H.o_ig2;
        const int t = nwv - 1 - wv;
        lv_k = t * 64 + lane;
        lv_i0 = nG;
        double lv_lnq = 0.0;
        if (lv_k < H.syn_nE) {
          lv_E = sm[H.o_synE + lv_k];
          lv_q = lv_E * qfac;
          int lo = 0, hi2 = nG;  // first i with q*ig2[i] <= 746 (ig2 decreases with i)
          if (S2) {
            // the grid is a comb in ln gamma (that is what S2 means): node i has ln x_i = ln q -
            // 2 ln gamma_0 - 2 i lx, so the first live node is where the comb crosses ln 746 -- from
            // the logarithm the items need anyway -- and the comparison itself, as the search made
            // it, settles the last place (two or three reads instead of ten dependent ones)
            // (several liveness waves -- more than 64 photon energies -- reach barrier 2 behind the
            // weights' waves: the short logarithm; one of them does not, and cfg3 measured 1 % slower
            // with it than with the library's)
            lv_lnq = syn_tiles > 1 ? hsr_ln_tab(lv_q, hs_lds_addr(sm + R.o_lnt)) : hsr_log(lv_q);
            // (node i sits z + i comb steps below T_top, z = s2_z0 - ln q / (2 lx); s2_r746 = (T_top - ln 746)
            // / (2 lx) - s2_z0, formed on the host: as an expression here it is a loop invariant the
            // compiler computes at the head of every slice and spills)
            const double r = fma(lv_lnq, R.s2_invd, R.s2_r746);
            if (r == r) {
              // (the first live node is ceil(r) -- ADVICE r5: the floor made the one-shot check below
              // fail for almost every energy and the fix-up loops run every slice)
              int c = r > 0.0 ? (r < (double)nG ? (int)__builtin_ceil(r) : nG) : 0;
              // (both neighbours of the estimate asked for at once: it is right, or one off, and the
              // two reads one after the other were two LDS round trips of this wave's chain)
              const double xa = lv_q * ig2[max(c - 1, 0)], xb = lv_q * ig2[min(c, nG - 1)];
              if (!(c > 0 && c < nG && !(xa <= 746.0) && xb <= 746.0)) {
                while (c > 0 && lv_q * ig2[c - 1] <= 746.0) --c;
                while (c < nG && !(lv_q * ig2[c] <= 746.0)) ++c;
              }
              lo = c;
            } else {
              // (a field that is not positive: ln q is NaN -- the search itself, whose answer for a
              // negative q is "every node", for a NaN "none")
              while (lo < hi2) {
                const int mid = (lo + hi2) >> 1;
                if (lv_q * ig2[mid] <= 746.0) hi2 = mid; else lo = mid + 1;
              }
            }
          } else {
            while (lo < hi2) {
              const int mid = (lo + hi2) >> 1;
              if (lv_q * ig2[mid] <= 746.0) hi2 = mid; else lo = mid + 1;
            }
          }
          lv_i0 = lo;
        }
        lv_live = lv_i0 < nG;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(lv_live);
        int ln = lv_live ? (nG - 1) - lv_i0 : 0;  // segments this energy walks
        ln = hs_wave_sum_i32_dpp(ln);  // (the total: in lane 63)
        if (lane == 63) {
          tcnt[t] = __popcll(m);
          if (ln > 0) atomicAdd(&hi[HI_LIVE], ln);
        }
        // The live energies compacted in order, and what the items need per live energy -- here,
        // ahead of the second barrier (the tile waves have no weights to form and used to reach it
        // early, then did this BEHIND it while the waves that had pulled the first synchrotron
        // items spun on HI_READY: the items phase began 1 us late for them, and this wave pulled
        // its own first item last of all).  Where this tile's energies start in the compacted
        // order needs the earlier tiles' live counts: an energy is live iff x <= 746 at the
        // grid's LAST node (1/gamma^2 decreases along it) -- one comparison per lane and tile.
        {
          const int nEs = H.syn_nE;
          int base = 0;
          for (int q = 0; q < t; ++q) {
            const int kq = q * 64 + lane;
            const bool lq = kq < nEs && (sm[H.o_synE + min(kq, nEs - 1)] * qfac) * ig2[nG - 1] <= 746.0;
            base += __popcll(__builtin_amdgcn_ballot_w64(lq));
          }
          int* amap = reinterpret_cast<int*>(sm + H.o_amap);
          int* ai0 = amap + nEs;
          double* sq = sm + H.o_sq;  // q | cbrt(q) | CS1 per live energy
          if (lv_live) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            amap[pos] = lv_k;
            ai0[pos] = lv_i0;
            // CS1 = sqrt(3) e^3 B / (2 pi m_e c^2 hbar E)          radiative.py:319-328
            const double cs1 = Bw * sm[R.o_synce + lv_k];
            if (!S2) {  // (the direct form's operands; the log-domain items take q itself)
              sq[pos] = lv_q;
              sq[nEs + pos] = hsr_cbrt(lv_q);
              sq[2 * nEs + pos] = cs1;
            }
            if (S2) {
              // where this energy's nodes sit on the comb (nh_syn2.h): node i at z + i steps below
              // T_top, z = Z + f; ln Gtilde's t / 3 + ln 1.808 rides with the energy
              const double lnq = lv_lnq;
              const double z = fma(-lnq, R.s2_invd, R.s2_z0);
              const double Zf = floor(z);
              double* s2q = sm + R.o_s2q;
              s2q[pos] = lv_q;
              s2q[nEs + pos] = (HS_S2_LAMBDA / 3.0) * lnq;  // (ln 1.808 rides in the table)
              s2q[2 * nEs + pos] = (z - Zf) * R.s2.im;
              s2q[3 * nEs + pos] = cs1 * qs[HS_O_LNA + 1];
              reinterpret_cast<int*>(sm + R.o_s2z)[pos] = (int)Zf;
            }
          } else if (lv_k < nEs) {
            if (DEEP && skip1) {
              // (barrier 1 was not taken: the walker before may still be in its likelihood -- its
              // columns of `spec` must have been read)
              const int* hxp = reinterpret_cast<const int*>((par ? sm : sm + R.o_small1) + HS_O_HX);
              while (__atomic_load_n(&hxp[HX_SPECRD], __ATOMIC_RELAXED) == 0) __builtin_amdgcn_s_sleep(1);
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            spec[H.syn_spec_off + lv_k] = 0.0;
          }
        }
      }
      HSR_FSTAMP(1);
      if (R.o_ut >= 0) {
        // ---- the grids' nodes are in LDS: units from the table built when the launch began ------
        // (rows split between the walker's workgroups: the units that hold this one's nodes, listed
        // when the launch began)
        const int* ul = reinterpret_cast<const int*>(sm + R.o_rs) + 9;
        const int u_end = R.rowsplit ? __builtin_amdgcn_readfirstlane(ul[-1]) : nunits;
        int fl = 0;  // (wave-uniform) grids with a non-zero weight | << 8: with one that is not finite
        // (tried, round 6: s_setprio 3 for the waves that decide when barrier 2 opens -- a pair of
        // units, the liveness search: cfg3 / 512 13.37 -> 13.37 M, cfg5 - 0.4 %, cfg2 + 1 %, cfg3 /
        // 2048 - 0.5 %: the phase is latency, not issue slots)
        for (int u = worker ? rank : u_end; u < u_end; u += 2 * nwork)
          fl |= hsr_weights_kind<(SYN && S2)>(
              D.kind, hs_lds_addr(sm + R.o_ut), R.rowsplit ? hs_lds_addr(reinterpret_cast<const double*>(ul)) : 0u,
              u, nwork, u_end, hs_lds_addr(row), hs_lds_addr(lg), hs_lds_addr(qs + HS_O_LNA),
              hs_lds_addr(sm + HS_O_T64), R.s2_lnw0);
        HSR_FSTAMP(4);
        fl = __builtin_amdgcn_readfirstlane(fl);
        if (lane == 0 && fl) atomicOr(&hi[HI_NZ], fl);
      } else {
      int nzmask = 0;
      for (int u = worker ? rank : nunits; u < nunits; u += 2 * nwork) {
        // this wave's units u and u + nwork (if any), node `lane` of each
        int gq[2], iq[2], bq[2] = {0, 0};
        double lrq[2] = {0.0, 0.0}, lneq[2] = {0.0, 0.0}, gxq[2] = {0.0, 0.0};
        bool onq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int uu = u + q * nwork;
          // (constant indices only: a dynamically indexed ub[] is an array in scratch, and the
          // search a chain of dependent scratch loads at the head of every slice's weights)
          // (tried: the grid whose nodes are cheap -- ln w only -- LAST in the order of units, so that
          // no wave computes two full units in a row: the weights phase 3.12 -> 3.36 us)
          int g = 0, u0 = 0;
#pragma unroll
          for (int gg = 1; gg < NH_MAX_GRIDS; ++gg)
            if (uu < nunits && uu >= ub[gg]) { g = gg; u0 = ub[gg]; }
          gq[q] = g;
          const int nG = H.nG[g], i = (uu - u0) * 64 + lane;
          iq[q] = i;
          onq[q] = uu < nunits && i < nG;
          if (onq[q]) {
            const bool last = i + 1 >= nG;
            lrq[q] = sm[H.o_lx[g] + i];  // (0 at the last node)
            const bool in_lds = R.o_gx[0] >= 0;  // (wave-uniform)
            lneq[q] = in_lds ? sm[R.o_lne[g] + i] : H.lne[g][i];
            gxq[q] = in_lds ? sm[R.o_gx[g] + i] : H.xg[g][i];
            if (broken) {
              const double E = in_lds ? sm[R.o_ge[g] + i] : H.e[g][i];
              const double E2 = last ? E : (in_lds ? sm[R.o_ge[g] + i + 1] : H.e[g][i + 1]);
              bq[q] = (E < p.eb ? 1 : 0) | (E2 < p.eb ? 2 : 0);
            }
          }
          // (the log-domain synchrotron items read ln w and nothing else of this grid: no
          // exponential, no expm1 for its nodes -- wave-uniform, a unit is one grid's)
          if (SYN && S2 && R.s2_own && g == H.syn_grid) bq[q] |= 4;
        }
        const bool two = u + nwork < nunits;  // (wave-uniform)
        double nnq[2], dshq[2], exq[2];
        HSR_FSTAMP(2);
        const unsigned t64 = hs_lds_addr(sm + HS_O_T64);
        if (two) {
          const hsr_node2 nd = hsr_pd_core2(D.kind, p.A, p.al, p.be, p.a2, lg[2] - lg[0],
                                            lneq[0] - lg[0], lneq[0] - lg[1], bq[0], lrq[0],
                                            lneq[1] - lg[0], lneq[1] - lg[1], bq[1], lrq[1],
                                            t64);
          nnq[0] = nd.n0; dshq[0] = nd.dsh0; nnq[1] = nd.n1; dshq[1] = nd.dsh1;
          exq[0] = nd.ex0; exq[1] = nd.ex1;
        } else {
          const hsr_node nd = hsr_pd_core(D.kind, p.A, p.al, p.be, p.a2, lneq[0] - lg[0],
                                          lneq[0] - lg[1], lg[2] - lg[0], bq[0], lrq[0],
                                          t64);
          nnq[0] = nd.n; dshq[0] = nd.dsh; nnq[1] = 0.0; dshq[1] = 0.0;
          exq[0] = nd.ex; exq[1] = 0.0;
        }
        HSR_FSTAMP(3);
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (onq[q]) {
            const int g = gq[q], i = iq[q];
            const bool last = i + 1 >= H.nG[g];
            const double nn = nnq[q] * H.scale[g];
            double wv_ = gxq[q] * nn;
            const double dv = last ? 0.0 : lrq[q] + dshq[q];
            bool plain = true;  // (wave-uniform: a unit is 64 nodes of one grid)
            if (SYN && S2 && g == H.syn_grid) {
              // the log-domain items' Lambda ln|w| + Lambda ln cbrt(1/gamma^2) (nh_syn2.h); a zero
              // weight (or amplitude) is the floor: an exact 0, and exact zeros for its segments
              const double lna = qs[HS_O_LNA] + exq[q];  // ln |n|
              const double lw = fma(HS_S2_LAMBDA, lna, sm[R.o_s2lg + i]);
              // (w = gamma n scale underflows to an exact 0 in the reference below ln w = -744.44:
              // the floor -- an exact zero node -- from there on)
              const bool nonzero = lna + lneq[q] > R.s2_lnw0;
              sm[R.o_s2lw + HS_S2_GUARD + i] = nonzero ? lw : HS_S2_FLOOR;
              if (R.s2_own) {  // (nobody reads this grid's w / dlw, and gx holds another array)
                plain = false;
                wv_ = !(lw < INFINITY) ? lw : (nonzero ? 1.0 : 0.0);  // (what the flags below look at)
              }
            }
            if (plain) {
              sm[H.o_w[g] + i] = wv_;
              sm[H.o_d[g] + i] = dv;
              if (H.o_dp[g] >= 0) {  // what the non-negative table items read
                if (R.o_il[g] >= 0) {  // (1 / lx and the threshold: in LDS since the launch began)
                  sm[H.o_dp[g] + i] = dv * sm[R.o_il[g] + i];
                } else {
                  const double il = last ? 0.0 : nh_rcp(lrq[q]);
                  sm[H.o_dp[g] + i] = dv * il;
                  sm[H.o_th[g] + i] = NH_SEG_SMALL_POS * il;
                }
              }
            }
            if (wv_ != 0.0) nzmask |= 1 << g;
            // (a weight that is not finite -- a far-off walker whose distribution overflows --
            // makes 0 x inf = NaN of a zero table entry, as in the reference: no row of such a
            // walker's tables is skipped)
            if (!isfinite(wv_)) nzmask |= 256 << g;
          }
      }
      HSR_FSTAMP(4);
      {
        int any = nzmask;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) any |= __shfl_xor(any, off, 64);
        if (lane == 0 && any) atomicOr(&hi[HI_NZ], any);
      }
      }
      HSR_FSTAMP(5);
      HSR_WSTAMP(1);
      __syncthreads();  // ---------------------------------------------------------------- #2
      HSR_STAMP(3);
      HSR_FSTAMP(6);
      if (RT > 0 && hi[HI_TICK] != 0) return;  // (a record never came: see behind barrier 1)
      // ---- the next walker's phase A, ahead of this wave's items -- here, where little is live (two walkers in flight) ---------
      if constexpr (DEEP) if (wv == 0) {
        int made = 0;
        const int jn = j + (int)gridDim.x;
        const int s_n = jn < H.nloc ? s : s + 1, j_n = jn < H.nloc ? jn : (int)blockIdx.x;
        if (s_n < R.nslices) {
          const hsr_a_out a = phase_a(s_n, j_n, par ^ 1);
          const double pv = a.pval;
          const int pc = a.pcol;
          if (a.ok) {
            made = 1;
            if (pc >= 0) {  // (the packs' rows in HBM: see behind barrier 1)
              const double* o = sm + R.o_pk + lane * HS_RUN_PKW;
              const long long ld = reinterpret_cast<const long long*>(o + 4)[0];
              double* out = reinterpret_cast<double* const*>(o + 5)[0];
              out[(long long)j_n * ld + pc] = pv;
            }
          }
        }
        if (lane == 0) reinterpret_cast<int*>((par ? sm : sm + R.o_small1) + HS_O_HX)[HX_PRE] = made;
        if (R.dbg && lane == 0 && blockIdx.x == 0) {  // (NH_HS_DEBUG: looks ahead | made ahead)
          R.dbg[256 * 64 * 8 + 3 * 16 + 0] += 1;
          R.dbg[256 * 64 * 8 + 3 * 16 + 1] += made;
        }
      }
      const int nz = hi[HI_DEAD] ? 0 : hi[HI_NZ];  // (forbidden by the prior: nothing is integrated)
      int nA = 0, Cd = 1, nS = 0;
      if (has_syn) {
        const int nEs = H.syn_nE;
        for (int q = 0; q < syn_tiles; ++q) nA += tcnt[q];
        // (the log-domain items take ln q and ln w: a magnetic field that is not positive, or a
        // weight that is not finite, makes the reference's spectrum NaN -- x < 0 overflows
        // exp(-x), inf x 0 -- and this one with it)
        const bool syn_nan = S2 && !hi[HI_DEAD] &&
                             (!(Bw > 0.0) || !(Bw < INFINITY) || (nz >> (8 + H.syn_grid) & 1) != 0);
        const bool syn_zero = !(nz >> H.syn_grid & 1) || syn_nan;
        if (nA > 0 && !syn_zero) {
          // (every wave forms these two quotients in every slice, ahead of its first item: as
          // 32-bit divisions they were ~60 vector instructions of the slice's serial part;
          // hs_div_small's ranges -- x < 2^20, d < 2^12 -- hold for any grid that fits LDS)
          const int live = hi[HI_LIVE];
          if (live < (1 << 20) && nA < (1 << 12) && R.syn_nodes < (1 << 12)) {
            const int per = hs_div_small(live, nA, __builtin_amdgcn_rcpf((float)nA));
            Cd = hs_div_small(per + R.syn_nodes - 1, R.syn_nodes, __builtin_amdgcn_rcpf((float)R.syn_nodes));
          } else {
            Cd = (live / nA + R.syn_nodes - 1) / R.syn_nodes;
          }
          Cd = min(max(Cd, 1), D.syn_cdmax);
          nS = (nA * Cd + 63) >> 6;
        }
        if (syn_zero) {
          for (int k = tid; k < nEs; k += T) spec[H.syn_spec_off + k] = syn_nan ? NAN : 0.0;
          nA = 0;
        }
      }
      // ---- single-row reductions (We, Wp), one wave each (from the back) ----------------------
      if (RT == 0 && nwv - 1 - wv < H.nmom) {
        // (the 1024-thread instances: the wave that has a reduction pulls work items behind it like
        // every other wave, and the look-up is hidden -- with the table below they measured 0.3 %
        // slower, ten more scalars spilled)
        const int m = nwv - 1 - wv;
        const int g = H.mgrid[m], nG = H.nG[g];
        int ko = H.o_mkt;
        for (int q = 0; q < m; ++q) ko += 2 * H.nG[H.mgrid[q]];
        const double* ws = sm + H.o_w[g];
        const double* ds = sm + H.o_d[g];
        const double* lxs = sm + H.o_lx[g];
        double acc = 0.0;
        int sg0 = 0, sg1 = nG - 1;
        if (R.rowsplit) {  // (this workgroup's rows: the walker's first workgroup adds the parts)
          const int* rs = reinterpret_cast<const int*>(sm + R.o_rs);
          sg0 = rs[2 * g];
          sg1 = rs[2 * g + 1];
        }
        for (int sgm = sg0 + lane; sgm < sg1; sgm += 64) {
          const double u1 = ws[sgm] * sm[ko + sgm];
          const double u2 = ws[sgm + 1] * sm[ko + sgm + 1];
          const double dl = ds[sgm] + sm[ko + nG + sgm];
          acc += nh_seg_term(u1, u2, dl, lxs[sgm]);
        }
        acc = hs_wave_sum(acc);
        if (lane == 0) sm[D.o_mrow + H.nE + m] = acc;
      }
      if (RT > 0 && nwv - 1 - wv < H.nmom) {
        const int m = nwv - 1 - wv;
        typedef int hsm_i4 __attribute__((ext_vector_type(4)));
        const hsm_i4* mt = reinterpret_cast<const hsm_i4*>(sm + R.o_mt) + 2 * m;  // (built when the launch began)
        const hsm_i4 m0 = mt[0], m1 = mt[1];
        const unsigned aw = (unsigned)m0.x, ad = (unsigned)m0.y, al = (unsigned)m0.z, ak = (unsigned)m0.w;
        const int nG = __builtin_amdgcn_readfirstlane(m1.x);
        const int sg0 = __builtin_amdgcn_readfirstlane(m1.y), sg1 = __builtin_amdgcn_readfirstlane(m1.z);
        double acc = 0.0;
        for (int sgm = sg0 + lane; sgm < sg1; sgm += 64) {
          const double u1 = hs_lds_at(aw, sgm) * hs_lds_at(ak, sgm);
          const double u2 = hs_lds_at(aw, sgm + 1) * hs_lds_at(ak, sgm + 1);
          const double dl = hs_lds_at(ad, sgm) + hs_lds_at(ak, nG + sgm);
          acc += nh_seg_term(u1, u2, dl, hs_lds_at(al, sgm));
        }
        acc = hs_wave_sum(acc);
        if (lane == 0) sm[D.o_mrow + H.nE + m] = acc;
      }
      // ---- C. work items: table reductions and synchrotron nodes, pulled from one counter -----
      {
        const int nT = D.nT;
        const int F0 = min(nT, nwv);
        const int both = 2 * min(nT - F0, nS), total = nT + nS;
        double* part_t = sm + H.o_part_t;
        double* part_s = sm + H.o_part_s;
        bool rt_first = true;
        if constexpr (RT > 0) {
          // The wave's own item, its rows in registers since the launch began (every slice but one
          // whose weights are not finite): NOTHING of the item is looked up again.  The loop below
          // -- which item, which table, its rows -- walks the by-value argument block (D.nT,
          // D.tab[t]: dependent loads from the kernel-argument segment, a division, the trailer):
          // at one item per wave and slice that was most of these instances' items phase.
          const int ixr = __builtin_amdgcn_readfirstlane(rt.ix);
          if (ixr >= 0) {
            const int tg = __builtin_amdgcn_readfirstlane(rt.tile);
            if (!(nz >> (8 + tg) & 1)) {
              double acc = 0.0;
              if (nz >> tg & 1)
                acc = __builtin_amdgcn_readfirstlane(rt.pre) != 0
                          ? hs_rt_compute<(RT > 0 ? RT : 1), false>(rt)
                          : hs_rt_compute<(RT > 0 ? RT : 1), true>(rt);
              // (its slot: the item's, or -- compact slots, four workgroups per walker on -- the wave's)
              sm[H.o_part_t + __builtin_amdgcn_readfirstlane(rt.slot) * 64 + lane] = acc;
              rt_first = false;  // (done: the loop below has nothing to do)
            }
          }
        }
        for (;;) {
          int item = 0, pulled = 0;
          if constexpr (RT > 0) {  // ONE item, the wave's own (its rows may sit in registers)
            if (!rt_first) break;
            rt_first = false;
            item = wv;
          } else {
            if (lane == 0) item = atomicAdd(&hi[HI_CNT], 1);
            item = __builtin_amdgcn_readfirstlane(item);
          }
          pulled = item;  // (K > 1: this workgroup's n-th item -- the slot of its partial sums)
          if (K > 1 && R.rowsplit) {  // this workgroup's share: its part of the rows
            const int per_part = total >> __builtin_ctz(K);  // (a shift: K is a power of two)
            if (item >= per_part) break;
            item += part * per_part;
          } else if (K > 1) {  // ... one of every K items, rotating
            if (item * K >= total) break;
            item = item * K + ((part + item) & (K - 1));
            if (item >= total) continue;
          }
          if (item >= total) break;
          bool is_tab;
          int ix;
          if (R.order == 2) {
            is_tab = item >= nS;
            ix = is_tab ? item - nS : item;
          } else if (item < F0) {
            is_tab = true;
            ix = item;
          } else if (item - F0 < both) {
            is_tab = ((item - F0) & 1) != 0;
            ix = is_tab ? F0 + ((item - F0) >> 1) : (item - F0) >> 1;
          } else {
            is_tab = nT - F0 > nS;
            ix = is_tab ? item - nS : item - nT;
          }
#ifdef NH_LAB  // (build.sh -DNH_LAB: instruction counts by kind of work item; every result is wrong)
          if (R.dbg_skip && (is_tab ? (R.dbg_skip & 2) : (R.dbg_skip & 1))) {
            if (is_tab) part_t[((K > 1 && R.tcompact) ? pulled : ix) * 64 + lane] = 0.0;
            continue;
          }
#endif
          if (is_tab && RT == 0) {  // (the descriptor table is part of every such instance's layout)
            typedef int hsi_i4 __attribute__((ext_vector_type(4)));
            const int* ttab = reinterpret_cast<const int*>(sm + R.o_it);
            const hsi_i4 ei = reinterpret_cast<const hsi_i4*>(ttab + HS_MAX_TAB * HST_N)[ix];
            const int tt = __builtin_amdgcn_readfirstlane(ei.x);
            const hsi_i4* e4 = reinterpret_cast<const hsi_i4*>(ttab + (tt & 15) * HST_N);
            const hsi_i4 e0 = e4[0], e1 = e4[1], e2 = e4[2];
            const int fl = __builtin_amdgcn_readfirstlane(e1.w);
            const int tg = fl & 15;
            const bool pre = (fl & 16) != 0;
            const bool inf = (nz >> (8 + tg) & 1) != 0;  // (a weight that is not finite: every row)
            const int s0 = __builtin_amdgcn_readfirstlane(inf ? (ei.w & 0xffff) : ei.y);
            const int s1 = __builtin_amdgcn_readfirstlane(inf ? (int)((unsigned)ei.w >> 16) : ei.z);
            const unsigned kd_lo = (unsigned)__builtin_amdgcn_readfirstlane(e0.x);
            const unsigned kd_hi = (unsigned)__builtin_amdgcn_readfirstlane(e0.y);
            const unsigned nK = (unsigned)__builtin_amdgcn_readfirstlane(e0.z);
            const int nG = __builtin_amdgcn_readfirstlane(e0.w);
            const int sub = __builtin_amdgcn_readfirstlane(e2.x);
            double acc;
            if (!(nz >> tg & 1) || s0 >= s1) {
              acc = 0.0;
            } else if (sub > 1) {
              const int nKp = __builtin_amdgcn_readfirstlane(e2.y);
              acc = pre ? hs_table_item_packed_v<false, SYN ? 4 : HS_RUN_PK>(kd_lo, kd_hi, nK, nKp, sub, nG, s0, s1,
                                                                            (unsigned)e1.x, (unsigned)e1.y, (unsigned)e1.z, lane)
                        : hs_table_item_packed_v<true, SYN ? 4 : HS_RUN_PK>(kd_lo, kd_hi, nK, nKp, sub, nG, s0, s1,
                                                                           (unsigned)e1.x, (unsigned)e1.y, (unsigned)e1.z, lane);
            } else {
              const int tile = tt >> 4;
              const unsigned o8 = 8u * (unsigned)s0;
              acc = pre ? hs_table_item_v<false>(kd_lo, kd_hi, nK, nG, tile, s0, s1, (unsigned)e1.x + o8,
                                                 (unsigned)e1.y + o8, (unsigned)e1.z + o8, lane)
                        : hs_table_item_v<true>(kd_lo, kd_hi, nK, nG, tile, s0, s1, (unsigned)e1.x + o8,
                                                (unsigned)e1.y + o8, (unsigned)e1.z + o8, lane);
            }
            part_t[((K > 1 && R.tcompact) ? pulled : ix) * 64 + lane] = acc;
          } else if (RT > 0 && is_tab) {
            int t = 0;
            while (t + 1 < D.ntab && ix >= D.tab[t + 1].item0) ++t;
            const hs_tab& tb = D.tab[t];
            const int loc = ix - tb.item0;
            const int tile = loc % tb.tiles, chunk = loc / tb.tiles;
            const int tg = __builtin_amdgcn_readfirstlane(tb.grid);
            const int nG = __builtin_amdgcn_readfirstlane(H.nG[tg]);
            int s0, s1;
            hs_chunk_range(tb.chunks, chunk, D.seg, nG - 1, s0, s1);
            // (rows below the tile's first non-zero one contribute exact zeros: not walked; the
            // tables' trailers -- identity for a table without one -- sit in LDS for the launch)
            {
              const bool inf = (nz >> (8 + tg) & 1) != 0;  // (a weight that is not finite: every row counts)
              const int r0 = __builtin_amdgcn_readfirstlane(
                  reinterpret_cast<const int*>(sm + R.o_trail)[t * HS_RUN_TRAIL + tile]);
              if (R.rebalance) {
                // the tile's chunks share the rows that are WALKED: cut into equal row ranges of
                // the whole grid, the chunks below a pi0 / inverse-Compton threshold are empty and
                // the waves that drew them idle while the others finish (cfg5: the sixteen waves
                // reached the barrier 8.9 ... 11.9 us into the slice)
                // (a weight that is not finite: the SAME ranges, the tile's first chunk reaching down to
                // row 0 -- 0 x inf = NaN for the zero rows, as the reference -- and not the unbalanced
                // chunk ranges: with the rows split between two workgroups those reach, for the second
                // one, below the nodes it formed -- ADVICE r5)
                const int nch = HS_CHUNKS(tb.chunks);
                const int per = (max(nG - 1 - r0, 0) + nch - 1) / nch;
                s0 = r0 + chunk * per;
                s1 = min(nG - 1, s0 + per);
                if (inf && chunk == 0) s0 = 0;
              } else if (!inf) {
                s0 = max(s0, r0);
              }
            }
            const double* kds = t == 0 ? R.kds[0] : t == 1 ? R.kds[1] : t == 2 ? R.kds[2] : R.kds[3];
            const double* ws = sm + __builtin_amdgcn_readfirstlane(H.o_w[tg]);
            const bool pre = __builtin_amdgcn_readfirstlane(tb.nonneg) != 0;
            const double* ds = sm + __builtin_amdgcn_readfirstlane(pre ? H.o_dp[tg] : H.o_d[tg]);
            const double* lxs = sm + __builtin_amdgcn_readfirstlane(pre ? H.o_th[tg] : H.o_lx[tg]);
            double acc;
            if (!(nz >> tg & 1) || s0 >= s1)
              acc = 0.0;
            else if (RT > 0 && rt.ix == ix && !(nz >> (8 + tg) & 1))
              acc = pre ? hs_rt_compute<(RT > 0 ? RT : 1), false>(rt)  // (the rows are in registers)
                        : hs_rt_compute<(RT > 0 ? RT : 1), true>(rt);
            else if (__builtin_amdgcn_readfirstlane(tb.sub) > 1)
              acc = pre ? hs_table_item_packed<false, SYN ? 4 : HS_RUN_PK>(tb, nG, s0, s1, ws, ds, lxs, lane, kds)
                        : hs_table_item_packed<true, SYN ? 4 : HS_RUN_PK>(tb, nG, s0, s1, ws, ds, lxs, lane, kds);
            else
              acc = pre ? hs_table_item<false>(tb, nG, tile, s0, s1, ws, ds, lxs, lane, kds)
                        : hs_table_item<true>(tb, nG, tile, s0, s1, ws, ds, lxs, lane, kds);
            part_t[((K > 1 && R.tcompact) ? pulled : ix) * 64 + lane] = acc;
          } else if (SYN) {
            // (the tile waves' constants were written ahead of barrier 2)
            const int g = H.syn_grid, nEs = H.syn_nE;
            if (S2) {
              hs_syn2_item(ix, lane, nA, Cd, nEs, R.s2, reinterpret_cast<const int*>(sm + H.o_amap) + nEs,
                           reinterpret_cast<const int*>(sm + R.o_s2z), sm + R.o_s2q,
                           hs_lds_addr(sm + R.o_s2lw + HS_S2_GUARD), hs_lds_addr(sm + R.o_s2ig + HS_S2_GUARD),
                           hs_lds_addr(sm + R.o_s2tab), hs_lds_addr(sm + R.o_s2t), part_s);
            } else {
              const hs_syn_lds L = {reinterpret_cast<const int*>(sm + H.o_amap), sm + H.o_ig2,
                                    sm + H.o_dig2, sm + H.o_ig23, sm + H.o_w[g], sm + H.o_d[g],
                                    sm + H.o_lx[g], sm + H.o_sq, sm + HS_O_T64};
              hs_syn_item(ix, lane, nA, Cd, H.nG[g], nEs, L, part_s);
            }
          }
        }
      }
      HSR_STAMP(4);
      HSR_WSTAMP(2);
      HSR_GSTAMP(0);
      __syncthreads();  // ---------------------------------------------------------------- #3
      HSR_STAMP(5);
      HSR_GSTAMP(1);
      // ---- the walker's spectra meet in LDS (every thread its column: one wave alone took 1.9 us
      // over it, 16 waves and a barrier 0.9) ------------------------------------------------
      {
        typedef int hss_i2 __attribute__((ext_vector_type(2)));
        const hss_i2* st = reinterpret_cast<const hss_i2*>(sm + R.o_sum);
        for (int k = tid; k < R.sum_cols; k += T) {
          const hss_i2 d = st[k];
          const int chunks = d.y & 0xff, dst = (int)((unsigned)d.y >> 8);
          double sum = 0.0;
          if (K > 1 && R.tcompact) {
            // this workgroup's items only, each in the slot of its group: chunk c of the column is
            // item ix0 + c tiles -- the g-th of all items (the synchrotron items come first) --
            // mine if it is my member of its group of K (or lies in my rows), slot = the group
            const int ix0 = d.x & 0x3ff, ln = (d.x >> 10) & 63, tiles = (int)((unsigned)d.x >> 16);
            const int lk = __builtin_ctz(K);
            const int per_part = (D.nT + nS) >> lk;
            for (int c0 = 0; c0 < chunks; c0 += 8) {
              double v[8];
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int g = ix0 + (c0 + q) * tiles + nS;
                int slot;
                bool mine;
                if (R.rowsplit) {
                  slot = g - part * per_part;
                  mine = slot >= 0 && slot < per_part;
                } else {
                  slot = g >> lk;
                  mine = g == (slot << lk) + ((part + slot) & (K - 1));
                }
                v[q] = (c0 + q < chunks && mine) ? sm[H.o_part_t + slot * 64 + ln] : 0.0;
              }
              sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
          } else {
            const double* pp = sm + (d.x & 0xffff);
            const int stride = (int)((unsigned)d.x >> 16);
            for (int c0 = 0; c0 < chunks; c0 += 8) {  // eight partial sums in flight, fixed order
              double v[8];
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = c0 + q < chunks ? pp[(c0 + q) * stride] : 0.0;
              sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
          }
          sum *= sm[H.o_scale + dst];
          spec[dst] = sum;
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
      HSR_GSTAMP(2);
      __syncthreads();  // ---------------------------------------------------------------- #4
      HSR_GSTAMP(3);
      if (K > 1) {
        // The K workgroups of this walker meet: workgroups 1 .. K - 1 hand their partial spectra to
        // workgroup 0 as tagged granules { 32 bits of payload | (launch, slice) } -- the records'
        // own hand-off: no drain, no ticket, no barrier on the sending side; a granule that has
        // not been written yet, or belongs to another launch, is recognised and never consumed --
        // and go on to their next slice; workgroup 0 adds the parts in the order of their index
        // (the result does not depend on who arrived when) and carries on to the likelihood.
        // (Round 6 tried the meeting by the likelihood waves alone for the rows-split instance -- a
        // lane per column sums its workgroup's chunks, hands over / takes in, no sum phase of all
        // threads, no barrier 4, no barrier behind the meeting: cfg5 / 256 9.05 -> 8.87 us per
        // half-step (+ 1.9 %), and 2 - 5 vector registers of the 256-register instance spilled;
        // with the second workgroup's ITEM waves handing their chunks over themselves 9.14 -> 9.95.
        // Not kept: profiles/NOTES_r06.md.)
        // (Round 4's meeting -- stores drained, a ticket drawn by an atomic, the last to arrive
        // reads everything back -- was 2.3 us of cfg2's 17; rows split between the workgroups
        // hand over the single-row reductions' parts as well.)
        int jx = j;  // (opaque: the cell's address is formed HERE, not carried -- spilled -- from
        asm volatile("" : "+s"(jx));  // the head of the slice)
        const long long cell = (long long)s * H.nloc + jx;
        const int nsp = D.nspec;
        const int nx = nsp + (R.rowsplit ? H.nmom : 0);
        unsigned long long* xg = reinterpret_cast<unsigned long long*>(R.xspec) +
                                 cell * (long long)(K - 1) * 2 * R.nxmax;
        const unsigned xtag = (R.seq << 8) | (unsigned)(s + 1);
        int tk = tid;  // (made opaque here: as the slice's `tid` its 64-bit multiples were formed at
        asm volatile("" : "+v"(tk));  // the head of the slice and carried -- spilled -- to this point)
        if (part != 0) {
          int px = part;  // (opaque too: a launch-wide invariant the compiler parks in a vector register)
          asm volatile("" : "+s"(px));
          unsigned long long* dst = xg + (long long)(px - 1) * 2 * R.nxmax;
          for (int k = tk; k < nx; k += T) {
            const double v = k < nsp ? spec[k] : sm[D.o_mrow + H.nE + (k - nsp)];
            hs_st_sc1(dst + 2 * k, hs_granule(v, 0, xtag));
            hs_st_sc1(dst + 2 * k + 1, hs_granule(v, 1, xtag));
          }
          continue;  // (the whole workgroup: on to its next slice)
        }
        for (int k = tk; k < nx; k += T) {
          double sum = k < nsp ? spec[k] : sm[D.o_mrow + H.nE + (k - nsp)];
          for (int q = 1; q < K; ++q) {
            const unsigned long long* src = xg + (long long)(q - 1) * 2 * R.nxmax + 2 * k;
            unsigned long long lo = 0, hiw = 0;
            int spins = 0;
            for (;;) {
              lo = hs_ld_sc1(src);
              hiw = hs_ld_sc1(src + 1);
              if ((unsigned)(lo >> 32) == xtag && (unsigned)(hiw >> 32) == xtag) break;
              if (++spins > R.spin_limit ||
                  ((spins & 255) == 0 &&
                   __hip_atomic_load(R.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(R.status, HS_RUN_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi[HI_TICK] = HS_RUN_ERR_TIMEOUT;
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
            sum += __hiloint2double((int)(unsigned)hiw, (int)(unsigned)lo);
          }
          if (k < nsp) spec[k] = sum;
          else sm[D.o_mrow + H.nE + (k - nsp)] = sum;
        }
        __syncthreads();
        if (hi[HI_TICK] != 0) return;  // (a part never came: the launch gives up, as for a record)
      }
      HSR_STAMP(6);
      HSR_GSTAMP(4);
      // ---- D. likelihood + priors (core.py:64-121), the accept, the record: one wave, while
      // wave 0 is already polling for the next slice and the others wait at its first barrier ----
      if (lik_wave) {
        // (two walkers in flight: this tail runs beside the next walker's weights, 2.0 us instead of
        // 0.9; s_setprio 3 for it changed nothing -- cfg3 / 2048 15.75 -> 15.75 M)
        const int nE = H.nE;
        const bool has_prior = D.lp || reinterpret_cast<const nh_prior_pack*>(sm + H.o_pri)->n > 0;
        const double prior = accs[3];
        // (what the accept and the record need besides the likelihood: asked for HERE, with the
        // components' descriptors -- behind the sums they were one more LDS round trip on the one
        // wave every other workgroup's next proposal may be waiting for)
        const double oldlp = accs[2], mlnu_ = accs[1], lg3 = lg[3];
        const int me2 = hi[HI_ME];
        const double q_new = qs[min(lane >> 1, HS_O_LNA - 1)], q_old = olds[min(lane >> 1, 63)];
        double acc = 0.0;
        int nviol = 0, nul = 0;
        const double* lik = sm + H.o_lik;
        // (the components' places in spec and their factors: from LDS, all of them asked for at
        // once -- every component is produced inside the launch, nh_half_step_run_create; a
        // component the model does not have reads a word that is always 0.0, with the factor 0:
        // eight reads in flight and one wait instead of a branch and a round trip per component)
        int coff[NH_MAX_COMP];
        double cscl[NH_MAX_COMP];
#pragma unroll
        for (int q = 0; q < NH_MAX_COMP; ++q) {
          if (RT > 0) {  // (kept in this wave's registers since the launch began: 256 of them per lane there)
            coff[q] = lk_coff[q];
            cscl[q] = lk_cscl[q];
          } else {
            coff[q] = (int)sm[R.o_cmp + 2 * q];
            cscl[q] = sm[R.o_cmp + 2 * q + 1];
          }
        }
        for (int k = lane; k < nE; k += 64) {
          double sv[NH_MAX_COMP];
#pragma unroll
          for (int q = 0; q < NH_MAX_COMP; ++q) sv[q] = sm[coff[q] + (coff[q] == R.o_cmp + 2 * NH_MAX_COMP ? 0 : k)];
          // (the data's columns: every one asked for before the first is used)
          const double cv = lik[k], f = lik[nE + k], wlo = lik[2 * nE + k], whi = lik[3 * nE + k];
          const bool isul = lik[4 * nE + k] != 0.0;
          double m = 0.0;
#pragma unroll
          for (int q = 0; q < NH_MAX_COMP; ++q) m = fma(cscl[q], sv[q], m);
          if (D.nblob) sm[D.o_mrow + k] = m;
          const double mc = m * cv;
          const double d = mc - f;
          const double term = (d * d) * ((d > 0.0) ? whi : wlo);
          nul += isul ? 1 : 0;
          nviol += (isul && mc > f) ? 1 : 0;
          acc += isul ? 0.0 : term;
        }
        if constexpr (DEEP) {
          // (two walkers in flight: this walker's columns of `spec` are in registers -- the next
          // walker's tile waves may zero theirs)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __atomic_store_n(&hx_cur[HX_SPECRD], 1, __ATOMIC_RELAXED);
        }
        int cnt = nviol | (nul << 16);
        hs_wave_sum_dpp(acc, cnt);  // (the totals: in lane 63)
        // (... read out as scalars: every lane carries on with the same numbers, no second trip
        // through the lanes to hand lane 0's result round)
        acc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(acc), 63),
                               __builtin_amdgcn_readlane(__double2loint(acc), 63));
        cnt = __builtin_amdgcn_readlane(cnt, 63);
        {
          nviol = cnt & 0xffff;
          nul = cnt >> 16;
          // quirk kept from core.py:89-92: cl is indexed by the violation count
          if (nul > 0) acc += (double)nviol * (nviol <= nE ? sm[R.o_lcl + nviol] : hsr_log(1.0 - H.cl[nviol]));
          if (has_prior) acc = isinf(prior) ? prior : acc + prior;  // core.py:115-119
        }
        HSR_GSTAMP(5);
        // emcee RedBlueMove.propose for this walker
        const double dd = lg3 + acc - oldlp;
        const bool ok = mlnu_ < dd;  // NaN compares false, as numpy
        // the record of the state after this step: row tl + 1
        double val = 0.0;
        if (lane < GRn) {
          const int d = lane >> 1;
          val = d < ndim ? (ok ? q_new : q_old) : (ok ? acc : oldlp);
          const long long roff = ((long long)(tl + 1) * N + me2) * R.gr + lane;
          const unsigned long long gv = hs_granule(val, lane, hs_tag(R.seq, tl + 1));
          if (!multi) {
            hs_st_sc1(R.ring + roff, gv);
          } else {
            if (R.publish_delay > 0) {
              const long long t0 = (long long)wall_clock64();
              while ((long long)wall_clock64() - t0 < R.publish_delay) __builtin_amdgcn_s_sleep(1);
            }
            // into every rank's ring, this one's included (constant indices: a dynamic one
            // would make the compiler copy the argument block to scratch)
#pragma unroll
            for (int pr = 0; pr < HS_RUN_MAX_RANKS; ++pr)
              if (pr < R.nrank) hs_st_sys(R.peer[pr] + roff, gv);
          }
        }
        HSR_GSTAMP(6);
        if (R.dbg && lane == 0 && blockIdx.x < 256 && it < 64)
          R.dbg[((long long)blockIdx.x * 64 + it) * 8 + 7] = (long long)wall_clock64();
        // chain history: this walker's entry of the step's row (nobody else writes it)
        const long long hrow = R.hrow0 + tl;
        const bool hist = R.hcoords != nullptr && hrow < R.hcap;
        if (hist) {
          if (lane < ndim) R.hcoords[(hrow * N + me2) * ndim + lane] = ok ? qs[lane] : olds[lane];
          if (lane == 0) R.hlogp[hrow * N + me2] = ok ? acc : oldlp;
        }
        if (lane == 0) {
          // (a shared ensemble: the flag says which launch it belongs to -- entries of walkers
          // other ranks move keep an older launch's number and are recognised by it, no memset)
          R.accw[(long long)tl * N + me2] = multi ? (int)((R.seq << 2) | (ok ? 2u : 1u)) : (ok ? 1 : 0);
          if (acc != acc) atomicAdd(R.lcnt, 1);  // (see nh_half_step_nan_count; committed by the epilogue)
          if (hi[HI_DEAD]) atomicAdd(R.lcnt + 1, 1);  // (forbidden by the prior: nothing was integrated)
        }
        if (ok) {  // the accepted position's blobs
          for (int b = 0; b < D.nblob; ++b) {
            const nh_hs_blob& bl = D.blob[b];
            double* hb = hist ? R.hblob[b] : nullptr;
            // (a scalar blob -- We, Wp -- is a lazy transform of a moment: ONE out-of-line call;
            // inlined at both of its uses the library's exp10 / log were 400 instructions of this
            // tail and the only scratch loads of the kernel)
            double sval = 0.0;
            if (bl.kind != 0 && lane == 0)
              sval = hsr_lazy_apply(bl.lazy.a, bl.lazy.b, bl.lazy.c, bl.lazy.tf, sm[D.o_mrow + H.nE + bl.mom]);
            if (hb) {  // (its row of the history; k_run_epilogue fills the rejected ones)
              double* dst = hb + (hrow * N + me2) * bl.m;
              if (bl.kind == 0) {
                for (int t = lane; t < bl.m; t += 64) dst[t] = sm[D.o_mrow + t];
              } else if (lane == 0) {
                dst[0] = sval;
              }
            }
            if (!hb || multi) {
              // no history: the current-blob array itself (also in a shared ensemble, whose
              // ranks each keep the blobs of the moves they accepted, stamped: merged by stamp
              // when somebody asks).  Several workgroups write a walker's row in the course of
              // a launch: write-through, so that the LAST write is the one memory keeps (dirty
              // lines of different L2s have no order)
              unsigned long long* dst = reinterpret_cast<unsigned long long*>(bl.cur + (long long)me2 * bl.m);
              if (bl.kind == 0) {
                for (int t = lane; t < bl.m; t += 64)
                  hs_st_sc1(dst + t, (unsigned long long)__double_as_longlong(sm[D.o_mrow + t]));
              } else if (lane == 0) {
                hs_st_sc1(dst, (unsigned long long)__double_as_longlong(sval));
              }
            }
          }
          if (multi && R.curstamp && lane == 0 && D.nblob > 0)
            __hip_atomic_store(R.curstamp + me2, R.stamp0 + tl, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

// After a launch of k_half_step_run: the flat coords / logp arrays := the ring's last row; the
// acceptance counters += the accept flags; blob history rows of rejected proposals := the
// previous step's (emcee keeps the blobs of the position a walker is AT), the current blobs :=
// the last row.  One thread per (walker, element).
// (its own small argument block: handed hs_hot and hs_run by value -- 3 KB -- the kernel opened
// with every field's scalar load and parked 650 of them in vector-register lanes: 12 us for a
// kernel that moves 5 MB)
struct hs_epi {
  int N, ndim, gr, nrank, nblob, spin_limit, report_launch, pad;
  unsigned seq;
  int* status; int* done; int* lcnt; volatile int* report;
  long long* clk;  // the span clock: the last workgroup out closes the span k_half_step_run opened
  const unsigned long long* ring;
  double* coords; double* logp;
  int* nacc; const int* accw; int* hacc;
  long long hrow0;
  double* hblob[NH_HS_MAX_BLOB]; double* bcur[NH_HS_MAX_BLOB]; int bm[NH_HS_MAX_BLOB];
};
__device__ __forceinline__ void run_epilogue_jobs(const hs_epi& R, int nsteps) {
  const int N = R.N, ndim = R.ndim;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gsz = (long long)gridDim.x * blockDim.x;
  // the three jobs -- ensemble, counters, one per blob -- side by side (blockIdx.y): each is a
  // chain of dependent round trips, one after the other they added up to 15 us of a 20-step
  // region
  const int job = blockIdx.y;
  if (blockIdx.x == 0 && job == 0 && threadIdx.x == 0) {
    // this launch's NaN / forbidden-proposal counts join the plan's counters if -- and only if --
    // it ended well; the report says how the counters stood before (a replay starts from there)
    const int st = *R.status;
    const int n0 = R.done[2], f0 = R.done[3];
    if (st == 0) {
      R.done[2] = n0 + R.lcnt[0];
      R.done[3] = f0 + R.lcnt[1];
    }
    R.lcnt[0] = 0;  // (the pair's next user is the launch after next: behind this kernel in the stream)
    R.lcnt[1] = 0;
    if (R.report) {
      volatile int* rp = R.report + 8 * (R.report_launch & 1);
      rp[0] = st;
      rp[1] = R.done[2];
      rp[2] = R.done[3];
      rp[4] = n0;
      rp[5] = f0;
      __threadfence_system();
      rp[3] = R.report_launch;  // (last: whoever sees the number sees the rest)
    }
  }
  // a launch that gave up (a record never came: nh_half_step_run_status) leaves the flat arrays,
  // the counters and the blob rows as they were before it -- whoever finds the status can replay
  // the block of moves from them
  // (the status is ASKED for here and looked at only where a job is about to store: its loads do
  // not wait for it -- one dependent round trip of three or four less, ~1.5 us each)
  const int st_now = __hip_atomic_load(R.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (R.nrank > 1 && st_now != 0) return;
  const unsigned n_ens = (unsigned)N * (unsigned)(ndim + 1);  // (32-bit: nh_half_step_run_create checks)
  for (unsigned e = job == 0 ? (unsigned)gid : n_ens; e < n_ens; e += (unsigned)gsz) {
    const int w = (int)(e / (unsigned)(ndim + 1)), d = (int)(e % (unsigned)(ndim + 1));
    const unsigned long long* rec = R.ring + ((long long)nsteps * N + w) * R.gr + 2 * d;
    unsigned long long lo = rec[0], hiw = rec[1];
    if (st_now != 0) return;
    if (R.nrank > 1) {
      // a shared ensemble: this rank's launch is over, another rank's last movers may not be --
      // their records are recognised by their tags like any other (bounded wait)
      const unsigned want = hs_tag(R.seq, nsteps);
      int spins = 0;
      for (;;) {
        lo = hs_ld_sys(rec);
        hiw = hs_ld_sys(rec + 1);
        if ((unsigned)(lo >> 32) == want && (unsigned)(hiw >> 32) == want) break;
        if (++spins > R.spin_limit ||
            ((spins & 255) == 0 &&
             __hip_atomic_load(R.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          __hip_atomic_store(R.status, HS_RUN_ERR_LAST_ROW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const double v = __hiloint2double((int)(unsigned)hiw, (int)(unsigned)lo);
    if (d < ndim) R.coords[(long long)w * ndim + d] = v;
    else R.logp[w] = v;
  }
  int* const nacc = R.nacc;
  if (nacc && job == 1)
    for (long long w = gid; w < N; w += gsz) {
      int f[HS_RUN_MAX_STEPS];  // (all the flags asked for at once: one round trip, not nsteps)
#pragma unroll
      for (int t = 0; t < HS_RUN_MAX_STEPS; ++t) f[t] = R.accw[(long long)(t < nsteps ? t : nsteps - 1) * N + w];
      if (R.nrank > 1) {  // (st_now == 0 here)
        // a shared ensemble: { launch number | 1 rejected, 2 accepted } -> -1 not moved by this
        // rank | 0 | 1, also into the history's flags (what the merge of the ranks' rows reads)
#pragma unroll
        for (int t = 0; t < HS_RUN_MAX_STEPS; ++t) {
          const bool mine = (unsigned)f[t] >> 2 == R.seq;
          f[t] = mine ? ((f[t] & 3) == 2 ? 1 : 0) : -1;
          if (R.hacc && t < nsteps) R.hacc[(R.hrow0 + t) * N + w] = f[t];
        }
      }
      int a = 0;
#pragma unroll
      for (int t = 0; t < HS_RUN_MAX_STEPS; ++t) a += (t < nsteps && f[t] > 0) ? 1 : 0;
      const int before = nacc[w];
      if (st_now != 0) return;
      nacc[w] = before + a;
    }
  // (a shared ensemble: the blobs of a walker's earlier steps may be on another rank -- the
  // rows of rejected proposals are filled when the ranks' histories are merged)
  for (int b = 0; b < R.nblob; ++b) {
    double* hb = R.hblob[b];  // (null: no history, or a shared ensemble)
    if (!hb || job != 2 + b) continue;
    const int bm = R.bm[b];
    double* const bcur = R.bcur[b];
    const unsigned n_bl = (unsigned)N * (unsigned)bm;
    for (unsigned e = (unsigned)gid; e < n_bl; e += (unsigned)gsz) {
      const int w = (int)(e / (unsigned)bm);
      // (every load of the column is issued before the first is used: walked one step at a time
      // the fill was a chain of 2 x nsteps dependent round trips, 24 us for 20 steps)
      double cell[HS_RUN_MAX_STEPS];
      int fl[HS_RUN_MAX_STEPS];
#pragma unroll
      for (int t = 0; t < HS_RUN_MAX_STEPS; ++t) {
        // (straight-line, every address valid: steps past the launch's re-read its last one.
        // Behind a condition per step the compiler waited for each load before the next.)
        const long long tt = t < nsteps ? t : nsteps - 1;
        cell[t] = hb[(R.hrow0 + tt) * (long long)N * bm + e];
        fl[t] = R.accw[tt * N + w];
      }
      double prev = bcur[e];
      if (st_now != 0) return;
      unsigned acc = 0;  // (first use of any load: after the last one has been issued)
#pragma unroll
      for (int t = 0; t < HS_RUN_MAX_STEPS; ++t) acc |= (t < nsteps && fl[t]) ? 1u << t : 0u;
#pragma unroll
      for (int t = 0; t < HS_RUN_MAX_STEPS; ++t)
        if (t < nsteps) {
          if (acc >> t & 1) prev = cell[t];
          else hb[(R.hrow0 + t) * (long long)N * bm + e] = prev;
        }
      bcur[e] = prev;
    }
  }
}
__global__ void k_run_epilogue(const hs_epi R, int nsteps) {
  run_epilogue_jobs(R, nsteps);
  // the span clock: whoever is the last workgroup out closes the span the launch opened
  __syncthreads();
  if (threadIdx.x == 0) {
    int* through = reinterpret_cast<int*>(R.clk + 3);
    const int total = (int)(gridDim.x * gridDim.y);
    if (__hip_atomic_fetch_add(through, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
      __hip_atomic_store(through, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nh_clk_close(R.clk);
    }
  }
}

// ---------------------------------------------------------------------------------------------
struct nh_halfstep_run {
  hs_run R;
  unsigned long long* ring;
  int* status;
  int* accw;
  long long* dbg;
  double* xspec;
  int* tick;
  int split;
  size_t lds_bytes;
  int grid, threads;
  int rt;  // the instance whose table items stay in registers (hs_rt_item)
  int deep;  // the instance with two walkers in flight (DEEP)
  int* report;        // page-locked host memory, two slots of eight ints (hs_run.report)
  int* lcnt;          // device: two pairs of launch-local counters (hs_run.lcnt)
  int nlaunch, fail_at;  // launches so far; NH_RUN_FAIL_AT = the launch whose first wait times out (tests)
  unsigned seq;
  // a shared ensemble (nh_half_step_run_create_shared): `base` is ONE fine-grained allocation
  // { HS_RUN_HEAD granules of probe slots | ring of even launches | ring of odd launches }
  // that the other ranks map (hipIpc), peer_base[p] is rank p's as mapped here
  int nrank, rank;
  unsigned long long* base;
  unsigned long long* peer_base[HS_RUN_MAX_RANKS];
  size_t ring_elems, base_bytes;
  int* nacc_own;
  int* curstamp;
  int* hacc;          // where the NEXT launch keeps its history flags (nh_half_step_run_hist_flags)
  int* probe_out;
  unsigned probe_seq;
  long long steps_total;
  double* s2_dev;     // the log-domain synchrotron items' table and grid constants (nh_syn2.h), or NULL
};

#define HS_RUN_RT HS_RT_NODES
static const void* hs_run_kernel(bool syn, bool shared, bool s2, bool rt = false, bool deep = false) {
  if (deep && !shared && !rt && syn) {
    return s2 ? (const void*)k_half_step_run<true, false, true, 0, true>
              : (const void*)k_half_step_run<true, false, false, 0, true>;
  }
  if (!syn && rt)
    return shared ? (const void*)k_half_step_run<false, true, false, HS_RUN_RT>
                  : (const void*)k_half_step_run<false, false, false, HS_RUN_RT>;
  if (!syn) return shared ? (const void*)k_half_step_run<false, true, false> : (const void*)k_half_step_run<false, false, false>;
  if (s2) return shared ? (const void*)k_half_step_run<true, true, true> : (const void*)k_half_step_run<true, false, true>;
  return shared ? (const void*)k_half_step_run<true, true, false> : (const void*)k_half_step_run<true, false, false>;
}

static int hs_run_create(nh_ctx* c, nh_halfstep_plan* P, int rank, int nrank, nh_halfstep_run** out) {
  NH_REQUIRE(c && P && out, "bad argument");
  const hs_hot& H = P->hot;
  const bool shared = nrank > 1;
  NH_REQUIRE(shared || H.C.do_accept, "the resident loop needs the in-launch accept");
  NH_REQUIRE(H.ndim <= 15, "at most 15 fit parameters in a record (32 granules per wave half)");
  NH_REQUIRE(shared || (H.lo == 0 && H.nloc == H.ns), "the resident loop moves whole half-ensembles");
  NH_REQUIRE(!shared || (nrank <= HS_RUN_MAX_RANKS && rank >= 0 && rank < nrank && H.nloc >= 1 &&
                         H.lo >= 0 && H.lo + H.nloc <= H.ns),
             "a shared ensemble: at most 8 ranks, each with at least one walker of every half-step");
  NH_REQUIRE(H.C.lp == nullptr, "a prior evaluated by a launch of its own cannot ride in the resident loop");
  for (int b = 0; b < H.C.nblob; ++b)
    NH_REQUIRE(2LL * H.ns * H.C.blob[b].m < (1LL << 31), "a blob's rows of one step: too large for 32-bit element offsets");
  for (int q = 0; q < H.C.ncomp; ++q)
    NH_REQUIRE(H.C.comp[q].off >= 0, "every component of the model must be produced inside the launch");
  hs_run R;
  memset(&R, 0, sizeof(R));
  R.N = 2 * H.ns;
  R.gr = ((2 * (H.ndim + 1) + 15) / 16) * 16;
  // LDS: the plan's layout, then what stays resident on top of it
  int off = (int)(P->lds_core / sizeof(double));
  // Workgroups of 1024 threads own their CU: the grids' nodes, ln E (and E where the particle
  // distribution has a break) stay in LDS for the whole launch.  Smaller workgroups (a half-step
  // of more walkers than CUs, k_half_step item 14) share a CU, and LDS decides how many fit:
  // they read the nodes from L2 every slice, as k_half_step does, while a neighbour computes.
  // RT: a table-only model in workgroups of <= 512 threads (256 vector registers per lane) keeps
  // its table items' rows in registers (nh_hs.h: hs_rt_item); a workgroup owns its CU then, too
  bool rt = P->rt != 0 && H.syn_grid < 0 && H.ntab > 0 && P->threads <= 512 && nh_env_int("NH_RUN_RT", 1) != 0;
  rt = rt && H.C.nT <= (P->threads / 64) * P->split;  // (one item per wave)
  const bool grids_in_lds = (P->threads >= 1024 || rt) && nh_env_int("NH_RUN_GRIDS_IN_LDS", 1) != 0;
  for (int g = 0; g < NH_MAX_GRIDS; ++g) R.o_gx[g] = R.o_lne[g] = R.o_ge[g] = -1;
  for (int g = 0; g < H.ngrids && grids_in_lds; ++g) {
    R.o_gx[g] = off; off += H.nG[g];
    R.o_lne[g] = off; off += H.nG[g];
    R.o_ge[g] = off;
    if (H.F.broken) off += H.nG[g];
  }
  // ---- the table grids' 1 / lx in LDS, and the weights' units (below)
  for (int g = 0; g < NH_MAX_GRIDS; ++g) R.o_il[g] = -1;
  R.order = nh_env_int("NH_RUN_ORDER", 2);
  R.rebalance = nh_env_int("NH_RUN_REBALANCE", H.syn_grid < 0 ? 1 : 0);
  const bool units_tab = grids_in_lds && nh_env_int("NH_RUN_UT", 1) != 0;
  // (the plan chose two workgroups per walker for a table-only model with one table: they halve
  // the grids' rows when the units' table is there -- the grids' nodes in LDS -- the chunks divide
  // the walked rows and come in a whole number per workgroup; else they interleave the items)
  R.rowsplit = (P->rowsplit && P->split >= 2 && units_tab && R.rebalance && H.ntab == 1 &&
                HS_CHUNKS(H.C.tab[0].chunks) % P->split == 0 && H.C.nT % P->split == 0 &&
                nh_env_int("NH_RUN_ROWSPLIT", 1) != 0) ? 1 : 0;
  // K > 1 workgroups per walker: a workgroup computes one work item of every K, and keeps its
  // table items' partial sums in the slot of the GROUP (its n-th item: slot n) -- a K-th of the
  // plan's block of them, which k_half_step fills in full; what that frees (54 KB at four
  // workgroups per walker) is the first place the loop's own arrays go
  // (from four workgroups per walker on: with two, clearing the whole block every slice and
  // summing it blindly measured faster than deciding per chunk whose it is -- cfg5 / 256 10.9
  // against 10.3 M walker-steps/s, cfg3 / 256 7.12 against 6.94; cfg3 / 128, four per walker: 3.95
  // against 4.07 the other way, and its log-domain synchrotron table fits LDS again)
  R.tcompact = (P->split >= nh_env_int("NH_RUN_TCOMPACT_MIN", 4) && (R.rowsplit || R.order == 2)) ? 1 : 0;
  int hole_lo = 0, hole_hi = 0;
  if (R.tcompact) {
    int nsmax = 0;  // synchrotron items of a slice at most
    if (H.syn_grid >= 0) nsmax = (H.syn_nE * H.C.syn_cdmax + 63) / 64;
    const int nslots = R.rowsplit ? H.C.nT / P->split : (H.C.nT + nsmax) / P->split + 2;
    if (nslots < H.C.nT) {
      hole_lo = H.o_part_t + nslots * 64;
      hole_hi = H.o_part_t + H.C.nT * 64;
    } else {
      R.tcompact = 0;
    }
  }
  auto take = [&](int n, bool align16) {
    if (align16) hole_lo += hole_lo & 1;
    if (hole_lo + n <= hole_hi) {
      const int at = hole_lo;
      hole_lo += n;
      return at;
    }
    if (align16) off += off & 1;
    const int at = off;
    off += n;
    return at;
  };
  R.o_ut = -1;
  if (units_tab) {
    int nunits = 0;
    for (int g = 0; g < H.ngrids; ++g) nunits += (H.nG[g] + 63) / 64;
    R.o_ut = take(nunits * (HSU_N / 2), true);  // (a unit's descriptor is read as four ds_read_b128)
  }
  {
    int ncols = 0;
    for (int t = 0; t < H.ntab; ++t) ncols += H.C.tab[t].nK;
    R.sum_cols = ncols;
    R.o_sum = take(ncols, false);
    R.o_cmp = take(2 * NH_MAX_COMP + 2, false);
    R.o_pci = take((NH_MAX_PRIOR + 1) / 2, false);
    R.o_synce = H.syn_grid >= 0 ? take(H.syn_nE, false) : -1;
    R.o_lnt = -1;  // (allotted with the log-domain block, below)
    R.o_mt = take(4 * NH_MAX_MOMENT, true);
    {
      int nun = 0;
      for (int g = 0; g < H.ngrids; ++g) nun += (H.nG[g] + 63) / 64;
      R.o_rs = take((9 + nun + 2 * NH_MAX_GRIDS + 1) / 2, false);
    }
    R.nxmax = H.C.nspec + NH_MAX_MOMENT;
  }
  for (int t = 0; t < H.ntab; ++t)
    NH_REQUIRE(H.nG[H.C.tab[t].grid] < 65536, "a table's grid has too many nodes for the items' descriptors");
  R.o_it = -1;
  if (!rt && H.C.nT > 0)  // (the register-resident instance keeps its rows per wave: no table)
    R.o_it = take((HS_MAX_TAB * HST_N + H.C.nT * HSI_N + 1) / 2, true);
  R.o_pk = take(NH_MAX_PACK * NH_MAX_LAZY * HS_RUN_PKW, false);
  R.o_small1 = take(HS_O_T64, false);
  R.o_olds = take(128, false);
  R.o_lcl = take(H.nE + 1, false);
  R.o_trail = take((HS_RUN_TRAIL * HS_MAX_TAB + H.C.nspec + 1) / 2, false);
  for (int t = 0; t < H.ntab; ++t)
    NH_REQUIRE(H.C.tab[t].tiles <= HS_RUN_TRAIL, "a table of more column tiles than the resident loop stages");
  // ---- the synchrotron items in the log domain (nh_syn2.h): a log-uniform grid only -------------
  std::vector<double> s2_host;
  R.syn2 = 0;
  R.s2_dev = nullptr;
  if (H.syn_grid >= 0 && nh_env_int("NH_RUN_SYN2", 1) != 0) {
    const int nG = H.nG[H.syn_grid];
    std::vector<double> gam((size_t)nG);
    int rc = nh_sync(c);
    if (rc) return rc;
    NH_CHECK_HIP(hipMemcpy(gam.data(), H.xg[H.syn_grid], (size_t)nG * sizeof(double), hipMemcpyDeviceToHost));
    bool ok = nG >= 8 && gam[0] > 0.0 && H.scale[H.syn_grid] > 0.0;
    long double lx = 0.0L, dev = 0.0L;
    if (ok) {
      const long double l0 = logl((long double)gam[0]);
      lx = (logl((long double)gam[nG - 1]) - l0) / (nG - 1);
      for (int i = 0; i < nG && ok; ++i) {
        ok = gam[i] > 0.0;
        if (ok) dev = fmaxl(dev, fabsl(logl((long double)gam[i]) - l0 - i * lx));
      }
      // (np.logspace, radiative.py:147-154: the nodes sit on i lx to a few 1e-15; 1e-12 in ln gamma
      // is 7e-13 in ln Gtilde)
      ok = ok && lx > 0.0L && dev <= 1e-12L;
    }
    int lm = 0;
    if (ok) {
      const double delta = (double)(2.0L * lx);
      lm = (int)lrint(log2(0.16 / delta));
      ok = lm >= 1 && lm <= 4;  // pieces of m = 2 .. 16 comb steps, h = m delta in [0.113, 0.226]
    }
    if (ok) {
      const int m = 1 << lm;
      const long double h = m * 2.0L * lx;
      const int P = (int)ceill(((long double)HS_S2_TTOP - (long double)HS_S2_TBOT) / h);
      s2_host.assign((size_t)(P + 1) * HS_S2_STRIDE + nG, 0.0);
      for (int pc = 0; pc < P; ++pc) hs_s2_piece(pc, h, &s2_host[(size_t)pc * HS_S2_STRIDE]);
      // (piece P, and every node below the table: rho = 0, i.e. Gtilde = 1.808 x^(1/3))
      s2_host[(size_t)P * HS_S2_STRIDE] = (double)((long double)HS_S2_LAMBDA * logl(1.808L));
      for (int i = 0; i < nG; ++i)
        s2_host[(size_t)(P + 1) * HS_S2_STRIDE + i] =
            (double)((long double)HS_S2_LAMBDA * (logl((long double)gam[i]) / 3.0L + logl((long double)H.scale[H.syn_grid])));
      R.s2.lm = lm; R.s2.P = P; R.s2.nG = nG; R.s2.pad = 0;
      R.s2.ilx = (double)(HS_S2_C / lx);  // (ln 2 / 1024) / lx
      R.s2.th = (double)((long double)NH_SEG_SMALL_POS / lx);
      R.s2.im = 1.0 / m;
      R.s2.lml = (double)(m - 1) / m;
      R.s2_invd = (double)(1.0L / (2.0L * lx));
      // w = gamma n scale = (E / mec2[eV]) n scale != 0  <=>  ln n + ln E > ln(2^-1075) + ln(mec2 / scale)
      R.s2_lnw0 = (double)(-1075.0L * logl(2.0L) + logl((long double)NH_MEC2_EV / (long double)H.scale[H.syn_grid]));
      R.s2_z0 = (double)(((long double)HS_S2_TTOP + 2.0L * logl((long double)gam[0])) / (2.0L * lx));
      R.s2_r746 = (double)(((long double)HS_S2_TTOP - logl(746.0L)) / (2.0L * lx) -
                           ((long double)HS_S2_TTOP + 2.0L * logl((long double)gam[0])) / (2.0L * lx));
      const int off_was = off, hole_was = hole_lo;  // (tentative: undone if it does not fit)
      R.o_s2tab = take((P + 1) * HS_S2_STRIDE, true);  // (16-byte aligned: the pieces are read as ds_read_b128)
      // the plan's LDS holds w | dlw of the synchrotron grid and 1/g^2 | its differences | its
      // cube roots, none of which these items read: when no table and no single-row reduction
      // shares the grid, the new arrays take their place (cfg3 with four workgroups per walker
      // has 142 KB of partial sums and arrays before this table)
      const int sg = H.syn_grid;
      bool own = nG >= 2 * HS_S2_GUARD && H.o_d[sg] == H.o_w[sg] + nG && H.o_ig23 == H.o_dig2 + nG;
      for (int t = 0; t < H.ntab; ++t) own = own && H.C.tab[t].grid != sg;
      for (int q = 0; q < H.nmom; ++q) own = own && H.mgrid[q] != sg;
      R.s2_own = own ? 1 : 0;
      if (own) {
        R.o_s2lw = H.o_w[sg];   // (w, dlw: 2 nG doubles in a row)
        R.o_s2ig = H.o_dig2;    // (dig2, ig23: 2 nG doubles in a row)
      } else {
        R.o_s2lw = take(nG + 2 * HS_S2_GUARD, false);
        R.o_s2ig = take(nG + 2 * HS_S2_GUARD, false);
      }
      if (own && grids_in_lds) {
        R.o_s2lg = R.o_gx[sg];  // (gamma itself: only the weights w = gamma n read it)
      } else {
        R.o_s2lg = take(nG, false);
      }
      R.o_s2q = take(4 * H.syn_nE, false);
      R.o_s2z = take((H.syn_nE + 1) / 2, false);
      R.o_s2t = take(HS_S2_TN, false);
      R.o_lnt = take(256, true);
      if ((size_t)off * sizeof(double) <= 160 * 1024) {
        R.syn2 = 1;
      } else {
        off = off_was;
        hole_lo = hole_was;
      }
    }
  }
  // (1 / lx of the table grids: optional -- where LDS has room behind everything else)
  if (nh_env_int("NH_RUN_IL", 1) != 0)
    for (int g = 0; g < H.ngrids; ++g)
      if (H.o_dp[g] >= 0 && (hole_lo + H.nG[g] <= hole_hi || (size_t)(off + H.nG[g]) * sizeof(double) <= 158 * 1024))
        R.o_il[g] = take(H.nG[g], false);
  R.syn_nodes = H.C.syn_nodes;
  if (P->split == 1 && R.syn_nodes < 32) R.syn_nodes = R.syn2 ? 48 : 32;
  if (P->split == 2 && R.syn_nodes < 10) R.syn_nodes = 10;  // (cfg2: 901 -> 885 us; 914 at 16)
  // (the log-domain items start on a piece boundary with a node and six coefficients of their
  // own: short items pay for that -- cfg3 / 128, four workgroups per walker, us per half-step:
  // 18.4 at the plan's 4 nodes, 16.9 at 16, 17.7 at 24, direct form 17.2; cfg3 / 256, two per
  // walker: 20.8 at 10, 20.8 at 16, 20.0 at 24; cfg2 / 256: 17.7 at 10 and at 16, 18.3 at 24)
  if (R.syn2 && P->split >= 2 && R.syn_nodes < 16) R.syn_nodes = 16;
  R.syn_nodes = nh_env_int("NH_RUN_SYN_NODES", R.syn_nodes);
#ifdef NH_LAB
  R.dbg_skip = nh_env_int("NH_RUN_DEBUG_SKIP", 0);
#else
  R.dbg_skip = 0;  // (the product build has no switch that drops work: -DNH_LAB, scripts/lab/r4_count.sh)
#endif
  R.pipeline = nh_env_int("NH_RUN_PIPELINE", 1);
  if (R.dbg_skip != 0)
    fprintf(stderr, "libnaima_hip: NH_RUN_DEBUG_SKIP=%d -- work items are DROPPED from the likelihood "
                    "(instruction-count experiments): every result of this loop is wrong\n", R.dbg_skip);
  if (nh_env_int("NH_RUN_FAIL_AT", 0) > 0)
    fprintf(stderr, "libnaima_hip: NH_RUN_FAIL_AT=%d -- that launch of the resident loop is made to time "
                    "out (fault injection of the tests)\n", nh_env_int("NH_RUN_FAIL_AT", 0));
  NH_REQUIRE(R.syn_nodes >= 1, "NH_RUN_SYN_NODES must be positive");
  const size_t lds = (size_t)off * sizeof(double);
  NH_REQUIRE(lds <= 160 * 1024, "the resident loop's working set does not fit in LDS");
  const void* fn = hs_run_kernel(H.syn_grid >= 0, shared, R.syn2 != 0, rt);
  if (lds > 64 * 1024)
    NH_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // every workgroup of the launch has to be resident (they wait for each other's records)
  int per_cu = 0, ncu = 0, devid = 0;
  NH_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, P->threads, lds));
  NH_CHECK_HIP(hipGetDevice(&devid));
  NH_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, devid));
  NH_REQUIRE(per_cu >= 1 && ncu >= 1, "the resident kernel does not fit a compute unit");
  // (MI355X_MICROARCH.md: the query can be one block per CU high where the SGPR file is what
  // limits residency -- 6 waves per SIMD at this kernel's ~110 SGPRs; its 128 VGPRs allow 4,
  // so registers or LDS bind first and the query is exact.  The bounded waits are the net.)
  {  // (ranks that rehearse a multi-GPU run on one device: a share of its CUs each, nh_halfstep.hip)
    const int share = nh_env_int("NAIMA_AMD_CU_SHARE", 1);
    if (share > 1) ncu = ncu / share > 1 ? ncu / share : 1;
  }
  long long cap = (long long)per_cu * ncu;
  if (const char* e = getenv("NH_RUN_GRID")) cap = atoi(e) > 0 ? atoi(e) : cap;
  // More walkers per half-step than resident workgroups: a workgroup takes several of a slice,
  // one after the other (dependencies still point backwards in (slice, walker) order).  Pays for
  // workgroups that own their CU -- cfg3 at 1024 / 2048 walkers: 7.8 / 8.1 M walker-steps/s
  // launched per half-step, 10.9 / 11.2 M resident -- not for the small workgroups of a
  // table-only model (cfg5 at 1024 walkers per half-step ran 19.6 M walker-steps/s with two
  // walkers per workgroup and slice, two 256-thread workgroups per CU, against 24.2 M launched
  // per half-step, four).  NH_RUN_MAX_PER_WG overrides (1: never).
  const long long per_wg = nh_env_int("NH_RUN_MAX_PER_WG", P->threads >= 1024 ? 8 : 1);
  NH_REQUIRE((long long)H.nloc * P->split <= cap * (per_wg > 1 ? per_wg : 1),
             "more walkers per half-step than resident workgroups");
  nh_halfstep_run* Q = new nh_halfstep_run();
  Q->R = R;
  Q->lds_bytes = lds;
  Q->threads = P->threads;
  Q->rt = rt ? 1 : 0;
  Q->grid = (int)(H.nloc < cap / P->split ? H.nloc : cap / P->split);
  // two walkers in flight (the DEEP instance): one GPU's loop whose workgroups own their CU, have
  // the walker to themselves and take more than one of a slice -- a model with a synchrotron
  // component (the 1024-thread instances; a table-only model's ensembles of that size run in small
  // workgroups that share a CU, where the hardware overlaps them).  NH_RUN_PIPELINE=0: never.
  // (With phase A alone made ahead cfg2 -- synchrotron items only, twenty of them for sixteen waves --
  // measured 1 % slower and models without table items were left out; with the priors made ahead as
  // well it gains like cfg3: cfg2 / 1024 13.61 -> 14.01 M, / 2048 14.01 -> 14.73 M.)
  Q->deep = 0;
  if (!shared && !rt && H.syn_grid >= 0 && P->split == 1 && P->threads >= 1024 && H.nloc > Q->grid &&
      R.pipeline != 0) {
    const void* fd = hs_run_kernel(H.syn_grid >= 0, false, R.syn2 != 0, false, true);
    int per_cu_d = 0;
    hipError_t ed = lds > 64 * 1024 ? hipFuncSetAttribute(fd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                    : hipSuccess;
    if (ed == hipSuccess) ed = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_d, fd, P->threads, lds);
    if (ed == hipSuccess && per_cu_d >= per_cu) Q->deep = 1;  // (as resident as the instance the grid was sized for)
    else (void)hipGetLastError();
  }
  Q->seq = 1;
  Q->ring = nullptr; Q->status = nullptr; Q->accw = nullptr; Q->dbg = nullptr;
  Q->xspec = nullptr; Q->tick = nullptr; Q->split = P->split;
  Q->nrank = shared ? nrank : 1; Q->rank = shared ? rank : 0;
  Q->base = nullptr; Q->nacc_own = nullptr; Q->curstamp = nullptr; Q->hacc = nullptr;
  Q->probe_out = nullptr; Q->steps_total = 0; Q->probe_seq = 1; Q->s2_dev = nullptr;
  Q->report = nullptr; Q->lcnt = nullptr; Q->nlaunch = 0; Q->fail_at = nh_env_int("NH_RUN_FAIL_AT", 0);
  Q->R.report = nullptr;
  for (int p = 0; p < HS_RUN_MAX_RANKS; ++p) Q->peer_base[p] = nullptr;
  Q->ring_elems = (size_t)(HS_RUN_MAX_STEPS + 1) * R.N * R.gr;
  const size_t ring_bytes = Q->ring_elems * sizeof(unsigned long long);
  hipError_t e = hipSuccess;
  if (!shared) {  // (a shared ensemble's ranks agree on a launch's fate by other means)
    void* rp = nullptr;
    e = hipHostMalloc(&rp, 16 * sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) {
      Q->report = static_cast<int*>(rp);
      memset(Q->report, 0, 16 * sizeof(int));
      Q->R.report = Q->report;
    }
  }
  if (shared) {
    // Fine-grained device memory: what another GPU stores into it over xGMI is visible to a
    // kernel that is already running here, and system-scope loads are served from memory, not
    // from an L2 that the incoming writes never pass (coarse-grained memory is coherent with
    // other agents at kernel boundaries only).  NH_RUN_SHARED_ALLOC = 1 fine-grained (default) |
    // 2 uncached | 0 plain hipMalloc (two processes on ONE GPU: same memory, same L2s).
    Q->base_bytes = HS_RUN_HEAD * sizeof(unsigned long long) + 2 * ring_bytes;
    const int how = nh_env_int("NH_RUN_SHARED_ALLOC", 1);
    void* b = nullptr;
    if (how == 0) e = hipMalloc(&b, Q->base_bytes);
    else e = hipExtMallocWithFlags(&b, Q->base_bytes, how == 2 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    Q->base = static_cast<unsigned long long*>(b);
    if (e == hipSuccess) e = nh_fill_now(c, Q->base, 0, Q->base_bytes);
    Q->peer_base[Q->rank] = Q->base;
    if (e == hipSuccess) e = hipMalloc(&Q->nacc_own, (size_t)R.N * sizeof(int));
    if (e == hipSuccess) e = nh_fill_now(c, Q->nacc_own, 0, (size_t)R.N * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&Q->curstamp, (size_t)R.N * sizeof(int));
    if (e == hipSuccess) e = nh_fill_now(c, Q->curstamp, 0xFF, (size_t)R.N * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&Q->probe_out, (4 + HS_RUN_MAX_RANKS) * sizeof(int));
  } else {
    if (e == hipSuccess) e = hipMalloc(&Q->ring, ring_bytes);
    if (e == hipSuccess) e = nh_fill_now(c, Q->ring, 0, ring_bytes);
  }
  if (e == hipSuccess && R.syn2) {
    e = hipMalloc(&Q->s2_dev, s2_host.size() * sizeof(double));
    if (e == hipSuccess)
      e = nh_put_now(c, Q->s2_dev, s2_host.data(), s2_host.size() * sizeof(double));
  }
  if (e == hipSuccess) e = hipMalloc(&Q->lcnt, 4 * sizeof(int));
  if (e == hipSuccess) e = nh_fill_now(c, Q->lcnt, 0, 4 * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&Q->status, sizeof(int));
  if (e == hipSuccess) e = nh_fill_now(c, Q->status, 0, sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&Q->accw, (size_t)HS_RUN_MAX_STEPS * R.N * sizeof(int));
  if (e == hipSuccess) e = nh_fill_now(c, Q->accw, 0, (size_t)HS_RUN_MAX_STEPS * R.N * sizeof(int));
  if (e == hipSuccess && P->split > 1) {
    // per (slice, walker): what workgroups 1 .. K - 1 hand over, two tagged granules per double
    // (no clearing between launches: the tags carry the launch's number)
    const size_t cellw = (size_t)(P->split - 1) * 2 * Q->R.nxmax;
    e = hipMalloc(&Q->xspec, (size_t)2 * HS_RUN_MAX_STEPS * H.nloc * cellw * sizeof(unsigned long long));
    if (e == hipSuccess)
      e = nh_fill_now(c, Q->xspec, 0, (size_t)2 * HS_RUN_MAX_STEPS * H.nloc * cellw * sizeof(unsigned long long));
  }
  if (e == hipSuccess)
    if (const char* dv = getenv("NH_HS_DEBUG"))
      if (atoi(dv) != 0) {
        e = hipMalloc(&Q->dbg, (256 * 64 * 8 + 64 * 4 * 16) * sizeof(long long));
        if (e == hipSuccess) e = nh_fill_now(c, Q->dbg, 0, (256 * 64 * 8 + 64 * 4 * 16) * sizeof(long long));
      }
  if (e != hipSuccess) {
    if (Q->ring) (void)hipFree(Q->ring);
    if (Q->base) (void)hipFree(Q->base);
    if (Q->nacc_own) (void)hipFree(Q->nacc_own);
    if (Q->curstamp) (void)hipFree(Q->curstamp);
    if (Q->probe_out) (void)hipFree(Q->probe_out);
    if (Q->status) (void)hipFree(Q->status);
    if (Q->lcnt) (void)hipFree(Q->lcnt);
    if (Q->accw) (void)hipFree(Q->accw);
    if (Q->dbg) (void)hipFree(Q->dbg);
    if (Q->xspec) (void)hipFree(Q->xspec);
    if (Q->tick) (void)hipFree(Q->tick);
    if (Q->s2_dev) (void)hipFree(Q->s2_dev);
    if (Q->report) (void)hipHostFree(Q->report);
    delete Q;
    return nh_set_error(NH_EHIP, "resident half-step loop: %s", hipGetErrorString(e));
  }
  Q->R.ring = Q->ring; Q->R.status = Q->status; Q->R.accw = Q->accw; Q->R.dbg = Q->dbg;
  Q->R.xspec = Q->xspec; Q->R.tick = Q->tick; Q->R.s2_dev = Q->s2_dev;
  Q->R.clk = c->clk;
  Q->R.spin_limit = 1 << 22;  // ~1 s of polling: a record that has not come by then never will
  // (a shared ensemble: the ranks' hosts launch on their own clocks; a rank may have to wait for
  // another one's launch to START -- ~16 s)
  if (shared) Q->R.spin_limit = 1 << 26;
  if (const char* sl = getenv("NH_RUN_SPIN_LIMIT")) Q->R.spin_limit = atoi(sl) > 0 ? atoi(sl) : Q->R.spin_limit;
  Q->R.nrank = Q->nrank; Q->R.rank = Q->rank;
  Q->R.publish_delay = shared ? nh_env_int("NH_RUN_PUBLISH_DELAY", 0) : 0;
  Q->R.nacc_own = Q->nacc_own; Q->R.curstamp = Q->curstamp;
  *out = Q;
  return NH_OK;
}

extern "C" int nh_half_step_run_create(nh_ctx* c, nh_halfstep_plan* P, nh_halfstep_run** out) {
  return hs_run_create(c, P, 0, 1, out);
}

// ---- The ensemble across GPUs -----------------------------------------------------------------
// Walkers shard over the ranks of a node as in the per-launch loop (rank r proposes positions
// [lo, lo + nloc) of every half-step; the move stream is replicated), but the one exchange of a
// half-step -- the new state of every moved walker -- is no longer a collective between launches:
// a mover stores its walker's record (96 bytes for cfg3) into EVERY rank's ring, its own
// included, with system-scope write-through stores (over xGMI for the others), and consumers
// poll their LOCAL ring exactly as on one GPU.  Tags make a record that has not arrived yet
// unmistakable; nothing else is exchanged and no rank waits for more than the two records a
// proposal needs.  Rings alternate between two buffers by the launch number's parity: a rank
// cannot finish launch L + 1 (its epilogue needs every rank's last records of L + 1) before
// every rank has started L + 1, i.e. finished reading launch L's ring, so whoever writes launch
// L + 2's records into that buffer finds no reader of launch L left.  Chain history, blobs and
// acceptance counts stay where they were produced (each rank: the moves it made, flagged in
// `hacc`, the blobs' current values stamped) and are merged when somebody reads them.
extern "C" int nh_half_step_run_create_shared(nh_ctx* c, nh_halfstep_plan* P, int rank, int nrank,
                                              nh_halfstep_run** out) {
  NH_REQUIRE(nrank >= 2, "a shared ensemble has at least two ranks");
  return hs_run_create(c, P, rank, nrank, out);
}

// the 64-byte handle other processes map this rank's rings with
extern "C" int nh_half_step_run_export(nh_ctx* c, nh_halfstep_run* Q, void* handle64) {
  NH_REQUIRE(c && Q && handle64 && Q->base, "not a shared-ensemble loop");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h;
  NH_CHECK_HIP(hipIpcGetMemHandle(&h, Q->base));
  memcpy(handle64, &h, 64);
  return NH_OK;
}

extern "C" int nh_half_step_run_attach(nh_ctx* c, nh_halfstep_run* Q, int peer, const void* handle64) {
  NH_REQUIRE(c && Q && handle64 && Q->base, "not a shared-ensemble loop");
  NH_REQUIRE(peer >= 0 && peer < Q->nrank && peer != Q->rank, "bad peer rank");
  NH_REQUIRE(Q->peer_base[peer] == nullptr, "peer attached twice");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  NH_CHECK_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  Q->peer_base[peer] = static_cast<unsigned long long*>(p);
  return NH_OK;
}

// `rounds` exchanges of a tagged granule with every peer inside ONE launch: round r is stored
// into slot [my rank] of every peer's head, then slot [p] of the local head is polled for every
// peer p until it shows round r or a later one (tags count up across rounds and probes).  It passes only if stores made by another GPU while this kernel runs become visible
// to its polling loads -- the property the shared loop stands on.
struct hs_probe_args {
  unsigned long long* peer[HS_RUN_MAX_RANKS];
  int nrank, rank, rounds, spin_limit;
  unsigned seq;
  int* out;  // status (0 ok) | rounds completed | 100 MHz ticks of all rounds | spins | then
             // per peer: the last round whose granule was seen (diagnostics of a failed probe)
};
__global__ void k_run_probe(const hs_probe_args A) {
  const int lane = threadIdx.x;
  const bool on = lane < A.nrank && lane != A.rank;
  unsigned long long* mine = nullptr;
  unsigned long long* theirs = nullptr;
#pragma unroll
  for (int p = 0; p < HS_RUN_MAX_RANKS; ++p) {
    if (p == A.rank) mine = A.peer[p];
    if (p == lane) theirs = A.peer[p];
  }
  int bad = 0, r = 0, spins_total = 0, seen = 0;
  const long long t0 = wall_clock64();
  for (r = 1; r <= A.rounds && !bad; ++r) {
    const unsigned tag = (A.seq << 12) | (unsigned)r;
    if (on) hs_st_sys(theirs + A.rank, ((unsigned long long)tag << 32) | (unsigned)A.rank);
    bool ok = !on;
    int spins = 0;
    for (;;) {
      if (!ok) {
        const unsigned long long v = hs_ld_sys(mine + lane);
        // (">=": a peer that has seen everybody's round r moves on and overwrites its slot with
        // round r + 1, or with the next probe's first round, before a slower rank has looked)
        ok = (int)((unsigned)(v >> 32) - tag) >= 0 && (unsigned)v == (unsigned)lane;
        if (ok) seen = r;
      }
      if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
      if (++spins > A.spin_limit) { bad = HS_RUN_ERR_TIMEOUT; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    spins_total += spins;
  }
  if (lane == 0) {
    A.out[0] = bad;
    A.out[1] = bad ? r - 2 : r - 1;
    A.out[2] = (int)(wall_clock64() - t0);
    A.out[3] = spins_total;
  }
  if (lane < HS_RUN_MAX_RANKS) A.out[4 + lane] = on ? seen : -1;
}

extern "C" int nh_half_step_run_probe(nh_ctx* c, nh_halfstep_run* Q, int rounds, int* status,
                                      double* us_per_round) {
  NH_REQUIRE(c && Q && Q->base && status, "not a shared-ensemble loop");
  NH_REQUIRE(rounds >= 1 && rounds < 4096, "1 .. 4095 rounds");
  hs_probe_args A;
  memset(&A, 0, sizeof(A));
  for (int p = 0; p < Q->nrank; ++p) {
    NH_REQUIRE(Q->peer_base[p] != nullptr, "a peer's rings are not attached");
    A.peer[p] = Q->peer_base[p];
  }
  A.nrank = Q->nrank; A.rank = Q->rank; A.rounds = rounds;
  A.spin_limit = Q->R.spin_limit;
  A.seq = Q->probe_seq++ & 0xFFFFFu;  // (its own count: the launches' decides the ring's parity)
  A.out = Q->probe_out;
  hipLaunchKernelGGL(k_run_probe, dim3(1), dim3(64), 0, c->stream, A);
  NH_CHECK_HIP(hipGetLastError());
  int rc = nh_sync(c);
  if (rc) return rc;
  int o[4 + HS_RUN_MAX_RANKS];
  NH_CHECK_HIP(hipMemcpy(o, Q->probe_out, sizeof(o), hipMemcpyDeviceToHost));
  *status = o[0];
  if (us_per_round) *us_per_round = o[1] > 0 ? (double)o[2] * 0.01 / o[1] : 0.0;
  if (o[0] != 0) {  // (what the caller's warning says: which peers' granules stopped coming)
    char msg[160];
    int n = snprintf(msg, sizeof(msg), "probe of rank %d: %d of %d rounds; last round seen per peer:",
                     Q->rank, o[1], rounds);
    for (int p = 0; p < Q->nrank && n < (int)sizeof(msg) - 8; ++p)
      n += snprintf(msg + n, sizeof(msg) - n, " %d", o[4 + p]);
    nh_set_error(NH_EHIP, "%s", msg);
  }
  return NH_OK;
}

// The loop's own copies of the plan's tables with SORTED columns (nh_hs.h): kds[t] = device
// pointer to [nG][nK][2] doubles + the trailer of ints { row0[8] | perm[nK] }, or NULL to keep
// the plan's table t.  The caller owns the buffers and keeps them alive with the loop.
extern "C" int nh_half_step_run_tables(nh_ctx* c, nh_halfstep_plan* P, nh_halfstep_run* Q,
                                       const double* const* kds, int ntab) {
  NH_REQUIRE(c && P && Q && kds && ntab >= 0 && ntab <= HS_MAX_TAB, "bad argument");
  const hs_hot& H = P->hot;
  int rc = nh_sync(c);
  if (rc) return rc;
  // the trailers decide where the kernel writes in LDS: checked here, once
  for (int t = 0; t < ntab && t < H.ntab; ++t) {
    if (!kds[t]) continue;
    const hs_tab& tb = H.C.tab[t];
    const int nG = H.nG[tb.grid], nK = tb.nK;
    NH_REQUIRE(tb.tiles <= HS_TRAIL_TILES, "a sorted table has at most 8 column tiles");
    std::vector<int> tr((size_t)HS_TRAIL_TILES + nK);
    NH_CHECK_HIP(hipMemcpy(tr.data(), kds[t] + 2 * (size_t)nG * nK, tr.size() * sizeof(int),
                           hipMemcpyDeviceToHost));
    std::vector<char> seen((size_t)nK, 0);
    for (int q = 0; q < HS_TRAIL_TILES; ++q)
      NH_REQUIRE(tr[q] >= 0 && tr[q] <= nG, "sorted table: a tile's first row is out of range");
    for (int k = 0; k < nK; ++k) {
      const int p = tr[HS_TRAIL_TILES + k];
      NH_REQUIRE(p >= 0 && p < nK && !seen[p], "sorted table: the column order is not a permutation");
      seen[p] = 1;
    }
  }
  for (int t = 0; t < HS_MAX_TAB; ++t) Q->R.kds[t] = (t < ntab && t < H.ntab) ? kds[t] : nullptr;
  return NH_OK;
}

// where the next launches of a shared ensemble keep their history flags ([hist_cap][N] ints,
// every row of a launch written by its epilogue; NULL: nowhere)
extern "C" int nh_half_step_run_hist_flags(nh_halfstep_run* Q, int* hacc) {
  NH_REQUIRE(Q && Q->base, "not a shared-ensemble loop");
  Q->hacc = hacc;
  return NH_OK;
}

// the shared loop's own bookkeeping: nacc_own[N] (moves this rank accepted since the last reset)
// and curstamp[N] (-1, or the stamp of the last move this rank accepted for the walker) to the
// host; reset != 0 zeroes / clears them afterwards.  Synchronises the stream.
extern "C" int nh_half_step_run_counters(nh_ctx* c, nh_halfstep_run* Q, int* nacc_own, int* curstamp,
                                         int reset) {
  NH_REQUIRE(c && Q && Q->base, "not a shared-ensemble loop");
  int rc = nh_sync(c);
  if (rc) return rc;
  const size_t nb = (size_t)Q->R.N * sizeof(int);
  if (nacc_own) NH_CHECK_HIP(hipMemcpy(nacc_own, Q->nacc_own, nb, hipMemcpyDeviceToHost));
  if (curstamp) NH_CHECK_HIP(hipMemcpy(curstamp, Q->curstamp, nb, hipMemcpyDeviceToHost));
  if (reset & 1) NH_CHECK_HIP(nh_fill_now(c, Q->nacc_own, 0, nb));
  if (reset & 2) NH_CHECK_HIP(nh_fill_now(c, Q->curstamp, 0xFF, nb));
  return NH_OK;
}

extern "C" int nh_half_step_run(nh_ctx* c, nh_halfstep_plan* P, nh_halfstep_run* Q, int slice0,
                                int nslices, double* hist_coords, double* hist_logp,
                                double* const* hist_blobs /*host, nblob pointers, or NULL*/,
                                long long hist_row0, long long hist_cap) {
  NH_REQUIRE(c && P && Q, "bad argument");
  NH_REQUIRE(slice0 >= 0 && (slice0 & 1) == 0 && nslices >= 2 && (nslices & 1) == 0 &&
                 nslices <= 2 * HS_RUN_MAX_STEPS, "whole ensemble steps, at most one block of moves");
  const hs_hot& H = P->hot;
  NH_REQUIRE(hist_coords == nullptr || (hist_logp && hist_row0 >= 0 && hist_row0 + nslices / 2 <= hist_cap),
             "the chain history must hold every step of the launch");
  hs_run R = Q->R;
  R.slice0 = slice0;
  R.nslices = nslices;
  R.seq = Q->seq++ & 0xFFFFFFu;
  if (R.seq == 0) R.seq = Q->seq++ & 0xFFFFFFu;
  R.report_launch = ++Q->nlaunch;
  R.clk = c->clk;
  R.lcnt = Q->lcnt + 2 * (Q->nlaunch & 1);
  if (Q->fail_at > 0 && Q->nlaunch == Q->fail_at) R.spin_limit = 0;  // (tests: this launch's first wait gives up)
  if (Q->base) {  // a shared ensemble: this launch's ring, here and on every other rank
    const size_t off = HS_RUN_HEAD + (size_t)(R.seq & 1u) * Q->ring_elems;
    for (int p = 0; p < Q->nrank; ++p) {
      NH_REQUIRE(Q->peer_base[p] != nullptr, "a peer's rings are not attached");
      R.peer[p] = Q->peer_base[p] + off;
    }
    R.ring = R.peer[Q->rank];
    R.hacc = hist_coords ? Q->hacc : nullptr;
    R.stamp0 = (int)(Q->steps_total & 0x3FFFFFFF);
    Q->steps_total += nslices / 2;
  }
  R.hcoords = hist_coords;
  R.hlogp = hist_logp;
  R.hrow0 = hist_row0;
  R.hcap = hist_cap;
  for (int b = 0; b < NH_HS_MAX_BLOB; ++b)
    R.hblob[b] = (hist_coords && hist_blobs && b < H.C.nblob) ? hist_blobs[b] : nullptr;
  {
    nh_prof_scope ps(c, NH_K_HALFSTEP);
    const dim3 grid((unsigned)Q->grid, (unsigned)Q->split);
    const dim3 thr(Q->threads);
    void* args[2] = {(void*)&H, (void*)&R};
    NH_CHECK_HIP(hipLaunchKernel(hs_run_kernel(H.syn_grid >= 0, Q->base != nullptr, R.syn2 != 0, Q->rt != 0, Q->deep != 0), grid, thr,
                                 args, Q->lds_bytes, c->stream));
  }
  {
    nh_prof_scope ps(c, NH_K_GLUE);
    const long long work = (long long)R.N * (H.nE > H.ndim + 1 ? H.nE : H.ndim + 1);
    const int blocks = (int)((work + 255) / 256 < 512 ? (work + 255) / 256 : 512);
    const int jobs = 2 + ((hist_coords && !Q->base) ? H.C.nblob : 0);
    hs_epi E;
    memset(&E, 0, sizeof(E));
    E.N = R.N; E.ndim = H.ndim; E.gr = R.gr; E.nrank = R.nrank; E.nblob = H.C.nblob;
    E.spin_limit = R.spin_limit; E.report_launch = R.report_launch; E.seq = R.seq;
    E.status = R.status; E.done = H.done; E.lcnt = R.lcnt; E.report = R.report;
    E.clk = c->clk;
    E.ring = R.ring; E.coords = const_cast<double*>(H.coords); E.logp = const_cast<double*>(H.logp);
    E.nacc = R.nrank > 1 ? R.nacc_own : H.C.naccepted; E.accw = R.accw; E.hacc = R.hacc;
    E.hrow0 = R.hrow0;
    for (int b = 0; b < NH_HS_MAX_BLOB && b < H.C.nblob; ++b) {
      E.hblob[b] = (R.hcoords != nullptr && R.nrank <= 1) ? R.hblob[b] : nullptr;
      E.bcur[b] = H.C.blob[b].cur;
      E.bm[b] = H.C.blob[b].m;
    }
    hipLaunchKernelGGL(k_run_epilogue, dim3(blocks, jobs), dim3(256), 0, c->stream, E, nslices / 2);
    NH_CHECK_HIP(hipGetLastError());
  }
  return NH_OK;
}

// 0 while every launch so far found its records; else the code of the first failure (the
// ensemble is then undefined from that launch on).  Synchronises the stream.
extern "C" int nh_half_step_run_status(nh_ctx* c, nh_halfstep_run* Q, int* status) {
  NH_REQUIRE(c && Q && status, "bad argument");
  int rc = nh_sync(c);
  if (rc) return rc;
  NH_CHECK_HIP(hipMemcpy(status, Q->status, sizeof(int), hipMemcpyDeviceToHost));
  return NH_OK;
}

// What k_run_epilogue reported about launch number `launch` (1, 2, ...: nh_half_step_run counts
// them) -- without touching the stream: *done = 0 while the launch has not ended (wait != 0:
// sleeps until it has, at most ~60 s), else its status (0: every record came), the plan's
// NaN / forbidden-proposal counters as they stood behind it and (before[2]) as it found them.  Only the last two launches have a
// slot; an older one reads as done with status -1.
extern "C" int nh_half_step_run_report(nh_ctx* c, nh_halfstep_run* Q, int launch, int wait, int* done,
                                       int* status, int* nan_count, int* forbidden, int* before) {
  NH_REQUIRE(c && Q && done && status && launch >= 1 && launch <= Q->nlaunch, "bad argument");
  NH_REQUIRE(Q->report != nullptr, "this loop keeps no launch reports");
  volatile int* rp = Q->report + 8 * (launch & 1);
  *done = 0;
  // (a launch is a millisecond: the wait spins on the word for the first 20 ms -- a sleep's
  // granularity is tens of microseconds, 5 % of a 20-step region of cfg3 -- and sleeps from there)
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const int seen = rp[3];
    if (seen == launch) break;
    if (seen > launch) {  // (its slot belongs to a later launch by now)
      *done = 1;
      *status = -1;
      return NH_OK;
    }
    if (!wait) return NH_OK;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 60e3) return nh_set_error(NH_EHIP, "launch %d of the resident loop never reported", launch);
    if (ms > 20.0) usleep(100);
  }
  __sync_synchronize();
  *done = 1;
  *status = rp[0];
  if (nan_count) *nan_count = rp[1];
  if (forbidden) *forbidden = rp[2];
  if (before) { before[0] = rp[4]; before[1] = rp[5]; }
  return NH_OK;
}

extern "C" int nh_half_step_run_info(const nh_halfstep_run* Q, int* grid, int* threads,
                                     long long* lds_bytes) {
  NH_REQUIRE(Q, "bad argument");
  if (grid) *grid = Q->grid;
  if (threads) *threads = Q->threads;
  if (lds_bytes) *lds_bytes = (long long)Q->lds_bytes;
  return NH_OK;
}

extern "C" int nh_half_step_run_syn_info(const nh_halfstep_run* Q, int* mode, int* nodes_per_piece,
                                         int* pieces) {
  NH_REQUIRE(Q, "bad argument");
  if (mode) *mode = Q->R.syn2;
  if (nodes_per_piece) *nodes_per_piece = Q->R.syn2 ? 1 << Q->R.s2.lm : 0;
  if (pieces) *pieces = Q->R.syn2 ? Q->R.s2.P + 1 : 0;
  return NH_OK;
}

// workgroups per walker of the loop's launches, and whether they split the grid's ROWS between
// them (a table-only model: each forms its part of the weights and reduces its part of the table)
// rather than taking every K-th work item of a walker whose weights every one of them forms
extern "C" int nh_half_step_run_pipeline_info(const nh_halfstep_run* Q, int* deep) {
  NH_REQUIRE(Q && deep, "bad argument");
  *deep = Q->deep;
  return NH_OK;
}

extern "C" int nh_half_step_run_split_info(const nh_halfstep_run* Q, int* split, int* rows) {
  NH_REQUIRE(Q, "bad argument");
  if (split) *split = Q->split;
  if (rows) *rows = Q->R.rowsplit;
  return NH_OK;
}

extern "C" int nh_half_step_run_table_info(const nh_halfstep_run* Q, int* in_registers, int* nodes_max) {
  NH_REQUIRE(Q, "bad argument");
  if (in_registers) *in_registers = Q->rt ? 1 : 0;
  if (nodes_max) *nodes_max = Q->rt ? HS_RUN_RT : 0;
  return NH_OK;
}

// NH_HS_DEBUG=1: out[256][64][8] wall-clock stamps (100 MHz) of the last launch's workgroups,
// per (workgroup, iteration): 0 start of the slice, 1 records in, 2 first barrier (packs
// done), 3 weights done, 4 this thread's items done, 5 all items done, 6 spectra summed,
// 7 record published
extern "C" int nh_half_step_run_stamps(nh_ctx* c, const nh_halfstep_run* Q, long long* out) {
  NH_REQUIRE(c && Q && out, "bad argument");
  memset(out, 0, (256 * 64 * 8 + 64 * 4 * 16) * sizeof(long long));
  if (!Q->dbg) return NH_OK;
  int rc = nh_sync(c);
  if (rc) return rc;
  NH_CHECK_HIP(hipMemcpy(out, Q->dbg, (256 * 64 * 8 + 64 * 4 * 16) * sizeof(long long), hipMemcpyDeviceToHost));
  return NH_OK;
}

extern "C" int nh_half_step_run_destroy(nh_ctx* c, nh_halfstep_run* Q) {
  NH_REQUIRE(c, "ctx is NULL");
  if (!Q) return NH_OK;
  int rc = nh_sync(c);
  for (int p = 0; p < HS_RUN_MAX_RANKS; ++p)
    if (Q->base && p != Q->rank && Q->peer_base[p]) (void)hipIpcCloseMemHandle(Q->peer_base[p]);
  if (Q->base) (void)hipFree(Q->base);
  else if (Q->ring) (void)hipFree(Q->ring);
  if (Q->nacc_own) (void)hipFree(Q->nacc_own);
  if (Q->curstamp) (void)hipFree(Q->curstamp);
  if (Q->probe_out) (void)hipFree(Q->probe_out);
  if (Q->status) (void)hipFree(Q->status);
  if (Q->lcnt) (void)hipFree(Q->lcnt);
  if (Q->accw) (void)hipFree(Q->accw);
  if (Q->dbg) (void)hipFree(Q->dbg);
  if (Q->xspec) (void)hipFree(Q->xspec);
  if (Q->tick) (void)hipFree(Q->tick);
  if (Q->s2_dev) (void)hipFree(Q->s2_dev);
  if (Q->report) (void)hipHostFree(Q->report);
  delete Q;
  return rc;
}
