// nh_hs.h -- what the two half-step kernels share: the plan's descriptor blocks, the LDS layout
// and the work items (table reductions, synchrotron nodes) one workgroup evaluates for its
// walker.  k_half_step (nh_halfstep.hip) is ONE launch per half-step; k_half_step_run
// (nh_persist.hip) keeps its workgroups resident over a whole block of moves.
#pragma once
#include "nh_front.h"
#include "nh_lnprob.h"
#include "nh_syn.h"

#define HS_MAX_TAB 4
#ifndef HS_SYN_NODES
#define HS_SYN_NODES 10  // synchrotron nodes per thread and work item (besides the start node)
#endif
#ifndef HS_ORDER
#define HS_ORDER 0  // work items: 0 = tables and synchrotron alternate, 1 = tables first, 2 = synchrotron first
#endif

struct hs_tab {
  const double* KD; const double* scale; double* out;  // KD: interleaved {K, dlnK}, [nG][nK][2]
  int grid, nK, ldo, nonneg, spec_off, tiles, item0, chunks;
  int sub, nKp;  // nK <= 32: sub = 64 / nKp sub-ranges of an item share a wave (nKp = 32, 16, ...)
  // `chunks` packs three numbers: chunks | nfull << 8 | seg2 << 16.  Chunks [0, nfull) hold `seg`
  // segments each, the rest seg2 (<= seg): the items pulled last decide how far apart the waves
  // reach the barrier, so they are the short ones.  (Packed into the existing word: a wider
  // struct made the compiler copy the whole by-value descriptor to scratch in k_half_step.)
};
#define HS_CHUNKS(pk) ((pk) & 0xff)
// A table copy with SORTED columns (the resident loop's own, nh_half_step_run_tables): its
// columns are ordered by the first row in which they are non-zero, and the [nG][nK][2] doubles
// are followed by a trailer of ints { row0[HS_TRAIL_TILES] | perm[nK] }: row0[tile] = the first
// row in which any column of the tile is non-zero -- segments below it contribute exact zeros
// (utils.py:347-348) and are not walked -- and perm[p] = the column of the spectrum that
// position p of the table holds.  (An inverse-Compton table is zero below gamma = E / mec2: for
// cfg3's TeV energies that is half the grid; sorted, the 64 highest-energy columns of its three
// seed fields share a tile.  k_half_step keeps the plain table: a read through one more pointer
// of its by-value descriptor has the compiler copy all 3 KB of it to scratch.)
#define HS_TRAIL_TILES 8
__device__ __forceinline__ const int* hs_tab_trailer(const double* KD, int nG, int nK) {
  return reinterpret_cast<const int*>(KD + 2 * (long long)nG * nK);
}

// segments [s0, s1) of chunk `chunk` of a table over a grid of nseg segments
__device__ __forceinline__ void hs_chunk_range(int packed_chunks, int chunk, int seg, int nseg,
                                               int& s0, int& s1) {
  const int nf = (packed_chunks >> 8) & 0xff, s2 = packed_chunks >> 16;  // (wave-uniform)
  const int over = max(chunk - nf, 0);
  s0 = (chunk - over) * seg + over * s2;
  s1 = min(nseg, s0 + (over > 0 ? s2 : seg));
}

struct hs_comp { const double* ptr; long long ld; double scale; int off; int pad; };

// The descriptor.  All of it travels BY VALUE with the launch (hs_hot below), except the
// parameter packs (1.6 KB), which stay in device memory: the threads that evaluate a pack
// column fetch it in the same round trip as the slice of the move block.  (A first version
// kept this block in device memory too: its first use -- the table descriptors at the head
// of the work items, the prior terms -- then cost every wave a cold ~1.5 us round trip.)
struct hs_dev {
  const nh_pack* pk;  // [NH_MAX_PACK], device
  int npk, kind;
  const double* params;
  double* w[NH_MAX_GRIDS]; double* dlw[NH_MAX_GRIDS];
  double* mom_out[NH_MAX_MOMENT];
  int* accepted; int* naccepted; int* sel;
  int do_accept, write_weights;
  hs_tab tab[HS_MAX_TAB];
  int ntab, nT, seg;  // table items in total; segments per item
  const double* synB; double* syn_out;
  // syn_ldo packs two numbers: ldo | n1 << 20.  n1 > 0: the energies [n1, nE) go to a second array
  // [nloc][nE - n1] that FOLLOWS the first (nh_hs_syn.n1 / out2: two Synchrotron.flux calls of one
  // model evaluation as one component).  (Packed: reading one more field of this block in
  // k_half_step makes the compiler copy all 3 KB of it to scratch -- scripts/hs_usage.sh.)
  int syn_ldo, syn_bcol, syn_ldB, syn_cdmax;
  hs_comp comp[NH_MAX_COMP];
  int ncomp, pad0;
  const double* lp;
  nh_prior_pack pri;
  double* model_out; double* total;
  nh_hs_blob blob[NH_HS_MAX_BLOB];
  int nblob, o_mrow;  // o_mrow: LDS offset of the model spectrum row + the moments' results
  int sendw, pad1;    // > 0: total[] holds rows { lnprob | blobs } of this width (sharded loop)
  long long* dbg;  // NH_HS_DEBUG=1: wall-clock stamps of the first 8 workgroups, [8][16]
  // gridDim.y = K > 1 workgroups share a walker (launches of fewer walkers than the chip has
  // CUs): their partial spectra meet in xspec[walker][K][nspec], the last to arrive (tick) sums
  // them in a fixed order and carries on to the likelihood
  double* xspec; int* tick;
  int nspec, syn_nodes;  // syn_nodes: synchrotron nodes per thread and work item
};

// ... and its HOT part, passed by value: every pointer and size the first phases touch, so
// that the kernel's first round trip to memory already fetches data, not descriptors.
// The kernel's first scalar round trip: every pointer and size its second (vector) trip
// needs, contiguous at the head of the argument block so that a few wide scalar loads and ONE
// wait fetch it.  pkd: byte t = the proposal coordinate pack thread t reads (0xFF: a constant).
struct hs_first {
  const nh_pack* pk; const int* hbase; const double* logp; const nh_hist* hist;
  int npk8, ngrids, nloc, ppk;  // ppk: the pack whose output is the particle rows
  int nG[NH_MAX_GRIDS];
  const double* e[NH_MAX_GRIDS]; const double* xg[NH_MAX_GRIDS];
  const double* lne[NH_MAX_GRIDS]; const double* lx[NH_MAX_GRIDS];
  unsigned pkd[8];
  double* qT; double* factors;
  const double* syn_c;  // [3][syn_nG]: 1/gamma^2 | its cube root | 1/g2^2 - 1/g1^2 (or NULL)
  int syn_nG;
  int broken;  // the particle distribution has a break energy (the only use of the grids' E)
};

struct hs_hot {
  hs_first F;
  hs_dev C;
  const double* coords; const double* logp; const double* blk;
  int* done; const int* hbase; int* cursor;
  double* qT; double* factors; const nh_hist* hist;
  int ns, ndim, lo, nloc;
  const double* e[NH_MAX_GRIDS]; const double* xg[NH_MAX_GRIDS];
  const double* lne[NH_MAX_GRIDS]; const double* lx[NH_MAX_GRIDS];
  double scale[NH_MAX_GRIDS];
  int nG[NH_MAX_GRIDS];
  int o_w[NH_MAX_GRIDS], o_d[NH_MAX_GRIDS], o_lx[NH_MAX_GRIDS];
  // grids that a non-negative table is reduced over: dlw / lx and 2^-10 / lx per node (-1: none)
  int o_dp[NH_MAX_GRIDS], o_th[NH_MAX_GRIDS];
  int ngrids, nmom;
  const double* mKt[NH_MAX_MOMENT]; const double* mdK[NH_MAX_MOMENT];
  int mgrid[NH_MAX_MOMENT];
  int o_mkt, o_part_t, o_spec, o_lik;
  int syn_grid, syn_nE, syn_spec_off, o_ig2, o_dig2, o_ig23, o_sq, o_amap, o_part_s;
  int o_s2;  // the log-domain synchrotron items' block in LDS (k_half_step<true, true>; nh_syn2.h), or 0
  const double* syn_E;
  const double* conv; const double* flux; const double* elo; const double* ehi;
  const int* ul; const double* cl;
  int nE, ntab;
  const double* tscale[HS_MAX_TAB];  // per-column factors of the table reductions (or NULL)
  int tnK[HS_MAX_TAB], tspec[HS_MAX_TAB];
  int o_scale, o_synE;  // o_synE: the synchrotron component's photon energies in LDS
  int o_pri, pad4;      // the prior terms, copied out of the kernel-argument segment
};

struct nh_halfstep_plan {
  hs_hot hot;
  nh_pack* dev;      // device copy of the parameter packs
  int* words;        // device: done counter | hbase
  double* syn_c;     // device: the synchrotron grid's constants (k_syn_consts), or NULL
  double* xspec;     // device: the partial spectra of a split launch, or NULL
  int* tick;         // device: arrival counters of a split launch, or NULL
  size_t lds_bytes;
  size_t lds_core;  // ... without k_half_step's own log-domain synchrotron block (the last thing in it)
  int threads, blocks, split;  // split = K workgroups per walker (gridDim.y)
  int rt;  // the workgroup size was chosen for register-resident table items (hs_rt_item)
  int rowsplit;  // split == 2 was chosen for a table-only model whose two workgroups halve the ROWS
  long long* dbg;
  int span;  // the span clock (nh_common.h): bit 0 a launch of this plan opens a span, bit 1 it closes one (nh_half_step_span)
};

// ints at the head of the LDS block (after qs/row/lg/acc)
enum { HI_ME = 0, HI_PA, HI_READY, HI_TICK, HI_DEAD, HI_CNT, HI_LIVE, HI_NZ };
#define HS_O_ROW 64
#define HS_O_LG 72
#define HS_O_ACC 76   // z, lnU, old logp, (pad)
#define HS_O_INT 80   // 16 ints
#define HS_O_T64 88   // 2^(j/64), j < 64: the table of nh_exp_tab
#define HS_O_FREE 152
#define HS_STAMP(k)                                                                    \
  do {                                                                                  \
    if (dbg_on && tid == 0 && j < 8) D.dbg[j * 16 + (k)] = (long long)wall_clock64();   \
    if (dbg_on && tid == 0 && j < 1024) D.dbg[2304 + j * 16 + (k)] = (long long)wall_clock64(); \
  } while (0)

// ln |a 10^(b x + c)| of a parameter-pack column that is a power of ten of a proposed coordinate
// (naima's fits walk in log10 of amplitudes and energies): ln|a| + y ln 10 with ln 10 in two
// pieces -- right to the last place of a number around 70, as the logarithm of the rounded power
// is, and not a library call BEHIND the library's exp10 on the one lane every wave waits for.
// Both half-step kernels take it, so that their weights are the same numbers.
__device__ __forceinline__ double hs_ln_pow10(double lna, double zb, double zc, double x) {
  const double y = fma(zb, x, zc);
  return lna + fma(y, 2.302585092994046, y * -2.1707562233822494e-16);
}

// Wave-wide sums over DPP (row shifts inside the 16-lane rows, then the two row broadcasts): the
// total lands in lane 63 -- a quarter of the latency of six ds_bpermute round trips, which on the
// one wave that finishes a slice alone is time every other wave waits.  (The order of the
// additions is fixed, like the shuffle tree's; it is a different order.)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int hs_dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double hs_dpp_f64(double v) {
  const int lo = hs_dpp_i32<CTRL, ROW_MASK, BANK_MASK>(__double2loint(v));
  const int hi = hs_dpp_i32<CTRL, ROW_MASK, BANK_MASK>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// (lanes a shift does not reach read 0: bound_ctrl off and old = 0 -- adding 0.0 / 0)
// the sum of an int over the wave, in lane 63 (row shifts, two row broadcasts: no LDS round trip)
__device__ __forceinline__ int hs_wave_sum_i32_dpp(int c) {
  c += hs_dpp_i32<0x111, 0xf, 0xf>(c);   // row_shr:1
  c += hs_dpp_i32<0x112, 0xf, 0xf>(c);   // row_shr:2
  c += hs_dpp_i32<0x114, 0xf, 0xe>(c);   // row_shr:4
  c += hs_dpp_i32<0x118, 0xf, 0xc>(c);   // row_shr:8
  c += hs_dpp_i32<0x142, 0xa, 0xf>(c);   // row_bcast:15
  c += hs_dpp_i32<0x143, 0xc, 0xf>(c);   // row_bcast:31
  return c;
}
__device__ __forceinline__ void hs_wave_sum_dpp(double& a, int& c) {
  a += hs_dpp_f64<0x111, 0xf, 0xf>(a);  c += hs_dpp_i32<0x111, 0xf, 0xf>(c);   // row_shr:1
  a += hs_dpp_f64<0x112, 0xf, 0xf>(a);  c += hs_dpp_i32<0x112, 0xf, 0xf>(c);   // row_shr:2
  a += hs_dpp_f64<0x114, 0xf, 0xe>(a);  c += hs_dpp_i32<0x114, 0xf, 0xe>(c);   // row_shr:4
  a += hs_dpp_f64<0x118, 0xf, 0xc>(a);  c += hs_dpp_i32<0x118, 0xf, 0xc>(c);   // row_shr:8
  a += hs_dpp_f64<0x142, 0xa, 0xf>(a);  c += hs_dpp_i32<0x142, 0xa, 0xf>(c);   // row_bcast:15
  a += hs_dpp_f64<0x143, 0xc, 0xf>(a);  c += hs_dpp_i32<0x143, 0xc, 0xf>(c);   // row_bcast:31
}

__device__ __forceinline__ double hs_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

typedef unsigned int hs_u32x4 __attribute__((ext_vector_type(4)));
// one 16-byte element {K[i][k], dlnK[i][k]} of the interleaved table
// (byte_off: the lane's; row_off: wave-uniform -- it rides in the instruction's scalar offset, where
// the rows of a trip, a multiple of the table's row apart, used to cost a v_add_u32 each: one
// vector instruction of a segment's eleven)
__device__ __forceinline__ void hs_buf_kd(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double& K,
                                          double& d, unsigned row_off = 0u) {
  typedef double hs_f64x2 __attribute__((ext_vector_type(2)));
  const hs_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, row_off, 0);
  const hs_f64x2 kd = __builtin_bit_cast(hs_f64x2, v);  // (register pairs as loaded: no moves)
  K = kd.x;
  d = kd.y;
}

// one table work item: columns [64 tile, 64 tile + 64) x segments [s0, s1) of table t for
// this workgroup's walker, whose w / dlw / lx live in LDS (wave-uniform reads).  The table
// is the interleaved copy KD[i][k] = {K, dlnK}: ONE 16-byte load per lane and node instead of
// two 8-byte ones (8-byte accesses reach 0.54-0.70 of the L2 rate of 16-byte ones; with one
// walker per workgroup the rows stream from L2 once per walker).  Eight nodes per trip = eight kilobytes in flight per
// wave (a double-buffered four-node version, half of that in flight, measured 25 % slower).
typedef __attribute__((address_space(3))) const double hs_lds_cd;
// LDS byte address of a pointer into the workgroup's shared block, parked in a VECTOR register:
// the walker's w / dlw / lx reads of a table item are wave-uniform, the compiler keeps such an
// address in SGPRs and re-materialises it with a v_mov before EVERY ds_read (3 of the 15 VALU
// instructions of a segment); from a VGPR base the reads of a trip are immediate offsets.
__device__ __forceinline__ unsigned hs_lds_addr(const double* p) {
  unsigned a = (unsigned)(unsigned long long)(hs_lds_cd*)p;
  asm volatile("" : "+v"(a));
  return a;
}
__device__ __forceinline__ double hs_lds_at(unsigned base, int idx) {
  return *(hs_lds_cd*)(unsigned long long)(base + 8u * (unsigned)idx);
}

// One segment of a NON-NEGATIVE table on pre-divided log-ratios: the table carries
// dlnK / lx (nh_table_interleave with lx), the walker dlw / lx, so that
//   (u2 - u1) lx / dl = (u2 - u1) / dl',   dl' = dl / lx
// -- no multiplication by lx, and the term joins the sum in the reciprocal's last FMA: 10
// instructions per segment against 12.  |dl| < 2^-10 <=> |dl'| < th = 2^-10 / lx (per segment, in
// LDS where lx was); the series (rare, wave-uniform branch) takes lx = 2^-10 / th.
__device__ __forceinline__ double hs_seg_pre(double acc, double u1, double u2, double dlp,
                                             double th) {
  double a2 = fma(u2 - u1, nh_rcp1f(dlp), acc);
  const bool small = fabs(dlp) < th;
  if (__builtin_amdgcn_ballot_w64(small) != 0) {
    asm volatile("" ::: "memory");  // keep this a branch: the compiler would if-convert it
    const double lx = NH_SEG_SMALL_POS * nh_rcp(th);
    const double d = dlp * lx;
    double f = fma(d, 8.333333333333333e-03, 4.166666666666666e-02);
    f = fma(f, d, 1.666666666666667e-01);
    f = fma(f, d, 0.5);
    f = fma(f, d, 1.0);
    a2 = small ? fma(u1 * lx, f, acc) : a2;
  }
  return a2;
}

// (the loop itself: table address (wave-uniform halves) and width, node count, column tile, row
// range, and the LDS byte addresses of w | dlw (/ lx) | lx (2^-10 / lx) AT row s0)
template <bool SIGNED>
__device__ __forceinline__ double hs_table_item_v(unsigned kd_lo, unsigned kd_hi, unsigned nK, int nG,
                                                  int tile, int s0, int s1, unsigned aw, unsigned ad,
                                                  unsigned al, int lane) {
  const void* KDu = (const void*)(((unsigned long long)kd_hi << 32) | kd_lo);
  const int k = tile * 64 + lane;
  const unsigned kk = (unsigned)k < nK ? (unsigned)k : nK - 1u;
  const unsigned tbytes = (unsigned)nG * nK * 16u;
  const __amdgpu_buffer_rsrc_t rKD =
      __builtin_amdgcn_make_buffer_rsrc((void*)KDu, 0, (int)tbytes, 0x00020000);
  const unsigned rowb = nK * 16u;
  unsigned ob = ((unsigned)s0 * nK + kk) * 16u;
  double acc = 0.0;
  double K1, d1;  // node s: its K and the log-ratio of the segment that starts there
  hs_buf_kd(rKD, ob, K1, d1);
  double u1 = hs_lds_at(aw, 0) * K1;
  int s = s0;
  for (; s + 8 <= s1; s += 8) {
    double K2[8], dK[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) hs_buf_kd(rKD, ob, K2[q], dK[q], (q + 1) * rowb);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double u2 = hs_lds_at(aw, q + 1) * K2[q];
      const double dl = hs_lds_at(ad, q) + d1;
      if (SIGNED) acc += nh_seg_signed(u1, u2, dl, hs_lds_at(al, q));
      else acc = hs_seg_pre(acc, u1, u2, dl, hs_lds_at(al, q));
      u1 = u2;
      d1 = dK[q];
    }
    ob += 8 * rowb;
    aw += 64u;
    ad += 64u;
    al += 64u;
  }
  if (s < s1) {  // tail: the remaining (< 8) nodes in one trip; rows past the table read 0
    double K2[7], dK[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) hs_buf_kd(rKD, ob, K2[q], dK[q], (q + 1) * rowb);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      if (s + q < s1) {
        const double u2 = hs_lds_at(aw, q + 1) * K2[q];
        const double dl = hs_lds_at(ad, q) + d1;
        if (SIGNED) acc += nh_seg_signed(u1, u2, dl, hs_lds_at(al, q));
        else acc = hs_seg_pre(acc, u1, u2, dl, hs_lds_at(al, q));
        u1 = u2;
        d1 = dK[q];
      }
    }
  }
  return acc;
}

template <bool SIGNED>
__device__ __forceinline__ double hs_table_item(const hs_tab& t, int nG, int tile, int s0, int s1,
                                                const double* ws, const double* ds,
                                                const double* lxs, int lane,
                                                const double* KD = nullptr) {
  // the table's address and width come out of the descriptor: the compiler cannot
  // prove them wave-uniform and would wrap EVERY load in a waterfall loop (four
  // v_readfirstlane + two v_cmp + exec juggling per load, 7 VALU instructions per segment of
  // the 25 the loop then costs) -- say so once per work item instead
  const unsigned nK = (unsigned)__builtin_amdgcn_readfirstlane(t.nK);
  const unsigned long long kd = (unsigned long long)(KD ? KD : t.KD);
  const unsigned kd_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kd);
  const unsigned kd_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kd >> 32));
  return hs_table_item_v<SIGNED>(kd_lo, kd_hi, nK, nG, tile, s0, s1, hs_lds_addr(ws + s0),
                                 hs_lds_addr(ds + s0), hs_lds_addr(lxs + s0), lane);
}

// The same for a table of at most 32 columns: a wave of 64 lanes would leave half of them (or
// more) idle, and the loop is bound by instruction issue -- so `sub` = 64 / nKp sub-ranges of
// the item's segments share the wave (lane = h nKp + k walks sub-range h of column k).  The
// walker's w / dlw / lx reads are then per lane (LDS, `sub` distinct addresses per wave).
// PK nodes per trip: 4 in the general instance of the kernel (123 VGPRs), 6 where there is no
// synchrotron component (99 VGPRs without them): a narrow table's items are bound by the round
// trips of their loads, half as many rows again in flight per trip (8 spills)
// (aw0 / ad0 / al0: LDS byte addresses of the arrays' row 0)
template <bool SIGNED, int PK>
__device__ __forceinline__ double hs_table_item_packed_v(unsigned kd_lo, unsigned kd_hi, unsigned nK,
                                                         int nKp, int sub, int nG, int s0, int s1,
                                                         unsigned aw0, unsigned ad0, unsigned al0,
                                                         int lane) {
  const void* KDu = (const void*)(((unsigned long long)kd_hi << 32) | kd_lo);
  const int k = lane & (nKp - 1), h = lane / nKp;
  const unsigned kk = (unsigned)k < nK ? (unsigned)k : nK - 1u;
  const unsigned tbytes = (unsigned)nG * nK * 16u;
  const __amdgpu_buffer_rsrc_t rKD =
      __builtin_amdgcn_make_buffer_rsrc((void*)KDu, 0, (int)tbytes, 0x00020000);
  const unsigned rowb = nK * 16u;
  const int len = (s1 - s0 + sub - 1) / sub;     // segments per sub-range (wave-uniform)
  const int sl = s0 + h * len;                   // this lane's first segment
  const int se = min(s1, sl + len);              // ... and the end of its sub-range
  unsigned ob = ((unsigned)sl * nK + kk) * 16u;  // (rows past the table read 0)
  // per-lane LDS bases: the reads of a trip are immediate offsets from them.  Lanes whose
  // sub-range is shorter than `len` read on past its end (another array of the block, or
  // nothing: LDS reads out of range return 0) and discard the term.
  unsigned aw = aw0 + 8u * (unsigned)sl, ad = ad0 + 8u * (unsigned)sl, al = al0 + 8u * (unsigned)sl;
  double acc = 0.0;
  double K1, d1;
  hs_buf_kd(rKD, ob, K1, d1);
  // PK nodes per trip, the NEXT trip's loads issued before the current one is consumed: a
  // narrow table means few work items -- one per wave, all waves in step -- so nothing else
  // hides the round trip.  Two trips per loop iteration: the two register sets swap roles by
  // name instead of being copied (eight 64-bit moves per trip otherwise).
  double KA[PK], dA[PK], KB[PK], dB[PK];
#pragma unroll
  for (int q = 0; q < PK; ++q) hs_buf_kd(rKD, ob, KA[q], dA[q], (q + 1) * rowb);
  double u1 = hs_lds_at(aw, 0) * K1;
  const int owed = se - sl;  // segments of this lane's sub-range (<= 0: idle from the start)
  int done = 0;              // (wave-uniform: lives in an SGPR)
  auto trip = [&](const double (&K2)[PK], const double (&dK)[PK]) {
#pragma unroll
    for (int q = 0; q < PK; ++q) {
      const double u2 = hs_lds_at(aw, q + 1) * K2[q];
      const double dl = hs_lds_at(ad, q) + d1;
      const double term = SIGNED ? nh_seg_signed(u1, u2, dl, hs_lds_at(al, q))
                                 : hs_seg_pre(0.0, u1, u2, dl, hs_lds_at(al, q));
      acc += done + q < owed ? term : 0.0;
      u1 = u2;
      d1 = dK[q];
    }
    done += PK;
    aw += 8u * PK;
    ad += 8u * PK;
    al += 8u * PK;
  };
  for (int q0 = 0; q0 < len; q0 += 2 * PK) {
    ob += PK * rowb;
    if (q0 + PK < len) {
#pragma unroll
      for (int q = 0; q < PK; ++q) hs_buf_kd(rKD, ob, KB[q], dB[q], (q + 1) * rowb);
    }
    trip(KA, dA);
    if (q0 + PK >= len) break;
    ob += PK * rowb;
    if (q0 + 2 * PK < len) {
#pragma unroll
      for (int q = 0; q < PK; ++q) hs_buf_kd(rKD, ob, KA[q], dA[q], (q + 1) * rowb);
    }
    trip(KB, dB);
  }
  // the sub-ranges of a column meet in its first lane group (fixed order: deterministic)
  for (int off = 32; off >= nKp; off >>= 1) acc += __shfl_down(acc, off, 64);
  return acc;
}

template <bool SIGNED, int PK>
__device__ __forceinline__ double hs_table_item_packed(const hs_tab& t, int nG, int s0, int s1,
                                                       const double* ws, const double* ds,
                                                       const double* lxs, int lane,
                                                       const double* KD = nullptr) {
  const unsigned nK = (unsigned)__builtin_amdgcn_readfirstlane(t.nK);
  const int nKp = __builtin_amdgcn_readfirstlane(t.nKp);
  const int sub = __builtin_amdgcn_readfirstlane(t.sub);
  const unsigned long long kd = (unsigned long long)(KD ? KD : t.KD);
  const unsigned kd_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kd);
  const unsigned kd_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kd >> 32));
  return hs_table_item_packed_v<SIGNED, PK>(kd_lo, kd_hi, nK, nKp, sub, nG, s0, s1, hs_lds_addr(ws),
                                            hs_lds_addr(ds), hs_lds_addr(lxs), lane);
}


// ---------------------------------------------------------------------------------------------
// A table item whose rows stay in REGISTERS for the whole launch (the resident loop of a
// table-only model, k_half_step_run<false, ., false, RT>): an emission table does not depend on
// the walker and a resident workgroup's waves take the same items slice after slice, so the
// round trip to the L2s at the head of every item -- for bytes that were the same bytes the
// slice before -- is paid once per launch (cfg1: 9.6 -> 9.0 us per half-step; what is left of an
// item is its arithmetic).  Workgroups of 512 threads own 256 vector registers per lane; RT
// nodes of {K, dlnK} take 4 RT of them.  The arithmetic is hs_table_item_packed's /
// hs_table_item's, operand for operand.
#define HS_RT_NODES 28  // nodes per lane a register-resident item may hold (112 registers)
template <int RT>
struct hs_rt_item {
  double K[RT], d[RT];   // node q of this lane's sub-range: K and the log-ratio of the segment that starts there
  int ix;                // the item (index into the partial sums), or -1: this wave has none / not cached
  int t, tile, len, owed, nKp, sub, pre, slot;
  unsigned aw, ad, al;   // LDS byte addresses of w | dlw / lx | 2^-10 / lx at the lane's first segment
};

template <int RT>
__device__ __forceinline__ bool hs_rt_load(hs_rt_item<RT>& c, const hs_tab& t, const double* KD,
                                           int nG, int tile, int s0, int s1, int lane,
                                           const double* ws, const double* ds, const double* lxs) {
  const unsigned nK = (unsigned)__builtin_amdgcn_readfirstlane(t.nK);
  const int sub = __builtin_amdgcn_readfirstlane(t.sub);
  const int nKp = sub > 1 ? __builtin_amdgcn_readfirstlane(t.nKp) : 64;
  const int len = (s1 - s0 + sub - 1) / sub;  // segments per lane (wave-uniform)
  if (len + 1 > RT || len < 1) return false;
  const int k = sub > 1 ? (lane & (nKp - 1)) : tile * 64 + lane, h = sub > 1 ? lane / nKp : 0;
  const unsigned kk = (unsigned)k < nK ? (unsigned)k : nK - 1u;
  const unsigned long long kd = (unsigned long long)KD;
  const unsigned kd_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kd);
  const unsigned kd_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kd >> 32));
  const void* KDu = (const void*)(((unsigned long long)kd_hi << 32) | kd_lo);
  const __amdgpu_buffer_rsrc_t rKD =
      __builtin_amdgcn_make_buffer_rsrc((void*)KDu, 0, (int)((unsigned)nG * nK * 16u), 0x00020000);
  const unsigned rowb = nK * 16u;
  const int sl = s0 + h * len, se = min(s1, sl + len);
  const unsigned ob = ((unsigned)sl * nK + kk) * 16u;  // (rows past the table read 0)
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    c.K[q] = 0.0;
    c.d[q] = 0.0;
    if (q <= len) hs_buf_kd(rKD, ob + (unsigned)q * rowb, c.K[q], c.d[q]);
  }
  c.len = len;
  c.owed = se - sl;
  c.nKp = nKp;
  c.sub = sub;
  c.aw = hs_lds_addr(ws + sl);
  c.ad = hs_lds_addr(ds + sl);
  c.al = hs_lds_addr(lxs + sl);
  return true;
}

// SIGNED = false: a non-negative table on pre-divided log-ratios (ds = dlw / lx, lxs = 2^-10 / lx);
// true: any table (ds = dlw, lxs = lx; a sign change is a NaN log-ratio: nh_seg_signed)
// A segment's term WITHOUT its rare branches (the series for |dl| < 2^-10, which the caller takes
// for a whole group of segments at once): SIGNED as nh_seg_signed (ds = dlw, lx in `l`), else as
// hs_seg_pre (pre-divided log-ratios, the threshold 2^-10 / lx in `l`)
template <bool SIGNED>
__device__ __forceinline__ double hs_seg_fast(double u1, double u2, double dl, double l, bool& small) {
  if (SIGNED) {
    const double t = ((u2 - u1) * l) * nh_rcp1f(dl);
    small = fabs(dl) < NH_SEG_SMALL_POS;  // (false for NaN)
    return (dl == dl) ? t : u1 * l;
  }
  small = fabs(dl) < l;
  return (u2 - u1) * nh_rcp1f(dl);
}
template <bool SIGNED>
__device__ __forceinline__ double hs_seg_exact(double u1, double u2, double dl, double l) {
  return SIGNED ? nh_seg_signed(u1, u2, dl, l) : hs_seg_pre(0.0, u1, u2, dl, l);
}

// Segments in GROUPS of four: a segment's rare branch (behind a ballot) ends the scheduler's view,
// so with one per segment the four dependent chains of a group -- log-ratio, reciprocal seed,
// Newton step, term: ~60 cycles each -- ran one after the other; with two waves per SIMD (the
// 512-thread instances that keep their rows in registers) nobody else filled the gaps.  One ballot
// per group: the chains interleave.
template <int RT, bool SIGNED>
__device__ __forceinline__ double hs_rt_compute(const hs_rt_item<RT>& c) {
  double acc = 0.0;
  double u1 = hs_lds_at(c.aw, 0) * c.K[0], d1 = c.d[0];
  const int len = __builtin_amdgcn_readfirstlane(c.len);  // (scalar branches below, no exec masks)
  const int sub = __builtin_amdgcn_readfirstlane(c.sub);
  const bool packed = sub > 1;  // hs_table_item_packed: lanes past their sub-range masked
#pragma unroll
  for (int q0 = 0; q0 < RT - 1; q0 += 4) {
    if (q0 + 4 <= len && q0 + 4 <= RT - 1) {
      double uu[5], dd[4], ll[4], t[4];
      bool sm[4];
      uu[0] = u1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uu[g + 1] = hs_lds_at(c.aw, q0 + g + 1) * c.K[q0 + g + 1];
        dd[g] = hs_lds_at(c.ad, q0 + g) + (g == 0 ? d1 : c.d[q0 + g]);
        ll[g] = hs_lds_at(c.al, q0 + g);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) t[g] = hs_seg_fast<SIGNED>(uu[g], uu[g + 1], dd[g], ll[g], sm[g]);
      if (__builtin_amdgcn_ballot_w64(sm[0] || sm[1] || sm[2] || sm[3]) != 0ull) {
        asm volatile("" ::: "memory");  // keep this a branch
#pragma unroll
        for (int g = 0; g < 4; ++g) t[g] = hs_seg_exact<SIGNED>(uu[g], uu[g + 1], dd[g], ll[g]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) acc += (!packed || q0 + g < c.owed) ? t[g] : 0.0;
      u1 = uu[4];
      d1 = c.d[q0 + 4 < RT ? q0 + 4 : RT - 1];
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int q = q0 + g;
        if (q < RT - 1 && q < len) {
          const double u2 = hs_lds_at(c.aw, q + 1) * c.K[q + 1];
          const double dl = hs_lds_at(c.ad, q) + d1;
          const double term = hs_seg_exact<SIGNED>(u1, u2, dl, hs_lds_at(c.al, q));
          acc += (!packed || q < c.owed) ? term : 0.0;
          u1 = u2;
          d1 = c.d[q + 1];
        }
      }
    }
  }
  if (packed) {
    const int nKp = __builtin_amdgcn_readfirstlane(c.nKp);
    for (int off = 32; off >= nKp; off >>= 1) acc += __shfl_down(acc, off, 64);
  }
  return acc;
}

// One synchrotron work item: 64 (live photon energy, chunk of the gamma grid) pairs of
// Synchrotron._spectrum's integrand (radiative.py:282-342) for the walker whose weights and
// log-ratios on the synchrotron grid sit in LDS.  amap / ai0 / sq: the live energies compacted
// in order, the first live node of each, and q | cbrt(q) | CS1 per live energy (written by the
// tile waves of the liveness search).
struct hs_syn_lds {
  const int* amap; const double* ig2; const double* dig2; const double* ig23;
  const double* wr; const double* dwr; const double* lxs; const double* sq; const double* T64;
};
__device__ __forceinline__ void hs_syn_item(int ix, int lane, int nA, int Cd, int nG, int nEs,
                                            const hs_syn_lds& L, double* part_s) {
  const int* amap = L.amap;
  const int* ai0 = amap + nEs;
  const double* ig2 = L.ig2;
  const double* dig2 = L.dig2;
  const double* ig23 = L.ig23;
  const double* wr = L.wr;
  const double* dwr = L.dwr;
  const double* lxs = L.lxs;
  const double* sq = L.sq;
  const double* T64 = L.T64;
  const int nseg = nG - 1;
  (void)amap;
  const int vt = ix * 64 + lane;
  const int a = vt % nA, ch = vt / nA;
  if (ch < Cd) {
    const int sbeg = ai0[a];
    const int per = (nseg - sbeg + Cd - 1) / Cd;
    const int s0 = sbeg + ch * per;
    const int s1 = min(nseg, s0 + per);
    const double q = sq[a], cbq = sq[nEs + a];
    double acc = 0.0;
    if (s0 < s1) {
      // One node: gamma nelec dNdE / CS1 (radiative.py:335-338) and P.  No branch and no
      // select: every node of the range is live (x <= 746, the liveness search), a zero
      // weight gives u = 0 by itself (P is finite), so TWO nodes per trip are one straight
      // block the scheduler can interleave.  (One node per trip, behind a branch, is a
      // dependent chain of ~70 FP64 instructions: a wave alone took 21 cycles per
      // instruction, four waves per SIMD kept it 40 % busy.)  A weight that has underflowed
      // to 0 ends the integrand like a table's zero entry does (nh_seg_pos<false>).
      auto node = [&](int sn, double& u, double& P) {
        const double x = q * ig2[sn];
        P = syn_P1(cbq * ig23[sn]);
        u = wr[sn] * (P * nh_exp_tab(-x, T64));  // (x <= 746: the range starts at the first live node)
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
        acc += nh_seg_pos<false>(u1, uA, dlA, lxs[s]);  // P(x) exp(-x) >= 0
        acc += nh_seg_pos<false>(uA, uB, dlB, lxs[s + 1]);
        u1 = uB;
        P1 = PB;
      }
      if (s < s1) {
        double uA, PA;
        node(s + 1, uA, PA);
        const double dlA = dwr[s] + syn_dlnP1(P1, PA) - q * dig2[s];
        acc += nh_seg_pos<false>(u1, uA, dlA, lxs[s]);
      }
    }
    part_s[ch * nEs + a] = acc * sq[2 * nEs + a];  // linear in u: CS1 once per thread
  }
}
