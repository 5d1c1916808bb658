// nh_common.h -- shared host/device helpers of libnaima_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/naima_hip.h"

// ---------------------------------------------------------------------------
// physical constants: the values the reference computes from astropy CODATA-2018
// (radiative.py:11,34-40) and the literals it hard-codes
// ---------------------------------------------------------------------------
#define NH_E_GAUSS 4.803204712570263e-10
#define NH_C_CGS 29979245800.0
#define NH_HBAR_CGS 1.0545718176461565e-27
#define NH_M_E_G 9.1093837015e-28
#define NH_ALPHA_FS 0.0072973525693
#define NH_MEC2_EV 510998.9499961643
#define NH_MEC2_ERG_ 8.187105776823886e-07   /* m_e c^2 [erg] */
#define NH_ERG_TO_EV_ 624150907446.0764      /* 1 erg in eV */
#define NH_R0_CM 2.817940324670788e-13
#define NH_ERG_PER_EV 1.602176634e-12
#define NH_M_P_GEV 0.9382720881604903
#define NH_M_PI_GEV 0.1349766        /* radiative.py:1212 */
#define NH_T_TH_GEV 0.27966184       /* radiative.py:1213 */
#define NH_K_TO_MEC2 1.6863699549e-10        /* literal, radiative.py:557 */
#define NH_IC_PLANCK_NORM 2.6318735743809104e16 /* literal, radiative.py:571 */
#define NH_SIGT_LIT 6.652458734983284e-25    /* literal, radiative.py:650 */
#define NH_PI 3.141592653589793
#define NH_PI26 1.6449340668482264           /* pi^2/6 */

struct nh_prof_rec { hipEvent_t a, b; int kid; };

#define NH_NSIDE 4

struct nh_ctx {
  int device;
  hipStream_t stream;        // the CURRENT stream: every launch goes here
  hipStream_t main_stream;   // uploads, downloads, sync, graph capture origin
  hipStream_t side[NH_NSIDE];
  hipEvent_t ev_fork, ev_side[NH_NSIDE];
  bool side_used[NH_NSIDE];
  hipEvent_t t0, t1;
  bool profiling;
  std::vector<nh_prof_rec> recs;
  std::vector<hipEvent_t> pool;
  double acc_ms[NH_K_COUNT];
  long long acc_n[NH_K_COUNT];
  void* comm;      // ncclComm_t
  void* rccl_lib;  // dlopen handle
  void* scratch;   // library-owned device scratch (grown outside graph capture)
  size_t scratch_bytes;
  hipStream_t copy_stream;  // uploads that run AHEAD of the main stream (nh_upload_ahead), or NULL
  int* nan_word;  // device: NaN log-probabilities met by the accepts of the separate kernels (nh_nan_count)
  // device: the span clock (nh_clock_read) -- { wall_clock64 at the open span's start | ticks of
  // all closed spans | closed spans | (int) blocks of a closing launch that are through }
  long long* clk;
};

// Fill / upload that has COMPLETED when it returns.  The context's streams are non-blocking
// streams: the null stream's hipMemset orders nothing against them, and for device memory it
// may return before the fill has happened -- a launch queued right behind it on c->stream can
// then run first and have its words zeroed under it afterwards (seen with three and more
// processes on one GPU: a plan's `done` counter reset in the middle of a block of moves).
static inline hipError_t nh_fill_now(nh_ctx* c, void* p, int byte, size_t n) {
  hipError_t e = hipMemsetAsync(p, byte, n, c->main_stream);
  return e == hipSuccess ? hipStreamSynchronize(c->main_stream) : e;
}
static inline hipError_t nh_put_now(nh_ctx* c, void* dst, const void* src, size_t n) {
  hipError_t e = hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, c->main_stream);
  return e == hipSuccess ? hipStreamSynchronize(c->main_stream) : e;
}

// device scratch of at least `bytes`; contents are only valid within one entry point
int nh_scratch(nh_ctx* c, size_t bytes, void** out);

int nh_set_error(int code, const char* fmt, ...);

// tuning overrides (scripts/, experiments): read from the environment ONCE per process, not
// on the launch path
static inline int nh_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

#define NH_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return nh_set_error(NH_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                          __FILE__, __LINE__);                                          \
  } while (0)

#define NH_REQUIRE(cond, msg)                                      \
  do {                                                             \
    if (!(cond)) return nh_set_error(NH_EINVAL, "%s: %s", __func__, msg); \
  } while (0)

// scoped per-kernel HIP-event bracket (only when nh_profile_enable(ctx,1))
struct nh_prof_scope {
  nh_ctx* c;
  nh_prof_rec r;
  bool on;
  nh_prof_scope(nh_ctx* ctx, int kid) : c(ctx), on(ctx->profiling) {
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); }
      else (void)hipEventCreate(&e);
      return e;
    };
    r.a = get(); r.b = get(); r.kid = kid;
    (void)hipEventRecord(r.a, c->stream);
  }
  ~nh_prof_scope() {
    if (!on) return;
    (void)hipEventRecord(r.b, c->stream);
    c->recs.push_back(r);
  }
};

// ---------------------------------------------------------------------------
// device: one segment of trapz_loglog (utils.py:336-348) in the variables
//   u = x*y (signed),  dl = ln|u2/u1|,  lx = ln(x2/x1) > 0.
// reference:  b = log10(y2/y1)/log10(x2/x1);  b+1 = dl/lx
//   |b+1| > 1e-10 : y1*(x2*(x2/x1)^b - x1)/(b+1) = (u2-u1)/(b+1) = (u2-u1)*lx/dl
//   else (also NaN b: sign change)          : x1*y1*ln(x2/x1) = u1*lx
//   y1 == 0 or y2 == 0                      : 0
// dl is always assembled from SMALL, separately accurate pieces (log-ratios of
// adjacent nodes), never as a difference of two large logarithms.
//
// Evaluation, branch-free (lanes of a wave sit at different photon energies, so a
// wave-level fast path would rarely be taken):
//   |dl| >= 2^-7 : (u2-u1)*lx/dl with 1/dl from v_rcp_f64 + two Newton steps (dl is
//                  well scaled); relative error <= 2e-16/2^-7 = 3e-14
//   |dl| <  2^-7 : u1*lx*expm1(dl)/dl by a 5-term series (next term 4e-14): smooth
//                  through the peak of u, and its dl -> 0 limit u1*lx IS the
//                  reference's |b+1| <= 1e-10 branch
// ---------------------------------------------------------------------------
// 1/x for a well-scaled x: v_rcp_f64 + two Newton steps, without the
// v_div_scale/v_div_fixup dance of a generic IEEE division
// ---- the device span clock (nh_clock_read, naima_hip.h) --------------------------------------
// What the step loop's launches spend ON the device, measured on the launches themselves: the
// first workgroup of a span's first kernel stamps clk[0]; whoever is the LAST workgroup out of the
// span's last kernel adds (now - clk[0]) to clk[1].  A span lies inside the host interval around
// its launch calls and the synchronisation behind them, and spans do not overlap (one stream), so
// host time - span time >= 0 by construction: that is bench.py's `region_overhead_us`.
// (agent-scope atomics: the opening and the closing workgroup may sit on different XCDs -- different
// L2s -- of the SAME launch)
__device__ __forceinline__ void nh_clk_open(long long* clk) {
  __hip_atomic_store(clk, (long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void nh_clk_close(long long* clk) {
  const long long t1 = (long long)wall_clock64();
  const long long t0 = __hip_atomic_load(clk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  clk[1] += t1 - t0;  // (one closer at a time: launches of a stream do not overlap)
  clk[2] += 1;
}

__device__ __forceinline__ double nh_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

// one Newton step: v_rcp_f64 is good to ~2^-23, so this is ~2^-46 = 1.4e-14 relative --
// enough for the factor 1/dl of a segment term
__device__ __forceinline__ double nh_rcp1(double x) {
  double r = __builtin_amdgcn_rcp(x);
  return fma(r, fma(-x, r, 1.0), r);
}

// 1/x seeded in single precision (v_cvt + v_rcp_f32 + v_cvt: 9 cycles against 18 for
// v_rcp_f64, same 2^-23 seed) + one Newton step: 3e-14 relative.  |x| beyond the float
// range gives 0 (the NH_DL_ZERO marker relies on it), |x| < 1e-38 is the caller's series.
__device__ __forceinline__ double nh_rcp1f(double x) {
  const double r = (double)__builtin_amdgcn_rcpf((float)x);
  return fma(r, fma(-x, r, 1.0), r);
}

// SIGNED = false promises u1, u2 >= 0 (non-negative table and amplitude): the
// sign-change / NaN-ratio test of the log branch is then dead code
template <bool SIGNED = true>
__device__ __forceinline__ double nh_seg_term(double u1, double u2, double dl, double lx) {
  // series for |dl| < 2^-7: 1 + d/2 + d^2/6 + d^3/24 + d^4/120
  double f = fma(dl, 8.333333333333333e-03, 4.166666666666666e-02);
  f = fma(f, dl, 1.666666666666667e-01);
  f = fma(f, dl, 0.5);
  f = fma(f, dl, 1.0);
  const double ul = u1 * lx;
  const double ts = ul * f;
  const double td = (u2 - u1) * lx * nh_rcp1f(dl);  // (discarded when |dl| < 2^-7)
  double t = (fabs(dl) < 0.0078125) ? ts : td;
  if (SIGNED) {
    // sign change or NaN ratio -> NaN b in the reference -> its log branch x1*y1*ln(x2/x1)
    const bool logb = ((__double2hiint(u1) ^ __double2hiint(u2)) < 0) || !(dl == dl);
    t = logb ? ul : t;
  }
  // zero node (utils.py:347-348)
  return (u1 == 0.0 || u2 == 0.0) ? 0.0 : t;
}

// The same term for the hot loops over non-negative integrands (u1, u2 >= 0).  The series
// is needed only where |dl| is tiny (the few segments around the peak of u, b + 1 -> 0 in
// utils.py:336-345): it sits behind a wave-uniform branch, so a wave whose 64 lanes are
// all away from their peaks pays for (u2-u1) lx / dl only.  The main form loses
// eps/|dl| to the cancellation in u2 - u1: 2e-13 at the threshold 2^-10.
// ZERO = true keeps the reference's explicit test (a zero node -> 0, utils.py:347-348);
// ZERO = false relies on the caller's encoding: dl >= NH_DL_ZERO where a node is zero
// because its TABLE entry is zero (then the term is (u2-u1) lx 1e-300 ~ 0; two zero
// nodes give exactly 0).
#ifndef NH_SEG_SMALL_POS
#define NH_SEG_SMALL_POS 0x1p-10
#endif
#define NH_DL_ZERO 1e300
// nh_seg_term<true> for the interleaved tables of the half-step kernel: the series (|dl| <
// 2^-10, rare: the segment at the peak of u) behind a wave-uniform branch, the reference's log
// branch for a sign change selected by dl = NaN.  The FITPACK ringing of the pi0 look-up table
// changes sign in a fifth of its segments: testing the signs of u1, u2 per segment and
// branching on "any lane" took the long way in nearly every trip.
__device__ __forceinline__ double nh_seg_signed(double u1, double u2, double dl, double lx) {
  double t = ((u2 - u1) * lx) * nh_rcp1f(dl);
  const bool small = fabs(dl) < NH_SEG_SMALL_POS;  // (false for NaN)
  if (__builtin_amdgcn_ballot_w64(small) != 0) {
    asm volatile("" ::: "memory");  // keep this a branch: the compiler would if-convert it
    double f = fma(dl, 8.333333333333333e-03, 4.166666666666666e-02);
    f = fma(f, dl, 1.666666666666667e-01);
    f = fma(f, dl, 0.5);
    f = fma(f, dl, 1.0);
    t = small ? (u1 * lx) * f : t;
  }
  // sign change of the integrand: the interleaved table carries it as dl = NaN (the sign
  // pattern of a table is walker-independent, nh_table_interleave) -> NaN b in the reference
  // -> its log branch x1 y1 ln(x2/x1); a NaN weight ratio takes the same way, as there
  // A zero node (utils.py:347-348) needs no test of its own: a zero TABLE entry is marked by
  // dl >= NH_DL_ZERO (k_table_dlog; never NaN), whose reciprocal underflows to 0, as in
  // nh_seg_pos<false>; a weight that has underflowed to 0 ends the integrand the same way.
  return (dl == dl) ? t : u1 * lx;
}

template <bool ZERO>
__device__ __forceinline__ double nh_seg_pos(double u1, double u2, double dl, double lx) {
  double t = ((u2 - u1) * lx) * nh_rcp1f(dl);
  const bool small = fabs(dl) < NH_SEG_SMALL_POS;
  if (__builtin_amdgcn_ballot_w64(small) != 0) {
    asm volatile("" ::: "memory");  // keep this a branch: the compiler would if-convert it
    double f = fma(dl, 8.333333333333333e-03, 4.166666666666666e-02);
    f = fma(f, dl, 1.666666666666667e-01);
    f = fma(f, dl, 0.5);
    f = fma(f, dl, 1.0);
    t = small ? (u1 * lx) * f : t;
  }
  if (ZERO) t = (u1 == 0.0 || u2 == 0.0) ? 0.0 : t;
  return t;
}

typedef unsigned int nh_u32x2 __attribute__((ext_vector_type(2)));
// 8-byte load through a buffer descriptor at a 32-bit byte offset
__device__ __forceinline__ double nh_buf_f64(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  nh_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}

__device__ __forceinline__ double nh_heaviside(double x) {  // radiative.py:1539-1540
  return x > 0.0 ? 1.0 : (x < 0.0 ? 0.0 : (x == 0.0 ? 0.5 : x));
}

// ---------------------------------------------------------------------------
// lazy per-walker scalars and the priors of core.py:34-58 (shared by the likelihood
// kernel and the glue kernels)
// ---------------------------------------------------------------------------
struct nh_prior_pack {
  nh_prior t[NH_MAX_PRIOR];
  int n;
};

// value[w] = a * tf(b * base[w*stride] + c);  base == NULL -> the constant a
__device__ __forceinline__ double nh_lazy_apply(const nh_lazy& z, double raw) {
  double x = z.b * raw + z.c;
  switch (z.tf) {
    case NH_TF_POW10: x = exp10(x); break;  // 10**x, <= 1 ulp, a fifth of the generic pow
    case NH_TF_EXP: x = exp(x); break;
    case NH_TF_LOG: x = log(x); break;
    case NH_TF_LOG10: x = log10(x); break;
    case NH_TF_SQRT: x = sqrt(x); break;
    case NH_TF_SQUARE: x = x * x; break;
    case NH_TF_RECIP: x = 1.0 / x; break;
    default: break;
  }
  return z.a * x;
}

__device__ __forceinline__ double nh_lazy_eval(const nh_lazy& z, long long w) {
  if (!z.base) return z.a;
  return nh_lazy_apply(z, z.base[w * z.stride]);
}

// sum of the prior terms of core.py:34-58 for walker w
__device__ __forceinline__ double nh_prior_sum(const nh_prior_pack& P, long long w) {
  double s = 0.0;
  for (int t = 0; t < P.n; ++t) {
    const double v = nh_lazy_eval(P.t[t].x, w);
    const double p0 = P.t[t].p0, p1 = P.t[t].p1;
    double r;
    switch (P.t[t].kind) {
      case NH_PRIOR_UNIFORM: r = (p0 <= v && v <= p1) ? 0.0 : -INFINITY; break;
      case NH_PRIOR_NORMAL: r = -0.5 * (2.0 * NH_PI * p1) - (v - p0) * (v - p0) / (2.0 * p1); break;
      case NH_PRIOR_LOGUNIFORM: r = (v > 0.0 && v >= p0 && v <= p1) ? 1.0 / v : -INFINITY; break;
      default: r = v; break;
    }
    s += r;
  }
  return s;
}
