// nh_device.hip -- the pieces that keep a whole sampler half-step on the device:
// lazy per-walker scalars (parameter transforms folded into the consumer),
// elementwise fallbacks, linear combinations of spectra, priors, the stretch
// move on device-resident ensembles, and hipGraph capture/replay of the launch
// sequence that a naima model function produces.
#include "nh_common.h"

struct lazy_pack { nh_lazy c[NH_MAX_LAZY]; int n; };

__global__ void k_pack_rows(lazy_pack P, int N, double* __restrict__ out, int ld) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * P.n) return;
  int w = idx / P.n, j = idx % P.n;
  out[(long long)w * ld + j] = nh_lazy_eval(P.c[j], w);
}

extern "C" int nh_pack_rows(nh_ctx* c, const nh_lazy* cols, int ncols, int N, double* out,
                            int ld) {
  NH_REQUIRE(c && cols && out && ncols >= 1 && ncols <= NH_MAX_LAZY && N >= 0 && ld >= ncols,
             "bad argument");
  if (N == 0) return NH_OK;
  lazy_pack P;
  P.n = ncols;
  for (int j = 0; j < ncols; ++j) P.c[j] = cols[j];
  nh_prof_scope ps(c, NH_K_GLUE);
  int tot = N * ncols;
  hipLaunchKernelGGL(k_pack_rows, dim3((tot + 255) / 256), dim3(256), 0, c->stream, P, N, out, ld);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// ---------------------------------------------------------------------------
// elementwise fallbacks for model functions that do arithmetic the lazy form
// cannot express: out = op(x, y) with x, y lazy per-walker scalars
// ---------------------------------------------------------------------------
__global__ void k_ew_binary(int op, nh_lazy x, nh_lazy y, int N, double* __restrict__ out) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= N) return;
  double a = nh_lazy_eval(x, w), b = nh_lazy_eval(y, w), r;
  switch (op) {
    case NH_OP_ADD: r = a + b; break;
    case NH_OP_SUB: r = a - b; break;
    case NH_OP_MUL: r = a * b; break;
    case NH_OP_DIV: r = a / b; break;
    case NH_OP_POW: r = pow(a, b); break;
    case NH_OP_MAX: r = fmax(a, b); break;
    case NH_OP_MIN: r = fmin(a, b); break;
    case NH_OP_LT: r = a < b ? 1.0 : 0.0; break;
    case NH_OP_LE: r = a <= b ? 1.0 : 0.0; break;
    case NH_OP_GT: r = a > b ? 1.0 : 0.0; break;
    case NH_OP_GE: r = a >= b ? 1.0 : 0.0; break;
    default: r = a; break;
  }
  out[w] = r;
}

extern "C" int nh_ew_binary(nh_ctx* c, int op, const nh_lazy* x, const nh_lazy* y, int N,
                            double* out) {
  NH_REQUIRE(c && x && y && out && N >= 0, "bad argument");
  if (N == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_GLUE);
  hipLaunchKernelGGL(k_ew_binary, dim3((N + 255) / 256), dim3(256), 0, c->stream, op, *x, *y, N,
                     out);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// ---------------------------------------------------------------------------
// out[w][k] = rowfac[w] * colfac[k] * sum_j scale_j * comp_j[w*ld_j + k]
// ---------------------------------------------------------------------------
struct comp_pack { nh_comp c[NH_MAX_COMP]; int n; };

__global__ void k_lincomb(comp_pack P, const double* __restrict__ colfac, nh_lazy rowfac, int N,
                          int m, double* __restrict__ out, int ldo) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)N * m) return;
  int w = (int)(idx / m), k = (int)(idx % m);
  double s = 0.0;
  for (int j = 0; j < P.n; ++j) s += P.c[j].scale * P.c[j].ptr[(long long)w * P.c[j].ld + k];
  if (colfac) s *= colfac[k];
  s *= nh_lazy_eval(rowfac, w);
  out[(long long)w * ldo + k] = s;
}

extern "C" int nh_lincomb(nh_ctx* c, const nh_comp* comps, int ncomp, const double* colfac,
                          const nh_lazy* rowfac, int N, int m, double* out, int ldo) {
  NH_REQUIRE(c && comps && out && ncomp >= 1 && ncomp <= NH_MAX_COMP && N >= 0 && m >= 1 &&
                 ldo >= m, "bad argument");
  if (N == 0) return NH_OK;
  comp_pack P;
  P.n = ncomp;
  for (int j = 0; j < ncomp; ++j) P.c[j] = comps[j];
  nh_lazy rf = {nullptr, 0, 1.0, 0.0, 0.0, NH_TF_ID, 0};  // constant 1
  if (rowfac) rf = *rowfac;
  nh_prof_scope ps(c, NH_K_GLUE);
  long long tot = (long long)N * m;
  hipLaunchKernelGGL(k_lincomb, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, P,
                     colfac, rf, N, m, out, ldo);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// ---------------------------------------------------------------------------
// priors (core.py:34-58) on lazy per-walker scalars, summed: lp[w] = sum_t term_t
// ---------------------------------------------------------------------------
__global__ void k_priors(nh_prior_pack P, int N, double* __restrict__ lp) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < N) lp[w] = nh_prior_sum(P, w);
}

extern "C" int nh_priors(nh_ctx* c, const nh_prior* terms, int nterms, int N, double* lp) {
  NH_REQUIRE(c && terms && lp && nterms >= 1 && nterms <= NH_MAX_PRIOR && N >= 0, "bad argument");
  if (N == 0) return NH_OK;
  nh_prior_pack P;
  P.n = nterms;
  for (int j = 0; j < nterms; ++j) P.t[j] = terms[j];
  nh_prof_scope ps(c, NH_K_GLUE);
  hipLaunchKernelGGL(k_priors, dim3((N + 255) / 256), dim3(256), 0, c->stream, P, N, lp);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// ---------------------------------------------------------------------------
// device-resident stretch move.  The host draws the random numbers (replicated
// stream) for MANY half-steps at once and ships them as one block
//   blk[h] = { z[ns] | lnU[ns] (float64) | S[ns] | partner[ns] (int32) },  h < nhalf
// (stride 3*ns doubles).  A device-side cursor says which slice the current
// half-step uses; k_move_accept advances it, so that a captured graph can be
// replayed for every half-step of the block without touching its arguments.
//   S        global indices of the active walkers
//   partner  global index of each active walker's partner in the other half
// ---------------------------------------------------------------------------
struct move_slice { const double* rnd; const int* idx; };

__device__ __forceinline__ move_slice move_get(const double* blk, const int* cursor, int ns) {
  const double* r = blk + (long long)cursor[0] * 3 * ns;
  return {r, reinterpret_cast<const int*>(r + 2 * ns)};
}

__global__ void k_move_propose(const double* __restrict__ coords, const double* __restrict__ blk,
                               const int* __restrict__ cursor, int ns, int ndim, int lo, int nloc,
                               double* __restrict__ qT, double* __restrict__ factors) {
  // proposals of this rank's block [lo, lo+nloc) of the active half, TRANSPOSED:
  // qT[d][j] so that pars[d] is a contiguous vector over walkers
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nloc * ndim) return;
  const move_slice m = move_get(blk, cursor, ns);
  int d = t / nloc, j = t % nloc;
  int g = lo + j;
  double z = m.rnd[g];
  double cj = coords[(long long)m.idx[ns + g] * ndim + d];
  double sj = coords[(long long)m.idx[g] * ndim + d];
  qT[(long long)d * nloc + j] = cj - (cj - sj) * z;
  if (d == 0) factors[j] = (ndim - 1.0) * log(z);
}

extern "C" int nh_move_propose(nh_ctx* c, const double* coords, const double* blk,
                               const int* cursor, int ns, int ndim, int lo, int nloc, double* qT,
                               double* factors) {
  NH_REQUIRE(c && coords && blk && cursor && qT && factors && ns >= 1 && ndim >= 1 && lo >= 0 &&
                 nloc >= 0 && lo + nloc <= ns, "bad argument");
  if (nloc == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_GLUE);
  int tot = nloc * ndim;
  hipLaunchKernelGGL(k_move_propose, dim3((tot + 255) / 256), dim3(256), 0, c->stream, coords,
                     blk, cursor, ns, ndim, lo, nloc, qT, factors);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

__global__ void k_move_accept(double* __restrict__ coords, double* __restrict__ logp,
                              const double* __restrict__ blk, int* __restrict__ cursor,
                              const double* __restrict__ newlp, int ns, int ndim,
                              int* __restrict__ accepted, int* __restrict__ naccepted,
                              int* __restrict__ sel, int advance, int* __restrict__ nan_count) {
  // every rank holds the full ensemble and all ns new log-probabilities: the
  // proposal is recomputed here from (coords, z, partner) so that no coordinates
  // ever have to be exchanged between ranks.  Single block (ns <= 1024 per pass).
  const move_slice m = move_get(blk, cursor, ns);
  for (int j = threadIdx.x; j < ns; j += blockDim.x) {
    int me = m.idx[j], pa = m.idx[ns + j];
    double z = m.rnd[j];
    double d = (ndim - 1.0) * log(z) + newlp[j] - logp[me];
    bool acc = m.rnd[ns + j] < d;  // NaN compares false, as numpy
    if (acc) {
      for (int k = 0; k < ndim; ++k) {
        double cj = coords[(long long)pa * ndim + k];
        double sj = coords[(long long)me * ndim + k];
        coords[(long long)me * ndim + k] = cj - (cj - sj) * z;
      }
      logp[me] = newlp[j];
      if (naccepted) atomicAdd(&naccepted[me], 1);
    }
    accepted[j] = acc ? 1 : 0;
    if (sel) sel[j] = me;  // the slice's active walkers, for the blob scatter that follows
    if (newlp[j] != newlp[j]) atomicAdd(nan_count, 1);  // (emcee raises here: nh_nan_count)
  }
  __syncthreads();
  if (advance && threadIdx.x == 0) cursor[0] += 1;
}

extern "C" int nh_move_accept(nh_ctx* c, double* coords, double* logp, const double* blk,
                              int* cursor, const double* newlp, int ns, int ndim, int* accepted,
                              int* naccepted, int* sel, int advance) {
  NH_REQUIRE(c && coords && logp && blk && cursor && newlp && accepted && ns >= 1 && ndim >= 1,
             "bad argument");
  nh_prof_scope ps(c, NH_K_GLUE);
  // one block: the partner rows it reads are never rows it writes (partners are in the
  // complementary half), so no ordering between threads is needed; the cursor is
  // advanced after the barrier
  hipLaunchKernelGGL(k_move_accept, dim3(1), dim3(1024), 0, c->stream, coords, logp, blk, cursor,
                     newlp, ns, ndim, accepted, naccepted, sel, advance, c->nan_word);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

struct accept_blobs { double* cur[4]; int m[4]; int n; };

// the same on gathered rows { lnprob | blob 0 | blob 1 ... }: one workgroup per proposal
__global__ void k_move_accept_rows(double* __restrict__ coords, double* __restrict__ logp,
                                   const double* __restrict__ blk, int* __restrict__ cursor,
                                   const double* __restrict__ rows, int width, int ns, int ndim,
                                   int* __restrict__ accepted, int* __restrict__ naccepted,
                                   int* __restrict__ sel, accept_blobs B, int* __restrict__ nan_count) {
  const move_slice m = move_get(blk, cursor, ns);
  const int j = blockIdx.x;
  const int me = m.idx[j], pa = m.idx[ns + j];
  const double z = m.rnd[j];
  const double* r = rows + (long long)j * width;
  const double d = (ndim - 1.0) * log(z) + r[0] - logp[me];
  const bool acc = m.rnd[ns + j] < d;  // NaN compares false, as numpy
  __syncthreads();  // (everybody has read logp[me] before anybody writes it)
  if (acc) {
    for (int k = threadIdx.x; k < ndim; k += blockDim.x) {
      const double cj = coords[(long long)pa * ndim + k];
      const double sj = coords[(long long)me * ndim + k];
      coords[(long long)me * ndim + k] = cj - (cj - sj) * z;
    }
    const double* rb = r + 1;
    for (int b = 0; b < B.n; ++b) {
      for (int t = threadIdx.x; t < B.m[b]; t += blockDim.x)
        B.cur[b][(long long)me * B.m[b] + t] = rb[t];
      rb += B.m[b];
    }
  }
  if (threadIdx.x == 0) {
    if (acc) {
      logp[me] = r[0];
      if (naccepted) atomicAdd(&naccepted[me], 1);
    }
    accepted[j] = acc ? 1 : 0;
    if (sel) sel[j] = me;
    if (r[0] != r[0]) atomicAdd(nan_count, 1);  // (emcee raises here: nh_nan_count)
  }
}

__global__ void k_cursor_advance(int* cursor) { cursor[0] += 1; }

// up to eight 64-bit words := values that travel as kernel arguments (stream-ordered; no
// host staging buffer, unlike an upload from pageable memory)
struct nh_words8 { long long v[8]; };
__global__ void k_set_words(long long* dst, nh_words8 w, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = w.v[threadIdx.x];
}

extern "C" int nh_set_words(nh_ctx* c, void* dev, const long long* values, int n) {
  NH_REQUIRE(c && dev && values && n >= 1 && n <= 8, "bad argument");
  nh_words8 w;
  for (int i = 0; i < 8; ++i) w.v[i] = i < n ? values[i] : 0;
  hipLaunchKernelGGL(k_set_words, dim3(1), dim3(8), 0, c->stream, static_cast<long long*>(dev), w, n);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_move_accept_rows(nh_ctx* c, double* coords, double* logp, const double* blk,
                                   int* cursor, const double* rows, int width, int ns, int ndim,
                                   int* accepted, int* naccepted, int* sel, int advance,
                                   int nblobs, double* const* cur, const int* m) {
  NH_REQUIRE(c && coords && logp && blk && cursor && rows && accepted && ns >= 1 && ndim >= 1 &&
                 width >= 1 && nblobs >= 0 && nblobs <= 4 && (nblobs == 0 || (cur && m)),
             "bad argument");
  accept_blobs B;
  B.n = nblobs;
  int wsum = 1;
  for (int b = 0; b < nblobs; ++b) {
    NH_REQUIRE(cur[b] && m[b] >= 1, "bad blob");
    B.cur[b] = cur[b];
    B.m[b] = m[b];
    wsum += m[b];
  }
  NH_REQUIRE(width >= wsum, "rows narrower than 1 + the blobs' lengths");
  nh_prof_scope ps(c, NH_K_GLUE);
  hipLaunchKernelGGL(k_move_accept_rows, dim3((unsigned)ns), dim3(64), 0, c->stream, coords, logp,
                     blk, cursor, rows, width, ns, ndim, accepted, naccepted, sel, B, c->nan_word);
  if (advance) hipLaunchKernelGGL(k_cursor_advance, dim3(1), dim3(1), 0, c->stream, cursor);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// blob bookkeeping: dst[idx[lo+j]][:] = src[j][:] where accepted[lo+j]
__global__ void k_scatter_rows(double* __restrict__ dst, int ldd, const double* __restrict__ src,
                               int lds, const int* __restrict__ idx,
                               const int* __restrict__ accepted, int lo, int nloc, int m) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)nloc * m) return;
  int j = (int)(t / m), k = (int)(t % m);
  if (accepted == nullptr || accepted[lo + j])
    dst[(long long)idx[lo + j] * ldd + k] = src[(long long)j * lds + k];
}

extern "C" int nh_scatter_rows(nh_ctx* c, double* dst, int ldd, const double* src, int lds,
                               const int* idx, const int* accepted, int lo, int nloc, int m) {
  NH_REQUIRE(c && dst && src && idx && nloc >= 0 && m >= 1 && ldd >= m && lds >= m,
             "bad argument");
  if (nloc == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_GLUE);
  long long tot = (long long)nloc * m;
  hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     c->stream, dst, ldd, src, lds, idx, accepted, lo, nloc, m);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_copy(nh_ctx* c, void* dst, const void* src, long long bytes) {
  NH_REQUIRE(c && dst && src && bytes >= 0, "bad argument");
  NH_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
  return NH_OK;
}

// ---------------------------------------------------------------------------
// hipGraph capture / replay of everything launched on the context's stream
// ---------------------------------------------------------------------------
extern "C" int nh_graph_begin(nh_ctx* c) {
  NH_REQUIRE(c, "ctx is NULL");
  NH_REQUIRE(!c->profiling, "disable per-kernel profiling before capturing a graph");
  NH_REQUIRE(c->stream == c->main_stream, "join the side streams before capturing");
  NH_CHECK_HIP(hipStreamBeginCapture(c->main_stream, hipStreamCaptureModeRelaxed));
  return NH_OK;
}

extern "C" int nh_graph_end(nh_ctx* c, void** exec_out) {
  NH_REQUIRE(c && exec_out, "bad argument");
  hipGraph_t g = nullptr;
  int rcj = nh_stream_join(c);
  if (rcj) return rcj;
  NH_CHECK_HIP(hipStreamEndCapture(c->main_stream, &g));
  hipGraphExec_t ex = nullptr;
  hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess)
    return nh_set_error(NH_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  *exec_out = ex;
  return NH_OK;
}

extern "C" int nh_graph_launch(nh_ctx* c, void* exec) {
  NH_REQUIRE(c && exec, "bad argument");
  NH_CHECK_HIP(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(exec), c->main_stream));
  return NH_OK;
}

extern "C" int nh_graph_destroy(nh_ctx* c, void* exec) {
  NH_REQUIRE(c, "ctx is NULL");
  if (exec) NH_CHECK_HIP(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(exec)));
  return NH_OK;
}
