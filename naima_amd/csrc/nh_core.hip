// nh_core.hip -- context/memory, trapz_loglog, particle weights, the generic
// per-walker table reduction, lnprobmodel and the stretch-move kernels.
// gfx950 (MI355X) only; FP64 throughout, denormals on, no fast-math.
#include "nh_common.h"
#include "nh_lnprob.h"
#include <cstdlib>

static thread_local char g_err[512] = "";

int nh_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* nh_last_error(void) { return g_err; }
extern "C" int nh_version(void) { return 100; }

// rocprofv3 (ROCm 7.2) + hipGraph replay from pre-built AQL packets dies inside hipGraphLaunch
// on this library's step graphs (sixteen kernel nodes, 3 KB by-value arguments); node-by-node
// replay does not (profiles/README.md, round 3).  Best effort for hosts that load the library
// without naima_amd/_lib.py (which does the same before the HIP runtime is loaded at all): only
// under a profiler, never over an explicit setting.
__attribute__((constructor(101))) static void nh_profiler_workaround() {
  const char* pre = getenv("LD_PRELOAD");
  if (getenv("ROCPROFILER_LIBRARY_CTOR") || getenv("ROCP_TOOL_LIBRARIES") ||
      (pre && strstr(pre, "rocprofiler-sdk")))
    setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0", 0);
}

extern "C" int nh_create(int device, nh_ctx** out) {
  NH_REQUIRE(out != nullptr, "out is NULL");
  int ndev = 0;
  NH_CHECK_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev)
    return nh_set_error(NH_EINVAL, "nh_create: device %d out of range (%d visible)", device, ndev);
  NH_CHECK_HIP(hipSetDevice(device));
  nh_ctx* c = new nh_ctx();
  c->device = device;
  c->profiling = false;
  c->comm = nullptr;
  c->rccl_lib = nullptr;
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  c->copy_stream = nullptr;
  c->nan_word = nullptr;
  c->clk = nullptr;
  memset(c->acc_ms, 0, sizeof(c->acc_ms));
  memset(c->acc_n, 0, sizeof(c->acc_n));
  NH_CHECK_HIP(hipStreamCreateWithFlags(&c->main_stream, hipStreamNonBlocking));
  c->stream = c->main_stream;
  for (int i = 0; i < NH_NSIDE; ++i) {
    NH_CHECK_HIP(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
    NH_CHECK_HIP(hipEventCreateWithFlags(&c->ev_side[i], hipEventDisableTiming));
    c->side_used[i] = false;
  }
  NH_CHECK_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  NH_CHECK_HIP(hipEventCreate(&c->t0));
  NH_CHECK_HIP(hipEventCreate(&c->t1));
  NH_CHECK_HIP(hipMalloc(&c->nan_word, sizeof(int)));
  NH_CHECK_HIP(nh_fill_now(c, c->nan_word, 0, sizeof(int)));
  NH_CHECK_HIP(hipMalloc(&c->clk, 8 * sizeof(long long)));
  NH_CHECK_HIP(nh_fill_now(c, c->clk, 0, 8 * sizeof(long long)));
  *out = c;
  return NH_OK;
}

// The device span clock: what the step loop's launches have spent on the device since the last
// reset, measured on those launches themselves (nh_common.h: nh_clk_open / nh_clk_close; a span =
// k_half_step_run + k_run_epilogue of one block of moves, or the launches of one half-step of the
// per-launch loop).  out = { ticks inside closed spans, closed spans, ticks per millisecond }.
// Synchronises the stream; nothing is added to any launch by reading.
extern "C" int nh_clock_read(nh_ctx* c, int reset, long long* out) {
  NH_REQUIRE(c && out, "bad argument");
  int rc = nh_sync(c);
  if (rc) return rc;
  long long w[4];
  NH_CHECK_HIP(hipMemcpy(w, c->clk, sizeof(w), hipMemcpyDeviceToHost));
  int khz = 0;
  NH_CHECK_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
  out[0] = w[1];
  out[1] = w[2];
  out[2] = khz > 0 ? khz : 100000;  // (wall_clock64: 100 MHz on gfx942 / gfx950)
  if (reset) NH_CHECK_HIP(nh_fill_now(c, c->clk, 0, 8 * sizeof(long long)));
  return NH_OK;
}

// NaN log-probabilities the accepts of the SEPARATE kernels have met since the last reset
// (nh_lnprob / nh_integrate_tables_lnprob / nh_synchrotron_lnprob with a move, nh_move_accept,
// nh_move_accept_rows; the one-launch kernels count per plan: nh_half_step_counts).  emcee raises
// ValueError("Probability function returned NaN") at the first one; a launch rejects the
// proposal (NaN compares false) and counts.  Synchronises the stream.
extern "C" int nh_nan_count(nh_ctx* c, int reset, int* count) {
  NH_REQUIRE(c && count, "bad argument");
  int rc = nh_sync(c);
  if (rc) return rc;
  NH_CHECK_HIP(hipMemcpy(count, c->nan_word, sizeof(int), hipMemcpyDeviceToHost));
  if (reset && *count) NH_CHECK_HIP(nh_fill_now(c, c->nan_word, 0, sizeof(int)));
  return NH_OK;
}

extern "C" int nh_destroy(nh_ctx* c) {
  if (!c) return NH_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->main_stream);
  if (c->nan_word) (void)hipFree(c->nan_word);
  if (c->clk) (void)hipFree(c->clk);
  nh_comm_destroy(c);
  for (int i = 0; i < NH_NSIDE; ++i) {
    (void)hipStreamSynchronize(c->side[i]);
    (void)hipStreamDestroy(c->side[i]);
    (void)hipEventDestroy(c->ev_side[i]);
  }
  (void)hipEventDestroy(c->ev_fork);
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamDestroy(c->copy_stream);
  }
  if (c->scratch) (void)hipFree(c->scratch);
  for (auto& r : c->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto& e : c->pool) (void)hipEventDestroy(e);
  (void)hipEventDestroy(c->t0);
  (void)hipEventDestroy(c->t1);
  (void)hipStreamDestroy(c->main_stream);
  delete c;
  return NH_OK;
}

int nh_scratch(nh_ctx* c, size_t bytes, void** out) {
  if (bytes > c->scratch_bytes) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(c->main_stream, &st);
    if (st != hipStreamCaptureStatusNone)
      return nh_set_error(NH_ENOMEM, "scratch must grow to %zu B during graph capture: run the "
                          "same call once eagerly first", bytes);
    NH_CHECK_HIP(hipStreamSynchronize(c->main_stream));
    if (c->scratch) NH_CHECK_HIP(hipFree(c->scratch));
    size_t want = bytes + bytes / 4;
    hipError_t e = hipMalloc(&c->scratch, want);
    if (e != hipSuccess) {
      c->scratch = nullptr; c->scratch_bytes = 0;
      return nh_set_error(NH_ENOMEM, "hipMalloc(%zu) for scratch: %s", want, hipGetErrorString(e));
    }
    c->scratch_bytes = want;
  }
  *out = c->scratch;
  return NH_OK;
}

// HIP devices this process sees (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES applied): what a
// launcher checks before it starts one rank per GPU.  No context needed.
extern "C" int nh_device_count(int* count) {
  NH_REQUIRE(count != nullptr, "count is NULL");
  int ndev = 0;
  NH_CHECK_HIP(hipGetDeviceCount(&ndev));
  *count = ndev;
  return NH_OK;
}

extern "C" int nh_device_info(nh_ctx* c, char* name, int name_len, int* cus, double* hbm,
                              int* clock_khz) {
  NH_REQUIRE(c, "ctx is NULL");
  hipDeviceProp_t p;
  NH_CHECK_HIP(hipGetDeviceProperties(&p, c->device));
  if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  if (cus) *cus = p.multiProcessorCount;
  if (hbm) *hbm = (double)p.totalGlobalMem;
  if (clock_khz) *clock_khz = p.clockRate;
  return NH_OK;
}

// The PCI bus id of the context's GPU ("0000:c1:00.0"): what tells two ranks apart that were
// meant to sit on different GPUs (bench.py prints every rank's and refuses a run whose ranks
// share a device unless NAIMA_AMD_DEVICE pinned them there on purpose).
extern "C" int nh_device_pci_bus_id(nh_ctx* c, char* out, int len) {
  NH_REQUIRE(c && out && len >= 16, "bad argument");
  NH_CHECK_HIP(hipDeviceGetPCIBusId(out, len, c->device));
  return NH_OK;
}

extern "C" int nh_alloc(nh_ctx* c, long long bytes, void** out) {
  NH_REQUIRE(c && out && bytes >= 0, "bad argument");
  NH_CHECK_HIP(hipSetDevice(c->device));
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, (size_t)(bytes > 0 ? bytes : 8));
  if (e != hipSuccess) return nh_set_error(NH_ENOMEM, "hipMalloc(%lld): %s", bytes, hipGetErrorString(e));
  *out = p;
  return NH_OK;
}

extern "C" int nh_free(nh_ctx* c, void* p) {
  NH_REQUIRE(c, "ctx is NULL");
  if (p) {
    int rc = nh_sync(c);
    if (rc) return rc;
    NH_CHECK_HIP(hipFree(p));
  }
  return NH_OK;
}

extern "C" int nh_upload(nh_ctx* c, void* dst, const void* src, long long bytes) {
  NH_REQUIRE(c && dst && src && bytes >= 0, "bad argument");
  // pageable source: hipMemcpyAsync stages it before returning, so the host
  // buffer may be reused immediately; ordering on the stream is preserved
  NH_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
  return NH_OK;
}

// Host -> device on a stream of its own, NOT ordered behind what the main stream is doing
// (and not waited for by nh_sync): the upload of the NEXT block of stretch-move random numbers
// while the current block's launch runs.  `marker` (nh_marker_create) is recorded behind the
// copy; nh_stream_wait_marker makes the main stream wait for it before the first launch that
// reads the data.  The caller keeps `dst` disjoint from what the launches behind `after` read.
extern "C" int nh_upload_ahead(nh_ctx* c, void* dst, const void* src, long long bytes,
                               void* marker, void* after) {
  NH_REQUIRE(c && dst && src && bytes >= 0 && marker, "bad argument");
  if (!c->copy_stream)
    NH_CHECK_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  // `after` (a marker recorded on the main stream, or NULL): the copy may not start before it --
  // the last launch that could still be reading the bytes about to be overwritten
  if (after)
    NH_CHECK_HIP(hipStreamWaitEvent(c->copy_stream, reinterpret_cast<hipEvent_t>(after), 0));
  NH_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->copy_stream));
  NH_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(marker), c->copy_stream));
  return NH_OK;
}

// everything launched on the context's stream from here on waits for `marker`
extern "C" int nh_stream_wait_marker(nh_ctx* c, void* marker) {
  NH_REQUIRE(c && marker, "bad argument");
  NH_CHECK_HIP(hipStreamWaitEvent(c->stream, reinterpret_cast<hipEvent_t>(marker), 0));
  return NH_OK;
}

extern "C" int nh_download(nh_ctx* c, void* dst, const void* src, long long bytes) {
  NH_REQUIRE(c && dst && src && bytes >= 0, "bad argument");
  int rc = nh_stream_join(c);
  if (rc) return rc;
  NH_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->main_stream));
  NH_CHECK_HIP(hipStreamSynchronize(c->main_stream));
  return NH_OK;
}

extern "C" int nh_memset(nh_ctx* c, void* p, int byte, long long bytes) {
  NH_REQUIRE(c && p && bytes >= 0, "bad argument");
  NH_CHECK_HIP(hipMemsetAsync(p, byte, (size_t)bytes, c->stream));
  return NH_OK;
}

extern "C" int nh_sync(nh_ctx* c) {
  NH_REQUIRE(c, "ctx is NULL");
  int rc = nh_stream_join(c);
  if (rc) return rc;
  NH_CHECK_HIP(hipStreamSynchronize(c->main_stream));
  return NH_OK;
}

// ---- fork / join over side streams (independent emission components of one model
// evaluation run concurrently; under graph capture these become graph branches) ----
extern "C" int nh_stream_fork(nh_ctx* c, int side) {
  NH_REQUIRE(c && side >= 0 && side < NH_NSIDE, "bad side stream");
  // the side stream starts after everything issued so far on the main stream
  NH_CHECK_HIP(hipEventRecord(c->ev_fork, c->main_stream));
  NH_CHECK_HIP(hipStreamWaitEvent(c->side[side], c->ev_fork, 0));
  c->side_used[side] = true;
  c->stream = c->side[side];
  return NH_OK;
}

// the side stream starts after a marker recorded earlier on the main stream (the launch
// that produced its inputs) instead of after everything issued so far: under capture the
// branch hangs off that node only and runs beside whatever followed it
extern "C" int nh_stream_fork_at(nh_ctx* c, int side, void* marker) {
  NH_REQUIRE(c && side >= 0 && side < NH_NSIDE && marker, "bad argument");
  NH_CHECK_HIP(hipStreamWaitEvent(c->side[side], reinterpret_cast<hipEvent_t>(marker), 0));
  c->side_used[side] = true;
  c->stream = c->side[side];
  return NH_OK;
}

extern "C" int nh_stream_switch(nh_ctx* c, int side) {
  NH_REQUIRE(c && side >= -1 && side < NH_NSIDE, "bad side stream");
  c->stream = side < 0 ? c->main_stream : c->side[side];
  return NH_OK;
}

extern "C" int nh_stream_wait(nh_ctx* c, int waiter, int producer) {
  NH_REQUIRE(c && waiter >= -1 && waiter < NH_NSIDE && producer >= -1 && producer < NH_NSIDE,
             "bad stream index");
  if (waiter == producer) return NH_OK;
  hipStream_t ws = waiter < 0 ? c->main_stream : c->side[waiter];
  if (producer < 0) {
    NH_CHECK_HIP(hipEventRecord(c->ev_fork, c->main_stream));
    NH_CHECK_HIP(hipStreamWaitEvent(ws, c->ev_fork, 0));
  } else {
    NH_CHECK_HIP(hipEventRecord(c->ev_side[producer], c->side[producer]));
    NH_CHECK_HIP(hipStreamWaitEvent(ws, c->ev_side[producer], 0));
  }
  return NH_OK;
}

extern "C" int nh_stream_join(nh_ctx* c) {
  NH_REQUIRE(c, "ctx is NULL");
  for (int i = 0; i < NH_NSIDE; ++i) {
    if (!c->side_used[i]) continue;
    NH_CHECK_HIP(hipEventRecord(c->ev_side[i], c->side[i]));
    NH_CHECK_HIP(hipStreamWaitEvent(c->main_stream, c->ev_side[i], 0));
    c->side_used[i] = false;
  }
  c->stream = c->main_stream;
  return NH_OK;
}

extern "C" int nh_timer_start(nh_ctx* c) {
  NH_REQUIRE(c, "ctx is NULL");
  NH_CHECK_HIP(hipEventRecord(c->t0, c->stream));
  return NH_OK;
}

extern "C" int nh_timer_stop(nh_ctx* c, double* ms) {
  NH_REQUIRE(c && ms, "bad argument");
  NH_CHECK_HIP(hipEventRecord(c->t1, c->stream));
  NH_CHECK_HIP(hipEventSynchronize(c->t1));
  float f = 0;
  NH_CHECK_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
  *ms = f;
  return NH_OK;
}

extern "C" int nh_profile_enable(nh_ctx* c, int on) {
  NH_REQUIRE(c, "ctx is NULL");
  c->profiling = on != 0;
  return NH_OK;
}

__global__ void k_empty() {}

// mean HIP-event time of an EMPTY kernel bracketed exactly like the profiled launches:
// the fixed cost the event pair adds to every measurement (subtract it to compare with
// rocprofv3's kernel durations)
extern "C" int nh_profile_calibrate(nh_ctx* c, int reps, double* overhead_us) {
  NH_REQUIRE(c && overhead_us && reps >= 1, "bad argument");
  hipEvent_t a, b;
  NH_CHECK_HIP(hipEventCreate(&a));
  NH_CHECK_HIP(hipEventCreate(&b));
  double tot = 0.0;
  for (int i = 0; i < reps + 3; ++i) {
    NH_CHECK_HIP(hipEventRecord(a, c->stream));
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, c->stream);
    NH_CHECK_HIP(hipEventRecord(b, c->stream));
    NH_CHECK_HIP(hipEventSynchronize(b));
    float f = 0;
    NH_CHECK_HIP(hipEventElapsedTime(&f, a, b));
    if (i >= 3) tot += f;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *overhead_us = tot / reps * 1e3;
  return NH_OK;
}

extern "C" int nh_profile_read(nh_ctx* c, double* ms, long long* n, int reset) {
  NH_REQUIRE(c, "ctx is NULL");
  int rcj = nh_sync(c);
  if (rcj) return rcj;
  for (auto& r : c->recs) {
    float f = 0;
    NH_CHECK_HIP(hipEventElapsedTime(&f, r.a, r.b));
    c->acc_ms[r.kid] += f;
    c->acc_n[r.kid] += 1;
    c->pool.push_back(r.a);
    c->pool.push_back(r.b);
  }
  c->recs.clear();
  for (int i = 0; i < NH_K_COUNT; ++i) {
    if (ms) ms[i] = c->acc_ms[i];
    if (n) n[i] = c->acc_n[i];
  }
  if (reset) {
    memset(c->acc_ms, 0, sizeof(c->acc_ms));
    memset(c->acc_n, 0, sizeof(c->acc_n));
  }
  return NH_OK;
}

// ---------------------------------------------------------------------------
// row 1: trapz_loglog as written (utils.py:336-348), one wave per row
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_trapz_loglog(const double* __restrict__ y,
                                                       const double* __restrict__ x, int nrows,
                                                       int n, double* __restrict__ out) {
  int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const double* yr = y + (long long)row * n;
  double acc = 0.0;
  for (int s = lane; s < n - 1; s += 64) {
    double y1 = yr[s], y2 = yr[s + 1], x1 = x[s], x2 = x[s + 1];
    double b = log10(y2 / y1) / log10(x2 / x1);
    double tp = (y1 * (x2 * pow(x2 / x1, b) - x1)) / (b + 1.0);
    double tl = x1 * y1 * log(x2 / x1);
    double t = (fabs(b + 1.0) > 1e-10) ? tp : tl;  // NaN b -> log branch
    if (y1 == 0.0 || y2 == 0.0 || x1 == x2) t = 0.0;
    acc += t;
  }
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc;
}

// the per-segment terms (utils.py:350-351, intervals=True)
__global__ __launch_bounds__(256) void k_trapz_loglog_intervals(const double* __restrict__ y,
                                                                 const double* __restrict__ x,
                                                                 int nrows, int n,
                                                                 double* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)nrows * (n - 1)) return;
  const int row = (int)(idx / (n - 1)), s = (int)(idx % (n - 1));
  const double y1 = y[(long long)row * n + s], y2 = y[(long long)row * n + s + 1];
  const double x1 = x[s], x2 = x[s + 1];
  const double b = log10(y2 / y1) / log10(x2 / x1);
  const double tp = (y1 * (x2 * pow(x2 / x1, b) - x1)) / (b + 1.0);
  const double tl = x1 * y1 * log(x2 / x1);
  double t = (fabs(b + 1.0) > 1e-10) ? tp : tl;  // NaN b -> log branch
  if (y1 == 0.0 || y2 == 0.0 || x1 == x2) t = 0.0;
  out[idx] = t;
}

extern "C" int nh_trapz_loglog_intervals(nh_ctx* c, const double* y, const double* x, int nrows,
                                         int n, double* out) {
  NH_REQUIRE(c && y && x && out && nrows >= 0 && n >= 2, "bad argument");
  if (nrows == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_INTEGRATE);
  const long long tot = (long long)nrows * (n - 1);
  hipLaunchKernelGGL(k_trapz_loglog_intervals, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     c->stream, y, x, nrows, n, out);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_trapz_loglog(nh_ctx* c, const double* y, const double* x, int nrows, int n,
                               double* out) {
  NH_REQUIRE(c && y && x && out && nrows >= 0 && n >= 1, "bad argument");
  if (nrows == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_INTEGRATE);
  hipLaunchKernelGGL(k_trapz_loglog, dim3((nrows + 3) / 4), dim3(256), 0, c->stream, y, x, nrows,
                     n, out);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// ---------------------------------------------------------------------------
// rows 2,3: per-walker particle spectrum on a grid -> weights w = xg*n and the
// log-ratios dlw[i] = ln|w[i+1]/w[i]| assembled analytically per segment
// ---------------------------------------------------------------------------
#include "nh_front.h"

__global__ __launch_bounds__(256) void k_particle_weights(
    int kind, const double* __restrict__ params, int N, const double* __restrict__ e,
    const double* __restrict__ xg, int nG, double unit_scale, double* __restrict__ w,
    double* __restrict__ dlw, double* __restrict__ nout) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)N * nG) return;
  int wi = (int)(idx / nG), i = (int)(idx % nG);
  const double* pr = params + (long long)wi * NH_PD_NPAR;
  pd_par p = {pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6]};
  const double E = e[i];
  const bool last = i + 1 >= nG;
  const double E2 = last ? E : e[i + 1];
  const double g = xg[i];
  const double lrE = last ? 0.0 : log(E2 / E);
  const double lrx = last ? 0.0 : log(xg[i + 1] / g);
  double n, dsh;
  pd_node(kind, p, E, E2, lrE, n, dsh);
  n *= unit_scale;
  w[idx] = g * n;
  if (nout) nout[idx] = n;
  dlw[idx] = last ? 0.0 : lrx + dsh;
}

// the same walkers on several grids (the components of one model evaluation use
// different electron grids: Synchrotron from 1 GeV, IC from Eemin, We(> 1 TeV) ...).
// Block = (walker, 256 nodes of the concatenated grids): the walker's parameter row
// arrives through the scalar cache and its logarithms are taken once per block; with
// the grid's own logarithms (ln e, lx: walker-independent, cached by the host) a node
// costs two exponentials instead of four logarithms, four divisions and two exponentials.
__global__ __launch_bounds__(256) void k_particle_weights_multi(
    int kind, const double* __restrict__ params, int N, pw_grids G) {
  const int wi = blockIdx.x;
  const double* pr = params + (long long)wi * NH_PD_NPAR;
  const pd_par p = {pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6]};
  __shared__ double lg[3];
  if (threadIdx.x < 3) {
    const double v = threadIdx.x == 0 ? p.e0 : (threadIdx.x == 1 ? p.ec : p.eb);
    lg[threadIdx.x] = v > 0.0 ? log(v) : 0.0;
  }
  __syncthreads();
  const int j = blockIdx.y * 256 + threadIdx.x;
  if (j >= G.off[G.n]) return;
  int g = 0;
  while (g + 1 < G.n && j >= G.off[g + 1]) ++g;
  const int i = j - G.off[g];
  const int nG = G.nG[g];
  const double* e = G.e[g];
  const double* xg = G.xg[g];
  const bool last = i + 1 >= nG;
  const double E = e[i];
  const double E2 = last ? E : e[i + 1];
  const double gx = xg[i];
  double lr = 0.0;
  if (!last) lr = G.lx[g] ? G.lx[g][i] : log(xg[i + 1] / gx);
  const double lnE = G.lne[g] ? G.lne[g][i] : log(E);
  double n, dsh;
  pd_core(kind, p, lnE - lg[0], lnE - lg[1], lg[2] - lg[0], E < p.eb, E2 < p.eb, lr, n, dsh);
  n *= G.scale[g];
  const long long loc = (long long)wi * nG + i;
  G.w[g][loc] = gx * n;
  G.dlw[g][loc] = last ? 0.0 : lr + dsh;
}

extern "C" int nh_particle_weights_multi(nh_ctx* c, int kind, const double* params, int N,
                                         const nh_grid* grids, int ngrids) {
  NH_REQUIRE(c && params && grids, "NULL pointer");
  NH_REQUIRE(kind >= NH_PD_POWERLAW && kind <= NH_PD_LOGPARABOLA, "unknown particle distribution kind");
  NH_REQUIRE(N >= 0 && ngrids >= 1 && ngrids <= NH_MAX_GRIDS, "bad sizes");
  if (N == 0) return NH_OK;
  pw_grids G;
  G.n = ngrids;
  G.off[0] = 0;
  for (int g = 0; g < ngrids; ++g) {
    NH_REQUIRE(grids[g].e_eV && grids[g].xg && grids[g].w && grids[g].dlw && grids[g].nG >= 2,
               "bad grid descriptor");
    G.e[g] = grids[g].e_eV; G.xg[g] = grids[g].xg; G.w[g] = grids[g].w; G.dlw[g] = grids[g].dlw;
    G.lne[g] = grids[g].ln_e; G.lx[g] = grids[g].lx;
    G.scale[g] = grids[g].unit_scale; G.nG[g] = grids[g].nG;
    NH_REQUIRE((long long)G.off[g] + grids[g].nG < (1LL << 30), "grids too long");
    G.off[g + 1] = G.off[g] + grids[g].nG;
  }
  nh_prof_scope ps(c, NH_K_PDIST);
  hipLaunchKernelGGL(k_particle_weights_multi,
                     dim3((unsigned)N, (unsigned)((G.off[ngrids] + 255) / 256)), dim3(256), 0,
                     c->stream, kind, params, N, G);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// ---------------------------------------------------------------------------
// The launch in FRONT of a model evaluation in the device step loop: proposal of the
// stretch move (emcee StretchMove.get_proposal; call sites core.py:128, 450-457), the
// parameter rows the model packs from it, the particle weights on all of the model's
// grids and the single-row reductions (We/Wp) -- four dependent small launches as one.
// One block per proposed walker; everything a later stage needs from an earlier one
// stays in LDS.  See include/naima_hip.h for the slice protocol.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_step_front(front_args A) {
  extern __shared__ double sm[];
  double* qs = sm;                       // [ndim]    this walker's proposal
  double* row = qs + A.ndim;             // [8]       its particle-distribution row
  double* lg = row + NH_PD_NPAR;         // [3]       ln e_0, ln e_cutoff, ln e_break
  double* ws = lg + 3;                   // [mom_nodes] w of the grids that have a reduction
  double* ds = ws + A.mom_nodes;         // [mom_nodes] dlw
  double* mlx = ds + A.mom_nodes;        // [mom_nodes] lx of the grids that have a reduction
  double* mkt = mlx + A.mom_nodes;       // [sum over reductions of nG] K | dlnK
  const int j = blockIdx.x, tid = threadIdx.x;
  const pw_grids& G = A.G;
  const bool skip_nodes = G.off[G.n] == 0;
  // Everything below is a chain of dependent ~1 us trips to memory that other launches
  // wrote (cursor -> slice -> coordinates -> ... ).  What does NOT depend on the proposal
  // -- the grids' own arrays for the weight nodes, the tables of the single-row
  // reductions -- is requested first, by the waves that are not on that chain.
  double nE_[NH_MAX_GRIDS], nE2_[NH_MAX_GRIDS], ngx_[NH_MAX_GRIDS], nlr_[NH_MAX_GRIDS],
      nln_[NH_MAX_GRIDS];
#pragma unroll
  for (int g = 0; g < NH_MAX_GRIDS; ++g) {
    nE_[g] = nE2_[g] = ngx_[g] = 1.0;
    nlr_[g] = nln_[g] = 0.0;
  }
#define NH_FRONT_LOAD_NODES()                                                              \
  _Pragma("unroll") for (int g = 0; g < NH_MAX_GRIDS; ++g) {                                \
    if (g < G.n && !skip_nodes && tid < G.nG[g]) {                                         \
      const int nG = G.nG[g], i = tid;                                                     \
      const bool last = i + 1 >= nG;                                                       \
      nE_[g] = G.e[g][i];                                                                  \
      nE2_[g] = last ? nE_[g] : G.e[g][i + 1];                                             \
      ngx_[g] = G.xg[g][i];                                                                \
      if (!last) nlr_[g] = G.lx[g] ? G.lx[g][i] : log(G.xg[g][i + 1] / ngx_[g]);           \
      nln_[g] = G.lne[g] ? G.lne[g][i] : log(nE_[g]);                                      \
    }                                                                                      \
  }
  if (tid >= 64) {
    NH_FRONT_LOAD_NODES()
    int ko = 0;
    for (int m = 0; m < A.nmom; ++m) {
      const int g = A.mom[m].grid, nG = G.nG[g], o = A.mom_off[g];
      for (int i = tid - 64; i < nG; i += blockDim.x - 64) {
        mkt[ko + i] = A.mom[m].Kt[i];
        mkt[ko + nG + i] = A.mom[m].dlnKt[i];
        if (i < nG - 1)
          mlx[o + i] = G.lx[g] ? G.lx[g][i] : log(G.xg[g][i + 1] / G.xg[g][i]);
      }
      ko += 2 * nG;
    }
  }
  const int c = A.cursor[0];             // slice accepted last (-1: none yet)
  const int cn = c + 1;                  // slice proposed here
  const double* r = A.blk + (long long)cn * 3 * A.ns;
  const int* idx = reinterpret_cast<const int*>(r + 2 * A.ns);
  // ---- chain history of the ensemble step that the last accept closed -------------
  bool appended = false;
  if (A.hist && c >= 0 && (c & 1)) {
    const long long rowh = A.hist->n;
    if (A.hist->coords && rowh < A.hist->cap) {
      appended = true;
      const long long N = 2LL * A.ns, nc = N * A.ndim;
      double* hc = A.hist->coords + rowh * nc;
      double* hl = A.hist->logp + rowh * N;
      for (long long t = (long long)j * blockDim.x + tid; t < nc; t += (long long)gridDim.x * blockDim.x)
        hc[t] = A.coords[t];
      for (long long t = (long long)j * blockDim.x + tid; t < N; t += (long long)gridDim.x * blockDim.x)
        hl[t] = A.logp[t];
    }
  }
  // ---- proposal -------------------------------------------------------------------
  if (tid < A.ndim) {
    const int g = A.lo + j;
    const double z = r[g];
    const double cj = A.coords[(long long)idx[A.ns + g] * A.ndim + tid];
    const double sj = A.coords[(long long)idx[g] * A.ndim + tid];
    const double q = cj - (cj - sj) * z;
    A.qT[(long long)tid * A.nloc + j] = q;
    qs[tid] = q;
    if (tid == 0) A.factors[j] = (A.ndim - 1.0) * log(z);
  }
  if (tid < 64) {
    NH_FRONT_LOAD_NODES()
  }
#undef NH_FRONT_LOAD_NODES
  __syncthreads();
  // ---- parameter packs (columns read this walker's proposal or are constants) -----
  if (tid < A.npk * NH_MAX_LAZY) {
    const int q = tid / NH_MAX_LAZY, col = tid % NH_MAX_LAZY;
    if (col < A.pk[q].ncols) {
      const nh_lazy& z = A.pk[q].cols[col];
      double v = z.a;
      if (z.base) v = nh_lazy_apply(z, qs[(z.base - A.qT) / A.nloc]);
      A.pk[q].out[(long long)j * A.pk[q].ld + col] = v;
      if (A.pk[q].out == A.params) {
        row[col] = v;
        // ln e_0, ln e_cutoff, ln e_break by the threads that hold those columns
        if (col == 1 || col == 3 || col == 5) lg[col >> 1] = v > 0.0 ? log(v) : 0.0;
      }
    }
  }
  __syncthreads();
  // ---- particle weights on every grid ---------------------------------------------
  const pd_par p = {row[0], row[1], row[2], row[3], row[4], row[5], row[6]};
#pragma unroll
  for (int g = 0; g < NH_MAX_GRIDS; ++g) {
    if (g < G.n && !skip_nodes && tid < G.nG[g]) {
      const int nG = G.nG[g], i = tid;
      const bool last = i + 1 >= nG;
      double nn, dsh;
      pd_core(A.kind, p, nln_[g] - lg[0], nln_[g] - lg[1], lg[2] - lg[0], nE_[g] < p.eb,
              nE2_[g] < p.eb, nlr_[g], nn, dsh);
      nn *= G.scale[g];
      const double wv = ngx_[g] * nn, dv = last ? 0.0 : nlr_[g] + dsh;
      G.w[g][(long long)j * nG + i] = wv;
      G.dlw[g][(long long)j * nG + i] = dv;
      if (A.mom_off[g] >= 0) {
        ws[A.mom_off[g] + i] = wv;
        ds[A.mom_off[g] + i] = dv;
      }
    }
  }
  for (int g = 0; g < G.n && !skip_nodes; ++g) {  // grids longer than the workgroup
    const int nG = G.nG[g];
    const double* e = G.e[g];
    const double* xg = G.xg[g];
    for (int i = tid + blockDim.x; i < nG; i += blockDim.x) {
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
      const double wv = gx * nn, dv = last ? 0.0 : lr + dsh;
      G.w[g][(long long)j * nG + i] = wv;
      G.dlw[g][(long long)j * nG + i] = dv;
      if (A.mom_off[g] >= 0) {
        ws[A.mom_off[g] + i] = wv;
        ds[A.mom_off[g] + i] = dv;
      }
    }
  }
  // ---- single-row reductions over those weights (We, Wp), one wave each ------------
  if (A.nmom > 0) {
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63;
    if (wv < A.nmom) {
      const nh_moment& m = A.mom[wv];
      const int g = m.grid, nG = G.nG[g], o = A.mom_off[g];
      int ko = 0;
      for (int q = 0; q < wv; ++q) ko += 2 * G.nG[A.mom[q].grid];
      double acc = 0.0;
      for (int sgm = lane; sgm < nG - 1; sgm += 64) {
        const double u1 = ws[o + sgm] * mkt[ko + sgm];
        const double u2 = ws[o + sgm + 1] * mkt[ko + sgm + 1];
        const double dl = ds[o + sgm] + mkt[ko + nG + sgm];
        acc += nh_seg_term(u1, u2, dl, mlx[o + sgm]);
      }
      acc = wave_sum(acc);
      if (lane == 0) m.out[j] = acc;
    }
  }
  // ---- the last block to finish moves the cursor on ---------------------------------
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(A.done, 1) == (int)gridDim.x - 1) {
      *A.done = 0;
      A.cursor[0] = cn;
      if (appended) A.hist->n += 1;
    }
  }
}

extern "C" int nh_step_front(nh_ctx* c, const double* coords, const double* logp,
                             const double* blk, int* cursor, int* done, int ns, int ndim, int lo,
                             int nloc, double* qT, double* factors, const nh_pack* packs,
                             int npacks, int kind, const double* params, const nh_grid* grids,
                             int ngrids, const nh_moment* moms, int nmoms, nh_hist* hist) {
  NH_REQUIRE(c && coords && logp && blk && cursor && done && qT && factors && params && grids,
             "NULL pointer");
  NH_REQUIRE(ns >= 1 && ndim >= 1 && ndim <= 64 && lo >= 0 && nloc >= 1 && lo + nloc <= ns,
             "bad proposal block");
  NH_REQUIRE(npacks >= 1 && npacks <= NH_MAX_PACK && packs, "bad pack plan");
  NH_REQUIRE(kind >= NH_PD_POWERLAW && kind <= NH_PD_LOGPARABOLA, "unknown particle distribution kind");
  NH_REQUIRE(ngrids >= 1 && ngrids <= NH_MAX_GRIDS, "bad grid count");
  NH_REQUIRE(nmoms >= 0 && nmoms <= NH_MAX_MOMENT && (nmoms == 0 || moms), "bad reductions");
  front_args A;
  A.coords = coords; A.logp = logp; A.blk = blk; A.cursor = cursor; A.done = done;
  A.ns = ns; A.ndim = ndim; A.lo = lo; A.nloc = nloc; A.qT = qT; A.factors = factors;
  A.hist = hist; A.npk = npacks; A.kind = kind; A.params = params; A.nmom = nmoms;
  bool have_params = false;
  for (int q = 0; q < npacks; ++q) {
    NH_REQUIRE(packs[q].out && packs[q].ncols >= 1 && packs[q].ncols <= NH_MAX_LAZY &&
                   packs[q].ld >= packs[q].ncols, "bad pack request");
    for (int k = 0; k < packs[q].ncols; ++k) {
      const double* b = packs[q].cols[k].base;
      NH_REQUIRE(b == nullptr || (b >= qT && b < qT + (long long)ndim * nloc &&
                                  (b - qT) % nloc == 0 && packs[q].cols[k].stride == 1),
                 "a pack column must read one proposal coordinate (or be a constant)");
    }
    if (packs[q].out == params) {
      NH_REQUIRE(packs[q].ncols >= 7, "the particle rows need 7 columns");
      have_params = true;
    }
    A.pk[q] = packs[q];
  }
  NH_REQUIRE(have_params, "params must be the output of one of the packs");
  pw_grids& G = A.G;
  G.n = ngrids;
  G.off[0] = 0;
  for (int g = 0; g < ngrids; ++g) {
    NH_REQUIRE(grids[g].e_eV && grids[g].xg && grids[g].w && grids[g].dlw && grids[g].nG >= 2,
               "bad grid descriptor");
    G.e[g] = grids[g].e_eV; G.xg[g] = grids[g].xg; G.w[g] = grids[g].w; G.dlw[g] = grids[g].dlw;
    G.lne[g] = grids[g].ln_e; G.lx[g] = grids[g].lx;
    G.scale[g] = grids[g].unit_scale; G.nG[g] = grids[g].nG;
    NH_REQUIRE((long long)G.off[g] + grids[g].nG < (1LL << 30), "grids too long");
    G.off[g + 1] = G.off[g] + grids[g].nG;
    A.mom_off[g] = -1;
  }
  A.mom_nodes = 0;
  for (int m = 0; m < nmoms; ++m) {
    NH_REQUIRE(moms[m].grid >= 0 && moms[m].grid < ngrids && moms[m].Kt && moms[m].dlnKt &&
                   moms[m].out, "bad reduction");
    A.mom[m] = moms[m];
    if (A.mom_off[moms[m].grid] < 0) {
      A.mom_off[moms[m].grid] = A.mom_nodes;
      A.mom_nodes += grids[moms[m].grid].nG;
    }
  }
  size_t momk = 0;
  for (int m = 0; m < nmoms; ++m) momk += 2 * (size_t)grids[moms[m].grid].nG;
  const size_t lds = ((size_t)ndim + NH_PD_NPAR + 3 + 3 * (size_t)A.mom_nodes + momk) * sizeof(double);
  NH_REQUIRE(lds <= 60 * 1024, "reduction grids do not fit in LDS");
  nh_prof_scope ps(c, NH_K_PDIST);
  int threads = 1024;
  static const int ov_threads = nh_env_int("NH_FRONT_T", 0);
  if (ov_threads > 0) threads = ov_threads;
  hipLaunchKernelGGL(k_step_front, dim3((unsigned)nloc), dim3(threads), lds, c->stream, A);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_particle_weights(nh_ctx* c, int kind, const double* params, int N,
                                   const double* e_eV, const double* xg, int nG,
                                   double unit_scale, double* w, double* dlw, double* n_out) {
  NH_REQUIRE(c && params && e_eV && xg && w && dlw, "NULL pointer");
  NH_REQUIRE(kind >= NH_PD_POWERLAW && kind <= NH_PD_LOGPARABOLA, "unknown particle distribution kind");
  NH_REQUIRE(N >= 0 && nG >= 2, "need N >= 0 and nG >= 2");
  if (N == 0) return NH_OK;
  nh_prof_scope ps(c, NH_K_PDIST);
  long long tot = (long long)N * nG;
  hipLaunchKernelGGL(k_particle_weights, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     c->stream, kind, params, N, e_eV, xg, nG, unit_scale, w, dlw, n_out);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

__global__ void k_grid_logratio(const double* __restrict__ xg, int nG, double* __restrict__ lx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nG - 1) lx[i] = log(xg[i + 1] / xg[i]);
}

extern "C" int nh_grid_logratio(nh_ctx* c, const double* xg, int nG, double* lx) {
  NH_REQUIRE(c && xg && lx && nG >= 2, "bad argument");
  nh_prof_scope ps(c, NH_K_GLUE);
  hipLaunchKernelGGL(k_grid_logratio, dim3((nG + 255) / 256), dim3(256), 0, c->stream, xg, nG, lx);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

// ---------------------------------------------------------------------------
// the generic per-walker reduction.  Lanes run over flattened (walker, k)
// pairs -- coalesced reads of the transposed table rows Kt[i][*] -- and the C
// waves of a block split the abscissa range; each thread walks its chunk
// sequentially carrying the previous node in registers (no shuffles), and the
// C partial sums meet in LDS.
// ---------------------------------------------------------------------------
// The hot reduction.  One wave = one tile of 64 consecutive k (lanes) x W walkers
// (register-blocked) x one chunk of the abscissa.  Per segment the wave issues TWO
// coalesced vector loads (Kt row, dlnKt row) that serve all W walkers; the walker
// data (w, dlw) and lx are wave-uniform and come through the scalar cache
// (s_load), not through the vector L1 -- the first version issued 5 vector loads
// per segment and walker and was bound by the L1 tag-lookup rate (TCP), at 40 % of
// the VALU.  The C waves of a block split the abscissa; partial sums meet in LDS.
// STG: the block first copies its walkers' w and dlw rows into LDS with one coalesced
// sweep.  They were written by the launch just before this one, i.e. they sit in
// HBM / Infinity Cache, not in this XCD's L2: read through the scalar cache inside the
// loop, every new 64-byte line was a serialized ~1 us miss (measured: +3 us per launch
// inside the step loop against a warm micro-benchmark).
// EPI (one tile, nsplit = 1: the block holds its walkers' complete spectra): waves 0..W-1 go
// on to evaluate the likelihood (+ priors, + accept) of walker w0 + wave, nh_lnprob.h, as
// the synchrotron kernel does when it is the last producer.
template <int C, int W, bool SIGNED, bool STG, bool EPI = false>
__global__ __launch_bounds__(64 * C) void k_integrate_tables(
    const double* __restrict__ w, const double* __restrict__ dlw, int N, int nG,
    const double* __restrict__ lx, const double* __restrict__ Kt,
    const double* __restrict__ dlnKt, int nK, const double* __restrict__ scale,
    double* __restrict__ out, int ldo, int nsplit, nh_lnprob_args L, int loc_comp) {
  __shared__ double part[C][W][64];
  extern __shared__ double stg[];  // [W][nG] w | [W][nG] dlw   (STG only)
  const int lane = threadIdx.x & 63;
  const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ktiles = (nK + 63) >> 6;
  // blockIdx -> (split h, tile, walker group): h selects 1/nsplit of the abscissa and
  // its own output plane out + h*N*ldo (the consumer adds the planes)
  const int h = blockIdx.x % nsplit, bx = blockIdx.x / nsplit;
  const int tile = bx % ktiles, grp = bx / ktiles;
  const int k = tile * 64 + lane;
  const bool kvalid = k < nK;
  const unsigned kk = kvalid ? (unsigned)k : (unsigned)(nK - 1);
  const int w0 = grp * W;
  nh_lnprob_pre PRE = {};
  const bool epw = EPI && ch < W && w0 + ch < N;  // this wave runs walker w0 + ch's epilogue
  if (epw) nh_lnprob64_prefetch_a(PRE, L, w0 + ch, lane, loc_comp);
  const int nseg = nG - 1;
  const int hper = (nseg + nsplit - 1) / nsplit;
  const int hs0 = h * hper, hs1 = min(nseg, hs0 + hper);
  const int per = (max(hs1 - hs0, 0) + C - 1) / C;
  const int s0 = hs0 + ch * per;
  const int s1 = min(hs1, s0 + per);
  double acc[W], u1[W];
  unsigned row[W];  // wave-uniform row offsets of the W walkers (tail clamped)
#pragma unroll
  for (int j = 0; j < W; ++j) {
    acc[j] = 0.0;
    row[j] = (unsigned)min(w0 + j, N - 1) * (unsigned)nG;
  }
  if (STG) {
    for (int t = threadIdx.x; t < W * nG; t += 64 * C) {
      const int j = t / nG, i = t - j * nG;
      stg[t] = w[row[j] + i];
      stg[W * nG + t] = dlw[row[j] + i];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) row[j] = (unsigned)(j * nG);
    __syncthreads();
  }
  if (epw) nh_lnprob64_prefetch_b(PRE, L, w0 + ch);
  if (s0 < s1) {
    // table rows through buffer descriptors: base in SGPRs + one 32-bit byte offset per
    // lane (a single v_add per load instead of 64-bit address arithmetic)
    const unsigned tbytes = (unsigned)nG * (unsigned)nK * 8u;
    const __amdgpu_buffer_rsrc_t rK =
        __builtin_amdgcn_make_buffer_rsrc((void*)Kt, 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD =
        __builtin_amdgcn_make_buffer_rsrc((void*)dlnKt, 0, (int)tbytes, 0x00020000);
    const unsigned rowb = (unsigned)nK * 8u;
    unsigned ob = ((unsigned)s0 * (unsigned)nK + kk) * 8u;
    {
      const double K0 = nh_buf_f64(rK, ob);
#pragma unroll
      for (int j = 0; j < W; ++j) u1[j] = (STG ? stg[row[j] + s0] : w[row[j] + s0]) * K0;
    }
    // four segments per trip: their eight table loads are in flight together, so the
    // L2 latency is paid once per four segments (the loop is latency-, not issue-bound)
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
      double K2[4], dK[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        K2[q] = nh_buf_f64(rK, ob + (q + 1) * rowb);
        dK[q] = nh_buf_f64(rD, ob + q * rowb);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double lxs = lx[s + q];
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const double u2 = (STG ? stg[row[j] + s + q + 1] : w[row[j] + s + q + 1]) * K2[q];
          const double dl = (STG ? stg[W * nG + row[j] + s + q] : dlw[row[j] + s + q]) + dK[q];
          acc[j] += SIGNED ? nh_seg_term<true>(u1[j], u2, dl, lxs)
                           : nh_seg_pos<false>(u1[j], u2, dl, lxs);
          u1[j] = u2;
        }
      }
      ob += 4 * rowb;
    }
    for (; s < s1; ++s) {
      const double K2 = nh_buf_f64(rK, ob + rowb);
      const double dK = nh_buf_f64(rD, ob);
      const double lxs = lx[s];
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const double u2 = (STG ? stg[row[j] + s + 1] : w[row[j] + s + 1]) * K2;
        const double dl = (STG ? stg[W * nG + row[j] + s] : dlw[row[j] + s]) + dK;
        acc[j] += SIGNED ? nh_seg_term<true>(u1[j], u2, dl, lxs)
                         : nh_seg_pos<false>(u1[j], u2, dl, lxs);
        u1[j] = u2;
      }
      ob += rowb;
    }
  }
#pragma unroll
  for (int j = 0; j < W; ++j) part[ch][j][lane] = acc[j];
  if (epw) nh_lnprob64_prefetch_c(PRE, L, lane);
  __syncthreads();
  if (ch == 0) {
    const double sc = (scale && kvalid) ? scale[k] : 1.0;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (w0 + j < N) {
        double sum = 0.0;
#pragma unroll
        for (int c2 = 0; c2 < C; ++c2) sum += part[c2][j][lane];
        sum *= sc;
        if (kvalid) out[((long long)h * N + (w0 + j)) * ldo + k] = sum;
        if (EPI) part[0][j][lane] = sum;  // (each lane overwrites only what it has read)
      }
    }
  }
  if (EPI) {
    __syncthreads();
    if (epw) nh_lnprob64_finish(L, PRE, w0 + ch, lane, &part[0][ch][0], loc_comp);
  }
}

// few table rows (We, Wp: nK = 1): one wave per (walker, k), lanes over the segments
__global__ __launch_bounds__(256) void k_integrate_rows(
    const double* __restrict__ w, const double* __restrict__ dlw, int N, int nG,
    const double* __restrict__ lx, const double* __restrict__ Kt,
    const double* __restrict__ dlnKt, int nK, const double* __restrict__ scale,
    double* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const long long pair = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (long long)N * nK) return;
  const unsigned wi = (unsigned)(pair / nK), k = (unsigned)(pair % nK);
  const double* wr = w + (long long)wi * nG;
  const double* dwr = dlw + (long long)wi * nG;
  double acc = 0.0;
  for (int s = lane; s < nG - 1; s += 64) {
    double u1 = wr[s] * Kt[(long long)s * nK + k];
    double u2 = wr[s + 1] * Kt[(long long)(s + 1) * nK + k];
    double dl = dwr[s] + dlnKt[(long long)s * nK + k];
    acc += nh_seg_term(u1, u2, dl, lx[s]);
  }
  acc = wave_sum(acc);
  if (lane == 0) out[(long long)wi * ldo + k] = scale ? acc * scale[k] : acc;
}

// recommended nsplit: with W walkers per thread the launch has ktiles*ceil(N/W) workgroups
// of up to 16 waves; when that is between one and two per CU (256 CUs) half of the CUs
// carry twice the work of the others -- two half-range workgroups of 8 waves each spread
// evenly instead
extern "C" int nh_integrate_tables_nsplit(int N, int nG, int nK) {
  static const int ov_split = nh_env_int("NH_INT_SPLIT", 0);
  if (ov_split > 0) return ov_split;
  const long long ktiles = (nK + 63) / 64;
  if ((long long)N * nK * 4 < 4096 || nG < 66) return 1;
  const int W = (ktiles * N >= 8192) ? 4 : (ktiles * N >= 512 ? 2 : 1);
  const long long blocks = ktiles * ((N + W - 1) / W);
  return (blocks > 256 && blocks < 512) ? 2 : 1;
}

// L != NULL asks for the likelihood epilogue; *fused says whether this launch could carry it
// (one tile, one plane, staged rows, at most two walkers per thread) -- if not, the caller
// launches the likelihood itself
static int integrate_impl(nh_ctx* c, const double* w, const double* dlw, int N, int nG,
                          const double* lx, const double* Kt, const double* dlnKt, int nK,
                          const double* scale, double* out, int ldo, int nonnegative, int nsplit,
                          const nh_lnprob_args* L, int loc_comp, bool* fused) {
  if (fused) *fused = false;
  NH_REQUIRE(c && w && dlw && lx && Kt && dlnKt && out, "NULL pointer");
  NH_REQUIRE(N >= 0 && nG >= 2 && nK >= 1 && ldo >= nK, "bad sizes");
  NH_REQUIRE(nsplit >= 1 && nsplit <= 8, "nsplit must be 1..8");
  NH_REQUIRE((long long)N * nG < (1LL << 31) && (long long)nG * nK < (1LL << 28),
             "arrays too large for 32-bit offsets");
  if (N == 0) return NH_OK;
  long long pairs = (long long)N * nK;
  nh_prof_scope ps(c, pairs * 4 < 4096 ? NH_K_ROWS : NH_K_INTEGRATE);
  if (pairs * 4 < 4096) {  // too few (walker, k) pairs to fill the chip with pair-lanes
    hipLaunchKernelGGL(k_integrate_rows, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0,
                       c->stream, w, dlw, N, nG, lx, Kt, dlnKt, nK, scale, out, ldo);
    NH_CHECK_HIP(hipGetLastError());
    if (nsplit > 1)  // everything is in plane 0
      NH_CHECK_HIP(hipMemsetAsync(out + (long long)N * ldo, 0,
                                  (size_t)(nsplit - 1) * N * ldo * sizeof(double), c->stream));
    return NH_OK;
  }
  const int nseg = nG - 1;
  const int ktiles = (nK + 63) / 64;
  // walkers per thread (register blocking: the table rows are loaded once for W
  // walkers).  Measured inside the cfg3 step loop (N = 256, nK = 192, C = 16): W = 1
  // 20.8 us, W = 2 17.0 us, W = 4 18.1 us -- halving the L2->L1 table traffic pays as
  // long as the launch keeps >= ~4 waves per SIMD.
  int W = ((long long)ktiles * N >= 8192) ? 4 : ((long long)ktiles * N >= 512 ? 2 : 1);
  static const int ov_W = nh_env_int("NH_INT_W", 0);
  if (ov_W > 0) W = ov_W;
  const unsigned blocks = (unsigned)(ktiles * ((N + W - 1) / W) * nsplit);
  // split the abscissa so that the launch has >= ~4 waves per SIMD (1024 SIMDs)
  // (split launches use 8-wave workgroups: four fit on a CU, so 768 of them spread evenly)
  int C = 1;
  const int Cmax = nsplit > 1 ? 8 : 16;
  while (C < Cmax && (long long)blocks * C < 12288 && nseg / nsplit / (2 * C) >= 8) C *= 2;
  static const int ov_C = nh_env_int("NH_INT_C", 0);
  if (ov_C > 0) C = ov_C;
  const size_t stg_bytes = 2 * (size_t)W * nG * sizeof(double);
  const bool stage = stg_bytes <= 48 * 1024;
  nh_lnprob_args none = {};
  const nh_lnprob_args& LA = L ? *L : none;
  const bool epi = L && stage && nsplit == 1 && ktiles == 1 && W <= 2 && (C == 16 || C == 8);
  if (fused) *fused = epi;
#define NH_LAUNCH_INT_T(CC, WW, SS, TT)                                                      \
  hipLaunchKernelGGL((k_integrate_tables<CC, WW, SS, TT>), dim3(blocks), dim3(64 * CC),      \
                     TT ? stg_bytes : 0, c->stream, w, dlw, N, nG, lx, Kt, dlnKt, nK, scale, \
                     out, ldo, nsplit, LA, loc_comp)
#define NH_LAUNCH_INT_E(CC, WW, SS)                                                          \
  hipLaunchKernelGGL((k_integrate_tables<CC, WW, SS, true, true>), dim3(blocks),             \
                     dim3(64 * CC), stg_bytes, c->stream, w, dlw, N, nG, lx, Kt, dlnKt, nK,  \
                     scale, out, ldo, nsplit, LA, loc_comp)
  if (epi) {
    if (C == 16) {
      if (W == 2) { if (nonnegative) NH_LAUNCH_INT_E(16, 2, false); else NH_LAUNCH_INT_E(16, 2, true); }
      else { if (nonnegative) NH_LAUNCH_INT_E(16, 1, false); else NH_LAUNCH_INT_E(16, 1, true); }
    } else {
      if (W == 2) { if (nonnegative) NH_LAUNCH_INT_E(8, 2, false); else NH_LAUNCH_INT_E(8, 2, true); }
      else { if (nonnegative) NH_LAUNCH_INT_E(8, 1, false); else NH_LAUNCH_INT_E(8, 1, true); }
    }
    NH_CHECK_HIP(hipGetLastError());
    return NH_OK;
  }
#define NH_LAUNCH_INT_S(CC, WW, SS) \
  do { if (stage) NH_LAUNCH_INT_T(CC, WW, SS, true); else NH_LAUNCH_INT_T(CC, WW, SS, false); } while (0)
#define NH_LAUNCH_INT(CC, WW) \
  do { if (nonnegative) NH_LAUNCH_INT_S(CC, WW, false); else NH_LAUNCH_INT_S(CC, WW, true); } while (0)
#define NH_LAUNCH_INT_C(CC) \
  do { if (W == 4) NH_LAUNCH_INT(CC, 4); else if (W == 2) NH_LAUNCH_INT(CC, 2); else NH_LAUNCH_INT(CC, 1); } while (0)
  switch (C) {
    case 16: NH_LAUNCH_INT_C(16); break;
    case 8: NH_LAUNCH_INT_C(8); break;
    case 4: NH_LAUNCH_INT_C(4); break;
    case 2: NH_LAUNCH_INT_C(2); break;
    default: NH_LAUNCH_INT_C(1); break;
  }
#undef NH_LAUNCH_INT_C
#undef NH_LAUNCH_INT
#undef NH_LAUNCH_INT_S
#undef NH_LAUNCH_INT_T
#undef NH_LAUNCH_INT_E
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_integrate_tables(nh_ctx* c, const double* w, const double* dlw, int N, int nG,
                                   const double* lx, const double* Kt, const double* dlnKt,
                                   int nK, const double* scale, double* out, int ldo,
                                   int nonnegative, int nsplit) {
  return integrate_impl(c, w, dlw, N, nG, lx, Kt, dlnKt, nK, scale, out, ldo, nonnegative,
                        nsplit, nullptr, -1, nullptr);
}

// ---------------------------------------------------------------------------
// row 11: lnprobmodel (core.py:64-94), one wave per walker
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lnprobmodel(nh_lnprob_args A) {
  const int lane = threadIdx.x & 63;
  const int wi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wi >= A.N) return;
  nh_lnprob_wave(A, wi, lane, nullptr, -1);
}

static int launch_lnprob(nh_ctx* c, const nh_comps& cs, int N, int nE, const double* conv,
                         const double* flux, const double* err_lo, const double* err_hi,
                         const int* ul, const double* cl, const double* lp,
                         const nh_prior* terms, int nterms, double* model_out, double* lnl,
                         const nh_accept* mv = nullptr) {
  nh_lnprob_args A;
  nh_lnprob_fill(A, cs, N, nE, conv, flux, err_lo, err_hi, ul, cl, lp, terms, nterms, model_out,
                 lnl, mv);
  A.nan_count = c->nan_word;
  nh_prof_scope ps(c, NH_K_LNPROB);
  hipLaunchKernelGGL(k_lnprobmodel, dim3((N + 3) / 4), dim3(256), 0, c->stream, A);
  NH_CHECK_HIP(hipGetLastError());
  return NH_OK;
}

extern "C" int nh_lnprobmodel(nh_ctx* c, const double* const* comps, const double* cscale,
                              int ncomp, int ldc, int N, int nE, const double* conv,
                              const double* flux, const double* err_lo, const double* err_hi,
                              const int* ul, const double* cl, double* model_out, double* lnl) {
  NH_REQUIRE(c && comps && cscale && conv && flux && err_lo && err_hi && ul && cl && lnl,
             "NULL pointer");
  NH_REQUIRE(ncomp >= 1 && ncomp <= NH_MAX_COMP, "ncomp must be 1..8");
  NH_REQUIRE(N >= 0 && nE >= 1 && ldc >= nE, "bad sizes");
  if (N == 0) return NH_OK;
  nh_comps cs;
  cs.n = ncomp;
  for (int j = 0; j < ncomp; ++j) cs.c[j] = {comps[j], ldc, cscale[j]};
  return launch_lnprob(c, cs, N, nE, conv, flux, err_lo, err_hi, ul, cl, nullptr, nullptr, 0,
                       model_out, lnl);
}

extern "C" int nh_lnprob(nh_ctx* c, const nh_comp* comps, int ncomp, int N, int nE,
                         const double* conv, const double* flux, const double* err_lo,
                         const double* err_hi, const int* ul, const double* cl, const double* lp,
                         const nh_prior* terms, int nterms, double* model_out, double* total) {
  NH_REQUIRE(c && comps && conv && flux && err_lo && err_hi && ul && cl && total, "NULL pointer");
  NH_REQUIRE(nterms >= 0 && nterms <= NH_MAX_PRIOR && (nterms == 0 || terms), "bad prior terms");
  NH_REQUIRE(ncomp >= 1 && ncomp <= NH_MAX_COMP, "ncomp must be 1..8");
  NH_REQUIRE(N >= 0 && nE >= 1, "bad sizes");
  if (N == 0) return NH_OK;
  nh_comps cs;
  cs.n = ncomp;
  for (int j = 0; j < ncomp; ++j) {
    NH_REQUIRE(comps[j].ptr && comps[j].ld >= nE, "bad component");
    cs.c[j] = comps[j];
  }
  return launch_lnprob(c, cs, N, nE, conv, flux, err_lo, err_hi, ul, cl, lp, terms, nterms,
                       model_out, total);
}

extern "C" int nh_lnprob_accept(nh_ctx* c, const nh_comp* comps, int ncomp, int N, int nE,
                                const double* conv, const double* flux, const double* err_lo,
                                const double* err_hi, const int* ul, const double* cl,
                                const double* lp, const nh_prior* terms, int nterms,
                                double* model_out, double* total, const nh_accept* mv) {
  NH_REQUIRE(c && comps && conv && flux && err_lo && err_hi && ul && cl && total && mv,
             "NULL pointer");
  NH_REQUIRE(mv->coords && mv->logp && mv->blk && mv->cursor && mv->accepted, "bad accept block");
  NH_REQUIRE(mv->ns >= 1 && mv->ndim >= 1 && mv->ndim <= 64 && mv->lo >= 0 && mv->lo + N <= mv->ns,
             "bad accept sizes");
  NH_REQUIRE(nterms >= 0 && nterms <= NH_MAX_PRIOR && (nterms == 0 || terms), "bad prior terms");
  NH_REQUIRE(ncomp >= 1 && ncomp <= NH_MAX_COMP, "ncomp must be 1..8");
  NH_REQUIRE(N >= 0 && nE >= 1, "bad sizes");
  if (N == 0) return NH_OK;
  nh_comps cs;
  cs.n = ncomp;
  for (int j = 0; j < ncomp; ++j) {
    NH_REQUIRE(comps[j].ptr && comps[j].ld >= nE, "bad component");
    cs.c[j] = comps[j];
  }
  return launch_lnprob(c, cs, N, nE, conv, flux, err_lo, err_hi, ul, cl, lp, terms, nterms,
                       model_out, total, mv);
}

extern "C" int nh_integrate_tables_lnprob(
    nh_ctx* c, const double* w, const double* dlw, int N, int nG, const double* lx,
    const double* Kt, const double* dlnKt, int nK, const double* scale, double* out, int ldo,
    int nonnegative, const nh_comp* comps, int ncomp, int loc_comp, const double* conv,
    const double* flux, const double* err_lo, const double* err_hi, const int* ul,
    const double* cl, const double* lp, const nh_prior* terms, int nterms, double* total,
    const nh_accept* mv) {
  NH_REQUIRE(c && comps && conv && flux && err_lo && err_hi && ul && cl && total, "NULL pointer");
  NH_REQUIRE(ncomp >= 1 && ncomp <= NH_MAX_COMP && loc_comp >= 0 && loc_comp < ncomp,
             "bad components");
  NH_REQUIRE(nterms >= 0 && nterms <= NH_MAX_PRIOR && (nterms == 0 || terms), "bad prior terms");
  NH_REQUIRE(comps[loc_comp].ptr == out && comps[loc_comp].ld == ldo,
             "component loc_comp must be this launch's output");
  if (mv) {
    NH_REQUIRE(mv->coords && mv->logp && mv->blk && mv->cursor && mv->accepted &&
                   mv->ns >= 1 && mv->ndim >= 1 && mv->ndim <= 64 && mv->lo >= 0 &&
                   mv->lo + N <= mv->ns, "bad accept block");
  }
  nh_comps cs;
  cs.n = ncomp;
  for (int j = 0; j < ncomp; ++j) {
    NH_REQUIRE(comps[j].ptr && comps[j].ld >= nK, "bad component");
    cs.c[j] = comps[j];
  }
  nh_lnprob_args A;
  nh_lnprob_fill(A, cs, N, nK, conv, flux, err_lo, err_hi, ul, cl, lp, terms, nterms, nullptr,
                 total, mv);
  A.nan_count = c->nan_word;
  bool fused = false;
  int rc = integrate_impl(c, w, dlw, N, nG, lx, Kt, dlnKt, nK, scale, out, ldo, nonnegative, 1,
                          nK <= 64 ? &A : nullptr, loc_comp, &fused);
  if (rc || fused || N == 0) return rc;
  // this shape cannot carry the epilogue (several tiles, very long grids ...): two launches
  return launch_lnprob(c, cs, N, nK, conv, flux, err_lo, err_hi, ul, cl, lp, terms, nterms,
                       nullptr, total, mv);
}

// ---------------------------------------------------------------------------
// pinned host staging: lets the host run ahead of the device (the random block of
// half-step h+1 is drawn and shipped while the graph of half-step h executes)
// ---------------------------------------------------------------------------
extern "C" int nh_host_alloc(nh_ctx* c, long long bytes, void** out) {
  NH_REQUIRE(c && out && bytes > 0, "bad argument");
  NH_CHECK_HIP(hipSetDevice(c->device));
  NH_CHECK_HIP(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
  return NH_OK;
}

extern "C" int nh_host_free(nh_ctx* c, void* p) {
  NH_REQUIRE(c, "ctx is NULL");
  if (p) NH_CHECK_HIP(hipHostFree(p));
  return NH_OK;
}

extern "C" int nh_marker_create(nh_ctx* c, void** out) {
  NH_REQUIRE(c && out, "bad argument");
  hipEvent_t e;
  NH_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *out = e;
  return NH_OK;
}

extern "C" int nh_marker_record(nh_ctx* c, void* m) {
  NH_REQUIRE(c && m, "bad argument");
  NH_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(m), c->stream));
  return NH_OK;
}

extern "C" int nh_marker_wait(nh_ctx* c, void* m) {  // host blocks until the marker is reached
  NH_REQUIRE(c && m, "bad argument");
  NH_CHECK_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(m)));
  return NH_OK;
}

extern "C" int nh_marker_destroy(nh_ctx* c, void* m) {
  NH_REQUIRE(c, "ctx is NULL");
  if (m) NH_CHECK_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(m)));
  return NH_OK;
}
