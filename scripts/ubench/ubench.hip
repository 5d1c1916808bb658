// instruction-throughput microbenchmarks for gfx950 (FP64 VALU ops used by the hot loops)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1);} } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3;
  double a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
  const double c = 1.0000001, d = 1e-9;
  for (int i = 0; i < iters; ++i) {
#define REP(x) x(a0) x(a1) x(a2) x(a3) x(a4) x(a5) x(a6) x(a7)
    if (OP == 0) {
#define F(v) v = fma(v, c, d);
      REP(F)
#undef F
    } else if (OP == 1) {
#define F(v) v = __builtin_amdgcn_rcp(v);
      REP(F)
#undef F
    } else if (OP == 2) {
#define F(v) v = v * c;
      REP(F)
#undef F
    } else if (OP == 3) {
#define F(v) v = v + d;
      REP(F)
#undef F
    } else if (OP == 4) {
#define F(v) v = __builtin_amdgcn_rsq(v);
      REP(F)
#undef F
    } else if (OP == 5) {
#define F(v) v = (double)__builtin_amdgcn_rcpf((float)v);
      REP(F)
#undef F
    } else if (OP == 6) {
#define F(v) v = (v > 1.5) ? c : v + d;
      REP(F)
#undef F
    } else if (OP == 7) {
#define F(v) v = sqrt(v);
      REP(F)
#undef F
    } else if (OP == 8) {
#define F(v) v = __builtin_amdgcn_ldexp(v, 1);
      REP(F)
#undef F
    } else if (OP == 9) {
#define F(v) v = __builtin_amdgcn_fract(v) + c;
      REP(F)
#undef F
    } else if (OP == 10) {
#define F(v) v = __builtin_amdgcn_trig_preop(v, 1);
      REP(F)
#undef F
    } else if (OP == 11) {
#define F(v) v = (double)(float)v;
      REP(F)
#undef F
    } else if (OP == 12) {
#define F(v) { float f_ = (float)v; f_ = __builtin_amdgcn_rcpf(f_); v = v + (double)f_; }
      REP(F)
#undef F
    } else if (OP == 13) {
#define F(v) { float f_ = (float)v; f_ = __builtin_amdgcn_rcpf(f_ + 1.0f); f_ = f_ * 1.5f + 0.5f; v = (double)f_; }
      REP(F)
#undef F
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char* name, int nops_per_rep) {
  const int blocks = 256 * 8, iters = 2000;
  double* out;
  CHECK(hipMalloc(&out, blocks * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.25);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.25);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * iters * 8 * nops
  double winstr = (double)blocks * 4 / 1024.0 * iters * 8 * nops_per_rep;
  double cyc = ms * 1e-3 * 2.4e9 / winstr;
  printf("%-28s %8.3f ms  -> %6.2f cycles(@2.4GHz)/wave-instr-group (%d ops)\n", name, ms, cyc, nops_per_rep);
  CHECK(hipFree(out));
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("%s clock %d kHz CUs %d\n", p.name, p.clockRate, p.multiProcessorCount);
  run<0>("v_fma_f64", 1);
  run<2>("v_mul_f64", 1);
  run<3>("v_add_f64", 1);
  run<1>("v_rcp_f64", 1);
  run<4>("v_rsq_f64", 1);
  run<7>("sqrt(double) ocml", 1);
  run<5>("cvt+rcp_f32+cvt", 3);
  run<11>("cvt f64->f32->f64", 2);
  run<6>("cmp+cndmask x2 + add", 1);
  run<8>("v_ldexp_f64", 1);
  run<9>("v_fract_f64 + add", 2);
  run<12>("cvt,rcp_f32,cvt,add_f64", 4);
  run<13>("cvt,addf,rcpf,fmaf,cvt", 5);
  return 0;
}
