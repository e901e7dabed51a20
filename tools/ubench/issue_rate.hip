// Microbenchmark: per-wave issue cost of fp64 / 32-bit VALU streams at 1, 2, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

template <int ILP, int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a, double b) {
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (MODE == 0) x[i] = __builtin_fma(x[i], a, b);             // v_fma_f64
        if (MODE == 1) x[i] = x[i] + a;                              // v_add_f64
        if (MODE == 2) x[i] = x[i] * a;                              // v_mul_f64
        if (MODE == 3) { float f = (float)x[i]; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"((float)a), "v"((float)b)); x[i] = f; }
        if (MODE == 4) { int lo = __double2loint(x[i]); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(lo)); x[i] = __hiloint2double(__double2hiint(x[i]), lo); }
        if (MODE == 5) x[i] = __builtin_fmax(x[i], a);               // v_max_f64
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP, int MODE>
int run(const char* name, double* d_out) {
  const int iters = 20000;
  for (int blocks : {1024, 2048, 4096, 8192}) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<ILP, MODE>), dim3(blocks), dim3(64), 0, 0, d_out, 100, 1.0000001, 1e-9);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<ILP, MODE>), dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double inst_per_wave = (double)iters * 8 * ILP;
    double waves_per_simd = blocks / 1024.0;
    double ns_per_inst_per_simd = ms * 1e6 / (inst_per_wave * waves_per_simd);
    printf("%-12s ILP=%d waves/SIMD=%.0f  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.1f cyc @2.4GHz)\n", name, ILP, waves_per_simd, ms, ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4);
  }
  return 0;
}

int main() {
  double* d_out; CK(hipMalloc(&d_out, 8192 * 64 * 8));
  run<1, 0>("fma_f64", d_out); run<2, 0>("fma_f64", d_out); run<4, 0>("fma_f64", d_out); run<8, 0>("fma_f64", d_out);
  run<8, 1>("add_f64", d_out); run<8, 2>("mul_f64", d_out); run<8, 5>("max_f64", d_out);
  run<1, 3>("fma_f32", d_out); run<8, 3>("fma_f32", d_out);
  run<8, 4>("cndmask", d_out);
  return 0;
}
