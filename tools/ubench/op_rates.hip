// Microbenchmark: issue cost of individual gfx950 instructions from ONE wave per SIMD with 8
// independent chains (the regime of the DP kernels). Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

#define OP8(ASM) \
  asm volatile(ASM : "+v"(x0) : "v"(a)); asm volatile(ASM : "+v"(x1) : "v"(a)); \
  asm volatile(ASM : "+v"(x2) : "v"(a)); asm volatile(ASM : "+v"(x3) : "v"(a)); \
  asm volatile(ASM : "+v"(x4) : "v"(a)); asm volatile(ASM : "+v"(x5) : "v"(a)); \
  asm volatile(ASM : "+v"(x6) : "v"(a)); asm volatile(ASM : "+v"(x7) : "v"(a));

#define KERNEL64(NAME, ASM)                                                          \
  __global__ __launch_bounds__(64) void NAME(double* out, int iters, double a) {     \
    double x0 = threadIdx.x + 1.5, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int it = 0; it < iters; ++it) { OP8(ASM) OP8(ASM) OP8(ASM) OP8(ASM) }       \
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;      \
  }
#define KERNEL32(NAME, ASM)                                                          \
  __global__ __launch_bounds__(64) void NAME(double* out, int iters, double ad) {    \
    int a = (int)ad + threadIdx.x;                                                   \
    int x0 = threadIdx.x + 1, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int it = 0; it < iters; ++it) { OP8(ASM) OP8(ASM) OP8(ASM) OP8(ASM) }       \
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;      \
  }

KERNEL64(k_fma, "v_fma_f64 %0, %0, %1, %1")
KERNEL64(k_add, "v_add_f64 %0, %0, %1")
KERNEL64(k_mul, "v_mul_f64 %0, %0, %1")
KERNEL64(k_max, "v_max_f64 %0, %0, %1")
KERNEL64(k_rcp, "v_rcp_f64 %0, %0")
KERNEL64(k_rndne, "v_rndne_f64 %0, %0")
KERNEL64(k_ldexp, "v_ldexp_f64 %0, %0, 1")
KERNEL64(k_mov64, "v_mov_b64 %0, %1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_bfi, "v_bfi_b32 %0, %1, %0, %1")
KERNEL32(k_add32, "v_add_u32 %0, %0, %1")
KERNEL32(k_mov32, "v_mov_b32 %0, %1")
KERNEL32(k_dpp_ror, "v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_dpp_rowshr, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_rcpf32, "v_rcp_f32 %0, %0")

template <class K> int run(const char* name, K kern, double* d_out) {
  const int iters = 5000;
  for (int blocks : {1024, 2048}) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, 50, 1.0000001);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0000001);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double inst = (double)iters * 32 * (blocks / 1024.0);
    printf("%-14s waves/SIMD=%d  %.3f ms -> %.2f ns/instr/SIMD (%.1f cyc @2.3GHz)\n", name, blocks / 1024, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.3);
  }
  return 0;
}

int main() {
  double* d; CK(hipMalloc(&d, 2048 * 64 * 8));
  run("v_fma_f64", k_fma, d); run("v_add_f64", k_add, d); run("v_mul_f64", k_mul, d); run("v_max_f64", k_max, d);
  run("v_rcp_f64", k_rcp, d); run("v_rndne_f64", k_rndne, d); run("v_ldexp_f64", k_ldexp, d);
  run("v_cndmask_b32", k_cndmask, d); run("v_bfi_b32", k_bfi, d); run("v_add_u32", k_add32, d); run("v_mov_b32", k_mov32, d);
  run("dpp wave_ror", k_dpp_ror, d); run("dpp row_shr", k_dpp_rowshr, d); run("v_rcp_f32", k_rcpf32, d);
  return 0;
}
