// Validates the LDS-DMA ring used by the forward sweep of k_read_queue: rows of 448 doubles are streamed HBM -> LDS with
// global_load_lds_dwordx4 (3 full + 1 half-wave instruction per 3584-byte row), D rows deep per
// wave, consumed with ds_read_b64 behind a hand-counted s_waitcnt vmcnt(N).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_test lds_dma_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
constexpr int P = 448, D = 4, ROWB = P * 8;

__device__ __forceinline__ void dma_row(const double* row_lane_ptr /* row + lane*2 doubles */, unsigned lds_slot) {
  // offset: applies to BOTH the global address and the LDS address
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off\n\t"
      "global_load_lds_dwordx4 %0, off offset:1024\n\t"
      "global_load_lds_dwordx4 %0, off offset:2048\n\t"
      "s_mov_b32 exec_hi, 0\n\t"
      "global_load_lds_dwordx4 %0, off offset:3072\n\t"
      "s_mov_b32 exec_hi, -1\n\t"
      :: "v"(row_lane_ptr), "s"(__builtin_amdgcn_readfirstlane(lds_slot)) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ void read_row(unsigned lds_lane_addr, double (&b)[7]) {
  asm volatile(
      "ds_read_b64 %0, %7\n\t"
      "ds_read_b64 %1, %7 offset:512\n\t"
      "ds_read_b64 %2, %7 offset:1024\n\t"
      "ds_read_b64 %3, %7 offset:1536\n\t"
      "ds_read_b64 %4, %7 offset:2048\n\t"
      "ds_read_b64 %5, %7 offset:2560\n\t"
      "ds_read_b64 %6, %7 offset:3072\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6])
      : "v"(lds_lane_addr) : "memory");
}

__global__ __launch_bounds__(256) void k(const double* __restrict__ in, float2* __restrict__ outst, double* __restrict__ sums, int T) {
  __shared__ __attribute__((aligned(16))) double ring[4][D][P];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + w;
  const double* base = in + (size_t)r * T * P;
  float2* ost = outst + (size_t)r * T * P + lane;
  const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)&ring[w][0][0];
  const unsigned lane_addr = ring_base + lane * 8;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int rr = 0; rr < D && rr < T; ++rr) dma_row(base + (size_t)rr * P + lane * 2, ring_base + (rr % D) * ROWB);
  for (int t = 0; t < T; ++t) {
    const bool steady = (t >= D) && (t + D < T);
    if (steady) wait_vm<WAITN>(); else wait_vm<0>();
    double b[7];
    read_row(lane_addr + (t % D) * ROWB, b);
    if (t + D < T) dma_row(base + (size_t)(t + D) * P + lane * 2, ring_base + (t % D) * ROWB);
#pragma unroll
    for (int j = 0; j < 7; ++j) { acc[j] += b[j] * (1.0 + j); ost[(size_t)t * P + j * 64] = make_float2((float)b[j], (float)acc[j]); }
  }
  double s = 0;
  for (int j = 0; j < 7; ++j) s += acc[j];
  sums[r * 64 + lane] = s;
}

int main() {
  const int R = 1024, T = 2000;
  const size_t n = (size_t)R * T * P;
  std::vector<double> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) % 1000003) * 1e-3;
  double *d_in, *d_s; float2* d_o;
  CK(hipMalloc(&d_in, n * 8)); CK(hipMalloc(&d_o, n * 8)); CK(hipMalloc(&d_s, R * 64 * 8));
  CK(hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(R / 4), dim3(256), 0, 0, d_in, d_o, d_s, T);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(R / 4), dim3(256), 0, 0, d_in, d_o, d_s, T);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<double> s(R * 64);
  CK(hipMemcpy(s.data(), d_s, R * 64 * 8, hipMemcpyDeviceToHost));
  long bad = 0;
  for (int r = 0; r < R; r += 37)
    for (int lane = 0; lane < 64; ++lane) {
      double acc[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int t = 0; t < T; ++t) for (int j = 0; j < 7; ++j) acc[j] += h[((size_t)r * T + t) * P + j * 64 + lane] * (1.0 + j);
      double want = 0; for (int j = 0; j < 7; ++j) want += acc[j];
      if (want != s[r * 64 + lane]) { if (bad < 8) printf("r %d lane %d want %.17g got %.17g\n", r, lane, want, s[r*64+lane]); ++bad; }
    }
  printf("lds-dma ring: %s (%ld mismatches), %.3f ms, %.1f GB/s read + same written\n", bad ? "FAIL" : "OK", bad, ms, n * 8 / ms / 1e6);
  return bad != 0;
}
