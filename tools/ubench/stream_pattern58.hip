// stream_pattern.hip with PACKED rows: would 58 active lanes x 7 cells = 406 band slots (for the 401 cells a row holds)
// instead of 64 x 7 = 448 pay on the memory side? A/B of the two row shapes in one binary, rows per second is the measure:
//   L = 64: rows of 448 slots, runs of 1 024 B (3 x 16 B per lane) + 512 B, row stride 3 584 B (bE) / 1 792 B (float LPE)
//   L = 58: rows of 406 slots, runs of   928 B + 464 B, row stride 3 264 B (51 lines) / 1 664 B (26 lines); lanes 58..63 off
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_pattern58 stream_pattern58.hip ; ./stream_pattern58 [rows_per_wave=20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
constexpr int D = 4;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int L> struct Shape {
  static constexpr int RUN = L * 2;            // doubles per pair run
  static constexpr int SLOTS = L * 7;
  static constexpr int STRIDE = (SLOTS * 8 + 63) / 64 * 8;       // doubles per stored row (whole cache lines)
  static constexpr int LSTRIDE = (SLOTS * 4 + 63) / 64 * 16;     // floats per stored LPE row
};

template <int L>
__device__ __forceinline__ void dma_row(const double* row_lane_ptr, unsigned lds_slot) {
  // three pair runs by L lanes, the single run by L/2 lanes (16 B each)
  constexpr unsigned RUNB = L * 16;
  constexpr unsigned long long FULL = L == 64 ? ~0ull : ((1ull << L) - 1), HALF = (1ull << (L / 2)) - 1;
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off nt\n\t"
      "global_load_lds_dwordx4 %0, off offset:%2 nt\n\t"
      "global_load_lds_dwordx4 %0, off offset:%3 nt\n\t"
      "s_mov_b64 exec, %5\n\t"
      "global_load_lds_dwordx4 %0, off offset:%4 nt\n\t"
      "s_mov_b64 exec, %6\n\t"
      :: "v"(row_lane_ptr), "s"(__builtin_amdgcn_readfirstlane(lds_slot)), "n"(RUNB), "n"(2 * RUNB), "n"(3 * RUNB), "s"(HALF), "s"(FULL) : "memory");
}

template <int L>
__device__ __forceinline__ void store_row(double* row, int lane, double v) {
  f4 q; q.x = q.y = q.z = q.w = (float)v;
  f2 h; h.x = h.y = (float)v;
#pragma unroll
  for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(q, reinterpret_cast<f4*>(row + k * Shape<L>::RUN + lane * 2));
  __builtin_nontemporal_store(h, reinterpret_cast<f2*>(row + 3 * Shape<L>::RUN + lane));
}

template <int L>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k(double* ws, float* lpe, unsigned long long* bits, int rows, int mode, double* sink) {
  constexpr int S = Shape<L>::STRIDE, LS = Shape<L>::LSTRIDE, RUN = Shape<L>::RUN;
  __shared__ __attribute__((aligned(16))) double ring[4][D][448];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const size_t slot = (size_t)blockIdx.x * 4 + wave;
  double* my_ws = ws + slot * (size_t)rows * S;
  float* my_lp = lpe + slot * (size_t)rows * LS;
  unsigned long long* my_bits = bits + slot * (size_t)rows * 7;
  const bool backward = mode == 0 || (mode == 2 && slot % 5 < 2);
  double acc = 0.0;
  if (lane < L) {
    if (backward) {
      for (int t = rows - 1; t >= 0; --t) store_row<L>(my_ws + (size_t)t * S, lane, (double)t);
    } else {
      const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)&ring[wave][0][0];
      const double* src = my_ws + lane * 2;
      for (int r = 0; r < D; ++r) dma_row<L>(src + (size_t)(r < rows ? r : rows - 1) * S, base + (r % D) * 3584);
      for (int t = 0; t < rows; ++t) {
        wait_vm<4 * (D - 1)>();
        acc += ring[wave][t % D][lane];
        const int nx = t + D < rows ? t + D : rows - 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_row<L>(src + (size_t)nx * S, base + (t % D) * 3584);
        float* o = my_lp + (size_t)t * LS;
        f2 h; h.x = h.y = (float)t;
#pragma unroll
        for (int q = 0; q < 3; ++q) __builtin_nontemporal_store(h, reinterpret_cast<f2*>(o + q * RUN + lane * 2));
        __builtin_nontemporal_store((float)t, o + 3 * RUN + lane);
        if (lane < 7) my_bits[(size_t)t * 7 + lane] = (unsigned long long)t;
      }
      wait_vm<0>();
    }
  }
  if (acc == 12345.678) sink[0] = acc;
}

template <int L>
int run(double* ws, float* lpe, unsigned long long* bits, double* sink, int rows) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[3] = {"B backward (write bE rows)", "F forward (read bE rows by LDS-DMA, write LPE + bits)", "M 2 of 5 waves B, 3 of 5 F"};
  for (int mode = 0; mode < 3; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k<L>, dim3(256), dim3(256), 0, 0, ws, lpe, bits, rows, mode, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double nb = mode == 0 ? 1.0 : mode == 1 ? 0.0 : 410.0 / 1024.0;
      const double rowb = Shape<L>::SLOTS * 8.0, bytes = 1024.0 * rows * (nb * rowb + (1.0 - nb) * (rowb + Shape<L>::SLOTS * 4 + 56));
      if (rep) printf("L=%d  %-56s rep %d  %8.3f ms  %6.1f Mrows/s  %.2f TB/s of useful bytes\n", L, names[mode], rep, ms, 1024.0 * rows / ms * 1e-3, bytes / ms * 1e-9);
    }
  return 0;
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 20000;
  const size_t slots = 1024;
  double *ws, *sink; float* lpe; unsigned long long* bits;
  CK(hipMalloc(&ws, slots * (size_t)rows * 448 * 8)); CK(hipMalloc(&lpe, slots * (size_t)rows * 448 * 4));
  CK(hipMalloc(&bits, slots * (size_t)rows * 56)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(ws, 0, slots * (size_t)rows * 448 * 8));
  if (run<64>(ws, lpe, bits, sink, rows)) return 1;
  if (run<58>(ws, lpe, bits, sink, rows)) return 1;
  if (run<64>(ws, lpe, bits, sink, rows)) return 1;
  if (run<58>(ws, lpe, bits, sink, rows)) return 1;
  return 0;
}
