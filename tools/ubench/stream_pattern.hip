// The HBM traffic of k_read_queue WITHOUT its arithmetic: what the access pattern alone allows.
// 1 024 waves (one per SIMD, 256 workgroups x 4 waves), each with a lattice region of its own:
//   B  "backward": per row 3 x 16 B/lane + 1 x 8 B/lane non-temporal stores (3 584 B), rows descending
//   F  "forward" : per row the same 3 584 B through the 4-deep LDS-DMA ring (global_load_lds_dwordx4), rows ascending,
//                  + 1 792 B of float stores (3 x 8 B/lane + 4 B/lane) + 56 B of decision bits
//   M  mix       : 2 of every 5 waves run B, the others F (the time shares of the two sweeps in the real launch)
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_pattern stream_pattern.hip ;  ./stream_pattern [rows_per_wave=20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
constexpr int P = 448, D = 4, ROWB = P * 8;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dma_row(const double* row_lane_ptr, unsigned lds_slot) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off nt\n\t"
      "global_load_lds_dwordx4 %0, off offset:1024 nt\n\t"
      "global_load_lds_dwordx4 %0, off offset:2048 nt\n\t"
      "s_mov_b32 exec_hi, 0\n\t"
      "global_load_lds_dwordx4 %0, off offset:3072 nt\n\t"
      "s_mov_b32 exec_hi, -1\n\t"
      :: "v"(row_lane_ptr), "s"(__builtin_amdgcn_readfirstlane(lds_slot)) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ void store_row(double* row, int lane, double v) {
  f4 q; q.x = q.y = q.z = q.w = (float)v;
  f2 h; h.x = h.y = (float)v;
#pragma unroll
  for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(q, reinterpret_cast<f4*>(row + k * 128 + lane * 2));
  __builtin_nontemporal_store(h, reinterpret_cast<f2*>(row + 384 + lane));
}

// mode: 0 = B on every wave, 1 = F on every wave, 2 = mix, 3 = F with the LPE / bits stores of two rows issued together
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k(double* ws, float* lpe, unsigned long long* bits, int rows, int mode, double* sink) {
  __shared__ __attribute__((aligned(16))) double ring[4][D][P];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const size_t slot = (size_t)blockIdx.x * 4 + wave;
  double* my_ws = ws + slot * (size_t)rows * P;
  float* my_lp = lpe + slot * (size_t)rows * P;
  unsigned long long* my_bits = bits + slot * (size_t)rows * 7;
  const bool backward = mode == 0 || (mode == 2 && slot % 5 < 2);  // (mode 3: forward on every wave)
  double acc = 0.0;
  if (backward) {
    for (int t = rows - 1; t >= 0; --t) store_row(my_ws + (size_t)t * P, lane, (double)t);
  } else {
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)&ring[wave][0][0];
    const double* src = my_ws + lane * 2;
    for (int r = 0; r < D; ++r) dma_row(src + (size_t)(r < rows ? r : rows - 1) * P, base + (r % D) * ROWB);
    for (int t = 0; t < rows; ++t) {
      wait_vm<4 * (D - 1)>();
      acc += ring[wave][t % D][lane];  // touch the row (one ds_read; the real sweep reads all of it)
      const int nx = t + D < rows ? t + D : rows - 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dma_row(src + (size_t)nx * P, base + (t % D) * ROWB);
      if (mode != 3 || (t & 1)) {
        for (int tt = (mode == 3 ? t - 1 : t); tt <= t; ++tt) {
          float* o = my_lp + (size_t)tt * P;
          f2 h; h.x = h.y = (float)tt;
#pragma unroll
          for (int q = 0; q < 3; ++q) __builtin_nontemporal_store(h, reinterpret_cast<f2*>(o + q * 128 + lane * 2));
          __builtin_nontemporal_store((float)tt, o + 384 + lane);
        }
        if (mode == 3) { if (lane < 14) my_bits[(size_t)(t - 1) * 7 + lane] = (unsigned long long)t; }
        else if (lane < 7) my_bits[(size_t)t * 7 + lane] = (unsigned long long)t;
      }
    }
    wait_vm<0>();
  }
  if (acc == 12345.678) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 20000;
  const size_t slots = 1024, cells = slots * (size_t)rows * P;
  double *ws, *sink; float* lpe; unsigned long long* bits;
  CK(hipMalloc(&ws, cells * 8)); CK(hipMalloc(&lpe, cells * 4)); CK(hipMalloc(&bits, slots * (size_t)rows * 56)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(ws, 0, cells * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"B (backward: 3584 B/row written)", "F (forward: 3584 B/row read by LDS-DMA + 1848 B/row written)", "M (2 of 5 waves B, 3 of 5 F)",
                          "F2 (forward, the float / bit stores of two rows issued together)"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, ws, lpe, bits, rows, mode, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double nb = mode == 0 ? 1.0 : (mode == 1 || mode == 3) ? 0.0 : 410.0 / 1024.0;  // 1024 slots: slot % 5 < 2 -> 410 of them
      const double bytes = (double)slots * rows * (nb * ROWB + (1.0 - nb) * (ROWB + P * 4 + 56));
      if (rep) printf("%-64s rep %d  %.3f ms  %.2f TB/s\n", names[mode], rep, ms, bytes / ms * 1e-9);
    }
  }
  return 0;
}
