// pool_stats.hip -- pooled Baum-Welch statistics of a batch in a FIXED summation order (BASELINE.json config 5).
//
// Reference: runTraining accumulates, per read, (w, s1, s2) per k-mer over the lattice columns in ascending order
// (src/cpp/NT_aligner_api.cpp:505-535); a pooled estimator adds those per-read sums read after read. This file computes
// exactly that association on the device:
//     pooled[code] = (((0 + P(read r0)) + P(read r1)) + ...),   P(read) = ((0 + c_a) + c_b) + ...  (columns ascending)
// over the ok reads in INPUT order -- bit for bit the sum dyn_batch_fetch_train forms on the host (finalise_train), and the
// same bits on every run. Rounds 1-3 added the columns with fp64 atomics, whose order is whatever the scheduler makes it:
// the all-reduced statistics of config 5, and the model file written from them, differed in the last digits run to run.
//
// Steps (all on the batch's compute stream):
//   k_pool_keys     one thread per lattice column: key = its k-mer code (columns of failed reads: a sentinel that sorts
//                   last), value = its flat index; col_read = the read it belongs to
//   rocprim radix sort of (key, index) pairs -- LSD radix sort is stable, so equal codes stay in flat-index order, which
//                   is input-read order, columns ascending
//   k_pool_gather   the columns' (w, s1, s2) and read numbers into sorted order (the sums below then read contiguous memory)
//   k_pool_segments one thread per k-mer that occurs: walks its run, closes a per-read partial sum whenever the read
//                   changes, adds the partials in order. 2 M columns spread over <= 4^k codes: runs are ~8 long; the
//                   one long run of an RNA batch (the polyA k-mer of every read's pad, a few thousand entries) costs
//                   its thread ~0.2 ms of sequential, cache-friendly loads.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "nt_kernels.hpp"

namespace dynk {

namespace {

__global__ void k_pool_keys(const ReadDesc* __restrict__ descs, const ReadState* __restrict__ st, const int32_t* __restrict__ kmers,
                            uint32_t sentinel, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx, uint32_t* __restrict__ col_read) {
  const ReadDesc rd = descs[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;  // column index - 1
  if (c >= (int)rd.N - 1) return;
  const uint64_t i = rd.par_off + c;
  keys[i] = st[rd.read].status == 0 ? (uint32_t)kmers[i] : sentinel;
  idx[i] = (uint32_t)i;
  col_read[i] = rd.read;
}

__global__ void k_pool_fill(uint32_t* __restrict__ keys, uint32_t* __restrict__ idx, uint32_t sentinel, uint64_t total) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) {
    keys[i] = sentinel;  // columns of reads that never reached the device (host-side failures) stay out of the sums
    idx[i] = (uint32_t)i;
  }
}

__global__ void k_pool_gather(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ col_read,
                              TrainBuffers tb, uint32_t sentinel, uint64_t total, double* __restrict__ sw, double* __restrict__ s1,
                              double* __restrict__ s2, uint32_t* __restrict__ sread) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= total || keys[j] == sentinel) return;
  const uint32_t i = idx[j];
  sw[j] = tb.col_w[i];
  s1[j] = tb.col_s1[i];
  s2[j] = tb.col_s2[i];
  sread[j] = col_read[i];
}

__global__ void k_pool_segments(const uint32_t* __restrict__ keys, const double* __restrict__ sw, const double* __restrict__ s1,
                                const double* __restrict__ s2, const uint32_t* __restrict__ sread, uint32_t sentinel, uint64_t total,
                                double* __restrict__ pooled, uint64_t num_kmers) {
  const uint64_t j0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j0 >= total) return;
  const uint32_t code = keys[j0];
  if (code == sentinel || (j0 > 0 && keys[j0 - 1] == code)) return;  // not the first entry of a k-mer's run
  double tw = 0.0, t1 = 0.0, t2 = 0.0;   // pooled sums: start from the zero the host's array starts from
  double pw = 0.0, p1 = 0.0, p2 = 0.0;   // the current read's partial sums
  uint32_t cur = sread[j0];
  for (uint64_t j = j0; j < total && keys[j] == code; ++j) {
    const uint32_t r = sread[j];
    if (r != cur) {
      tw += pw;
      t1 += p1;
      t2 += p2;
      pw = p1 = p2 = 0.0;
      cur = r;
    }
    pw += sw[j];
    p1 += s1[j];
    p2 += s2[j];
  }
  tw += pw;
  t1 += p1;
  t2 += p2;
  pooled[code] = tw;
  pooled[num_kmers + code] = t1;
  pooled[2 * num_kmers + code] = t2;
}

int key_bits(uint64_t num_kmers) {  // the sentinel num_kmers itself must be representable
  int b = 1;
  while ((1ull << b) <= num_kmers) ++b;
  return b;
}

}  // namespace

size_t pool_stats_temp_bytes(uint64_t total_cols, uint64_t num_kmers) {
  size_t bytes = 0;
  uint32_t* nul = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)total_cols, 0, (unsigned)key_bits(num_kmers), (hipStream_t) nullptr);
  return bytes;
}

size_t pool_stats_work_bytes(uint64_t total_cols) {
  // keys in/out, idx in/out, col_read, sread: 6 x u32; sw, s1, s2: 3 x f64 -- each region 8-byte aligned
  return (size_t)(6 * ((total_cols * 4 + 7) / 8 * 8) + 3 * total_cols * 8);
}

hipError_t launch_pool_stats(const ReadDesc* descs, int n_reads, uint32_t max_N, const ReadState* st, const int32_t* kmers,
                             TrainBuffers tb, double* pooled, uint64_t num_kmers, uint64_t total_cols, void* work, void* temp,
                             size_t temp_bytes, hipStream_t s) {
  if (!total_cols || n_reads <= 0) return hipSuccess;
  const size_t u32r = (total_cols * 4 + 7) / 8 * 8;
  char* p = static_cast<char*>(work);
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(p);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(p + u32r);
  uint32_t* idx_in = reinterpret_cast<uint32_t*>(p + 2 * u32r);
  uint32_t* idx_out = reinterpret_cast<uint32_t*>(p + 3 * u32r);
  uint32_t* col_read = reinterpret_cast<uint32_t*>(p + 4 * u32r);
  uint32_t* sread = reinterpret_cast<uint32_t*>(p + 5 * u32r);
  double* sw = reinterpret_cast<double*>(p + 6 * u32r);
  double* s1 = sw + total_cols;
  double* s2 = s1 + total_cols;
  const uint32_t sentinel = (uint32_t)num_kmers;
  const unsigned blocks = (unsigned)((total_cols + 255) / 256);
  hipLaunchKernelGGL(k_pool_fill, dim3(blocks), dim3(256), 0, s, keys_in, idx_in, sentinel, total_cols);
  constexpr int MAX_GRID_Y = 65535;
  for (int r0 = 0; r0 < n_reads; r0 += MAX_GRID_Y) {
    const int nr = std::min(MAX_GRID_Y, n_reads - r0);
    hipLaunchKernelGGL(k_pool_keys, dim3((max_N + 255) / 256, nr), dim3(256), 0, s, descs + r0, st, kmers, sentinel, keys_in, idx_in, col_read);
  }
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, (size_t)total_cols, 0,
                                           (unsigned)key_bits(num_kmers), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_pool_gather, dim3(blocks), dim3(256), 0, s, keys_out, idx_out, col_read, tb, sentinel, total_cols, sw, s1, s2, sread);
  hipLaunchKernelGGL(k_pool_segments, dim3(blocks), dim3(256), 0, s, keys_out, sw, s1, s2, sread, sentinel, total_cols, pooled, num_kmers);
  return hipGetLastError();
}

}  // namespace dynk
