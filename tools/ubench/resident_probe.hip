// GPU box: can a RESIDENT read-queue kernel (one 4-wave workgroup per CU, 150 KB of LDS, 360 registers per lane, one wave
// per SIMD -- the footprint of k_read_queue<JOB_ALIGN, true>) be fed while it runs?
//   1. auxiliary kernels of another stream (parameter tables, preprocessing, medians: <= 112 VGPRs, no LDS to speak of)
//      must be SCHEDULED beside it -- the resident kernel waits for what they produce, so if they queued behind it the
//      design would deadlock;
//   2. control words in device memory written by a one-lane kernel of another stream (agent-scope atomics) must become
//      visible to the resident waves' sc1 polls, and payload written by such kernels / by an H2D copy must be read
//      fresh after an agent-scope acquire -- with the SAME addresses rewritten ticket after ticket;
//   3. which stream kinds keep the resident kernel out of the auxiliary streams' hardware queue (plain streams, more
//      streams than hardware queues, a CU-masked stream).
// Every wait is bounded (the resident kernel leaves after `limit_s` whatever happens).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

constexpr int CTL_TAIL = 0, CTL_CLOSED = 1, CTL_ARRIVED = 2, CTL_BAD = 3, CTL_SEEN = 4, CTL_TIMEOUT = 5;
constexpr int PAYLOAD = 4096;  // doubles per ticket, the same buffer every ticket

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the footprint of the real kernel: 150 544 B of LDS, 256 VGPRs + 104 AGPRs, one wave per SIMD
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_resident(uint32_t* ctl, const double* __restrict__ payload,
                                                                                               double* sink, unsigned long long limit_ticks) {
  __shared__ double lds[150544 / 8];
  lds[threadIdx.x] = 1.0;
  asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a103, 0" ::: "v255", "a103");
  const int lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&ctl[CTL_ARRIVED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t seen = 0;
  double acc = 0.0;
  for (;;) {
    const uint32_t tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_agent(&ctl[CTL_TAIL]));
    if (tail != seen) {
      // a new "ticket": its payload (written by a kernel of another stream, or by an H2D copy, into the SAME buffer as every
      // ticket before) must read as value == tail everywhere
      __builtin_amdgcn_s_dcache_inv();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      int bad = 0;
      for (int i = lane; i < PAYLOAD; i += 64) bad += payload[i] != (double)tail;
      // and one wave-uniform (scalar) load of it
      const double u = payload[__builtin_amdgcn_readfirstlane((int)(tail % PAYLOAD))];
      bad += u != (double)tail;
      if (bad) __hip_atomic_fetch_add(&ctl[CTL_BAD], (uint32_t)bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) __hip_atomic_fetch_add(&ctl[CTL_SEEN], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += u;
      seen = tail;
      continue;
    }
    if (__builtin_amdgcn_readfirstlane((int)ld_agent(&ctl[CTL_CLOSED]))) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) {
      if (lane == 0) __hip_atomic_store(&ctl[CTL_TIMEOUT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(32);
  }
  if (acc == 12345.678) sink[0] = acc + lds[lane];
}

__global__ void k_publish(uint32_t* ctl, uint32_t tail, uint32_t closed) {
  if (threadIdx.x == 0) {
    if (tail) __hip_atomic_store(&ctl[CTL_TAIL], tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (closed) __hip_atomic_store(&ctl[CTL_CLOSED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// an auxiliary kernel of the preprocessing kind: ~100 VGPRs, 1 024 blocks
__global__ void k_fill(double* __restrict__ p, int n, double v) {
  asm volatile("v_mov_b32 v107, 0" ::: "v107");  // k_hampel<double> allocates 108
  double r[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) r[k] = v + (double)k * 0.0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 48; ++k) s += r[k];
  if (i < n) p[i] = s / 48.0;
}

// an auxiliary kernel of RCCL's footprint: 40 KB of LDS, 248 VGPRs -- fits no CU a resident workgroup occupies
__global__ __launch_bounds__(256) void k_big(double* __restrict__ p, int n) {
  __shared__ double lds[40 * 1024 / 8];
  asm volatile("v_mov_b32 v247, 0" ::: "v247");
  lds[threadIdx.x] = (double)threadIdx.x;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = lds[(threadIdx.x + 1) & 255];
}

// the same with SCRATCH (a dynamically indexed private array), as RCCL's and rocPRIM's kernels have
__global__ __launch_bounds__(256) void k_scratch(double* __restrict__ p, int n, int rot) {
  double a[96];
  for (int k = 0; k < 96; ++k) a[k] = (double)(k * rot);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double sacc = 0.0;
  for (int k = 0; k < 8; ++k) sacc += a[(i * 7 + k * rot) % 96];   // dynamic index: the array lives in scratch
  if (i < n) p[i] = sacc;
}

// small register footprint, KB kilobytes of LDS: where is the line between 'fits beside a resident workgroup' (9.5 KB are
// free on its CU) and 'needs a CU of its own'?
template <int KB>
__global__ __launch_bounds__(256) void k_lds(double* __restrict__ p, int n) {
  __shared__ double lds[KB * 1024 / 8];
  lds[threadIdx.x] = (double)threadIdx.x;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = lds[(threadIdx.x + 1) & 255];
}

// rocPRIM's radix sort shape: 1 024-thread workgroups, 20 KB of LDS, scratch
__global__ __launch_bounds__(1024) void k_wide(double* __restrict__ p, int n, int rot) {
  __shared__ double lds[20 * 1024 / 8];
  double a[40];
  for (int k = 0; k < 40; ++k) a[k] = (double)(k * rot);
  lds[threadIdx.x] = a[(threadIdx.x * rot) % 40];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = lds[(threadIdx.x + 1) & 1023];
}

// waits (bounded) until every resident wave has seen ticket `tail`: the k_wait_ticket of the design
__global__ void k_wait_seen(uint32_t* ctl, uint32_t want, unsigned long long limit_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (ld_agent(&ctl[CTL_SEEN]) < want) {
    if (__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) {
      __hip_atomic_store(&ctl[CTL_TIMEOUT], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(64);
  }
}

static int run(const char* name, int stream_kind, int extra_streams, bool payload_by_copy) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cus = prop.multiProcessorCount;
  hipStream_t s_res = nullptr, s_in = nullptr, s_out = nullptr;
  if (stream_kind == 1) {
    std::vector<uint32_t> mask((n_cus + 31) / 32, 0xffffffffu);
    CK(hipExtStreamCreateWithCUMask(&s_res, (uint32_t)mask.size(), mask.data()));
  } else {
    CK(hipStreamCreateWithFlags(&s_res, hipStreamNonBlocking));
  }
  std::vector<hipStream_t> extra(extra_streams);
  for (auto& s : extra) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
  uint32_t* ctl = nullptr;
  double *payload = nullptr, *sink = nullptr, *h_payload = nullptr;
  CK(hipMalloc(&ctl, 256));
  CK(hipMalloc(&payload, PAYLOAD * 8));
  CK(hipMalloc(&sink, 8));
  CK(hipHostMalloc(&h_payload, PAYLOAD * 8, hipHostMallocDefault));
  CK(hipMemset(ctl, 0, 256));
  CK(hipMemset(payload, 0, PAYLOAD * 8));
  CK(hipDeviceSynchronize());
  const double limit_s = 4.0;
  const unsigned long long limit_ticks = (unsigned long long)(limit_s * 1e8);  // s_memrealtime: 100 MHz (s_memtime is the shader clock, ~2.2 GHz)
  const double t0 = now();
  hipLaunchKernelGGL(k_resident, dim3(n_cus), dim3(256), 0, s_res, ctl, payload, sink, limit_ticks);
  CK(hipGetLastError());
  // keep the extra streams busy with something short, so that they own hardware queues
  for (auto& s : extra) hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, s, sink + 0, 0, 0.0);
  const int tickets = 50;
  double worst_pub = 0, sum_pub = 0, worst_rt = 0, sum_rt = 0;
  int done_tickets = 0;
  for (int t = 1; t <= tickets; ++t) {
    const double a = now();
    if (payload_by_copy) {
      for (int i = 0; i < PAYLOAD; ++i) h_payload[i] = (double)t;
      CK(hipMemcpyAsync(payload, h_payload, PAYLOAD * 8, hipMemcpyHostToDevice, s_in));
    } else {
      hipLaunchKernelGGL(k_fill, dim3((PAYLOAD + 255) / 256), dim3(256), 0, s_in, payload, PAYLOAD, (double)t);
    }
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s_in, ctl, (uint32_t)t, 0u);
    CK(hipStreamSynchronize(s_in));
    const double b = now();
    hipLaunchKernelGGL(k_wait_seen, dim3(1), dim3(1), 0, s_out, ctl, (uint32_t)(t * n_cus * 4), limit_ticks / 4);
    CK(hipStreamSynchronize(s_out));
    const double c = now();
    worst_pub = std::max(worst_pub, b - a);
    sum_pub += b - a;
    worst_rt = std::max(worst_rt, c - a);
    sum_rt += c - a;
    ++done_tickets;
    if (c - t0 > limit_s - 0.5) break;  // the auxiliary kernels are queuing behind the resident one: stop feeding
  }
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s_in, ctl, 0u, 1u);
  CK(hipStreamSynchronize(s_in));
  CK(hipStreamSynchronize(s_res));
  const double t_end = now() - t0;
  uint32_t h[8];
  CK(hipMemcpy(h, ctl, sizeof h, hipMemcpyDeviceToHost));
  printf("%-44s tickets %2d/%d  publish mean %7.1f us worst %8.1f us | publish+all %d waves seen it: mean %7.1f us worst %8.1f us | resident kernel %.3f s, "
         "arrived %u, seen %u (want %u), STALE %u, timeout %u\n",
         name, done_tickets, tickets, 1e6 * sum_pub / done_tickets, 1e6 * worst_pub, n_cus * 4, 1e6 * sum_rt / done_tickets, 1e6 * worst_rt, t_end,
         h[CTL_ARRIVED], h[CTL_SEEN], (unsigned)(done_tickets * n_cus * 4), h[CTL_BAD], h[CTL_TIMEOUT]);
  fflush(stdout);
  const int ok = h[CTL_TIMEOUT] == 0 && h[CTL_BAD] == 0 && h[CTL_SEEN] == (unsigned)(done_tickets * n_cus * 4) && done_tickets == tickets;
  CK(hipFree(ctl));
  CK(hipFree(payload));
  CK(hipFree(sink));
  CK(hipHostFree(h_payload));
  for (auto s : extra) CK(hipStreamDestroy(s));
  CK(hipStreamDestroy(s_res));
  CK(hipStreamDestroy(s_in));
  CK(hipStreamDestroy(s_out));
  return ok;
}

// Kernels that need a WHOLE free CU (and scratch) beside a resident kernel that occupies n_cus - free_cus CUs: are they served,
// and how fast? (RCCL's exchange kernels at N > 1; rocPRIM's radix sort behind a training ticket.)
// partition: 0 = the resident grid merely leaves `free_cus` CUs unused (any of them); 1 = the CUs are PARTITIONED by CU masks:
// the resident stream may use the first n - free_cus CUs only, the auxiliary stream the last free_cus only; 2 = the same with
// the free CUs spread (every (n / free_cus)-th CU)
static void run_big(const char* name, int free_cus, int kind, int partition = 0) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cus = prop.multiProcessorCount;
  hipStream_t s_res = nullptr, s_aux = nullptr;
  std::vector<uint32_t> mask((n_cus + 31) / 32, 0xffffffffu), aux_mask((n_cus + 31) / 32, 0u);
  if (partition) {
    for (int k = 0; k < free_cus; ++k) {
      const int cu = partition == 1 ? n_cus - 1 - k : k * (n_cus / free_cus);
      mask[cu / 32] &= ~(1u << (cu % 32));
      aux_mask[cu / 32] |= 1u << (cu % 32);
    }
  }
  CK(hipExtStreamCreateWithCUMask(&s_res, (uint32_t)mask.size(), mask.data()));
  if (partition) CK(hipExtStreamCreateWithCUMask(&s_aux, (uint32_t)aux_mask.size(), aux_mask.data()));
  else CK(hipStreamCreateWithFlags(&s_aux, hipStreamNonBlocking));
  uint32_t* ctl = nullptr;
  double *payload = nullptr, *sink = nullptr, *buf = nullptr;
  CK(hipMalloc(&ctl, 256));
  CK(hipMalloc(&payload, PAYLOAD * 8));
  CK(hipMalloc(&sink, 8));
  CK(hipMalloc(&buf, 1 << 22));
  CK(hipMemset(ctl, 0, 256));
  CK(hipDeviceSynchronize());
  const double limit_s = 2.0;
  const double t0 = now();
  hipLaunchKernelGGL(k_resident, dim3(n_cus - free_cus), dim3(256), 0, s_res, ctl, payload, sink, (unsigned long long)(limit_s * 1e8));
  CK(hipGetLastError());
  double worst = 0, sum = 0, first = 0;
  int done = 0;
  for (int it = 0; it < 20; ++it) {
    const double a = now();
    if (kind == 0) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s_aux, buf, 64 * 256);
    else if (kind == 1) hipLaunchKernelGGL(k_scratch, dim3(64), dim3(256), 0, s_aux, buf, 64 * 256, it + 1);
    else if (kind == 2) hipLaunchKernelGGL(k_wide, dim3(64), dim3(1024), 0, s_aux, buf, 64 * 1024, it + 1);
    else if (kind == 3) hipLaunchKernelGGL(k_lds<8>, dim3(64), dim3(256), 0, s_aux, buf, 64 * 256);
    else if (kind == 4) hipLaunchKernelGGL(k_lds<16>, dim3(64), dim3(256), 0, s_aux, buf, 64 * 256);
    else if (kind == 5) hipLaunchKernelGGL(k_lds<16>, dim3(1), dim3(256), 0, s_aux, buf, 256);
    else hipLaunchKernelGGL(k_big, dim3(kind - 100), dim3(256), 0, s_aux, buf, (kind - 100) * 256);  // kind 100 + n: n workgroups of RCCL's footprint
    CK(hipStreamSynchronize(s_aux));
    const double dt = now() - a;
    if (!it) first = dt;
    else {
      worst = std::max(worst, dt);
      sum += dt;
    }
    ++done;
    if (now() - t0 > limit_s - 0.5) break;
  }
  uint32_t arrived = 0;
  CK(hipMemcpyAsync(&arrived, ctl + CTL_ARRIVED, 4, hipMemcpyDeviceToHost, s_aux));
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s_aux, ctl, 0u, 1u);
  CK(hipStreamSynchronize(s_aux));
  CK(hipStreamSynchronize(s_res));
  printf("[%3u resident workgroups running] ", arrived);
  printf("%-44s %2d/20 launches beside a resident kernel on %d of %d CUs: first %9.1f us, then mean %8.1f us worst %9.1f us%s\n", name, done, n_cus - free_cus, n_cus,
         1e6 * first, done > 1 ? 1e6 * sum / (done - 1) : 0.0, 1e6 * worst, done < 20 ? "   <-- WAITED for the resident kernel to leave" : "");
  fflush(stdout);
  CK(hipFree(ctl));
  CK(hipFree(payload));
  CK(hipFree(sink));
  CK(hipFree(buf));
  CK(hipStreamDestroy(s_res));
  CK(hipStreamDestroy(s_aux));
}

// where does bit i of a CU mask point? One workgroup on a stream whose mask has only that bit reports its XCC and HW_ID
__global__ void k_where(uint32_t* out) {
  if (threadIdx.x == 0) {
    out[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    out[1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
  }
}

static void run_map() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cus = prop.multiProcessorCount;
  uint32_t* out = nullptr;
  CK(hipHostMalloc(&out, 64, hipHostMallocDefault));
  const int bits[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 24, 31, 32, 33, 40, 63, 64, 96, 128, 160, 192, 224, 247, 248, 249, 250, 251, 252, 253, 254, 255};
  for (int bit : bits) {
    if (bit >= n_cus) continue;
    std::vector<uint32_t> mask((n_cus + 31) / 32, 0u);
    mask[bit / 32] = 1u << (bit % 32);
    hipStream_t s = nullptr;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    out[0] = out[1] = 0xffffffffu;
    hipLaunchKernelGGL(k_where, dim3(1), dim3(64), 0, s, out);
    CK(hipStreamSynchronize(s));
    const uint32_t hw = out[1];
    printf("mask bit %3d -> XCC %u, SE %u, SH %u, CU %2u   (XCC_ID %08x HW_ID %08x)\n", bit, out[0] & 0xf, (hw >> 13) & 0x7, (hw >> 12) & 1, (hw >> 8) & 0xf, out[0], hw);
  }
  fflush(stdout);
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  if (argc > 1 && !strcmp(argv[1], "map")) {
    run_map();
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "big")) {
    run_big("40 KB LDS + 248 VGPRs, no CU free", 0, 0);
    run_big("40 KB LDS + 248 VGPRs, 8 CUs free", 8, 0);
    run_big("scratch kernel, no CU free", 0, 1);
    run_big("scratch kernel, 8 CUs free", 8, 1);
    run_big("1024-thread workgroups + scratch, no CU free", 0, 2);
    run_big("1024-thread workgroups + scratch, 8 CUs free", 8, 2);
    run_big("8 KB LDS, 64 workgroups, no CU free", 0, 3);
    run_big("16 KB LDS, 64 workgroups, no CU free", 0, 4);
    run_big("16 KB LDS, 64 workgroups, 8 CUs free", 8, 4);
    run_big("16 KB LDS, 64 workgroups, 64 CUs free", 64, 4);
    run_big("16 KB LDS, 64 workgroups, 128 CUs free", 128, 4);
    run_big("16 KB LDS, ONE workgroup, 8 CUs free", 8, 5);
    run_big("16 KB LDS, ONE workgroup, 128 CUs free", 128, 5);
    run_big("40 KB LDS + 248 VGPRs, 128 CUs free", 128, 0);
    run_big("40 KB+248 VGPRs x1, 8 CUs free", 8, 101);
    run_big("40 KB+248 VGPRs x8, 8 CUs free", 8, 108);
    run_big("40 KB+248 VGPRs x16, 8 CUs free", 8, 116);
    run_big("40 KB+248 VGPRs x32, 8 CUs free", 8, 132);
    run_big("40 KB+248 VGPRs x8, 16 CUs free", 16, 108);
    run_big("40 KB+248 VGPRs x16, 16 CUs free", 16, 116);
    run_big("40 KB+248 VGPRs x32, 32 CUs free", 32, 132);
    run_big("40 KB+248 VGPRs x64, 8 CUs by CU MASK (last 8)", 8, 0, 1);
    run_big("40 KB+248 VGPRs x64, 8 CUs by CU MASK (spread)", 8, 0, 2);
    run_big("1024-thread x64, 8 CUs by CU MASK (last 8)", 8, 2, 1);
    run_big("1024-thread x64, 8 CUs by CU MASK (spread)", 8, 2, 2);
    run_big("16 KB LDS x64, 4 CUs by CU MASK (spread)", 4, 4, 2);
    return 0;
  }
  int ok = 1;
  ok &= run("plain streams, payload by kernel", 0, 0, false);
  ok &= run("plain streams, payload by H2D copy", 0, 0, true);
  ok &= run("plain streams + 6 busy extra streams, kernel", 0, 6, false);
  ok &= run("CU-masked resident stream + 6 extra, kernel", 1, 6, false);
  ok &= run("CU-masked resident stream + 6 extra, copy", 1, 6, true);
  printf(ok ? "RESULT: resident waves can be fed\n" : "RESULT: some configuration FAILED\n");
  return ok ? 0 : 1;
}
