// GPU box: can a lattice pool be MAPPED progressively while a resident kernel already works in the part that is there?
// (cold start of a session: hipMalloc of the 127 GB pool takes 1-5 s before the first read runs.)
//   1. hipMalloc of `total` GiB in one piece: the time to beat.
//   2. HIP virtual memory management: reserve `total` GiB of address space, then hipMemCreate + hipMemMap + hipMemSetAccess
//      `chunk` GiB at a time; the time of every chunk.
//   3. the same while a resident kernel (one workgroup per CU, like a session) waits on a device word for "chunks mapped" and
//      touches every page of a chunk as soon as the word says it is there: do the mappings proceed beside a running kernel,
//      and does the kernel see the memory?
// Usage: vmm_probe [total GiB = 96] [chunk GiB = 4]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      std::printf("FAILED %s: %s\n", #x, hipGetErrorString(e_));                               \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

// every workgroup polls `mapped` (chunks mapped so far, written by the host through a pinned word) and touches the chunks
// as they arrive: one 8-byte store per 4 KB page, workgroups striped over the pages
__global__ void resident(volatile unsigned* mapped, unsigned n_chunks, char* base, size_t chunk_bytes, unsigned long long* touched,
                         volatile unsigned* give_up) {
  unsigned seen = 0;
  while (seen < n_chunks && !*give_up) {
    unsigned m = __hip_atomic_load((unsigned*)mapped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    m = __builtin_amdgcn_readfirstlane(m);
    while (seen < m) {
      char* c = base + (size_t)seen * chunk_bytes;
      const size_t pages = chunk_bytes / 4096;
      for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < pages; p += (size_t)gridDim.x * blockDim.x)
        *reinterpret_cast<unsigned long long*>(c + p * 4096) = p + seen;
      if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(touched, 1ull);
      ++seen;
    }
    __builtin_amdgcn_s_sleep(64);
  }
}

int main(int argc, char** argv) {
  const size_t GiB = 1ull << 30;
  const size_t total = (size_t)(argc > 1 ? std::atoi(argv[1]) : 96) * GiB;
  const size_t chunk = (size_t)(argc > 2 ? std::atoi(argv[2]) : 4) * GiB;
  CK(hipSetDevice(0));
  {
    void* p = nullptr;
    double t0 = now();
    CK(hipMalloc(&p, total));
    std::printf("1. hipMalloc %zu GiB in one piece: %.3f s\n", total / GiB, now() - t0);
    t0 = now();
    CK(hipFree(p));
    std::printf("   hipFree: %.3f s\n", now() - t0);
  }
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  std::printf("2. VMM: allocation granularity %zu bytes\n", gran);
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const unsigned n_chunks = (unsigned)(total / chunk);
  for (int pass = 0; pass < 2; ++pass) {
    void* va = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    std::printf("%s reserve %zu GiB of address space: %.3f ms\n", pass ? "3. (beside a resident kernel)" : "  ", total / GiB, (now() - t0) * 1e3);
    unsigned* mapped = nullptr;
    unsigned* give_up = nullptr;
    unsigned long long* touched = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&mapped), 64, hipHostMallocCoherent));
    give_up = mapped + 8;
    *mapped = 0;
    *give_up = 0;
    CK(hipMalloc(reinterpret_cast<void**>(&touched), 8));
    CK(hipMemset(touched, 0, 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (pass) {
      hipLaunchKernelGGL(resident, dim3(256), dim3(256), 0, s, mapped, n_chunks, static_cast<char*>(va), chunk, touched, give_up);
      CK(hipGetLastError());
    }
    std::vector<hipMemGenericAllocationHandle_t> handles;
    const double t_all = now();
    for (unsigned c = 0; c < n_chunks; ++c) {
      hipMemGenericAllocationHandle_t h;
      double a = now();
      CK(hipMemCreate(&h, chunk, &prop, 0));
      double b = now();
      CK(hipMemMap(static_cast<char*>(va) + (size_t)c * chunk, chunk, 0, h, 0));
      double d = now();
      CK(hipMemSetAccess(static_cast<char*>(va) + (size_t)c * chunk, chunk, &acc, 1));
      double e = now();
      handles.push_back(h);
      __atomic_store_n(mapped, c + 1, __ATOMIC_RELEASE);
      if (c < 4 || c + 1 == n_chunks || (e - a) > 0.02)
        std::printf("   chunk %2u: create %.1f ms, map %.1f ms, set access %.1f ms (since start %.3f s)\n", c, (b - a) * 1e3, (d - b) * 1e3, (e - d) * 1e3, e - t_all);
    }
    std::printf("   all %u chunks of %zu GiB mapped after %.3f s\n", n_chunks, chunk / GiB, now() - t_all);
    if (pass) {
      double t1 = now();
      hipError_t q;
      while ((q = hipStreamQuery(s)) == hipErrorNotReady && now() - t1 < 20.0) {
      }
      if (q == hipErrorNotReady) {
        *give_up = 1;
        std::printf("   the resident kernel did NOT see all chunks within 20 s: giving up\n");
      }
      CK(hipStreamSynchronize(s));
      unsigned long long n = 0;
      CK(hipMemcpy(&n, touched, 8, hipMemcpyDeviceToHost));
      std::printf("   resident kernel touched %llu of %u chunks; finished %.3f s after the last mapping\n", n, n_chunks, now() - t1);
    }
    t0 = now();
    for (unsigned c = 0; c < n_chunks; ++c) {
      CK(hipMemUnmap(static_cast<char*>(va) + (size_t)c * chunk, chunk));
      CK(hipMemRelease(handles[c]));
    }
    CK(hipMemAddressFree(va, total));
    std::printf("   unmap + release + free: %.3f s\n", now() - t0);
    CK(hipStreamDestroy(s));
    CK(hipFree(touched));
    CK(hipHostFree(mapped));
  }
  return 0;
}
