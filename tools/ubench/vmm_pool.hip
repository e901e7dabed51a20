// GPU box: a lattice pool as ONE virtual range backed by physical chunks mapped one at a time (HIP virtual memory
// management). Times reserve / create / map / set-access per chunk and a first touch, twice in a row (the second pass
// allocates VRAM the first one has just released: the driver scrubs freed memory, which is what makes a plain
// hipMalloc of 110 GB take 1-4 s on a box that has been used before).
//   hipcc --offload-arch=gfx950 -O2 -o vmm_pool vmm_pool.hip && ./vmm_pool [chunk_GiB] [total_GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t GiB = 1ull << 30;
  const size_t chunk = (argc > 1 ? atoi(argv[1]) : 8) * GiB, total = (argc > 2 ? atoi(argv[2]) : 112) * GiB;
  int dev = 0, vmm = 0;
  CK(hipSetDevice(dev));
  CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
  printf("virtual memory management supported: %d\n", vmm);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity %zu bytes\n", gran);
  for (int pass = 0; pass < 2; ++pass) {
    void* base = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&base, total, gran, nullptr, 0));
    printf("pass %d: reserve %zu GiB of address space: %.4f s\n", pass, total / GiB, now() - t0);
    std::vector<hipMemGenericAllocationHandle_t> hs;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const double tall = now();
    for (size_t off = 0; off < total; off += chunk) {
      hipMemGenericAllocationHandle_t h;
      double a = now();
      CK(hipMemCreate(&h, chunk, &prop, 0));
      double b = now();
      CK(hipMemMap((char*)base + off, chunk, 0, h, 0));
      CK(hipMemSetAccess((char*)base + off, chunk, &acc, 1));
      double c = now();
      CK(hipMemset((char*)base + off, 0, chunk));
      CK(hipDeviceSynchronize());
      double d = now();
      hs.push_back(h);
      printf("  chunk at %3zu GiB: create %.3f s, map + access %.3f s, first touch %.3f s\n", off / GiB, b - a, c - b, d - c);
    }
    printf("pass %d: %zu GiB mapped in %.2f s\n", pass, total / GiB, now() - tall);
    t0 = now();
    CK(hipMemUnmap(base, total));
    for (auto h : hs) CK(hipMemRelease(h));
    CK(hipMemAddressFree(base, total));
    printf("pass %d: unmap + release: %.3f s\n", pass, now() - t0);
  }
  return 0;
}
