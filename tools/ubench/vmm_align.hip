// GPU box: why the lattice pool is NOT built from HIP virtual-memory ranges (yet). One thread reserves a range, maps
// chunks into it, lets the GPU write them, unmaps them chunk by chunk, frees the range and starts over with other chunk
// sizes -- and ROCm 7.2 refuses hipMemSetAccess ("invalid argument") for some chunks whose address was part of an
// earlier mapping (2 of 31 here; which ones changes with the history). Earlier variants of this file showed two more
// refusals: a chunk mapped by a thread other than the one whose kernel is running on the neighbouring chunk, and the
// same with a dedicated mapping thread. A pool that grows while kernels run would need exactly those calls.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>
static hipMemAllocationProp prop{};
static std::vector<hipMemGenericAllocationHandle_t> g_handles;
static int add_chunk(void* base, size_t off, size_t bytes) {
  hipMemGenericAllocationHandle_t h;
  hipError_t e1 = hipMemCreate(&h, bytes, &prop, 0);
  hipError_t e2 = e1 == hipSuccess ? hipMemMap((char*)base + off, bytes, 0, h, 0) : e1;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  hipError_t e3 = e2 == hipSuccess ? hipMemSetAccess((char*)base + off, bytes, &acc, 1) : e2;
  printf("   %5zu MiB at %6zu MiB: create %d map %d access %d%s\n", bytes >> 20, off >> 20, (int)e1, (int)e2, (int)e3, e3 ? "  <-- refused" : "");
  fflush(stdout);
  (void)hipGetLastError();
  if (e3 != hipSuccess && e2 == hipSuccess) (void)hipMemUnmap((char*)base + off, bytes);
  if (e1 == hipSuccess) (void)hipMemRelease(h);  // (the mapping keeps the memory alive until it is unmapped)
  return (int)e3;
}
int main() {
  (void)hipSetDevice(0);
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  const size_t MiB = 1 << 20;
  std::vector<std::vector<size_t>> plans = {{4096, 798}, {4096, 1024}, {4096, 512, 286}, {64, 14}, {64, 16}, {210, 14}, {2048, 400}, {4096, 800}, {4096, 768}, {4096, 1024}, {4096, 1024, 1024}, {1024, 1024, 1024, 1024, 1024}, {4096, 1024}};
  for (auto& plan : plans) {
    void* base = nullptr;
    (void)hipMemAddressReserve(&base, 16ull << 30, 2 * MiB, nullptr, 0);
    printf("plan at base %p (the GPU writes every chunk that was accepted before the next one is mapped):\n", base);
    size_t off = 0;
    std::vector<size_t> done;
    for (size_t m : plan) {
      if (add_chunk(base, off, m * MiB)) break;
      (void)hipMemset((char*)base + off, 0, m * MiB);
      (void)hipDeviceSynchronize();
      off += m * MiB;
      done.push_back(m * MiB);
    }
    off = 0;
    for (size_t b : done) {  // unmapped the way it was mapped: chunk by chunk (one call over the union leaves stale state)
      hipError_t e = hipMemUnmap((char*)base + off, b);
      if (e != hipSuccess) printf("   unmap of %zu MiB at %zu MiB: %d\n", b >> 20, off >> 20, (int)e);
      off += b;
    }
    (void)hipMemAddressFree(base, 16ull << 30);
  }
  return 0;
}
