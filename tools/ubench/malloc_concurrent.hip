// GPU box: does a large hipMalloc in one thread (seconds, when the driver has to scrub VRAM) hold up kernel launches and
// small allocations of another thread? Pass 1 dirties 112 GiB and frees them; pass 2 allocates 110 GiB in a worker
// thread while the main thread keeps launching a short kernel and timing each launch + synchronise.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(double* p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}
int main() {
  (void)hipSetDevice(0);
  const size_t GiB = 1ull << 30;
  {  // dirty the memory
    void* big = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&big, 112 * GiB);
    printf("pass 1: hipMalloc 112 GiB: %.3f s (rc %d)\n", now() - t0, (int)e);
    (void)hipMemset(big, 1, 112 * GiB);
    (void)hipDeviceSynchronize();
    (void)hipFree(big);
  }
  double* small = nullptr;
  (void)hipMalloc(&small, 1 * GiB);
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  std::atomic<int> done{0};
  double t_alloc = 0;
  void* big2 = nullptr;
  std::thread worker([&] {
    (void)hipSetDevice(0);
    double t0 = now();
    hipError_t e = hipMalloc(&big2, 110 * GiB);
    t_alloc = now() - t0;
    printf("worker: hipMalloc 110 GiB: %.3f s (rc %d)\n", t_alloc, (int)e);
    done = 1;
  });
  double worst = 0, sum = 0;
  int n = 0;
  const double t_begin = now();
  while (!done) {
    double t0 = now();
    hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, s, small, (size_t)1 << 20);
    (void)hipStreamSynchronize(s);
    double dt = now() - t0;
    worst = dt > worst ? dt : worst;
    sum += dt;
    ++n;
  }
  worker.join();
  printf("main: %d launch+sync round trips in %.3f s while the worker allocated: mean %.1f us, worst %.1f ms\n", n, now() - t_begin, 1e6 * sum / (n ? n : 1), 1e3 * worst);
  // and a medium allocation from the main thread while a second big one is in flight
  (void)hipFree(big2);
  done = 0;
  std::thread worker2([&] {
    (void)hipSetDevice(0);
    double t0 = now();
    hipError_t e = hipMalloc(&big2, 110 * GiB);
    printf("worker: second hipMalloc 110 GiB: %.3f s (rc %d)\n", now() - t0, (int)e);
    done = 1;
  });
  std::this_thread::sleep_for(std::chrono::milliseconds(50));
  void* mid = nullptr;
  double t0 = now();
  hipError_t e = hipMalloc(&mid, 8 * GiB);
  printf("main: hipMalloc 8 GiB while the worker allocates: %.3f s (rc %d)\n", now() - t0, (int)e);
  worker2.join();
  return 0;
}
