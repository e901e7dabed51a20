// Sanitizer builds of the HOST side (tools/sanitize/Makefile): the .hip translation units cannot be compiled by g++, and
// nothing in a CPU-only run may reach them -- every entry point of the device code is a stub that aborts loudly.
#include <cstdio>
#include <cstdlib>

#include "../../dynamont_amd/csrc/nt_kernels.hpp"

namespace dynk {

[[noreturn]] static void no_device(const char* what) {
  std::fprintf(stderr, "sanitizer build: %s needs the GPU build of the library\n", what);
  std::abort();
}

void launch_session(bool, int, const SessionArgs&, const void*, void*, const dynmath::SoftplusNode*, int, hipStream_t) { no_device("launch_session"); }
void launch_session_publish(SessionTicket*, uint32_t*, const SessionTicket&, uint32_t, uint32_t, hipStream_t) { no_device("launch_session_publish"); }
void launch_session_close(uint32_t*, hipStream_t) { no_device("launch_session_close"); }
void launch_preprocess(const void*, int, int, const uint64_t*, const double*, const double*, const float*, const float*, void*, double*, int, uint64_t,
                       int, double, hipStream_t) {
  no_device("launch_preprocess");
}
void launch_prep_params(const int32_t*, const Emis*, Emis*, uint64_t, uint32_t, hipStream_t) { no_device("launch_prep_params"); }
void launch_pool_init(const PagePool&, uint32_t, int, hipStream_t) { no_device("launch_pool_init"); }
void launch_read_queue(QueueJob, bool, const QueueArgs&, int, hipStream_t) { no_device("launch_read_queue"); }
void launch_segments(const ReadDesc*, int, uint64_t, uint32_t, const ReadState*, TraceBuffers, SegRow*, int, hipStream_t) { no_device("launch_segments"); }
size_t pool_stats_temp_bytes(uint64_t, uint64_t) { return 8; }
size_t pool_stats_work_bytes(uint64_t) { return 8; }
hipError_t launch_pool_stats(const ReadDesc*, int, uint32_t, const ReadState*, const int32_t*, TrainBuffers, double*, uint64_t, uint64_t, void*, void*,
                             size_t, hipStream_t) {
  no_device("launch_pool_stats");
}
uint64_t wide_arena_bytes(uint64_t T, uint64_t bw, bool calc) { return T * (2 * bw + 3) * (16 + (calc ? 9 : 0)); }
void launch_wide_reads(int, const WideArgs&, int, hipStream_t) { no_device("launch_wide_reads"); }

}  // namespace dynk
