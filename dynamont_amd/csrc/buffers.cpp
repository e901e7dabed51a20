// buffers.cpp -- device / pinned buffers of the host engine: the per-handle buffer cache (BufCache, DevBuf, PinnedBuf), the lattice
// pools and batch buffers that destroyed handles PARK for their successors (hipMalloc / hipFree of ~100 GB take seconds and hipFree
// waits for the whole device), dyn_release_cached_memory, and the helper-thread pool. Reference counterpart: std::vector<double>
// matrices allocated per read (NT_aligner_api.cpp:249-262).
#include "engine_internal.hpp"
#include "dp_math_strict.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <queue>

using dynhost::PoreModel;
using dynk::ReadDesc;
using dynk::ReadState;
using dynk::SegRow;
using dynmath::Emis;
using namespace dyneng;

namespace dyneng {

// ---- buffers -----------------------------------------------------------------------------------
size_t BufCache::round_up(size_t want) {
  size_t g = (size_t)1 << 16;
  while (g * 16 < want) g <<= 1;  // granule between want/16 and want/8: <= 12.5 % over-allocation
  return (want + g - 1) / g * g;
}

namespace {
// buffers parked by destroyed handles, per device (BufCache::park); g_bufpark_m guards them
constexpr int BUFPARK_DEVICES = 32;
constexpr size_t BUFPARK_MAX_BYTES = (size_t)16 << 30;  // per device and kind
std::mutex g_bufpark_m;
std::multimap<size_t, void*> g_bufpark[BUFPARK_DEVICES][2];  // [device][pinned]
size_t g_bufpark_bytes[BUFPARK_DEVICES][2];
}  // namespace

hipError_t BufCache::take(bool pinned, size_t want, void** p, size_t* got) {
  {
    std::lock_guard<std::mutex> lk(m);
    auto& mp = pinned ? pin : dev;
    auto it = mp.lower_bound(want);
    if (it != mp.end() && it->first <= 2 * want + ((size_t)1 << 20)) {
      *p = it->second;
      *got = it->first;
      mp.erase(it);
      return hipSuccess;
    }
  }
  if (device >= 0 && device < BUFPARK_DEVICES) {
    std::lock_guard<std::mutex> lk(g_bufpark_m);
    auto& mp = g_bufpark[device][pinned ? 1 : 0];
    auto it = mp.lower_bound(want);
    if (it != mp.end() && it->first <= 2 * want + ((size_t)1 << 20)) {
      *p = it->second;
      *got = it->first;
      g_bufpark_bytes[device][pinned ? 1 : 0] -= it->first;
      mp.erase(it);
      return hipSuccess;
    }
  }
  const size_t ask = round_up(want);
  hipError_t e = pinned ? hipHostMalloc(p, ask, hipHostMallocDefault) : hipMalloc(p, ask);
  if (e != hipSuccess && !(session_open && session_open->load())) {  // give everything cached back to the runtime and try once more
    (void)hipGetLastError();
    purge();
    e = pinned ? hipHostMalloc(p, ask, hipHostMallocDefault) : hipMalloc(p, ask);
  }
  if (e == hipSuccess) *got = ask;
  return e;
}

void BufCache::give(bool pinned, void* p, size_t bytes) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(m);
  (pinned ? pin : dev).emplace(bytes, p);
}

void BufCache::purge() {
  std::lock_guard<std::mutex> lk(m);
  for (auto& kv : dev) (void)hipFree(kv.second);
  for (auto& kv : pin) (void)hipHostFree(kv.second);
  dev.clear();
  pin.clear();
}

void BufCache::park(int dev_id) {
  if (dev_id < 0 || dev_id >= BUFPARK_DEVICES || std::getenv("DYN_NO_POOL_CACHE") != nullptr) {
    purge();
    return;
  }
  std::lock_guard<std::mutex> lk(m);
  std::lock_guard<std::mutex> lk2(g_bufpark_m);
  for (int k = 0; k < 2; ++k) {
    auto& from = k ? pin : dev;
    for (auto& kv : from) {
      if (g_bufpark_bytes[dev_id][k] + kv.first <= BUFPARK_MAX_BYTES) {
        g_bufpark[dev_id][k].emplace(kv.first, kv.second);
        g_bufpark_bytes[dev_id][k] += kv.first;
      } else if (k) {
        (void)hipHostFree(kv.second);
      } else {
        (void)hipFree(kv.second);
      }
    }
    from.clear();
  }
}

static void release_parked_buffers() {
  std::lock_guard<std::mutex> lk(g_bufpark_m);
  for (int d = 0; d < BUFPARK_DEVICES; ++d)
    for (int k = 0; k < 2; ++k) {
      if (g_bufpark[d][k].empty()) continue;
      (void)hipSetDevice(d);
      for (auto& kv : g_bufpark[d][k]) (void)(k ? hipHostFree(kv.second) : hipFree(kv.second));
      g_bufpark[d][k].clear();
      g_bufpark_bytes[d][k] = 0;
    }
}

// ---- parked lattice pools ---------------------------------------------------------------------------------
// Allocating (and freeing) a lattice pool of ~100 GB costs seconds (hipMalloc / hipFree of that size: 1.3-2.5 s each
// way on an MI355X). The reference's training loop builds a new Aligner for every batch (train.py:179,227), and so
// does its counterpart here: with 1 024-read batches that was 2.7 s of allocation around 25 ms of kernels. A handle
// that is destroyed therefore PARKS its three pool buffers, per device, and the next handle on that device takes them
// over if they are large enough (the page count in use is still capped by the handle's own memory budget). At most
// one set is parked per device; dyn_release_cached_memory() frees it. DYN_NO_POOL_CACHE=1 switches parking off.
namespace {
struct ParkedBuf {
  void* p = nullptr;
  size_t bytes = 0;
};
constexpr int PARK_DEVICES = 32, PARK_KINDS = 3;  // kinds: ws, lpe, bits
std::mutex g_park_m;
ParkedBuf g_park[PARK_DEVICES][PARK_KINDS];

bool parking_enabled() {
  static const bool on = std::getenv("DYN_NO_POOL_CACHE") == nullptr;
  return on;
}

}  // namespace

// the handle's buffer goes to the parking slot (the larger of the two stays, the other is freed)
void park_pool_buffer(int device, int kind, DevBuf& b) {
  if (!b.p) return;
  if (!parking_enabled() || device < 0 || device >= PARK_DEVICES) {
    b.release();
    return;
  }
  std::lock_guard<std::mutex> lk(g_park_m);
  ParkedBuf& slot = g_park[device][kind];
  if (slot.p && slot.bytes >= b.bytes) {
    (void)hipFree(b.p);
  } else {
    if (slot.p) (void)hipFree(slot.p);
    slot.p = b.p;
    slot.bytes = b.bytes;
  }
  b.p = nullptr;
  b.bytes = 0;
}

// bytes parked on a device: memory this process holds that a handle can take over (or have freed) on demand -- part of
// what is available to the next handle, although hipMemGetInfo reports it as used
size_t parked_bytes(int device) {
  if (!parking_enabled() || device < 0 || device >= PARK_DEVICES) return 0;
  std::lock_guard<std::mutex> lk(g_park_m);
  size_t n = 0;
  for (int k = 0; k < PARK_KINDS; ++k) n += g_park[device][k].bytes;
  return n;
}

// give back what destroyed handles have parked on `device` (their pools are this process's to use, for whatever needs the room)
void free_parked(int device) {
  if (device < 0 || device >= PARK_DEVICES) return;
  std::lock_guard<std::mutex> lk(g_park_m);
  for (int k = 0; k < PARK_KINDS; ++k)
    if (g_park[device][k].p) {
      (void)hipFree(g_park[device][k].p);
      g_park[device][k] = ParkedBuf{};
    }
}

// grow `b` to at least `want` bytes: a parked buffer that is large enough, else a fresh allocation (after the parked
// one has been freed: its memory may be what the larger buffer needs)
hipError_t ensure_pool_buffer(int device, int kind, DevBuf& b, size_t want, double headroom) {
  if (want <= b.bytes) return hipSuccess;
  if (parking_enabled() && device >= 0 && device < PARK_DEVICES) {
    std::lock_guard<std::mutex> lk(g_park_m);
    ParkedBuf& slot = g_park[device][kind];
    if (slot.p && slot.bytes >= want) {
      b.release();
      b.p = slot.p;
      b.bytes = slot.bytes;
      slot = ParkedBuf{};
      return hipSuccess;
    }
    if (slot.p) {
      (void)hipFree(slot.p);
      slot = ParkedBuf{};
    }
  }
  hipError_t e = b.ensure(want, headroom);
  if (e == hipErrorOutOfMemory && parking_enabled() && device >= 0 && device < PARK_DEVICES) {
    // The planner counts parked bytes as available (they are): a parked buffer of ANOTHER kind -- the separate posterior plane
    // of a predecessor whose successor keeps its posteriors in place -- may be what this allocation needs.
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_park_m);
    for (int k = 0; k < PARK_KINDS; ++k)
      if (g_park[device][k].p) {
        (void)hipFree(g_park[device][k].p);
        g_park[device][k] = ParkedBuf{};
      }
    e = b.ensure(want, headroom);
  }
  return e;
}

// The three arrays of a lattice pool together. A buffer taken over from a predecessor's parked pool may be far larger than
// asked for (a handle that served reads of 100 k samples, in place: one 250 GB array) and leave no room for the others:
// on out-of-memory everything the handle holds of the pool is released and the three are allocated again at their sizes.
hipError_t ensure_pool(int device, DevBuf& ws, size_t ws_bytes, DevBuf& lpe, size_t lpe_bytes, DevBuf& bits, size_t bits_bytes, double headroom) {
  auto all = [&]() -> hipError_t {
    hipError_t e = ensure_pool_buffer(device, 0, ws, ws_bytes, headroom);
    if (e == hipSuccess && lpe_bytes) e = ensure_pool_buffer(device, 1, lpe, lpe_bytes, headroom);
    if (e == hipSuccess && bits_bytes) e = ensure_pool_buffer(device, 2, bits, bits_bytes, headroom);
    return e;
  };
  hipError_t e = all();
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    ws.release();
    lpe.release();
    bits.release();
    e = all();
  }
  return e;
}

extern "C" void dyn_release_cached_memory(void) {
  std::lock_guard<std::mutex> lk(g_park_m);
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  dyneng::release_parked_buffers();
  for (int d = 0; d < PARK_DEVICES; ++d)
    for (int k = 0; k < PARK_KINDS; ++k)
      if (g_park[d][k].p) {
        (void)hipSetDevice(d);
        (void)hipFree(g_park[d][k].p);
        g_park[d][k] = ParkedBuf{};
      }
  if (have_cur) (void)hipSetDevice(cur);
}

hipError_t DevBuf::ensure(size_t want, double headroom) {
  if (want <= bytes) return hipSuccess;
  release();
  if (cache) return cache->take(false, want, &p, &bytes);
  size_t ask = (size_t)((double)want * headroom);
  if (ask < want) ask = want;
  hipError_t e = hipMalloc(&p, ask);
  if (e != hipSuccess && ask > want) {  // no room for the headroom: take exactly what is needed
    (void)hipGetLastError();
    ask = want;
    e = hipMalloc(&p, ask);
  }
  if (e == hipSuccess) bytes = ask;
  else p = nullptr;
  return e;
}

void DevBuf::release() {
  if (p) {
    if (cache) cache->give(false, p, bytes);
    else (void)hipFree(p);
  }
  p = nullptr;
  bytes = 0;
}

hipError_t PinnedBuf::ensure(size_t want) {
  if (want <= bytes) return hipSuccess;
  release();
  if (cache) return cache->take(true, want, &p, &bytes);
  const size_t ask = want + want / 8;
  hipError_t e = hipHostMalloc(&p, ask, hipHostMallocDefault);
  if (e == hipSuccess) bytes = ask;
  else p = nullptr;
  return e;
}

void PinnedBuf::release() {
  if (p) {
    if (cache) cache->give(true, p, bytes);
    else (void)hipHostFree(p);
  }
  p = nullptr;
  bytes = 0;
}

// ---- helper pool -------------------------------------------------------------------------------
HelperPool::HelperPool(int n_threads) {
  for (int i = 1; i < n_threads; ++i) workers_.emplace_back([this] { worker(); });
}

HelperPool::~HelperPool() {
  {
    std::lock_guard<std::mutex> lk(m_);
    stop_ = true;
  }
  cv_work_.notify_all();
  for (auto& t : workers_) t.join();
}

void HelperPool::worker() {
  uint64_t seen = 0;
  std::unique_lock<std::mutex> lk(m_);
  for (;;) {
    cv_work_.wait(lk, [&] { return stop_ || (gen_ != seen && next_ < n_); });
    if (stop_) return;
    seen = gen_;
    while (next_ < n_) {
      const int task = next_++;
      ++active_;
      const auto* fn = fn_;
      lk.unlock();
      (*fn)(task);
      lk.lock();
      --active_;
    }
    if (active_ == 0) cv_done_.notify_all();
  }
}

void HelperPool::parallel_for(int n_tasks, const std::function<void(int)>& fn) {
  if (n_tasks <= 0) return;
  if (n_tasks == 1 || workers_.empty()) {
    for (int i = 0; i < n_tasks; ++i) fn(i);
    return;
  }
  // ONE job slot: concurrent callers (the pipeline's front and back threads share a pool) take turns. Without
  // this a second caller overwrote fn_/n_/next_ of a job in progress -- tasks of the first were lost and its
  // cv_done_ wait could hang (seen once staging became 24 tasks of page-faulting copies).
  std::lock_guard<std::mutex> turn(call_m_);
  std::unique_lock<std::mutex> lk(m_);
  fn_ = &fn;
  n_ = n_tasks;
  next_ = 0;
  ++gen_;
  cv_work_.notify_all();
  while (next_ < n_) {  // the caller takes part
    const int task = next_++;
    ++active_;
    lk.unlock();
    fn(task);
    lk.lock();
    --active_;
  }
  cv_done_.wait(lk, [&] { return active_ == 0 && next_ >= n_; });
  fn_ = nullptr;
  n_ = 0;
}

}  // namespace dyneng
