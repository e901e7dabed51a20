// dynamont_mi.cpp -- C-ABI entry points (include/dynamont_mi.h) and the host batch engine:
// validation + k-mer coding per read, HBM planning, kernel launches, result marshalling.
//
// Reference driver being replaced: NTAligner::align / NTAligner::train
// (src/cpp/NT_aligner_api.cpp:230-312, 567-639) and the pybind marshalling around them
// (src/cpp/aligner_bindings.cpp:53-107,132-165). Per-read failures are isolated exactly as the
// reference's per-read try/except does (src/dynamont/segmentation/segment.py:160-187).
//
// Every GPU stage of a batch is ENQUEUED without a host synchronisation (enqueue_job): the
// synchronous staged API (dyn_batch_create / _align / _fetch) waits for the stream itself, the
// asynchronous pipeline (async_engine.cpp) lets batch k+1's host work and copies run under batch
// k's kernels.
//
// There is NO CPU compute path here: without a bound GPU every compute entry point fails with
// DYN_ERR_DEVICE.
#include "engine.hpp"
#include "dp_math_strict.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <queue>
#include <stdexcept>

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
}  // namespace

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

namespace {

void copy_msg(char* buf, uint64_t cap, const std::string& s) {
  if (!buf || !cap) return;
  const size_t n = std::min<size_t>(cap - 1, s.size());
  std::memcpy(buf, s.data(), n);
  buf[n] = 0;
}

// (last_error is written under err_mu: the pipeline's back thread runs session_finish_enqueue / collect_timing without the
// handle's lock, while the front thread and the caller's thread report their own failures)
#define HIP_TRY(a, expr)                                                                  \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      {                                                                                   \
        std::lock_guard<std::mutex> _elk((a)->err_mu);                                    \
        (a)->last_error = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr; \
      }                                                                                   \
      return _e == hipErrorOutOfMemory ? DYN_ERR_OUT_OF_MEMORY : DYN_ERR_DEVICE;          \
    }                                                                                     \
  } while (0)

// A getter's copy off the device: on the handle's own non-blocking stream and waited for there. The data is complete when a
// getter may be called (the synchronous job calls return after the compute stream has passed the batch, dyn_batch_wait after
// the copy-out stream has); a null-stream hipMemcpy would ALSO wait for the resident session kernel that later tickets keep
// open (the session stream is not a non-blocking one) and serialise a caller's stream of batches.
hipError_t copy_out(dyn_aligner* a, void* dst, const void* src, size_t bytes) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, a->s_get);
  return e != hipSuccess ? e : hipStreamSynchronize(a->s_get);
}

}  // namespace

namespace dyneng {
void attach_cache(dyn_batch* b) {
  dyneng::BufCache* c = &b->a->cache;
  for (DevBuf* d : {&b->d_sig, &b->d_kmers, &b->d_par, &b->d_state, &b->d_rows, &b->d_segrow, &b->d_medhi,
                    &b->d_medlo, &b->d_descs, &b->d_colw, &b->d_cols1, &b->d_cols2, &b->d_trans, &b->d_pooled, &b->d_poolwork, &b->d_pooltemp, &b->d_pp,
                    &b->d_pathn, &b->d_norm, &b->d_meta, &b->d_wide})
    d->cache = c;
  for (PinnedBuf* h : {&b->h_kmers, &b->h_descs, &b->h_state, &b->h_rows, &b->h_stats, &b->h_sig}) h->cache = c;
}

int need_device(dyn_aligner* a) {
  if (a->host_only) {
    a->last_error = "no GPU bound to this handle (created with DYN_DEVICE_HOST_ONLY); the MI355X "
                    "build has no CPU compute path";
    return DYN_ERR_DEVICE;
  }
  hipError_t e = hipSetDevice(a->device);
  if (e != hipSuccess) {
    a->last_error = std::string("HIP error: ") + hipGetErrorString(e) + " at hipSetDevice";
    return DYN_ERR_DEVICE;
  }
  return DYN_OK;
}

// Host front half of align()/train(): validateInput then sequenceToKmers
// (NT_aligner_api.cpp:236-238). Reads are independent: k-mer coding runs on the helper pool.
// Every read that passes validateInput gets its slice of the flat k-mer array up front; a read that
// then fails in sequenceToKmers ("Invalid nucleotide") simply leaves its slice unused.
int host_prepare(dyn_batch* b, const PoreModel& m, bool pinned, uint64_t n, const uint64_t* sig_offsets,
                 const char* seqs, const uint64_t* seq_offsets, HelperPool* pool) {
  b->n = n;
  b->reads.assign(n, HostRead());
  b->n_wide = 0;
  uint64_t cap = 0, flat = 0;
  for (uint64_t i = 0; i < n; ++i) {
    HostRead& r = b->reads[i];
    r.S = sig_offsets[i + 1] - sig_offsets[i];
    r.L = seq_offsets[i + 1] - seq_offsets[i];
    r.sig_off = sig_offsets[i] - sig_offsets[0];
    r.seg_off = cap;
    r.kc = r.L >= (uint64_t)m.k ? r.L - (uint64_t)m.k + 1 : 0;
    cap += r.kc;
    r.status = m.validate(r.S, r.L);
    // 448 band slots per lattice row in the register sweeps: a read whose half band min(band / 2, columns / 2) does not fit
    // them takes the generic kernel (wide_band.hip), which holds a row of up to 4 096 band columns in LDS
    if (r.status == DYN_READ_OK) {
      const uint64_t hb = std::min<uint64_t>(m.half_band, (r.kc + 1) / 2);
      if (hb > (uint64_t)dynk::WIDE_MAX_HALF_BAND) r.status = DYN_READ_BAND_TOO_WIDE;
      else if (hb > (uint64_t)dynk::MAX_HALF_BAND) {
        r.wide = true;
        ++b->n_wide;
      }
    }
    if (r.status == DYN_READ_OK) {
      r.flat_off = flat;
      flat += r.kc;
    }
  }
  b->capacity = cap;
  b->total_cols = flat;
  if (pinned) {
    if (b->h_kmers.ensure(std::max<uint64_t>(4, flat * 4)) != hipSuccess) {
      (void)hipGetLastError();
      if (b->a) b->a->last_error = "out of pinned host memory for the k-mer codes";
      return DYN_ERR_OUT_OF_MEMORY;
    }
  } else {  // dyn_validate_batch needs no GPU: plain memory, no HIP call
    b->h_kmers.release();
    b->h_kmers.cache = nullptr;
    b->h_kmers.p = std::malloc(std::max<uint64_t>(4, flat * 4));
    b->h_kmers.bytes = 0;  // marks "malloc'ed": freed by the caller of host_prepare
    if (!b->h_kmers.p) return DYN_ERR_OUT_OF_MEMORY;
  }
  int32_t* km = b->h_kmers.as<int32_t>();
  auto encode_range = [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) {
      HostRead& r = b->reads[i];
      if (r.status != DYN_READ_OK) continue;
      r.status = m.encode(seqs + seq_offsets[i], r.L, km + r.flat_off, &r.bad);
      // an invalid nucleotide ends the encoding half-way: the rest of the read's columns would keep whatever the
      // (recycled) buffer held, and k_prep_params turns every column of the batch into a table index
      if (r.status != DYN_READ_OK) std::fill(km + r.flat_off, km + r.flat_off + r.kc, 0);
    }
  };
  const int parts = (pool && n >= 64) ? std::min<int>(pool->size() * 2, (int)(n / 16)) : 1;
  if (parts <= 1) encode_range(0, n);
  else pool->parallel_for(parts, [&](int t) { encode_range(n * t / parts, n * (t + 1) / parts); });
  b->max_T = b->max_N = 0;
  for (HostRead& r : b->reads) {
    if (r.status != DYN_READ_OK) continue;
    if (r.S + 1 > 0x7fffffffull || r.kc + 1 > 0x7fffffffull) {
      r.status = DYN_READ_TOO_LARGE;  // 32-bit lattice indices; isolated per read like any other failure
      continue;
    }
    b->max_T = std::max<uint32_t>(b->max_T, (uint32_t)(r.S + 1));
    b->max_N = std::max<uint32_t>(b->max_N, (uint32_t)(r.kc + 1));
  }
  return DYN_OK;
}

static int alloc_batch_buffers_once(dyn_batch* b, uint64_t total_sig) {
  dyn_aligner* a = b->a;
  HIP_TRY(a, b->d_sig.ensure(std::max<uint64_t>(8, total_sig * 8)));
  HIP_TRY(a, b->d_kmers.ensure(std::max<uint64_t>(4, b->total_cols * 4)));
  HIP_TRY(a, b->d_par.ensure(std::max<uint64_t>(sizeof(Emis), b->total_cols * sizeof(Emis))));
  HIP_TRY(a, b->d_state.ensure(std::max<uint64_t>(sizeof(ReadState), b->n * sizeof(ReadState))));
  HIP_TRY(a, b->d_rows.ensure(std::max<uint64_t>(sizeof(SegRow), b->capacity * sizeof(SegRow))));
  HIP_TRY(a, b->h_state.ensure(std::max<uint64_t>(sizeof(ReadState), b->n * sizeof(ReadState))));
  return DYN_OK;
}

// (caller holds a->mu) While a session is open the buffer cache does not free anything to make room (BufCache::session_open):
// out of memory then means: let the resident waves finish what is published, leave, THEN purge and allocate.
int alloc_batch_buffers(dyn_batch* b, uint64_t total_sig) {
  dyn_aligner* a = b->a;
  int rc = alloc_batch_buffers_once(b, total_sig);
  if (rc == DYN_ERR_OUT_OF_MEMORY && a->sess_open_hint.load()) {
    (void)hipGetLastError();
    if (int q = session_quiesce(a)) return q;
    a->cache.purge();
    rc = alloc_batch_buffers_once(b, total_sig);
  }
  return rc;
}

}  // namespace dyneng

extern "C" {

int dyn_pore_from_string(const char* s, int* pore_out, char* err, uint64_t errcap) {
  try {
    *pore_out = dynhost::pore_from_string(s ? s : "");
    return DYN_OK;
  } catch (const std::invalid_argument& e) {
    copy_msg(err, errcap, e.what());
    return DYN_ERR_INVALID_ARGUMENT;
  }
}

// The certified emission forms (x - mean) / stdev from stdev and RN(1 / stdev); the one divisor that construction cannot
// serve is a significand of all ones (dp_math_strict.hpp) -- no decimal of a model file parses to one. A handle whose
// table holds one runs the plain kernels (strict mode 0) and refuses dyn_aligner_set_strict(1 | 2).
static bool model_allows_strict(const dynhost::PoreModel& m) {
  for (double sd : m.stdev)
    if (dynmath::div_by_const_excluded(sd)) return false;
  return true;
}

// The session stream: CU-masked (every CU enabled), hence a hardware queue of its own. `reserved_cus` CUs are left free by the
// session's GRID (one workgroup of 150 KB of LDS per CU, n_cus - reserved of them), not by the mask: partitioning by masks
// leaves resident workgroups unplaced (tools/ubench/resident_probe.hip). What starts beside a session
// is what fits beside a resident workgroup (<= 9.5 KB of LDS, <= 152 registers per lane) or has no more workgroups than
// there are free CUs.
// CU-masked streams are PARKED per device, never destroyed: the second hipStreamDestroy of such a stream in a process did
// not return on this runtime (ROCm 7.2: set_session_mode(0) after a destroyed handle, and a CLI run in a loop, both
// stopped there), and a parked stream costs one idle hardware queue.
static std::mutex g_sess_stream_m;
static std::vector<hipStream_t> g_sess_streams[32];

extern "C++" {
namespace dyneng {
// shared with rccl_comm.cpp: a dyn_comm's stream is of the same kind and is parked the same way
hipStream_t take_masked_stream(int device, int n_cus) {
  if (device >= 0 && device < 32) {
    std::lock_guard<std::mutex> lk(g_sess_stream_m);
    if (!g_sess_streams[device].empty()) {
      hipStream_t s = g_sess_streams[device].back();
      g_sess_streams[device].pop_back();
      return s;
    }
  }
  if (n_cus <= 0) return nullptr;
  std::vector<uint32_t> mask((size_t)(n_cus + 31) / 32, 0u);
  for (int c = 0; c < n_cus; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return s;
}

void park_masked_stream(int device, hipStream_t s) {
  if (!s) return;
  if (std::getenv("DYN_DESTROY_SESSION_STREAM") || device < 0 || device >= 32) {
    // for processes that create ONE handle and run under rocprofv3 (tools/profile_round.sh): the profiler's exit handler
    // faults on a queue that is still alive, and the first destroy of a process has always returned
    (void)hipStreamDestroy(s);
    return;
  }
  std::lock_guard<std::mutex> lk(g_sess_stream_m);
  g_sess_streams[device].push_back(s);
}
}  // namespace dyneng
}  // extern "C++"

static void park_session_stream(dyn_aligner* a) {
  if (!a->s_session) return;
  a->sess_enabled.store(false);
  dyneng::park_masked_stream(a->device, a->s_session);
  a->s_session = nullptr;
}

static int make_session_stream(dyn_aligner* a, int reserved_cus) {
  // Reserved CUs are not masked out: a session is ONE workgroup per CU it uses (150 KB of LDS each), so a grid of
  // n_cus - reserved workgroups leaves `reserved` CUs empty wherever the dispatcher puts it. The mask enables every CU (the
  // stream serves any mode); what it buys is the hardware queue.
  a->sess_cus = std::max(1, a->n_cus - std::max(0, reserved_cus));
  if (!a->s_session) a->s_session = dyneng::take_masked_stream(a->device, a->n_cus);
  if (!a->s_session) return DYN_ERR_DEVICE;
  if (!a->sess_flags && hipHostMalloc(reinterpret_cast<void**>(&a->sess_flags), SESSION_FLAGS * 4, hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();
    park_session_stream(a);
    a->sess_flags = nullptr;
    return DYN_ERR_DEVICE;
  }
  a->sess_enabled.store(true);
  return DYN_OK;
}

int dyn_aligner_set_session_mode(dyn_aligner* a, int enabled, int reserved_cus) {
  if (!a || reserved_cus < 0) return DYN_ERR_INVALID_ARGUMENT;
  if (a->host_only) return DYN_OK;
  std::lock_guard<std::mutex> lk(a->mu);
  if (int rc = need_device(a)) return rc;
  if (int rc = session_quiesce(a)) return rc;
  if (!enabled) {
    park_session_stream(a);
    return DYN_OK;
  }
  if (reserved_cus >= a->n_cus) return DYN_ERR_INVALID_ARGUMENT;
  if (make_session_stream(a, reserved_cus) != DYN_OK) {
    a->last_error = "hipExtStreamCreateWithCUMask failed: the handle runs one launch per batch";
    return DYN_ERR_DEVICE;
  }
  return DYN_OK;
}

int dyn_aligner_create(const char* model_path, int pore, const char* mode, int threads,
                       uint64_t band, int device, dyn_aligner** out, char* err, uint64_t errcap) {
  *out = nullptr;
  const std::string md = mode ? mode : "basic";
  const bool ntk = md == "resquiggle" || md == "ntk";  // aligner_bindings.cpp:46-49 -> NTKAligner
  if (!ntk && md != "basic" && md != "nt") {
    copy_msg(err, errcap, "Unknown aligner mode: " + md);  // aligner_bindings.cpp:50
    return DYN_ERR_INVALID_ARGUMENT;
  }
  dyn_aligner* a = new dyn_aligner();
  a->threads = threads;
  a->ntk = ntk;
  // DYN_TRACE_HOST=1: where a handle's creation goes (tools/cold_start_trace.py)
  const bool trace_create = std::getenv("DYN_TRACE_HOST") != nullptr;
  const auto tc0 = std::chrono::steady_clock::now();
  auto tc_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count(); };
  // The HIP runtime comes up (50-60 ms in a fresh process) on a helper thread while this one parses the model file (~110 ms
  // for a 9-mer table): a cold dynamont-resquiggle run is start-up bound below ~10 k reads.
  std::thread hip_warm;
  if (device != DYN_DEVICE_HOST_ONLY)
    hip_warm = std::thread([device] {
      int dv = device;
      if (dv < 0 && hipGetDevice(&dv) != hipSuccess) return;
      if (hipSetDevice(dv) == hipSuccess) (void)hipFree(nullptr);
      (void)hipGetLastError();
    });
  struct Joiner {
    std::thread& t;
    ~Joiner() {
      if (t.joinable()) t.join();
    }
  } hip_warm_join{hip_warm};
  try {
    a->model.load(model_path ? model_path : "", pore, band);
    if (!model_allows_strict(a->model)) a->strict_mode = 0;
    if (trace_create) std::fprintf(stderr, "[dyn] create: model parsed at %.1f ms\n", tc_ms());
  } catch (const std::invalid_argument& e) {
    copy_msg(err, errcap, e.what());
    delete a;
    return DYN_ERR_INVALID_ARGUMENT;
  } catch (const std::exception& e) {
    copy_msg(err, errcap, e.what());
    delete a;
    return DYN_ERR_RUNTIME;
  }
  if (device == DYN_DEVICE_HOST_ONLY) {
    a->host_only = true;
    *out = a;
    return DYN_OK;
  }
  if (hip_warm.joinable()) hip_warm.join();
  auto fail = [&](hipError_t e, const char* what) {
    copy_msg(err, errcap, std::string("HIP error: ") + hipGetErrorString(e) + " at " + what +
                              " (the MI355X build has no CPU compute path)");
    park_session_stream(a);
    for (hipStream_t s : {a->stream, a->s_in, a->s_out, a->s_get})
      if (s) (void)hipStreamDestroy(s);
    if (a->sess_flags) (void)hipHostFree(a->sess_flags);
    a->d_model.release();
    a->d_sptab.release();
    delete a;
    return (int)DYN_ERR_DEVICE;
  };
  hipError_t e;
  if (device < 0) {
    e = hipGetDevice(&device);
    if (e != hipSuccess) return fail(e, "hipGetDevice");
  }
  a->device = device;
  a->cache.device = device;
  a->cache.session_open = &a->sess_open_hint;
  if ((e = hipSetDevice(device)) != hipSuccess) return fail(e, "hipSetDevice");
  if (trace_create) std::fprintf(stderr, "[dyn] create: HIP runtime up (hipSetDevice) at %.1f ms\n", tc_ms());
  if ((e = hipDeviceGetAttribute(&a->n_cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return fail(e, "hipDeviceGetAttribute");
  if (const char* f = std::getenv("DYN_QUEUE_CUS")) a->n_cus = std::max(1, std::atoi(f));  // experiments: fewer persistent workgroups
  if ((e = hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = hipStreamCreateWithFlags(&a->s_in, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = hipStreamCreateWithFlags(&a->s_out, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = hipStreamCreateWithFlags(&a->s_get, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  // The resident read queue needs a hardware queue of its own: plain streams share four per process, and a kernel that stays
  // resident blocks whatever is queued behind it in the same one -- the copies and small kernels it is waiting for included
  // (tools/ubench/resident_probe.hip). A CU-masked stream (all CUs enabled) always gets its own. No such stream, or
  // DYN_NO_SESSION=1: the handle runs one launch per batch, as before round 5.
  if (!std::getenv("DYN_NO_SESSION")) {
    const char* rsv = std::getenv("DYN_SESSION_RESERVE_CUS");
    (void)make_session_stream(a, rsv ? std::atoi(rsv) : 0);  // failure: the handle simply has no resident read queue
    if (const char* f = std::getenv("DYN_SESSION_IDLE_S")) a->sess_idle_s = std::max(0.001, std::atof(f));
  }
  if (trace_create) std::fprintf(stderr, "[dyn] create: streams at %.1f ms\n", tc_ms());
  if ((e = a->d_model.ensure(sizeof(Emis) * a->model.table.size())) != hipSuccess) return fail(e, "hipMalloc(model)");
  // (uploads on the handle's own non-blocking stream: a null-stream copy would wait for another handle's resident session)
  auto upload = [&](void* dst, const void* src, size_t bytes) {
    hipError_t ue = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, a->stream);
    return ue != hipSuccess ? ue : hipStreamSynchronize(a->stream);
  };
  if ((e = upload(a->d_model.p, a->model.table.data(), sizeof(Emis) * a->model.table.size())) != hipSuccess)
    return fail(e, "hipMemcpy(model)");
  {
    std::vector<dynmath::SoftplusNode> tab(dynmath::SP_NODES + dynmath::EXP128_NODES + dynmath::STRICT_EXP_WORDS / 2);
    dynmath::softplus_build_table(tab.data());
    dynmath::exp128_build_table(reinterpret_cast<double*>(tab.data() + dynmath::SP_NODES));  // 2^(i/128), training sweeps
    std::memcpy(tab.data() + dynmath::SP_NODES + dynmath::EXP128_NODES, dynmath::strict_exp_table(),
                dynmath::STRICT_EXP_WORDS * 8);  // 2^(k/128) of the strict exp
    if ((e = a->d_sptab.ensure(sizeof(dynmath::SoftplusNode) * tab.size())) != hipSuccess) return fail(e, "hipMalloc(softplus table)");
    if ((e = upload(a->d_sptab.p, tab.data(), sizeof(dynmath::SoftplusNode) * tab.size())) != hipSuccess)
      return fail(e, "hipMemcpy(softplus table)");
  }
  if (trace_create) std::fprintf(stderr, "[dyn] create: tables uploaded, done at %.1f ms\n", tc_ms());
  *out = a;
  return DYN_OK;
}

void dyn_aligner_destroy(dyn_aligner* a) {
  if (!a) return;
  const bool trace = std::getenv("DYN_TRACE_HOST") != nullptr;
  if (trace) std::fprintf(stderr, "[dyn] destroy %p: joining the pipeline\n", (void*)a);
  a->pipe.reset();  // drains and joins the pipeline threads
  if (!a->host_only) {
    (void)hipSetDevice(a->device);
    if (trace) std::fprintf(stderr, "[dyn] destroy %p: session open %d, pending %d %d; device synchronise\n", (void*)a, (int)a->sess.open, (int)a->sess.pending[0],
                            (int)a->sess.pending[1]);
    {
      std::lock_guard<std::mutex> lk(a->mu);
      (void)session_quiesce(a);  // (an open session would make the device-wide wait below wait for its idle watchdog)
    }
    (void)hipDeviceSynchronize();
    if (trace) std::fprintf(stderr, "[dyn] destroy %p: device idle; releasing buffers\n", (void*)a);
    a->d_model.release();
    a->d_sptab.release();
    park_pool_buffer(a->device, 0, a->ws);
    park_pool_buffer(a->device, 1, a->lpe);
    park_pool_buffer(a->device, 2, a->bits);
    a->free_list.release();
    a->ctl.release();
    a->h_rows.release();
    a->sess_anchor.release();
    for (int k = 0; k < 2; ++k) {
      a->sess_ctl[k].release();
      a->sess_ring[k].release();
      for (hipEvent_t ev : {a->sess.ev_begin[k], a->sess.ev_end[k]})
        if (ev) (void)hipEventDestroy(ev);
    }
    a->sess_hctl.release();
    if (a->sess_flags) (void)hipHostFree(a->sess_flags);
    a->cache.park(a->device);
    if (trace) std::fprintf(stderr, "[dyn] destroy %p: buffers parked; destroying streams\n", (void*)a);
    park_session_stream(a);
    for (hipStream_t s : {a->stream, a->s_in, a->s_out, a->s_get})
      if (s) (void)hipStreamDestroy(s);
    if (trace) std::fprintf(stderr, "[dyn] destroy %p: done\n", (void*)a);
  }
  delete a;
}

int dyn_aligner_info(const dyn_aligner* a, dyn_info* info) {
  if (!a || !info) return DYN_ERR_INVALID_ARGUMENT;
  info->abi_version = DYN_ABI_VERSION;
  info->pore = a->model.pore;
  info->rna = a->model.rna ? 1 : 0;
  info->kmer_size = a->model.k;
  info->alphabet_size = a->model.alphabet;
  info->device = a->host_only ? DYN_DEVICE_HOST_ONLY : a->device;
  info->num_kmers = a->model.num_kmers;
  info->half_band = a->model.half_band;
  info->log_m1 = a->model.log_m1;
  info->log_e1 = a->model.log_e1;
  info->log_e2 = a->model.log_e2;
  info->max_half_band = dynk::WIDE_MAX_HALF_BAND;
  return DYN_OK;
}

int dyn_aligner_model(const dyn_aligner* a, double* out2n) {
  if (!a || !out2n) return DYN_ERR_INVALID_ARGUMENT;
  for (uint64_t i = 0; i < a->model.num_kmers; ++i) {
    out2n[2 * i] = a->model.mean[i];
    out2n[2 * i + 1] = a->model.stdev[i];
  }
  return DYN_OK;
}

int dyn_aligner_set_model(dyn_aligner* a, const double* in2n) {
  if (!a || !in2n) return DYN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(a->mu);
  dynhost::PoreModel& m = a->model;
  for (uint64_t i = 0; i < m.num_kmers; ++i) {
    m.mean[i] = in2n[2 * i];
    m.stdev[i] = in2n[2 * i + 1];
    m.table[i] = dynmath::make_emis(m.mean[i], m.stdev[i], std::log(m.stdev[i]));  // as PoreModel::load does
  }
  if (a->strict_mode != 0 && !model_allows_strict(m)) a->strict_mode = 0;
  if (!a->host_only) {
    HIP_TRY(a, hipSetDevice(a->device));
    if (int rc = session_quiesce(a)) return rc;  // (a device-wide wait would never return while resident waves are waiting for tickets)
    HIP_TRY(a, hipDeviceSynchronize());  // nothing that reads the old table may still be running
    HIP_TRY(a, hipMemcpy(a->d_model.p, m.table.data(), sizeof(dynmath::Emis) * m.table.size(), hipMemcpyHostToDevice));
  }
  return DYN_OK;
}

int dyn_aligner_set_mem_budget(dyn_aligner* a, uint64_t bytes) {
  if (!a) return DYN_ERR_INVALID_ARGUMENT;
  a->mem_budget = bytes;
  return DYN_OK;
}

int dyn_aligner_set_strict(dyn_aligner* a, int mode) {
  if (!a || mode < 0 || mode > 2) return DYN_ERR_INVALID_ARGUMENT;
  if (mode != 0 && !model_allows_strict(a->model)) {
    a->last_error = "strict mode: a model stdev with an all-ones significand has no division-free exact quotient";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  if (mode != a->strict_mode && !a->host_only) {
    std::lock_guard<std::mutex> lk(a->mu);
    if (a->sess.open) {  // the resident kernel's variant was chosen for the mode it was opened in
      (void)hipSetDevice(a->device);
      if (int rc = session_quiesce(a)) return rc;
    }
    a->strict_mode = mode;
    return DYN_OK;
  }
  a->strict_mode = mode;
  return DYN_OK;
}

int dyn_aligner_set_train_zcheck(dyn_aligner* a, int on) {
  if (!a) return DYN_ERR_INVALID_ARGUMENT;
  a->train_zcheck = on != 0;
  return DYN_OK;
}

}  // extern "C"

namespace dyneng {

// Which reads need the reference's arithmetic bit for bit, and for how many forward rows (dyn_aligner_set_strict, mode 1).
// A STRUCTURAL TIE: two neighbouring lattice columns with the same emission parameters (equal k-mers -- the polyA pad
// followed by A, any homopolymer of k+1 bases -- or distinct k-mers whose table entries coincide). Moving the border
// between their segments leaves the exact score unchanged, so the traceback's comparison vE == vM_prev + LPE
// (NT_aligner_api.cpp:445-448) is a tie in exact arithmetic wherever the path crosses them, decided by the last bits of
// the reference's sums. Columns are compared by their (mean, stdev) bits, not by k-mer code.
// Returns 0 for a read without such a pair; otherwise the number of forward rows that must be exact: up to the row in
// which the LAST tied pair has left the band for good (no decision is taken on a column outside the band), UINT32_MAX
// when that is the whole read. Z enters every posterior, so a flagged read's backward sweep is exact in full.
uint32_t tie_rows(const dynhost::PoreModel& m, const int32_t* km, uint64_t kc, uint64_t S) {
  int64_t last = -1;
  for (uint64_t j = 0; j + 1 < kc; ++j) {
    const int32_t a = km[j], b = km[j + 1];
    if (a == b || (m.mean[a] == m.mean[b] && m.stdev[a] == m.stdev[b])) last = (int64_t)j;
  }
  if (last < 0) return 0;
  const uint64_t T = S + 1, N = kc + 1;
  const uint64_t bw = std::min<uint64_t>(m.half_band, N / 2);
  // k-mers last, last+1 are lattice columns last+1, last+2. Row t's band starts at size_t(t N/T) - bw (NT_aligner_api.cpp:
  // 100-104): column n is below every later band once t N/T >= n + bw + 1.
  const uint64_t n = (uint64_t)last + 2;
  const double ratio = (double)N / (double)T;
  const double t_out = std::ceil((double)(n + bw + 1) / ratio) + 1.0;
  if (!(t_out < (double)(T - 1))) return 0xffffffffu;
  return (uint32_t)t_out;
}

}  // namespace dyneng

extern "C" {

uint32_t dyn_tie_rows(const dyn_aligner* a, const int32_t* kmers, uint64_t n_kmers, uint64_t signal_len) {
  if (!a || (!kmers && n_kmers)) return 0;
  return dyneng::tie_rows(a->model, kmers, n_kmers, signal_len);
}

const char* dyn_aligner_last_error(const dyn_aligner* a) { return a ? a->last_error.c_str() : ""; }

int dyn_read_strerror(int read_status, char bad_char, char* buf, uint64_t cap) {
  std::string s;
  switch (read_status) {
    case DYN_READ_OK: s = ""; break;
    case DYN_READ_SIGNAL_EMPTY: s = "Signal is empty"; break;
    case DYN_READ_SEQ_SHORT: s = "Sequence shorter than model kmer size"; break;
    case DYN_READ_SIGNAL_SHORT: s = "Signal too short compared to sequence"; break;
    case DYN_READ_INVALID_NT: s = std::string("Invalid nucleotide: ") + bad_char; break;
    case DYN_READ_Z_MISMATCH: s = "Alignment failed: alignment scores do not match"; break;
    case DYN_READ_TRAIN_Z_MISMATCH: s = "Training failed: alignment scores do not match"; break;
    case DYN_READ_INTERNAL: s = "Traceback left the lattice"; break;
    case DYN_READ_TOO_LARGE: s = "Read too large for the device memory budget"; break;
    case DYN_READ_NTK_MISMATCH: s = "NTK alignment failed: alignment scores do not match"; break;
    case DYN_READ_BAD_SIGNAL: s = "Signal could not be decoded"; break;
    case DYN_READ_BAND_TOO_WIDE: s = "Band wider than this build's 4096 band columns for a read of this length"; break;
    default: copy_msg(buf, cap, "unknown read status"); return DYN_ERR_INVALID_ARGUMENT;
  }
  copy_msg(buf, cap, s);
  return DYN_OK;
}

uint64_t dyn_segment_capacity(const dyn_aligner* a, uint64_t n_reads, const uint64_t* seq_offsets) {
  uint64_t cap = 0;
  for (uint64_t i = 0; i < n_reads; ++i) {
    const uint64_t L = seq_offsets[i + 1] - seq_offsets[i];
    if (L >= (uint64_t)a->model.k) cap += L - (uint64_t)a->model.k + 1;
  }
  return cap;
}

int dyn_validate_batch(const dyn_aligner* a, uint64_t n_reads, const uint64_t* sig_offsets,
                       const char* seqs, const uint64_t* seq_offsets, int32_t* status,
                       char* bad_char, int32_t* kmers_out, uint64_t kmers_cap) {
  if (!a) return DYN_ERR_INVALID_ARGUMENT;
  dyn_batch b;  // scratch, host memory only
  int rc = host_prepare(&b, a->model, false, n_reads, sig_offsets, seqs, seq_offsets, nullptr);
  if (rc == DYN_OK) {
    for (uint64_t i = 0; i < n_reads; ++i) {
      if (status) status[i] = b.reads[i].status;
      if (bad_char) bad_char[i] = b.reads[i].bad;
      if (kmers_out && b.reads[i].status == DYN_READ_OK) {
        // k-mers of read i are written at the read's segment offset (capacity layout)
        if (b.reads[i].seg_off + b.reads[i].kc > kmers_cap) {
          rc = DYN_ERR_INVALID_ARGUMENT;
          break;
        }
        std::memcpy(kmers_out + b.reads[i].seg_off, b.kmers() + b.reads[i].flat_off, sizeof(int32_t) * b.reads[i].kc);
      }
    }
  }
  std::free(b.h_kmers.p);
  b.h_kmers.p = nullptr;
  return rc;
}

void* dyn_host_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void dyn_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

}  // extern "C"

namespace {
int create_impl(dyn_aligner* a, uint64_t n_reads, const double* signals, const RawSource* rs,
                const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                dyn_batch** out) {
  if (!a || !out) return DYN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(a->mu);
  int rc = need_device(a);
  if (rc != DYN_OK) return rc;
  dyn_batch* b = new dyn_batch();
  b->a = a;
  attach_cache(b);
  // DYN_TRACE_HOST=1: wall time of the host stages of batch creation on stderr (tools/batch_latency.py)
  static const bool trace_host = std::getenv("DYN_TRACE_HOST") != nullptr;
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now_ms();
  auto cleanup = [&](int code) {
    dyn_batch_destroy(b);
    return code;
  };
  rc = host_prepare(b, a->model, true, n_reads, sig_offsets, seqs, seq_offsets, nullptr);
  if (rc != DYN_OK) return cleanup(rc);
  const double t_prepared = now_ms();
#define B_TRY(expr)                                                                        \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      a->last_error = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr;   \
      return cleanup(_e == hipErrorOutOfMemory ? DYN_ERR_OUT_OF_MEMORY : DYN_ERR_DEVICE);  \
    }                                                                                      \
  } while (0)
  const uint64_t total_sig = n_reads ? sig_offsets[n_reads] - sig_offsets[0] : 0;
  rc = alloc_batch_buffers(b, total_sig);
  if (rc != DYN_OK) return cleanup(rc);
  DevBuf d_raw, d_norm, d_offs, d_shift, d_scale;  // preprocessing scratch, back to the cache on return
  for (DevBuf* d : {&d_raw, &d_norm, &d_offs, &d_shift, &d_scale}) d->cache = &a->cache;
  struct Scratch {
    DevBuf* v[5];
    ~Scratch() { for (DevBuf* d : v) d->release(); }
  } scratch{{&d_raw, &d_norm, &d_offs, &d_shift, &d_scale}};
  if (!rs) {
    if (total_sig) B_TRY(hipMemcpyAsync(b->d_sig.p, signals + sig_offsets[0], total_sig * 8, hipMemcpyHostToDevice, a->stream));
  } else if (total_sig) {
    // P1/P2 on the device (segment.py:146-153): upload the raw slices, normalise, Hampel-filter
    const size_t esz = rs->elem_size();
    std::vector<uint64_t> offs(n_reads + 1);
    uint64_t max_len = 0;
    for (uint64_t i = 0; i <= n_reads; ++i) offs[i] = sig_offsets[i] - sig_offsets[0];
    for (uint64_t i = 0; i < n_reads; ++i) max_len = std::max(max_len, offs[i + 1] - offs[i]);
    B_TRY(d_raw.ensure(total_sig * esz));
    B_TRY(d_norm.ensure(total_sig * (rs->compute_f32 ? 4 : 8)));
    B_TRY(d_offs.ensure((n_reads + 1) * 8));
    B_TRY(d_shift.ensure(n_reads * 8));
    B_TRY(d_scale.ensure(n_reads * 8));
    B_TRY(hipMemcpyAsync(d_raw.p, (const char*)rs->raw + sig_offsets[0] * esz, total_sig * esz, hipMemcpyHostToDevice, a->stream));
    B_TRY(hipMemcpyAsync(d_offs.p, offs.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, a->stream));
    B_TRY(hipMemcpyAsync(d_shift.p, rs->shift, n_reads * 8, hipMemcpyHostToDevice, a->stream));
    B_TRY(hipMemcpyAsync(d_scale.p, rs->scale, n_reads * 8, hipMemcpyHostToDevice, a->stream));
    B_TRY(hipStreamSynchronize(a->stream));  // offs is a local vector
    dynk::launch_preprocess(d_raw.p, rs->dtype, rs->compute_f32, d_offs.as<uint64_t>(), d_shift.as<double>(),
                            d_scale.as<double>(), nullptr, nullptr, d_norm.p, b->d_sig.as<double>(), (int)n_reads, max_len, rs->window,
                            rs->n_sigmas, a->stream);
    B_TRY(hipGetLastError());
  }
  if (b->total_cols) {
    B_TRY(hipMemcpyAsync(b->d_kmers.p, b->h_kmers.p, b->total_cols * 4, hipMemcpyHostToDevice, a->stream));
    dynk::launch_prep_params(b->d_kmers.as<int32_t>(), a->d_model.as<Emis>(), b->d_par.as<Emis>(), b->total_cols, (uint32_t)a->model.num_kmers, a->stream);
    B_TRY(hipGetLastError());
  }
  const double t_enqueued = now_ms();
  B_TRY(hipStreamSynchronize(a->stream));  // the scratch buffers go back to the cache on return
#undef B_TRY
  if (trace_host)
    std::fprintf(stderr, "[dyn] batch create: validate+encode %.2f ms, alloc+enqueue %.2f ms, drain %.2f ms\n",
                 t_prepared - t_start, t_enqueued - t_prepared, now_ms() - t_enqueued);
  *out = b;
  return DYN_OK;
}
}  // namespace

extern "C" {

int dyn_batch_create(dyn_aligner* a, uint64_t n_reads, const double* signals,
                     const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                     dyn_batch** out) {
  return create_impl(a, n_reads, signals, nullptr, sig_offsets, seqs, seq_offsets, out);
}

int dyn_batch_create_raw(dyn_aligner* a, uint64_t n_reads, const void* raw, int raw_dtype,
                         const uint64_t* raw_offsets, const double* shift, const double* scale,
                         int hampel_window, double hampel_n_sigmas, int compute_f32, const char* seqs,
                         const uint64_t* seq_offsets, dyn_batch** out) {
  if (!a) return DYN_ERR_INVALID_ARGUMENT;
  if (raw_dtype < 0 || raw_dtype > 2 || hampel_window < 1 || hampel_window > 16 || !raw || !shift || !scale) {
    a->last_error = "dyn_batch_create_raw: raw_dtype must be 0 (f32), 1 (i16) or 2 (f64) and 1 <= window <= 16";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  RawSource rs;
  rs.raw = raw;
  rs.dtype = raw_dtype;
  rs.shift = shift;
  rs.scale = scale;
  rs.window = hampel_window;
  rs.n_sigmas = hampel_n_sigmas;
  rs.compute_f32 = compute_f32;
  return create_impl(a, n_reads, nullptr, &rs, raw_offsets, seqs, seq_offsets, out);
}

// Normalised + filtered signals of a batch created with dyn_batch_create_raw (tests, debugging).
int dyn_batch_signals(dyn_batch* b, double* out, uint64_t count) {
  if (!b || !out) return DYN_ERR_INVALID_ARGUMENT;
  dyn_aligner* a = b->a;
  int rc = need_device(a);
  if (rc != DYN_OK) return rc;
  if (b->group && b->group->g) {  // a member of a merged launch: its samples are a slice of the group's signal pool
    const dyn_batch* g = b->group->g;
    const uint64_t first = b->n ? g->reads[b->g_read0].sig_off : 0;
    if ((first + count) * 8 > g->d_sig.bytes) return DYN_ERR_INVALID_ARGUMENT;
    HIP_TRY(a, copy_out(a, out, g->d_sig.as<double>() + first, count * 8));
    return DYN_OK;
  }
  if (count * 8 > b->d_sig.bytes) return DYN_ERR_INVALID_ARGUMENT;
  HIP_TRY(a, copy_out(a, out, b->d_sig.p, count * 8));
  return DYN_OK;
}

void dyn_batch_destroy(dyn_batch* b) {
  if (!b) return;
  if (b->async && b->a && b->a->pipe) (void)b->a->pipe->wait(b);  // returns at once when the batch is done
  if (b->a && !b->a->host_only) (void)hipSetDevice(b->a->device);
  for (DevBuf* d : {&b->d_sig, &b->d_kmers, &b->d_par, &b->d_state, &b->d_rows, &b->d_segrow, &b->d_medhi,
                    &b->d_medlo, &b->d_descs, &b->d_colw, &b->d_cols1, &b->d_cols2, &b->d_trans, &b->d_pooled, &b->d_poolwork, &b->d_pooltemp, &b->d_pp,
                    &b->d_pathn, &b->d_norm, &b->d_meta, &b->d_tctl})
    d->release();
  for (PinnedBuf* h : {&b->h_kmers, &b->h_descs, &b->h_state, &b->h_rows, &b->h_stats, &b->h_sig}) h->release();
  for (hipEvent_t e : b->events) (void)hipEventDestroy(e);
  for (hipEvent_t e : {b->ev_in, b->ev_done, b->ev_out})
    if (e) (void)hipEventDestroy(e);
  delete b;
}

}  // extern "C"

namespace dyneng {

// ---- queue planning for page-starved launches ---------------------------------------------------
// The persistent waves take reads off the queue in order; a wave keeps its arena and exchanges pages with
// the pool only when its next read needs more (it then waits with no pages until the pool can serve it)
// or much less while somebody waits. Every wave sweeps rows at the same rate, so the whole launch can be
// replayed on the host: simulate_queue returns the makespan in rows for a given queue order.
static uint64_t simulate_queue(const std::vector<uint32_t>& need, const std::vector<uint64_t>& rows, size_t n_slots,
                               uint64_t pool) {
  const size_t n = need.size();
  struct Ev { uint64_t t; uint32_t slot; bool operator>(const Ev& o) const { return t > o.t; } };
  std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> events;
  struct Wait { uint32_t slot, need; size_t idx; };
  std::vector<Wait> waiting;
  std::vector<uint32_t> have(n_slots, 0);
  uint64_t free_pages = pool, end = 0;
  size_t head = 0;
  auto serve = [&](uint64_t now) {
    for (size_t i = 0; i < waiting.size();) {
      if (free_pages >= waiting[i].need) {
        free_pages -= waiting[i].need;
        have[waiting[i].slot] = waiting[i].need;
        events.push(Ev{now + rows[waiting[i].idx], waiting[i].slot});
        waiting.erase(waiting.begin() + i);
      } else {
        ++i;
      }
    }
  };
  bool reserving = true;
  for (size_t s = 0; s < n_slots && s < n; ++s) {  // first round: pages reserved by the host while they last
    head = s + 1;
    if (reserving && free_pages >= need[s]) {
      free_pages -= need[s];
      have[s] = need[s];
      events.push(Ev{rows[s], (uint32_t)s});
    } else {
      reserving = false;
      waiting.push_back(Wait{(uint32_t)s, need[s], s});
    }
  }
  while (!events.empty()) {
    const Ev e = events.top();
    events.pop();
    end = std::max(end, e.t);
    uint32_t& hv = have[e.slot];
    if (head < n) {
      const size_t idx = head++;
      if (hv >= need[idx]) {
        if (!waiting.empty() && hv - need[idx] >= 8 && 8 * (hv - need[idx]) >= hv) {
          free_pages += hv - need[idx];
          hv = need[idx];
          serve(e.t);
        }
        events.push(Ev{e.t + rows[idx], e.slot});
      } else {
        free_pages += hv;
        hv = 0;
        waiting.push_back(Wait{e.slot, need[idx], idx});
        serve(e.t);
      }
    } else {
      free_pages += hv;
      hv = 0;
      serve(e.t);
    }
  }
  return waiting.empty() ? end : ~0ull;  // a plan that strands a read is no plan
}

// `order` comes in longest first. When the pool cannot hold a lattice for every wave slot, longest-first
// leaves the slots beyond the pool's capacity idle until the first long reads finish (config 3: 7.6 % of
// all wave time, measured). Candidate plans give those slots BRIDGE reads -- shorter reads whose arenas fit
// beside L long ones -- and start the displaced long reads when the first round's memory comes back:
//   queue = [ L longest | bridge = ranks [first, last), longest first | everything else, longest first ]
// The shortest quarter of the batch is never used as bridge (it keeps the launch's tail short). The plan
// with the smallest simulated makespan wins; plain longest-first is one of the candidates.
static void plan_queue(std::vector<uint32_t>& order, const std::vector<uint32_t>& need, const std::vector<uint64_t>& rows,
                       size_t n_slots, uint64_t pool) {
  const size_t n = order.size();
  std::vector<uint64_t> pre(n + 1, 0), rpre(n + 1, 0);
  for (size_t k = 0; k < n; ++k) {
    pre[k + 1] = pre[k] + need[k];
    rpre[k + 1] = rpre[k] + rows[k];
  }
  size_t L0 = 0;
  while (L0 < n_slots && pre[L0 + 1] <= pool) ++L0;
  if (L0 >= n_slots || L0 < 2) return;  // every slot gets its lattice (or nothing sensible to plan)
  const size_t lo_rank = n - n / 4;
  auto permute = [&](size_t L, size_t first, size_t last, std::vector<uint32_t>& nd, std::vector<uint64_t>& rw,
                     std::vector<uint32_t>* ord) {
    nd.clear();
    rw.clear();
    if (ord) ord->clear();
    auto put = [&](size_t lo, size_t hi) {
      for (size_t k = lo; k < hi; ++k) {
        nd.push_back(need[k]);
        rw.push_back(rows[k]);
        if (ord) ord->push_back(order[k]);
      }
    };
    put(0, L);
    put(first, last);
    put(L, first);
    put(last, n);
  };
  std::vector<uint32_t> nd;
  std::vector<uint64_t> rw;
  uint64_t best = simulate_queue(need, rows, n_slots, pool);
  size_t bL = 0, bfirst = 0, blast = 0;
  const size_t step = std::max<size_t>(1, n_slots / 32);
  for (size_t L = L0; L + step > step && L >= n_slots / 4; L -= step) {
    const uint64_t per_slot = (pool - pre[L]) / (n_slots - L);
    size_t first = std::lower_bound(need.begin() + L, need.begin() + lo_rank, per_slot,
                                    [](uint32_t a, uint64_t v) { return a > v; }) - need.begin();  // first rank that fits
    if (lo_rank - first < n_slots - L) continue;
    const uint64_t target = (uint64_t)(n_slots - L) * rows[L - 1];
    for (int f = 2; f <= 6; ++f) {  // bridge rows = 0.5 .. 1.5 x "one long read per bridge slot"
      size_t last = std::lower_bound(rpre.begin() + first, rpre.begin() + lo_rank, rpre[first] + target * f / 4) - rpre.begin();
      last = std::min(std::max(last, first + (n_slots - L)), lo_rank);
      permute(L, first, last, nd, rw, nullptr);
      const uint64_t t = simulate_queue(nd, rw, n_slots, pool);
      if (t < best) {
        best = t;
        bL = L;
        bfirst = first;
        blast = last;
      }
    }
  }
  if (bL) {
    std::vector<uint32_t> planned;
    permute(bL, bfirst, blast, nd, rw, &planned);
    order.swap(planned);
  }
}

}  // namespace dyneng

namespace dyneng {
// SPREAD (paged sessions; `order` comes in longest first): in a stream of tickets the waves never start together, so what
// matters is that any ~n_waves consecutive reads ask for about the AVERAGE number of pages (config 3: 205 GB against a 250 GB
// pool) instead of the maximum (370 GB for the 1 024 longest). The longer (tail_div - 1) / tail_div of the reads are dealt out
// in a low-discrepancy order (rank k * phi mod m); the shortest 1 / tail_div follow, longest first, so that a ticket nobody
// follows still ends on short reads (tail_div 0: every read is spread). Config 3, same box: tail 1/8 507, 1/4 499-504, 1/2 493,
// none 509; the planned order (plan_queue) 465; one planned launch per batch 446-457 Msamp/s.
constexpr int SESSION_TAIL_DIV = 8;
static void spread_order(std::vector<uint32_t>& order, int tail_div) {
  const size_t n = order.size();
  const size_t m = tail_div > 0 ? n - n / (size_t)tail_div : n;
  if (m < 3) return;
  size_t step = (size_t)((double)m * 0.6180339887498949) | 1;
  auto gcd = [](size_t x, size_t y) { while (y) { const size_t t = x % y; x = y; y = t; } return x; };
  while (gcd(step, m) != 1) step += 2;
  std::vector<uint32_t> spread(order);
  for (size_t k = 0; k < m; ++k) spread[k] = order[(k * step) % m];
  order.swap(spread);
}
}  // namespace dyneng

extern "C" int dyn_session_order(uint64_t n_reads, uint32_t* order_out) {
  if (!order_out) return DYN_ERR_INVALID_ARGUMENT;
  std::vector<uint32_t> order(n_reads);
  for (uint64_t k = 0; k < n_reads; ++k) order[k] = (uint32_t)k;
  dyneng::spread_order(order, dyneng::SESSION_TAIL_DIV);
  if (n_reads) std::memcpy(order_out, order.data(), n_reads * sizeof(uint32_t));  // (an empty vector's data() may be null)
  return DYN_OK;
}

extern "C" int dyn_plan_queue(uint64_t n_reads, const uint32_t* pages, const uint64_t* rows, uint64_t n_slots,
                              uint64_t pool_pages, uint32_t* order_out, uint64_t* makespan_longest_first,
                              uint64_t* makespan_planned) {
  if (!pages || !rows || !order_out || !n_slots) return DYN_ERR_INVALID_ARGUMENT;
  std::vector<uint32_t> need(pages, pages + n_reads), order(n_reads);
  std::vector<uint64_t> rw(rows, rows + n_reads);
  for (uint64_t k = 0; k < n_reads; ++k) {
    order[k] = (uint32_t)k;
    if (k && need[k] > need[k - 1]) return DYN_ERR_INVALID_ARGUMENT;  // longest first
  }
  if (makespan_longest_first) *makespan_longest_first = dyneng::simulate_queue(need, rw, n_slots, pool_pages);
  if (n_reads > n_slots) dyneng::plan_queue(order, need, rw, n_slots, pool_pages);
  if (makespan_planned) {
    std::vector<uint32_t> nd(n_reads);
    std::vector<uint64_t> r2(n_reads);
    for (uint64_t k = 0; k < n_reads; ++k) {
      nd[k] = need[order[k]];
      r2[k] = rw[order[k]];
    }
    *makespan_planned = dyneng::simulate_queue(nd, r2, n_slots, pool_pages);
  }
  std::memcpy(order_out, order.data(), n_reads * sizeof(uint32_t));
  return DYN_OK;
}

namespace dyneng {

// Shared engine of align / train. Every ok read of the batch goes, longest first, into ONE launch of
// persistent waves (k_read_queue): a wave runs a read's whole pipeline and then takes the next read
// off the queue. The lattice of a read lives in pages of a pool that only has to hold the reads in
// flight (at most 4 per CU); the pages of the first round are reserved here, later reads take theirs
// from the pool's free list on the device. Everything is ENQUEUED on the handle's compute stream
// without a host synchronisation; host-side inputs of the launches (read descriptors, initial
// per-read state) live in pinned per-batch buffers until the batch is destroyed.
int enqueue_job(dyn_batch* b, DynJob job) {
  dyn_aligner* a = b->a;
  const bool lattice = job != DynJob::AlignZ;
  const bool calc = job == DynJob::AlignFull;
  const PoreModel& m = a->model;
  const int z_fail = job == DynJob::Train ? DYN_READ_TRAIN_Z_MISMATCH : DYN_READ_Z_MISMATCH;

  // mode "resquiggle"/"ntk": what the reference's NTKAligner does in this snapshot, as observed with the compiled
  // reference (tests/golden/g11_ntk_messages.json): validateInput / sequenceToKmers errors first, then EVERY read fails
  // its Zf/Zb check (NTK_aligner_api.cpp:911-917), and train() is the base class's "not implemented" (aligner.cpp:38-44).
  // No kernel runs; the reads get the per-read status whose message is the reference's exception text.
  if (a->ntk && job == DynJob::Train) {
    a->last_error = "Training is not implemented for this aligner";
    return DYN_ERR_RUNTIME;
  }
  // one launch per batch on the compute stream: the lattice pool must not be in the hands of resident waves
  if (int rc = session_quiesce(a)) return rc;
  // Strict reads (align(calc=true) only; dyn_aligner_set_strict) take the sweeps whose every sum is certified to be the
  // reference's bit for bit (dp_math_strict.hpp). Mode 2: every read, every row. Mode 1 (the default): the reads that carry
  // a structural tie (tie_rows above) -- their backward sweep in full and their forward sweep up to the row in which the
  // last tied column pair has left the band: the Viterbi values of a row depend on forward values of earlier rows only,
  // so every decision up to that row is the reference's own; later decisions have the ordinary >= 1e-6 margins. Strict
  // reads run in the SAME launch as the others (a per-read flag, kernel variant k_read_queue<JOB, true>).
  // Queue order: most expensive reads first, so that the tail of the launch is made of the cheapest ones.
  // Cost of a certified row relative to a default one (ISA instruction counts of the row loops, confirmed on the device:
  // profiles/r04/strict_mode_cost.json): backward 1.3, forward 1.4; a read spends 0.4 / 0.6 of its time in the two sweeps.
  const int32_t* km = b->kmers();
  std::vector<uint32_t> strict_rows(b->n, 0);
  std::vector<uint32_t> order, wide;
  uint64_t n_strict = 0;
  for (uint64_t i = 0; i < b->n; ++i) {
    const HostRead& r = b->reads[i];
    if (r.status != DYN_READ_OK) continue;
    if (a->ntk) continue;  // no read reaches the device; its status is set below
    if (r.wide) {  // the generic kernel (wide_band.hip): the reference's own arithmetic in every cell, no queue, no pages
      wide.push_back((uint32_t)i);
      continue;
    }
    if (calc && a->strict_mode == 2) strict_rows[i] = 0xffffffffu;
    else if (calc && a->strict_mode == 1) strict_rows[i] = tie_rows(a->model, km + r.flat_off, r.kc, r.S);
    n_strict += strict_rows[i] != 0;
    order.push_back((uint32_t)i);
  }
  auto is_strict = [&](uint32_t i) { return strict_rows[i] != 0; };
  auto cost_rows_strict = [&](uint32_t i) -> uint64_t {
    const uint64_t T = b->reads[i].S + 1, fr = std::min<uint64_t>(T, strict_rows[i]);
    return (T * 100 + T * 12 + fr * 24) / 100;  // 0.4 T x 1.3 + 0.6 (T + 0.4 fr) = T (1 + 0.12) + 0.24 fr
  };
  auto cost_rows = [&](uint32_t i) { return is_strict(i) ? cost_rows_strict(i) : b->reads[i].S + 1; };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cost_rows(x) > cost_rows(y); });

  if (calc) {
    HIP_TRY(a, b->d_segrow.ensure(std::max<uint64_t>(4, b->capacity * 4)));
    HIP_TRY(a, b->d_medhi.ensure(std::max<uint64_t>(8, b->capacity * 8)));
    HIP_TRY(a, b->d_medlo.ensure(std::max<uint64_t>(8, b->capacity * 8)));
  }
  if (job == DynJob::Train) {
    HIP_TRY(a, b->d_colw.ensure(std::max<uint64_t>(8, b->total_cols * 8)));
    HIP_TRY(a, b->d_cols1.ensure(std::max<uint64_t>(8, b->total_cols * 8)));
    HIP_TRY(a, b->d_cols2.ensure(std::max<uint64_t>(8, b->total_cols * 8)));
    HIP_TRY(a, b->d_trans.ensure(std::max<uint64_t>(16, b->n * 16)));
    // (the DEVICE-resident pooled statistics -- a radix sort and a segmented sum behind every training launch, 2.4 % of it --
    //  have one reader, dyn_batch_device_pooled for the multi-GPU all-reduce: they are computed when it asks, round 5)
    b->pooled_on_device = false;
  }

  // HBM budget for the page pool
  uint64_t budget = a->mem_budget;
  if (lattice) {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(a, hipMemGetInfo(&free_b, &total_b));
    // the handle's own pool and the pools destroyed handles have parked on this device are not "free", but they are
    // this launch's to use: without the parked share the budget of a second handle depended on the process's history
    const uint64_t pool = a->ws.bytes + a->lpe.bytes + a->bits.bytes + parked_bytes(a->device);
    const uint64_t avail = (uint64_t)((double)(free_b + pool) * 0.90);
    if (budget == 0 || budget > avail) budget = avail;
  }

  // rows per page: the longest read must fit the waves' PT_MAX-entry page tables
  uint32_t max_T = 0;
  for (uint32_t i : order) max_T = std::max<uint32_t>(max_T, (uint32_t)(b->reads[i].S + 1));
  int log_r = 8;
  while (((uint64_t)max_T + 1 + ((1ull << log_r) - 1)) >> log_r > (uint64_t)dynk::PT_MAX) ++log_r;
  const uint64_t page_rows = 1ull << log_r;
  auto pages_of = [&](uint64_t S) { return (uint32_t)((S + 2 + page_rows - 1) >> log_r); };  // rows 0 .. T = S+1
  const size_t n_slots = std::min<size_t>(order.size(), (size_t)a->n_cus * dynk::WAVES_PER_CU);
  // pages that keep every wave slot busy: the largest lattices at once (with strict reads in the batch the queue is
  // not in length order, hence the explicit selection)
  auto pages_wanted = [&]() {
    std::vector<uint32_t> pg(order.size());
    for (size_t k = 0; k < order.size(); ++k) pg[k] = pages_of(b->reads[order[k]].S);
    const size_t top = std::min(n_slots, pg.size());
    std::partial_sort(pg.begin(), pg.begin() + top, pg.end(), std::greater<uint32_t>());
    uint64_t w = 0;
    for (size_t k = 0; k < top; ++k) w += pg[k];
    return w;
  };
  uint64_t wanted = pages_wanted();

  // Posterior layout (nt_kernels.hip, forward_sweep): the separate float LPE array makes the forward sweep
  // 17 % faster but costs 12 instead of 8 bytes of HBM per band slot. When the pool cannot hold a
  // separate-layout lattice for every wave slot, waves wait for pages; in place then, if the wider
  // concurrency is worth more than the faster sweep.
  const uint64_t row_sep = (uint64_t)dynk::P * 12 + dynk::CPL * 8, row_inp = (uint64_t)dynk::P * 8 + dynk::CPL * 8;
  bool lpe_separate = calc;
  if (calc) {
    const double c_sep = std::min(1.0, (double)budget / ((double)wanted * page_rows * row_sep + 1.0));
    const double c_inp = std::min(1.0, (double)budget / ((double)wanted * page_rows * row_inp + 1.0));
    if (c_sep < 1.0 && c_inp * 0.92 > c_sep) lpe_separate = false;
    if (const char* f = std::getenv("DYN_FORCE_LAYOUT")) lpe_separate = std::string(f) != "inplace";
  }
  const uint64_t row_bytes = calc ? (lpe_separate ? row_sep : row_inp) : (uint64_t)dynk::P * 8;
  const uint64_t page_bytes = page_rows * row_bytes;

  // per-read state (status of host-side failures is final; ok reads start at 0). A read whose lattice
  // alone exceeds the budget fails on its own (the reference would die of std::bad_alloc for that read
  // only, segment.py:172-176), it does not take the batch with it.
  ReadState* st = b->h_state.as<ReadState>();
  for (uint64_t i = 0; i < b->n; ++i) {
    st[i].Zb = 0.0;
    st[i].Zf = 0.0;
    st[i].status = (a->ntk && b->reads[i].status == DYN_READ_OK) ? DYN_READ_NTK_MISMATCH : b->reads[i].status;
    st[i].n_segments = 0;
  }
  if (lattice) {
    size_t wr = 0;
    for (uint32_t i : order) {
      if ((uint64_t)pages_of(b->reads[i].S) * page_bytes > budget) st[i].status = DYN_READ_TOO_LARGE;
      else order[wr++] = i;
    }
    if (wr != order.size()) {
      order.resize(wr);
      wanted = pages_wanted();
    }
  }
  if (b->n) HIP_TRY(a, hipMemcpyAsync(b->d_state.p, st, b->n * sizeof(ReadState), hipMemcpyHostToDevice, a->stream));
  const size_t n_ok = order.size();

  // the pool: grow-only, shared by every batch of the handle (stream order serialises them)
  dynk::PagePool pool{};
  pool.log_rows = log_r;
  if (lattice && n_ok) {
    const uint64_t cap_pages = budget / page_bytes;
    const uint64_t target = std::min<uint64_t>(wanted, cap_pages);
    const uint64_t ws_pp = page_rows * dynk::P * 8, lpe_pp = page_rows * dynk::P * 4, bits_pp = page_rows * dynk::CPL * 8;
    const bool grow = a->ws.bytes < target * ws_pp || (calc && lpe_separate && a->lpe.bytes < target * lpe_pp) ||
                      (calc && a->bits.bytes < target * bits_pp);
    if (grow) {  // growing releases the old buffer, which earlier work on the compute stream may still be using
      HIP_TRY(a, hipStreamSynchronize(a->stream));
      const double headroom = std::min(1.15, std::max(1.0, (double)cap_pages / (double)std::max<uint64_t>(1, target)));
      HIP_TRY(a, ensure_pool(a->device, a->ws, target * ws_pp, a->lpe, (calc && lpe_separate) ? target * lpe_pp : 0, a->bits,
                             calc ? target * bits_pp : 0, headroom));
    }
    uint64_t n_pages = std::min<uint64_t>(a->ws.bytes / ws_pp, cap_pages);  // (a buffer taken over from a parked pool may exceed this handle's budget)
    if (calc && lpe_separate) n_pages = std::min<uint64_t>(n_pages, a->lpe.bytes / lpe_pp);
    if (calc) n_pages = std::min<uint64_t>(n_pages, a->bits.bytes / bits_pp);
    n_pages = std::min<uint64_t>(n_pages, 0xfffffff0ull >> log_r);  // pool rows are 32-bit
    if (a->free_list.bytes < n_pages * 4) {
      HIP_TRY(a, hipStreamSynchronize(a->stream));
      HIP_TRY(a, a->free_list.ensure(n_pages * 4, 1.0));
    }
    pool.ws = a->ws.as<double>();
    pool.lpe = (calc && lpe_separate) ? a->lpe.as<float>() : nullptr;
    pool.bits = calc ? a->bits.as<uint64_t>() : nullptr;
    pool.free_list = a->free_list.as<uint32_t>();
    pool.n_pages = (uint32_t)n_pages;
  }
  HIP_TRY(a, a->ctl.ensure(dynk::QUEUE_CTL_WORDS * 4, 1.0));
  pool.ctl = a->ctl.as<uint32_t>();

  // Page-starved launches: the queue order is planned (plan_queue above); `rows` is each read's duration. The planner
  // works on a queue in LENGTH order (pages descending). A launch with strict reads is in cost order: when its first
  // round does not fit the pool it goes back to length order first -- a strict read costs 1.2x a plain one of its length,
  // which matters far less than idle slots in a starved launch (config 3 holds ~50 tie reads in 4 096; round 4's first
  // builds skipped the planner for such launches).
  if (lattice && order.size() > n_slots && !std::getenv("DYN_NO_BRIDGE")) {
    bool plan = true;
    if (n_strict) {
      uint64_t first_round = 0;
      for (size_t k = 0; k < n_slots; ++k) first_round += pages_of(b->reads[order[k]].S);
      plan = first_round > pool.n_pages;
      if (plan)
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return b->reads[x].S > b->reads[y].S; });
    }
    if (plan) {
      std::vector<uint32_t> need(order.size());
      std::vector<uint64_t> rows(order.size());
      for (size_t k = 0; k < order.size(); ++k) {
        need[k] = pages_of(b->reads[order[k]].S);
        rows[k] = cost_rows(order[k]);
      }
      plan_queue(order, need, rows, n_slots, pool.n_pages);
    }
  }

  // (Dealing the first round's strict reads out across the CUs instead of four to a CU was measured: 50.6 vs 50.9 ms on
  //  cfg2 with 26 % tie reads -- the certified sweeps do not get in each other's way inside a CU. Not kept.)

  // read descriptors in processing order; pages of the first round reserved here
  HIP_TRY(a, b->h_descs.ensure(std::max<size_t>(sizeof(ReadDesc), (n_ok + wide.size()) * sizeof(ReadDesc))));
  ReadDesc* descs = b->h_descs.as<ReadDesc>();
  dyn_timing tm{};
  uint64_t rows_total = 0;
  uint32_t used_pages = 0, n_static = 0, max_N = 0;
  bool reserving = true;
  for (size_t k = 0; k < order.size(); ++k) {
    const uint32_t i = order[k];
    const HostRead& r = b->reads[i];
    ReadDesc d{};
    d.T = (uint32_t)(r.S + 1);
    d.N = (uint32_t)(r.kc + 1);
    d.bw = (uint32_t)std::min<uint64_t>(m.half_band, d.N / 2);
    d.read = i;
    d.ratio = (double)d.N / (double)d.T;
    d.sig_off = r.sig_off;
    d.par_off = r.flat_off;
    d.seg_off = r.seg_off;
    d.path_off = rows_total;
    d.n_pages = lattice ? pages_of(r.S) : 0;
    d.first_page = dynk::NO_PAGE;
    d.flags = !is_strict(i) ? 0u : strict_rows[i] == 0xffffffffu ? dynk::READ_STRICT : dynk::READ_STRICT_START;
    d.strict_rows = strict_rows[i];
    if (reserving && k < n_slots && (!lattice || (uint64_t)used_pages + d.n_pages <= pool.n_pages)) {
      d.first_page = lattice ? used_pages : 0;
      used_pages += d.n_pages;
      n_static = (uint32_t)(k + 1);
    } else {
      reserving = false;  // later reads get their pages on the device
    }
    rows_total += d.T;
    max_N = std::max(max_N, d.N);
    descs[k] = d;
    tm.cells += (uint64_t)d.T * std::min<uint64_t>(2ull * d.bw + 1, d.N);
    tm.samples += r.S;
  }
  // wide-band reads: their descriptors FOLLOW the queue's (the read queue sees the first n_ok, the per-segment kernels all)
  uint64_t wide_arena = 0;
  int wide_groups = 0;
  if (!wide.empty()) {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(a, hipMemGetInfo(&free_b, &total_b));
    const uint64_t room = (uint64_t)((double)(free_b + b->d_wide.bytes + parked_bytes(a->device)) * 0.8);
    size_t wr = 0;
    for (uint32_t i : wide) {
      const HostRead& r = b->reads[i];
      const uint64_t need = dynk::wide_arena_bytes(r.S + 1, std::min<uint64_t>(m.half_band, (r.kc + 1) / 2), calc);
      if (need > room) st[i].status = DYN_READ_TOO_LARGE;
      else {
        wide_arena = std::max(wide_arena, need);
        wide[wr++] = i;
      }
    }
    if (wr != wide.size()) {
      wide.resize(wr);
      if (b->n) HIP_TRY(a, hipMemcpyAsync(b->d_state.p, st, b->n * sizeof(ReadState), hipMemcpyHostToDevice, a->stream));
    }
    if (!wide.empty()) wide_groups = (int)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)wide.size(), (uint64_t)a->n_cus, room / wide_arena}));
  }
  const size_t n_all = n_ok + wide.size();
  for (size_t k = 0; k < wide.size(); ++k) {
    const uint32_t i = wide[k];
    const HostRead& r = b->reads[i];
    ReadDesc d{};
    d.T = (uint32_t)(r.S + 1);
    d.N = (uint32_t)(r.kc + 1);
    d.bw = (uint32_t)std::min<uint64_t>(m.half_band, d.N / 2);
    d.read = i;
    d.ratio = (double)d.N / (double)d.T;
    d.sig_off = r.sig_off;
    d.par_off = r.flat_off;
    d.seg_off = r.seg_off;
    d.path_off = rows_total;
    d.first_page = dynk::NO_PAGE;
    rows_total += d.T;
    max_N = std::max(max_N, d.N);
    descs[n_ok + k] = d;
    tm.cells += (uint64_t)d.T * std::min<uint64_t>(2ull * d.bw + 1, d.N);
    tm.samples += r.S;
  }
  if (calc) {
    HIP_TRY(a, b->d_pp.ensure(std::max<uint64_t>(8, rows_total * 8)));
    HIP_TRY(a, b->d_pathn.ensure(std::max<uint64_t>(4, rows_total * 4)));
  }
  HIP_TRY(a, b->d_descs.ensure(std::max<size_t>(sizeof(ReadDesc), n_all * sizeof(ReadDesc))));
  if (n_all)
    HIP_TRY(a, hipMemcpyAsync(b->d_descs.p, descs, n_all * sizeof(ReadDesc), hipMemcpyHostToDevice, a->stream));
  HIP_TRY(a, b->h_stats.ensure(dynk::QUEUE_CTL_WORDS * 4));
  std::memset(b->h_stats.p, 0, dynk::QUEUE_CTL_WORDS * 4);

  while (b->events.size() < 3) {
    hipEvent_t e = nullptr;
    HIP_TRY(a, hipEventCreate(&e));
    b->events.push_back(e);  // owned by the batch from here on: destroyed with it whatever happens next
  }
  hipEvent_t* ev = b->events.data();
  const int nr = (int)n_ok;
  dynk::QueueArgs q{};
  q.descs = b->d_descs.as<ReadDesc>();
  q.n_reads = nr;
  q.n_static = (int)n_static;
  q.sig = b->d_sig.as<double>();
  q.par = b->d_par.as<Emis>();
  q.pool = pool;
  q.st = b->d_state.as<ReadState>();
  q.tb = dynk::TraceBuffers{b->d_pp.as<double>(), b->d_pathn.as<uint32_t>(), b->d_segrow.as<uint32_t>(),
                            b->d_medhi.as<double>(), b->d_medlo.as<double>()};
  q.tr = dynk::TrainBuffers{b->d_colw.as<double>(), b->d_cols1.as<double>(), b->d_cols2.as<double>(), b->d_trans.as<double>()};
  q.m1 = m.log_m1;
  q.e2 = m.log_e2;
  q.sp_tab = a->d_sptab.as<dynmath::SoftplusNode>();
  q.z_fail_status = z_fail;
  const dynk::QueueJob qjob = job == DynJob::Train ? (a->train_zcheck ? dynk::JOB_TRAIN_ZCHECK : dynk::JOB_TRAIN)
                              : !calc              ? dynk::JOB_Z
                              : lpe_separate       ? dynk::JOB_ALIGN
                                                   : dynk::JOB_ALIGN_INPLACE;
  if (!b->ev_done) HIP_TRY(a, hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming));
  dynk::launch_pool_init(pool, used_pages, (int)n_static, a->stream);
  HIP_TRY(a, hipEventRecord(ev[0], a->stream));
  dynk::launch_read_queue(qjob, n_strict != 0, q, a->n_cus, a->stream);
  HIP_TRY(a, hipEventRecord(ev[1], a->stream));
  // the statistics leave the control words before the next batch's k_pool_init resets them (same stream)
  HIP_TRY(a, hipMemcpyAsync(b->h_stats.p, pool.ctl, dynk::QUEUE_CTL_WORDS * 4, hipMemcpyDeviceToHost, a->stream));
  // (Running the per-segment kernels on a stream of their own, beside the next batch's read queue, was measured:
  //  the 0.35 ms gap it closes comes back as a 0.4 ms slower start of that read queue -- same-box A/B, no gain.)
  if (!wide.empty()) {
    // one workgroup per wide read at a time, each with a lattice arena for the largest of them; behind the read queue on the
    // compute stream (its results feed the same per-segment kernels / the same host finalisation)
    {
      const uint64_t want = 256 + (uint64_t)wide_groups * wide_arena;
      size_t free_b = 0, total_b = 0;
      HIP_TRY(a, hipMemGetInfo(&free_b, &total_b));
      if (want > b->d_wide.bytes && want > (uint64_t)((double)free_b * 0.95)) free_parked(a->device);  // (counted as room above)
      HIP_TRY(a, b->d_wide.ensure(want));
    }
    dynk::WideArgs wa{};
    wa.descs = q.descs + n_ok;
    wa.n_reads = (int)wide.size();
    wa.sig = q.sig;
    wa.par = q.par;
    wa.st = q.st;
    wa.tb = q.tb;
    wa.tr = q.tr;
    wa.head = b->d_wide.as<uint32_t>();
    wa.arena = b->d_wide.as<char>() + 256;
    wa.arena_bytes = wide_arena;
    wa.exp_tab = reinterpret_cast<const uint64_t*>(a->d_sptab.as<dynmath::SoftplusNode>() + dynmath::SP_NODES + dynmath::EXP128_NODES);
    wa.m1 = m.log_m1;
    wa.e2 = m.log_e2;
    wa.z_fail_status = z_fail;
    dynk::launch_wide_reads(job == DynJob::Train ? 2 : calc ? 1 : 0, wa, wide_groups, a->stream);
  }
  const int nr_all = (int)n_all;
  if (calc) dynk::launch_segments(q.descs, nr_all, rows_total, max_N, q.st, q.tb, b->d_rows.as<SegRow>(), m.k, a->stream);
  if (job == DynJob::Train) {
    b->pool_nr = nr_all;
    b->pool_max_N = max_N;
  }
  HIP_TRY(a, hipEventRecord(ev[2], a->stream));
  HIP_TRY(a, hipEventRecord(b->ev_done, a->stream));
  HIP_TRY(a, hipGetLastError());
  tm.reads_ok = n_all;
  tm.reads_strict = (uint32_t)n_strict;
  tm.launch_share = 1.0;
  b->strict_flag.assign(b->n, 0);
  for (uint64_t i = 0; i < b->n; ++i) b->strict_flag[i] = strict_rows[i] != 0;
  tm.launches = (nr || !wide.empty()) ? 1 : 0;
  tm.lp_inplace = (calc && !lpe_separate) ? 1 : 0;
  tm.pool_pages = pool.n_pages;
  tm.page_rows = (uint32_t)page_rows;
  tm.n_static = n_static;
  tm.n_waves = (uint32_t)std::min<size_t>((order.size() + dynk::WAVES_PER_CU - 1) / dynk::WAVES_PER_CU * dynk::WAVES_PER_CU,
                                         (size_t)a->n_cus * dynk::WAVES_PER_CU);
  b->timing = tm;
  b->n_chunks = (nr || !wide.empty()) ? 1 : 0;
  b->aligned = job != DynJob::Train;
  b->trained = job == DynJob::Train;
  b->last_calc = calc ? 1 : 0;
  return DYN_OK;
}

// After the compute stream has passed the batch (and the statistics copy behind it).
int collect_timing(dyn_batch* b) {
  dyn_aligner* a = b->a;
  dyn_timing& tm = b->timing;
  tm.ms_backward = tm.ms_forward = tm.ms_trace = tm.ms_total = tm.ms_dp = 0.0;
  tm.wave_wait_share = tm.wave_occupancy = 0.0;
  tm.ms_backward_strict = tm.ms_forward_strict = 0.0;
  tm.cert_fallbacks = tm.cert_rows = 0;
  if (!b->n_chunks) return DYN_OK;
  hipEvent_t* ev = b->events.data();
  float ms01 = 0, ms12 = 0;
  HIP_TRY(a, hipEventElapsedTime(&ms01, ev[0], ev[1]));
  HIP_TRY(a, hipEventElapsedTime(&ms12, ev[1], ev[2]));
  // wave-cycles per phase, summed over all waves of the launch: backward, forward, traceback (+ state
  // write-back and page release), waiting for a read / for pages, lifetime; [5] = longest lifetime
  if (b->h_stats.as<uint32_t>()[3] != 0) {
    std::lock_guard<std::mutex> elk(a->err_mu);
    a->last_error = "the read queue aborted: a wave waited for the queue lock or for lattice pages for seconds";
    return DYN_ERR_DEVICE;
  }
  const uint64_t* s = reinterpret_cast<const uint64_t*>(b->h_stats.as<uint32_t>() + dynk::QUEUE_STATS);
  const double life = (double)s[4];
  tm.ms_dp = ms01;
  tm.ms_total = ms01 + ms12;
  if (life > 0) {
    tm.ms_backward = ms01 * (double)s[0] / life;
    tm.ms_forward = ms01 * (double)s[1] / life;
    tm.ms_trace = ms01 * (double)s[2] / life + ms12;
    tm.wave_wait_share = (double)s[3] / life;
    tm.ms_backward_strict = ms01 * (double)s[6] / life;
    tm.ms_forward_strict = ms01 * (double)s[7] / life;
    tm.cert_fallbacks = s[8];
    tm.cert_rows = s[9];
    if (s[5] && tm.n_waves) tm.wave_occupancy = life / ((double)s[5] * tm.n_waves);
  } else {
    tm.ms_trace = ms12;
  }
  return DYN_OK;
}

// ==== the resident read queue (engine.hpp: Session; nt_kernels.hpp: k_session) ============================================
namespace {

// workgroups of a session = CUs it occupies (dyn_aligner_set_session_mode leaves the others free)
int session_wgs(const dyn_aligner* a) { return std::max(1, a->sess_cus); }
// bytes of one lattice row in the pool (separate LPE layout): bE 8 B + float LPE 4 B per slot + the decision ballots
constexpr uint64_t SESSION_ROW_BYTES = (uint64_t)dynk::P * 12 + dynk::CPL * 8;

uint32_t session_pages_of(uint64_t S, int log_r) { return (uint32_t)((S + 2 + (1ull << log_r) - 1) >> log_r); }

struct SessionNeed {
  uint64_t n_ok = 0;
  uint64_t max_S = 0;
};
SessionNeed session_need(const dyn_batch* b) {
  SessionNeed n;
  for (uint64_t i = 0; i < b->n; ++i)
    if (b->reads[i].status == DYN_READ_OK) {
      ++n.n_ok;
      n.max_S = std::max(n.max_S, b->reads[i].S);
    }
  return n;
}

// statistics of the session that ran on control block `blk` (its kernel has finished: ev_end has been waited for)
int session_collect(dyn_aligner* a, int blk) {
  Session& ss = a->sess;
  if (!ss.pending[blk]) return DYN_OK;
  float ms = 0.f;
  HIP_TRY(a, hipEventElapsedTime(&ms, ss.ev_begin[blk], ss.ev_end[blk]));
  HIP_TRY(a, a->sess_hctl.ensure(dynk::SESSION_CTL_WORDS * 4));
  // (not a null-stream copy: that one would also wait for a later session that is still open)
  HIP_TRY(a, hipMemcpyAsync(a->sess_hctl.p, a->sess_ctl[blk].p, dynk::SESSION_CTL_WORDS * 4, hipMemcpyDeviceToHost, a->s_out));
  HIP_TRY(a, hipStreamSynchronize(a->s_out));
  const uint32_t* cw = a->sess_hctl.as<uint32_t>();
  const uint64_t* st = reinterpret_cast<const uint64_t*>(cw + dynk::SESSION_STATS);
  dyn_session_stats& t = a->sess_total;
  t.sessions += 1;
  t.tickets += ss.pend_tickets[blk];
  t.reads += ss.pend_reads[blk];
  t.cells += ss.pend_cells[blk];
  t.ms += ms;
  t.wave_cycles_busy += st[0];
  t.wave_cycles_idle += st[1];
  t.wave_cycles_life += st[2];
  a->sess_page_wait_cycles += st[5];
  t.waves += ss.pend_waves[blk];
  if (cw[dynk::S_ABORT]) t.aborted += 1;
  ss.pending[blk] = false;
  return DYN_OK;
}


int session_open(dyn_aligner* a, bool mixed, const SessionGeom& g) {
  const int log_r = g.log_r;
  const uint32_t arena_pages = g.arena_pages, n_pages_total = g.n_pages;
  const bool paged = g.layout != 0, separate = g.layout != 2;
  Session& ss = a->sess;
  const int blk = ss.blk ^ 1;
  // the session before the last one used this block: it has long ended, but its statistics may still be waiting
  if (ss.pending[blk]) {
    HIP_TRY(a, hipEventSynchronize(ss.ev_end[blk]));
    if (int rc = session_collect(a, blk)) return rc;
  }
  for (int k = 0; k < 2; ++k) {
    if (!ss.ev_begin[k]) HIP_TRY(a, hipEventCreate(&ss.ev_begin[k]));
    if (!ss.ev_end[k]) HIP_TRY(a, hipEventCreate(&ss.ev_end[k]));
  }
  HIP_TRY(a, a->sess_anchor.ensure(256, 1.0));
  HIP_TRY(a, a->sess_ctl[blk].ensure(dynk::SESSION_CTL_WORDS * 4, 1.0));
  HIP_TRY(a, a->sess_ring[blk].ensure((size_t)SESSION_RING * sizeof(dynk::SessionTicket), 1.0));
  // The lattice pool: an arena for every wave. Growing releases the old buffers -- whatever used them must have left: the
  // classic launches of the compute stream and the previous session (its kernel precedes this one on the session stream
  // anyway; the host-side wait is for the hipFree).
  const uint64_t page_rows = 1ull << log_r;
  const uint64_t ws_pp = page_rows * dynk::P * 8, lpe_pp = page_rows * dynk::P * 4, bits_pp = page_rows * dynk::CPL * 8;
  HIP_TRY(a, hipStreamSynchronize(a->stream));
  if (a->ws.bytes < n_pages_total * ws_pp || (separate && a->lpe.bytes < n_pages_total * lpe_pp) || a->bits.bytes < n_pages_total * bits_pp ||
      (paged && a->free_list.bytes < (size_t)n_pages_total * 4)) {
    if (ss.pending[ss.blk]) HIP_TRY(a, hipEventSynchronize(ss.ev_end[ss.blk]));
    HIP_TRY(a, ensure_pool(a->device, a->ws, n_pages_total * ws_pp, a->lpe, separate ? n_pages_total * lpe_pp : 0, a->bits,
                           n_pages_total * bits_pp, 1.0));
    if (paged) HIP_TRY(a, a->free_list.ensure((size_t)n_pages_total * 4, 1.0));
  }
  if (paged) HIP_TRY(a, a->ctl.ensure(dynk::QUEUE_CTL_WORDS * 4, 1.0));
  // control words cleared IN the session stream, and waited for: the first publish (copy-in stream) must not be wiped
  HIP_TRY(a, hipMemsetAsync(a->sess_ctl[blk].p, 0, dynk::SESSION_CTL_WORDS * 4, a->s_session));
  HIP_TRY(a, hipStreamSynchronize(a->s_session));  // (also: the previous session's kernel has left -- a->s_session is in order)
  if (ss.pending[ss.blk]) {
    if (int rc = session_collect(a, ss.blk)) return rc;
  }
  dynk::SessionArgs sa{};
  sa.ring = a->sess_ring[blk].as<dynk::SessionTicket>();
  sa.ring_size = SESSION_RING;
  sa.arena_pages = arena_pages;
  sa.give_always = std::getenv("DYN_SESSION_GIVE_ALWAYS") ? 1u : 0u;
  sa.ctl = a->sess_ctl[blk].as<uint32_t>();
  sa.pool.ws = a->ws.as<double>();
  sa.pool.lpe = separate ? a->lpe.as<float>() : nullptr;
  sa.pool.bits = a->bits.as<uint64_t>();
  sa.pool.free_list = paged ? a->free_list.as<uint32_t>() : nullptr;
  sa.pool.ctl = paged ? a->ctl.as<uint32_t>() : nullptr;
  sa.pool.log_rows = log_r;
  sa.pool.n_pages = n_pages_total;
  sa.pool.reserve_after = paged ? 64u : 0u;  // a stream never ends: a large request must not starve behind small ones
  sa.m1 = a->model.log_m1;
  sa.e2 = a->model.log_e2;
  sa.idle_limit_ticks = (uint64_t)(a->sess_idle_s * 1e8);
  if (paged) dynk::launch_pool_init(sa.pool, 0, 0, a->s_session);  // every page on the free list, control words cleared
  HIP_TRY(a, hipEventRecord(ss.ev_begin[blk], a->s_session));
  dynk::launch_session(mixed, g.layout, sa, a->d_model.p, a->sess_anchor.p, a->d_sptab.as<dynmath::SoftplusNode>(), session_wgs(a), a->s_session);
  HIP_TRY(a, hipGetLastError());
  HIP_TRY(a, hipEventRecord(ss.ev_end[blk], a->s_session));
  ss.open = true;
  a->sess_open_hint.store(true);
  ss.mixed = mixed;
  ss.blk = blk;
  ss.blk_gen[blk].store(++ss.gen);
  ss.published = 0;
  ss.next_base = 0;
  ss.log_r = log_r;
  ss.arena_pages = arena_pages;
  ss.layout = g.layout;
  ss.n_pages = n_pages_total;
  ss.n_waves = (uint32_t)session_wgs(a) * dynk::WAVES_PER_CU;
  ss.cells = ss.reads = ss.tickets = 0;
  return DYN_OK;
}

}  // namespace

bool session_candidate(const dyn_batch* b) {
  const dyn_aligner* a = b->a;
  // (b->async: a caller's ticket. The batch of a MERGED launch is the engine's own and stays one launch: its members report
  //  that launch and their share of it.)
  // align(calc_probabilities=1) only. Training tickets were tried twice (round 5, k_session<JOB_TRAIN>). First the kernels that
  // followed each of them -- rocPRIM's radix sort for the device-resident pooled statistics -- did not start beside resident
  // waves; those statistics are computed on demand since (dyn_batch_device_pooled). Then, with nothing following a training
  // ticket, sessions measured 805.6 / 810.5 against 810.7 / 813.9 Msamp/s for one launch per batch: 1 024 reads on 1 024
  // waves keep a launch's waves busy 0.98 of it already. Training stays one launch per batch.
  return a->sess_enabled.load() && !a->host_only && !a->ntk && b->async && b->job == DynJob::AlignFull &&
         (a->sess_open_hint.load() || b->n >= SESSION_MIN_READS);
}

int session_close(dyn_aligner* a) {
  Session& ss = a->sess;
  if (!ss.open) return DYN_OK;
  dynk::launch_session_close(a->sess_ctl[ss.blk].as<uint32_t>(), a->s_in);  // behind every publish: same stream
  HIP_TRY(a, hipGetLastError());
  ss.open = false;
  a->sess_open_hint.store(false);
  ss.pending[ss.blk] = true;
  ss.pend_cells[ss.blk] = ss.cells;
  ss.pend_reads[ss.blk] = ss.reads;
  ss.pend_tickets[ss.blk] = ss.tickets;
  ss.pend_waves[ss.blk] = ss.n_waves;
  return DYN_OK;
}

int session_quiesce(dyn_aligner* a) {
  if (!a->s_session) return DYN_OK;
  if (int rc = session_close(a)) return rc;
  Session& ss = a->sess;
  for (int k = 0; k < 2; ++k)
    if (ss.pending[k]) {
      HIP_TRY(a, hipEventSynchronize(ss.ev_end[k]));
      if (int rc = session_collect(a, k)) return rc;
    }
  return DYN_OK;
}

// The arena geometry a ticket asks for: pages of 2^log_r rows such that its longest read (plus an eighth: later tickets
// of the same kind should fit as well) stays within a wave's PT_MAX-entry page table.
static void session_geometry(uint64_t max_S, int* log_r, uint32_t* arena_pages) {
  const uint64_t cap_S = max_S + max_S / 8 + 64;
  int lr = 8;
  while (session_pages_of(cap_S, lr) > (uint32_t)dynk::PT_MAX) ++lr;
  *log_r = lr;
  *arena_pages = session_pages_of(cap_S, lr);
}

// The geometry of a session that could take the ticket: an arena for every wave if the memory budget allows (layout 0),
// else the pool's pages shared through the free list, with the posterior layout enqueue_job would choose for such a launch.
static int session_choose(dyn_aligner* a, const dyn_batch* b, const SessionNeed& need, SessionGeom* g) {
  *g = SessionGeom{};
  session_geometry(need.max_S, &g->log_r, &g->arena_pages);
  const uint64_t n_waves = (uint64_t)session_wgs(a) * dynk::WAVES_PER_CU;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(a, hipMemGetInfo(&free_b, &total_b));
  const uint64_t pool = a->ws.bytes + a->lpe.bytes + a->bits.bytes + parked_bytes(a->device);
  uint64_t budget = (uint64_t)((double)(free_b + pool) * 0.90);
  if (a->mem_budget && a->mem_budget < budget) budget = a->mem_budget;
  const uint64_t page_rows = 1ull << g->log_r, max_pages = 0xfffffff0ull >> g->log_r;  // pool rows are 32-bit
  const uint64_t want = n_waves * g->arena_pages * (page_rows * SESSION_ROW_BYTES);
  if (want <= budget && n_waves * g->arena_pages <= max_pages) {
    g->layout = 0;
    g->n_pages = (uint32_t)(n_waves * g->arena_pages);
    g->ok = true;
    return DYN_OK;
  }
  if (std::getenv("DYN_NO_PAGED_SESSION")) return DYN_OK;  // page-starved batches as one launch each (round 4's path)
  // page-starved. The pages that would keep every wave busy with this ticket's largest lattices:
  std::vector<uint32_t> pg;
  pg.reserve(need.n_ok);
  for (uint64_t i = 0; i < b->n; ++i)
    if (b->reads[i].status == DYN_READ_OK) pg.push_back(session_pages_of(b->reads[i].S, g->log_r));
  const size_t top = std::min<size_t>(n_waves, pg.size());
  std::partial_sort(pg.begin(), pg.begin() + top, pg.end(), std::greater<uint32_t>());
  uint64_t wanted = 0;
  for (size_t k = 0; k < top; ++k) wanted += pg[k];
  wanted = std::max<uint64_t>(wanted, 1);
  // separate float LPE: the forward sweep is 17 % faster, 12 instead of 8 bytes per band slot (enqueue_job's rule)
  const uint64_t row_sep = (uint64_t)dynk::P * 12 + dynk::CPL * 8, row_inp = (uint64_t)dynk::P * 8 + dynk::CPL * 8;
  const double c_sep = std::min(1.0, (double)budget / ((double)wanted * page_rows * row_sep + 1.0));
  const double c_inp = std::min(1.0, (double)budget / ((double)wanted * page_rows * row_inp + 1.0));
  bool separate = !(c_sep < 1.0 && c_inp * 0.92 > c_sep);
  if (const char* f = std::getenv("DYN_FORCE_LAYOUT")) separate = std::string(f) != "inplace";
  const uint64_t page_bytes = page_rows * (separate ? row_sep : row_inp);
  // later tickets of the stream are served from the same pool: everything the budget gives, up to an arena per wave
  const uint64_t n_pages = std::min<uint64_t>({budget / page_bytes, n_waves * g->arena_pages, max_pages});
  if (n_pages < g->arena_pages) return DYN_OK;  // the longest read alone does not fit: the classic launch gives it its status
  g->layout = separate ? 1 : 2;
  g->n_pages = (uint32_t)n_pages;
  g->ok = true;
  return DYN_OK;
}

static bool session_fits(const dyn_aligner* a, const Session& ss, const SessionNeed& need) {
  const uint32_t cap = ss.layout == 0 ? ss.arena_pages : std::min<uint32_t>((uint32_t)dynk::PT_MAX, ss.n_pages);
  return session_pages_of(need.max_S, ss.log_r) <= cap && ss.published < SESSION_RING && (uint64_t)ss.next_base + need.n_ok < 0x7fffffffull &&
         (ss.mixed || a->strict_mode == 0);
}

int session_plan(dyn_batch* b, bool* use) {
  dyn_aligner* a = b->a;
  *use = false;
  b->sess_geom = SessionGeom{};
  if (!session_candidate(b)) return DYN_OK;
  if (b->n_wide) return DYN_OK;  // wide-band reads take the generic kernel behind a classic launch
  const SessionNeed need = session_need(b);
  if (!need.n_ok) return DYN_OK;  // nothing to launch
  Session& ss = a->sess;
  if (ss.open) {
    if (session_fits(a, ss, need)) {
      *use = true;
      return DYN_OK;
    }
    if (int rc = session_close(a)) return rc;  // a new one is opened below if this ticket deserves it
  }
  if (need.n_ok < SESSION_MIN_READS) return DYN_OK;
  SessionGeom g;
  if (int rc = session_choose(a, b, need, &g)) return rc;
  *use = g.ok;  // (false: the longest read does not fit the pool at all -- the planned classic launch)
  // the geometry is decided ONCE: session_publish opens the session with it (a second look at hipMemGetInfo could disagree
  // with this one -- another process, the buffer cache -- and leave an accepted ticket without a session)
  b->sess_geom = g;
  return DYN_OK;
}

int session_publish(dyn_batch* b) {
  dyn_aligner* a = b->a;
  const PoreModel& m = a->model;
  Session& ss = a->sess;
  // strict reads and the queue order: as enqueue_job
  const int32_t* km = b->kmers();
  std::vector<uint32_t> strict_rows(b->n, 0), order;
  uint64_t n_strict = 0, max_S = 0;
  for (uint64_t i = 0; i < b->n; ++i) {
    const HostRead& r = b->reads[i];
    if (r.status != DYN_READ_OK) continue;
    if (a->strict_mode == 2) strict_rows[i] = 0xffffffffu;
    else if (a->strict_mode == 1) strict_rows[i] = tie_rows(a->model, km + r.flat_off, r.kc, r.S);
    n_strict += strict_rows[i] != 0;
    max_S = std::max(max_S, r.S);
    order.push_back((uint32_t)i);
  }
  auto cost_rows = [&](uint32_t i) -> uint64_t {
    const uint64_t T = b->reads[i].S + 1;
    if (!strict_rows[i]) return T;
    const uint64_t fr = std::min<uint64_t>(T, strict_rows[i]);
    return (T * 100 + T * 12 + fr * 24) / 100;
  };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cost_rows(x) > cost_rows(y); });
  const size_t n_ok = order.size();

  if (!ss.open) {
    SessionNeed need;
    need.n_ok = n_ok;
    need.max_S = max_S;
    SessionGeom g = b->sess_geom;  // session_plan's (the open session it may have counted on has been closed since: its own)
    if (!g.ok) {
      if (int rc = session_choose(a, b, need, &g)) return rc;
    }
    if (!g.ok) {
      a->last_error = "session_publish: no session geometry for a ticket session_plan had accepted";
      return DYN_ERR_RUNTIME;
    }
    if (int rc = session_open(a, a->strict_mode != 0, g)) return rc;
  }
  if (ss.layout != 0 && n_ok > ss.n_waves && !std::getenv("DYN_NO_BRIDGE")) {
    // a PAGED session: the ticket's longest reads one after the other would ask for more pages than the pool has, and the
    // waves that claim them would wait while the short reads behind them could run: the reads are dealt out in SPREAD order
    // (spread_order; the planned order of a page-starved LAUNCH, plan_queue, assumes waves that all start empty-handed and
    // a launch that must end on short reads -- measured here as well, DYN_SESSION_PLANNED: 465 against 507 Msamp/s).
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return b->reads[x].S > b->reads[y].S; });
    std::vector<uint32_t> need(n_ok);
    std::vector<uint64_t> rows(n_ok);
    for (size_t k = 0; k < n_ok; ++k) {
      need[k] = session_pages_of(b->reads[order[k]].S, ss.log_r);
      rows[k] = cost_rows(order[k]);
    }
    if (std::getenv("DYN_SESSION_PLANNED")) {
      plan_queue(order, need, rows, ss.n_waves, ss.n_pages);
    } else {
      const char* tail_env = std::getenv("DYN_SESSION_TAIL_DIV");  // experiments: 0 = spread every read
      std::vector<uint32_t> rank(b->n, 0);  // position in the longest-first order
      for (size_t k = 0; k < order.size(); ++k) rank[order[k]] = (uint32_t)k;
      spread_order(order, tail_env ? std::atoi(tail_env) : SESSION_TAIL_DIV);
      // The END of a ticket nobody follows (round 6): the last n_waves reads of the order are in flight together whatever their
      // order -- each wave takes one -- so their pages are asked for together either way; taken LONGEST FIRST the long ones among
      // them start as early as they can and the waves finish within a short read of each other, instead of one 100 k-sample read,
      // claimed last, keeping 1 023 waves waiting for the session's close (config 3: ~5 % of an 8-batch run; measured A/B below).
      if (!std::getenv("DYN_SESSION_NO_TAIL_LPT") && order.size() > ss.n_waves) {
        auto tail = order.end() - (ptrdiff_t)ss.n_waves;
        std::stable_sort(tail, order.end(), [&](uint32_t x, uint32_t y) { return rank[x] < rank[y]; });
      }
    }
  }

  HIP_TRY(a, b->d_segrow.ensure(std::max<uint64_t>(4, b->capacity * 4)));
  HIP_TRY(a, b->d_medhi.ensure(std::max<uint64_t>(8, b->capacity * 8)));
  HIP_TRY(a, b->d_medlo.ensure(std::max<uint64_t>(8, b->capacity * 8)));
  ReadState* st = b->h_state.as<ReadState>();
  for (uint64_t i = 0; i < b->n; ++i) {
    st[i].Zb = 0.0;
    st[i].Zf = 0.0;
    st[i].status = b->reads[i].status;
    st[i].n_segments = 0;
  }
  HIP_TRY(a, hipMemcpyAsync(b->d_state.p, st, b->n * sizeof(ReadState), hipMemcpyHostToDevice, a->s_in));
  HIP_TRY(a, b->h_descs.ensure(std::max<size_t>(sizeof(ReadDesc), n_ok * sizeof(ReadDesc))));
  ReadDesc* descs = b->h_descs.as<ReadDesc>();
  dyn_timing tm{};
  uint64_t rows_total = 0;
  uint32_t max_N = 0;
  for (size_t k = 0; k < n_ok; ++k) {
    const uint32_t i = order[k];
    const HostRead& r = b->reads[i];
    ReadDesc d{};
    d.T = (uint32_t)(r.S + 1);
    d.N = (uint32_t)(r.kc + 1);
    d.bw = (uint32_t)std::min<uint64_t>(m.half_band, d.N / 2);
    d.read = i;
    d.ratio = (double)d.N / (double)d.T;
    d.sig_off = r.sig_off;
    d.par_off = r.flat_off;
    d.seg_off = r.seg_off;
    d.path_off = rows_total;
    d.n_pages = session_pages_of(r.S, ss.log_r);
    d.first_page = dynk::NO_PAGE;  // the wave's own arena
    d.flags = !strict_rows[i] ? 0u : strict_rows[i] == 0xffffffffu ? dynk::READ_STRICT : dynk::READ_STRICT_START;
    d.strict_rows = strict_rows[i];
    rows_total += d.T;
    max_N = std::max(max_N, d.N);
    descs[k] = d;
    tm.cells += (uint64_t)d.T * std::min<uint64_t>(2ull * d.bw + 1, d.N);
    tm.samples += r.S;
  }
  HIP_TRY(a, b->d_pp.ensure(std::max<uint64_t>(8, rows_total * 8)));
  HIP_TRY(a, b->d_pathn.ensure(std::max<uint64_t>(4, rows_total * 4)));
  HIP_TRY(a, b->d_descs.ensure(std::max<size_t>(sizeof(ReadDesc), n_ok * sizeof(ReadDesc))));
  HIP_TRY(a, hipMemcpyAsync(b->d_descs.p, descs, n_ok * sizeof(ReadDesc), hipMemcpyHostToDevice, a->s_in));
  b->d_tctl.cache = &a->cache;
  HIP_TRY(a, b->d_tctl.ensure(dynk::SESSION_TCTL_WORDS * 4));
  HIP_TRY(a, hipMemsetAsync(b->d_tctl.p, 0, dynk::SESSION_TCTL_WORDS * 4, a->s_in));
  HIP_TRY(a, b->h_stats.ensure(dynk::SESSION_TCTL_WORDS * 4));
  std::memset(b->h_stats.p, 0, dynk::SESSION_TCTL_WORDS * 4);
  volatile uint32_t* flag = a->sess_flags + (ss.flag_seq++ % SESSION_FLAGS);
  __atomic_store_n(const_cast<uint32_t*>(flag), 0u, __ATOMIC_SEQ_CST);  // (before the record that names it is published)

  const char* in_base = static_cast<const char*>(a->d_model.p);
  const char* out_base = static_cast<const char*>(a->sess_anchor.p);
  auto off_in = [&](const void* p) { return (int64_t)(static_cast<const char*>(p) - in_base); };
  auto off_out = [&](const volatile void* p) { return (int64_t)(static_cast<const char*>(const_cast<const void*>(p)) - out_base); };
  dynk::SessionTicket tk{};
  tk.descs_off = off_in(b->d_descs.p);
  tk.sig_off = off_in(b->d_sig.p);
  tk.par_off = off_in(b->d_par.p);
  tk.st_off = off_out(b->d_state.p);
  tk.pp_off = off_out(b->d_pp.p);
  tk.pathn_off = off_out(b->d_pathn.p);
  tk.segrow_off = off_out(b->d_segrow.p);
  tk.medhi_off = off_out(b->d_medhi.p);
  tk.medlo_off = off_out(b->d_medlo.p);
  tk.tctl_off = off_out(b->d_tctl.p);
  tk.flag_off = off_out(flag);
  tk.n_reads = (uint32_t)n_ok;
  tk.base = ss.next_base;
  tk.z_fail_status = DYN_READ_Z_MISMATCH;
  dynk::launch_session_publish(a->sess_ring[ss.blk].as<dynk::SessionTicket>(), a->sess_ctl[ss.blk].as<uint32_t>(), tk, ss.published, SESSION_RING,
                               a->s_in);
  HIP_TRY(a, hipGetLastError());
  ss.published += 1;
  ss.next_base += (uint32_t)n_ok;
  ss.cells += tm.cells;
  ss.reads += n_ok;
  ss.tickets += 1;

  while (b->events.size() < 3) {
    hipEvent_t e = nullptr;
    HIP_TRY(a, hipEventCreate(&e));
    b->events.push_back(e);
  }
  // events[0]: the ticket's counter has been cleared and its record published. Until then d_tctl holds what the buffer's last
  // ticket left there (its full count, often the same number of reads): wait_resident must not read it earlier.
  HIP_TRY(a, hipEventRecord(b->events[0], a->s_in));
  tm.reads_ok = n_ok;
  tm.reads_strict = (uint32_t)n_strict;
  tm.launch_share = 0.0;
  tm.launches = 0;
  tm.lp_inplace = ss.layout == 2 ? 1 : 0;
  tm.pool_pages = ss.n_pages;
  tm.page_rows = 1u << ss.log_r;
  tm.n_static = 0;
  tm.n_waves = ss.n_waves;
  b->strict_flag.assign(b->n, 0);
  for (uint64_t i = 0; i < b->n; ++i) b->strict_flag[i] = strict_rows[i] != 0;
  b->timing = tm;
  b->n_chunks = 1;
  b->aligned = true;
  b->trained = false;
  b->last_calc = 1;
  b->in_session = true;
  b->sess_reads = (uint32_t)n_ok;
  b->sess_waves = ss.n_waves;
  b->sess_blk = ss.blk;
  b->sess_gen = ss.gen;
  b->sess_flag = flag;
  b->sess_max_N = max_N;
  b->sess_rows_total = rows_total;
  return DYN_OK;
}

int session_recover(dyn_batch* b, bool* republished) {
  dyn_aligner* a = b->a;
  Session& ss = a->sess;
  *republished = false;
  const bool mine_open = ss.open && ss.gen == b->sess_gen;
  if (mine_open) {
    // the host still believes in the session that aborted: close it and wait until its kernel has left
    if (int rc = session_quiesce(a)) return rc;
  } else if (ss.pending[b->sess_blk] && ss.blk_gen[b->sess_blk].load() == b->sess_gen) {
    HIP_TRY(a, hipEventSynchronize(ss.ev_end[b->sess_blk]));
    if (int rc = session_collect(a, b->sess_blk)) return rc;
  }
  // (otherwise the block has been cleared for a later session: the lost one ended long ago)
  // The abort word may have been raised by a wave that idled while OTHERS were still busy with this ticket's last reads: now
  // that the kernel has ended, the counter says whether anything is missing.
  // (on the copy-out stream: a null-stream copy would wait for a LATER session that is open, and that one waits for us)
  uint32_t* count = b->h_stats.as<uint32_t>();
  HIP_TRY(a, hipMemcpyAsync(count, b->d_tctl.p, 4, hipMemcpyDeviceToHost, a->s_out));
  HIP_TRY(a, hipStreamSynchronize(a->s_out));
  if (*count == b->sess_reads) return DYN_OK;
  if (b->sess_retries >= 2) {
    char msg[200];
    std::snprintf(msg, sizeof msg, "the resident read queue aborted under this ticket three times (its waves found no work for DYN_SESSION_IDLE_S "
                  "seconds while it was pending): %u of %u reads done", *count, b->sess_reads);
    a->last_error = msg;
    return DYN_ERR_DEVICE;
  }
  const SessionNeed need = session_need(b);
  if (ss.open && !session_fits(a, ss, need))
    if (int rc = session_quiesce(a)) return rc;  // (the pool may have to grow: nothing may be using it)
  b->sess_retries += 1;
  a->sess_total.republished += 1;
  *republished = true;
  return session_publish(b);
}

// the ticket's reads are done (its completion word has been seen): per-segment kernels, statistics
int session_finish_enqueue(dyn_batch* b, hipStream_t s) {
  dyn_aligner* a = b->a;
  hipEvent_t* ev = b->events.data();
  HIP_TRY(a, hipEventRecord(ev[1], s));
  dynk::TraceBuffers tb{b->d_pp.as<double>(), b->d_pathn.as<uint32_t>(), b->d_segrow.as<uint32_t>(), b->d_medhi.as<double>(),
                        b->d_medlo.as<double>()};
  dynk::launch_segments(b->d_descs.as<ReadDesc>(), (int)b->sess_reads, b->sess_rows_total, b->sess_max_N, b->d_state.as<ReadState>(), tb,
                        b->d_rows.as<SegRow>(), a->model.k, s);
  HIP_TRY(a, hipGetLastError());
  HIP_TRY(a, hipEventRecord(ev[2], s));
  HIP_TRY(a, hipMemcpyAsync(b->h_stats.p, b->d_tctl.p, dynk::SESSION_TCTL_WORDS * 4, hipMemcpyDeviceToHost, s));
  return DYN_OK;
}

int session_collect_timing(dyn_batch* b) {
  dyn_aligner* a = b->a;
  dyn_timing& tm = b->timing;
  float ms12 = 0;
  HIP_TRY(a, hipEventElapsedTime(&ms12, b->events[1], b->events[2]));
  const uint64_t* st = reinterpret_cast<const uint64_t*>(b->h_stats.as<uint32_t>() + dynk::SESSION_TSTATS);
  // the ticket's wave time: its reads' durations (10 ns ticks) spread over the session's waves; the phases by their share of
  // the shader-clock cycles
  tm.ms_dp = (double)st[3] / 1e5 / (double)std::max<uint32_t>(1, b->sess_waves);
  const double cyc = (double)(st[0] + st[1] + st[2]);
  const double per_cyc = cyc > 0 ? tm.ms_dp / cyc : 0.0;
  tm.ms_backward = (double)st[0] * per_cyc;
  tm.ms_forward = (double)st[1] * per_cyc;
  tm.ms_trace = (double)st[2] * per_cyc + ms12;
  tm.ms_total = tm.ms_dp + ms12;
  tm.wave_wait_share = 0.0;
  tm.wave_occupancy = 0.0;  // a session's, not a ticket's: dyn_aligner_session_stats
  tm.ms_backward_strict = (double)st[6] * per_cyc;
  tm.ms_forward_strict = (double)st[7] * per_cyc;
  tm.cert_fallbacks = st[8];
  tm.cert_rows = st[9];
  return DYN_OK;
}

}  // namespace dyneng

extern "C" int dyn_aligner_session_stats(dyn_aligner* a, dyn_session_stats* out) {
  if (!a || !out) return DYN_ERR_INVALID_ARGUMENT;
  if (!a->host_only && a->s_session) {
    std::lock_guard<std::mutex> lk(a->mu);
    if (int rc = need_device(a)) return rc;
    if (int rc = session_quiesce(a)) return rc;
  }
  *out = a->sess_total;
  return DYN_OK;
}

extern "C" int dyn_aligner_session_page_wait(dyn_aligner* a, uint64_t* wave_cycles_waiting_for_pages) {
  if (!a || !wave_cycles_waiting_for_pages) return DYN_ERR_INVALID_ARGUMENT;
  if (!a->host_only && a->s_session) {
    std::lock_guard<std::mutex> lk(a->mu);
    if (int rc = need_device(a)) return rc;
    if (int rc = session_quiesce(a)) return rc;
  }
  *wave_cycles_waiting_for_pages = a->sess_page_wait_cycles;
  return DYN_OK;
}

namespace dyneng {

// array-of-rows -> the caller's columns; reads are independent, so contiguous ranges of reads go to
// the helper threads (2 M segments per 1 024-read batch take ~10 ms on one core)
void unpack_align(const dyn_batch* b, const ReadState* st, const SegRow* rows, dyn_align_out* out,
                  HelperPool* pool, uint64_t read0, uint64_t n, uint64_t seg0) {
  if (n == ~0ull) n = b->n - read0;
  const bool want_rows = rows != nullptr;
  uint64_t cap = 0;
  for (uint64_t i = read0; i < read0 + n; ++i) cap += b->reads[i].kc;
  auto unpack = [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) {
      const HostRead& r = b->reads[i];
      const uint64_t j = i - read0, so = r.seg_off - seg0;  // the caller's indices
      const bool ok = st[i].status == DYN_READ_OK;
      out->status[j] = st[i].status;
      out->Z[j] = ok ? st[i].Zb : 0.0;  // Result::Z = Zb (NT_aligner_api.cpp:293)
      if (out->bad_char) out->bad_char[j] = r.bad;
      if (out->seg_offsets) out->seg_offsets[j] = so;
      const uint64_t ns = (ok && b->last_calc) ? st[i].n_segments : 0;
      if (out->n_segments) out->n_segments[j] = ns;
      if (want_rows) {
        for (uint64_t s = 0; s < ns; ++s) {
          const SegRow& row = rows[r.seg_off + s];
          if (out->sequence_positions) out->sequence_positions[so + s] = row.sequence_pos;
          if (out->signal_positions) out->signal_positions[so + s] = row.signal_pos;
          if (out->probabilities) out->probabilities[so + s] = row.probability;
          if (out->states) out->states[so + s] = 'M';
        }
      }
    }
  };
  const int parts = (pool && want_rows && cap > (1u << 16)) ? (int)std::min<uint64_t>(pool->size(), std::max<uint64_t>(1, n / 64)) : 1;
  if (parts <= 1) unpack(read0, read0 + n);
  else pool->parallel_for(parts, [&](int t) { unpack(read0 + n * t / parts, read0 + n * (t + 1) / parts); });
  if (out->seg_offsets) out->seg_offsets[n] = cap;
}

// Host finalisation of runTraining (NT_aligner_api.cpp:516-535) from per-column sums, and of
// trainTransition (:703-722) from the two linear-domain transition sums.
void finalise_train(const dyn_batch* b, const ReadState* st, const double* cw, const double* c1,
                    const double* c2, const double* tr, dyn_train_out* out, double* pooled3n) {
  const PoreModel& m = b->a->model;
  const bool want_em = out->em_code && out->em_mean && out->em_stdev;
  const int32_t* kmers = b->kmers();
  std::vector<std::pair<int32_t, uint64_t>> keyed;
  for (uint64_t i = 0; i < b->n; ++i) {
    const HostRead& r = b->reads[i];
    const bool ok = st[i].status == DYN_READ_OK;
    out->status[i] = st[i].status;
    out->Z[i] = ok ? st[i].Zb : 0.0;
    if (out->bad_char) out->bad_char[i] = r.bad;
    if (out->em_offsets) out->em_offsets[i] = r.seg_off;
    uint64_t count = 0;
    if (out->transitions) {
      double m1 = 0.0, e2 = 0.0;
      if (ok) {
        const double sm = tr[2 * i], se = tr[2 * i + 1];
        const double tot = sm + se;
        if (tot > 0.0 && !std::isinf(tot)) {
          m1 = sm / tot;
          e2 = se / tot;
        }
      }
      out->transitions[3 * i] = m1;
      out->transitions[3 * i + 1] = ok ? std::exp(m.log_e1) : 0.0;
      out->transitions[3 * i + 2] = e2;
    }
    if (out->trans_counts) {
      out->trans_counts[2 * i] = ok ? tr[2 * i] : 0.0;
      out->trans_counts[2 * i + 1] = ok ? tr[2 * i + 1] : 0.0;
    }
    if (ok && (want_em || pooled3n)) {
      // group the read's lattice columns by k-mer code, columns in ascending order
      keyed.clear();
      for (uint64_t c = 0; c < r.kc; ++c) keyed.emplace_back(kmers[r.flat_off + c], r.flat_off + c);
      std::stable_sort(keyed.begin(), keyed.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
      size_t p = 0;
      while (p < keyed.size()) {
        const int32_t code = keyed[p].first;
        double w = 0.0, s1 = 0.0, s2 = 0.0;
        for (; p < keyed.size() && keyed[p].first == code; ++p) {
          w += cw[keyed[p].second];
          s1 += c1[keyed[p].second];
          s2 += c2[keyed[p].second];
        }
        if (pooled3n) {
          pooled3n[code] += w;
          pooled3n[m.num_kmers + code] += s1;
          pooled3n[2 * m.num_kmers + code] += s2;
        }
        if (want_em && w > 0.0) {
          const double mean = s1 / w;
          double var = s2 / w - mean * mean;
          if (var < 1e-12) var = 1e-12;
          const uint64_t o = r.seg_off + count;
          out->em_code[o] = code;
          out->em_mean[o] = mean;
          out->em_stdev[o] = std::sqrt(var);
          if (out->em_weight) out->em_weight[o] = w;
          if (out->em_sum) out->em_sum[o] = s1;
          if (out->em_sumsq) out->em_sumsq[o] = s2;
          ++count;
        }
      }
    }
    if (out->em_count) out->em_count[i] = count;
  }
  if (out->em_offsets) out->em_offsets[b->n] = b->capacity;
}

}  // namespace dyneng

namespace {

int run_job_sync(dyn_batch* b, DynJob job) {
  dyn_aligner* a = b->a;
  if (b->async) {
    // An asynchronous ticket is a one-shot submission: its inputs were staged by the pipeline, and when it shared a launch
    // with other tickets it owns neither device buffers nor a read table (dyn_batch.group) -- there is nothing to run again.
    a->last_error = "dyn_batch_align / dyn_batch_train on an asynchronous ticket: submit a new ticket, or use dyn_batch_create";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  {
    std::lock_guard<std::mutex> lk(a->mu);
    int rc = need_device(a);
    if (rc != DYN_OK) return rc;
    rc = enqueue_job(b, job);
    if (rc != DYN_OK) {
      (void)hipStreamSynchronize(a->stream);
      return rc;
    }
  }
  HIP_TRY(a, hipStreamSynchronize(a->stream));
  return collect_timing(b);
}

}  // namespace

extern "C" {

int dyn_batch_align(dyn_batch* b, int calc_probabilities) {
  if (!b) return DYN_ERR_INVALID_ARGUMENT;
  return run_job_sync(b, calc_probabilities ? DynJob::AlignFull : DynJob::AlignZ);
}

int dyn_batch_train(dyn_batch* b) {
  if (!b) return DYN_ERR_INVALID_ARGUMENT;
  return run_job_sync(b, DynJob::Train);
}

int dyn_batch_timing(const dyn_batch* b, dyn_timing* t) {
  if (!b || !t) return DYN_ERR_INVALID_ARGUMENT;
  *t = b->timing;
  return DYN_OK;
}

int dyn_batch_device_results(dyn_batch* b, void** d_rows, uint64_t* capacity, void** d_z_status) {
  if (!b || !b->aligned) return DYN_ERR_INVALID_ARGUMENT;
  if (b->group && b->group->g) {  // a member of a merged launch: its slice of the group's device arrays
    const dyn_batch* g = b->group->g;
    if (d_rows) *d_rows = static_cast<char*>(g->d_rows.p) + b->g_seg0 * sizeof(SegRow);
    if (capacity) *capacity = b->capacity;
    if (d_z_status) *d_z_status = static_cast<char*>(g->d_state.p) + b->g_read0 * sizeof(ReadState);
    return DYN_OK;
  }
  if (d_rows) *d_rows = b->d_rows.p;
  if (capacity) *capacity = b->capacity;
  if (d_z_status) *d_z_status = b->d_state.p;
  return DYN_OK;
}

int dyn_batch_fetch(dyn_batch* b, dyn_align_out* out) {
  if (!b || !out || !out->Z || !out->status) return DYN_ERR_INVALID_ARGUMENT;
  dyn_aligner* a = b->a;
  if (!b->aligned) {
    a->last_error = "dyn_batch_fetch before dyn_batch_align";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  int rc = need_device(a);
  if (rc != DYN_OK) return rc;
  // A ticket that shared its launch with others (async_engine.cpp, merged launches) owns no device buffers and no read
  // table of its own: its results are reads [g_read0, g_read0 + n) and rows [g_seg0, g_seg0 + capacity) of the group's batch.
  const dyn_batch* src = (b->group && b->group->g) ? b->group->g : b;
  const uint64_t read0 = src == b ? 0 : b->g_read0, seg0 = src == b ? 0 : b->g_seg0;
  std::vector<ReadState> st(src->n);
  if (b->n)
    HIP_TRY(a, copy_out(a, st.data() + read0, static_cast<const ReadState*>(src->d_state.p) + read0, b->n * sizeof(ReadState)));
  const bool want_rows = b->last_calc && (out->sequence_positions || out->signal_positions || out->probabilities || out->states);
  if (want_rows && out->capacity < b->capacity) {
    a->last_error = "dyn_align_out.capacity is smaller than dyn_segment_capacity()";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  const SegRow* rows = nullptr;
  static HelperPool pool(4);
  static std::mutex pool_mu;
  std::lock_guard<std::mutex> lk(pool_mu);  // also guards the handle's h_rows staging
  if (want_rows && b->capacity) {
    HIP_TRY(a, a->h_rows.ensure(b->capacity * sizeof(SegRow)));
    HIP_TRY(a, copy_out(a, a->h_rows.p, static_cast<const SegRow*>(src->d_rows.p) + seg0, b->capacity * sizeof(SegRow)));
    rows = static_cast<const SegRow*>(a->h_rows.p) - seg0;  // unpack_align indexes rows by the batch's own segment offsets
  }
  unpack_align(src, st.data(), rows, out, &pool, read0, b->n, seg0);
  return DYN_OK;
}

int dyn_align_batch(dyn_aligner* a, uint64_t n_reads, const double* signals,
                    const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                    int calc_probabilities, dyn_align_out* out) {
  dyn_batch* b = nullptr;
  int rc = dyn_batch_create(a, n_reads, signals, sig_offsets, seqs, seq_offsets, &b);
  if (rc != DYN_OK) return rc;
  rc = dyn_batch_align(b, calc_probabilities);
  if (rc == DYN_OK) rc = dyn_batch_fetch(b, out);
  dyn_batch_destroy(b);
  return rc;
}

int dyn_batch_fetch_train(dyn_batch* b, dyn_train_out* out, double* pooled3n) {
  if (!b || !out || !out->Z || !out->status) return DYN_ERR_INVALID_ARGUMENT;
  dyn_aligner* a = b->a;
  if (!b->trained) {
    a->last_error = "dyn_batch_fetch_train before dyn_batch_train";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  int rc = need_device(a);
  if (rc != DYN_OK) return rc;
  const bool want_em = out->em_code && out->em_mean && out->em_stdev;
  if (want_em && out->capacity < b->capacity) {
    a->last_error = "dyn_train_out.capacity is smaller than dyn_segment_capacity()";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  std::vector<ReadState> st(b->n);
  std::vector<double> cw(b->total_cols), c1(b->total_cols), c2(b->total_cols), tr(2 * b->n);
  if (b->n) {
    HIP_TRY(a, copy_out(a, st.data(), b->d_state.p, b->n * sizeof(ReadState)));
    HIP_TRY(a, copy_out(a, tr.data(), b->d_trans.p, b->n * 16));
  }
  if (b->total_cols) {
    HIP_TRY(a, copy_out(a, cw.data(), b->d_colw.p, b->total_cols * 8));
    HIP_TRY(a, copy_out(a, c1.data(), b->d_cols1.p, b->total_cols * 8));
    HIP_TRY(a, copy_out(a, c2.data(), b->d_cols2.p, b->total_cols * 8));
  }
  finalise_train(b, st.data(), cw.data(), c1.data(), c2.data(), tr.data(), out, pooled3n);
  return DYN_OK;
}

int dyn_train_batch(dyn_aligner* a, uint64_t n_reads, const double* signals,
                    const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                    dyn_train_out* out, double* pooled3n) {
  dyn_batch* b = nullptr;
  int rc = dyn_batch_create(a, n_reads, signals, sig_offsets, seqs, seq_offsets, &b);
  if (rc != DYN_OK) return rc;
  rc = dyn_batch_train(b);
  if (rc == DYN_OK) rc = dyn_batch_fetch_train(b, out, pooled3n);
  dyn_batch_destroy(b);
  return rc;
}

int dyn_batch_device_pooled(dyn_batch* b, void** d_pooled3n, uint64_t* count) {
  if (!b || !b->trained) return DYN_ERR_INVALID_ARGUMENT;
  // Computed on the first call (k_pool_stats: the per-column sums of the training launch, sorted by k-mer and summed in a
  // fixed order -- the host's pooled sum bit for bit), on the compute stream, and waited for.
  dyn_aligner* a = b->a;
  std::lock_guard<std::mutex> lk(a->mu);
  if (!b->pooled_on_device) {
    if (int rc = need_device(a)) return rc;
    if (int rc = dyneng::session_quiesce(a)) return rc;  // (rocPRIM's sort does not start beside resident waves)
    const dynhost::PoreModel& m = a->model;
    HIP_TRY(a, b->d_pooled.ensure(3 * m.num_kmers * 8));
    HIP_TRY(a, hipMemsetAsync(b->d_pooled.p, 0, 3 * m.num_kmers * 8, a->stream));
    HIP_TRY(a, b->d_poolwork.ensure(std::max<size_t>(8, dynk::pool_stats_work_bytes(b->total_cols))));
    HIP_TRY(a, b->d_pooltemp.ensure(std::max<size_t>(8, dynk::pool_stats_temp_bytes(b->total_cols, m.num_kmers))));
    const dynk::TrainBuffers tr{b->d_colw.as<double>(), b->d_cols1.as<double>(), b->d_cols2.as<double>(), b->d_trans.as<double>()};
    HIP_TRY(a, dynk::launch_pool_stats(b->d_descs.as<ReadDesc>(), b->pool_nr, b->pool_max_N, b->d_state.as<ReadState>(), b->d_kmers.as<int32_t>(), tr,
                                       b->d_pooled.as<double>(), m.num_kmers, b->total_cols, b->d_poolwork.p, b->d_pooltemp.p,
                                       dynk::pool_stats_temp_bytes(b->total_cols, m.num_kmers), a->stream));
    HIP_TRY(a, hipStreamSynchronize(a->stream));
    b->pooled_on_device = true;
  }
  if (d_pooled3n) *d_pooled3n = b->d_pooled.p;
  if (count) *count = 3 * b->a->model.num_kmers;
  return DYN_OK;
}

}  // extern "C"

// DYN_BACKTRACE=1: the call stack of the thread that raises SIGABRT / SIGSEGV, written to stderr before the default action
// (debugging aid on boxes without a debugger; async-signal-safe calls only).
#include <execinfo.h>
#include <unistd.h>
#include <signal.h>
namespace {
void dyn_backtrace_handler(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "[dynamont_mi] fatal signal, backtrace:\n";
  (void)!write(2, msg, sizeof msg - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
struct DynBacktraceInstaller {
  DynBacktraceInstaller() {
    const char* e = std::getenv("DYN_BACKTRACE");
    if (e && *e == '1') {
      void* warm[2];
      (void)backtrace(warm, 2);  // loads libgcc now, not inside the handler
      signal(SIGABRT, dyn_backtrace_handler);
      signal(SIGSEGV, dyn_backtrace_handler);
    }
  }
} dyn_backtrace_installer;
}  // namespace
