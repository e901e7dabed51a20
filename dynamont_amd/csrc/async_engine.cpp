// async_engine.cpp -- the asynchronous batch pipeline behind dyn_batch_align_async /
// dyn_batch_train_async / dyn_batch_wait (include/dynamont_mi.h).
//
// What it replaces: the reference overlaps reading, preprocessing, alignment and writing by running
// `processes` worker processes behind a multiprocessing pool and a listener process
// (src/dynamont/segmentation/segment.py:301-325). One GPU replaces the workers; to keep it fed, the
// host stages of neighbouring batches have to run under the kernels of the current one:
//
//   caller        submit(k+2) ............................................. wait(k)
//   front thread  validate + k-mer code (k+1) | H2D (k+1) on s_in | launches (k+1) on the compute stream
//   GPU           kernels (k) ............................. | kernels (k+1) ...
//   copy-out      ............... D2H rows/state (k-1) on s_out
//   back thread   ............... unpack (k-1) into the caller's arrays -> done
//
// Stream order does all GPU-side synchronisation (events between the three streams); the only host
// waits are the back thread's hipEventSynchronize on a batch's last D2H and the caller's wait().
#include "engine.hpp"
#include "vbz_decode.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using dynk::ReadState;
using dynk::SegRow;
using dynmath::Emis;

namespace dyneng {

namespace {

int hip_fail(dyn_batch* b, hipError_t e, const char* what) {
  b->error = std::string("HIP error: ") + hipGetErrorString(e) + " at " + what;
  (void)hipGetLastError();
  return e == hipErrorOutOfMemory ? DYN_ERR_OUT_OF_MEMORY : DYN_ERR_DEVICE;
}

#define P_TRY(b, expr)                                        \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) return hip_fail((b), _e, #expr);    \
  } while (0)

// a copy of the handle's last error text (written under err_mu by whichever thread failed)
static std::string last_error_of(dyn_aligner* a) {
  std::lock_guard<std::mutex> lk(a->err_mu);
  return a->last_error;
}

// DYN_TRACE_HOST=1: wall time of the pipeline stages of every batch on stderr
const bool g_trace = std::getenv("DYN_TRACE_HOST") != nullptr;
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// page-locked host memory (hipHostMalloc / dyn_host_alloc): DMA can read it directly
bool is_pinned(const void* p) {
  hipPointerAttribute_t attr;
  std::memset(&attr, 0, sizeof attr);
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // unregistered memory reports an error on some runtimes
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

int helper_threads() {
  const unsigned hc = std::thread::hardware_concurrency();
  return (int)std::min<unsigned>(6, std::max<unsigned>(2, hc / 2));
}

}  // namespace

Pipeline::Pipeline(dyn_aligner* al) : a(al), helpers(helper_threads()) {
  t_front = std::thread([this] { front_loop(); });
  t_back = std::thread([this] { back_loop(); });
}

Pipeline::~Pipeline() {
  drain();
  {
    std::lock_guard<std::mutex> lk(m);
    stop = true;
  }
  cv_front.notify_all();
  cv_back.notify_all();
  if (t_front.joinable()) t_front.join();
  if (t_back.joinable()) t_back.join();
}

void Pipeline::submit(dyn_batch* b) {
  {
    std::unique_lock<std::mutex> lk(m);
    // Completion words of resident tickets are a ring of SESSION_FLAGS entries (pinned host memory): a caller that kept
    // that many tickets in flight would have a word reused while its first owner still waits on it. Back-pressure, not an
    // error: the submit waits until the pipeline has completed enough (no realistic caller gets here: a ticket in flight
    // holds its batch's buffers).
    cv_done.wait(lk, [&] { return in_flight + 8 < SESSION_FLAGS; });
    q_front.push_back(b);
    ++in_flight;
    peak_in_flight = std::max(peak_in_flight, in_flight);
  }
  cv_front.notify_one();
}

int Pipeline::wait(dyn_batch* b) {
  std::unique_lock<std::mutex> lk(m);
  cv_done.wait(lk, [&] { return b->done; });
  return b->rc;
}

void Pipeline::drain() {
  std::unique_lock<std::mutex> lk(m);
  cv_done.wait(lk, [&] { return in_flight == 0; });
}

// ---- merged launches ---------------------------------------------------------------------------------
// A read-queue launch of FEWER reads than waves cannot balance: every wave holds one read and the launch lasts as long
// as its slowest one (BASELINE config 2 with its 26 % certified reads: a fifth of the wave time idles). Tickets that are
// waiting while the GPU still has work queued are therefore merged into ONE batch -- one launch, one queue, the waves
// that finish early take the next reads -- instead of one launch each. Rules (front_loop):
//   * at most one launch is queued behind the one that runs: a third waits as tickets, where it can still be merged;
//   * with the GPU busy the front thread lingers up to 2 ms for tickets that are about to arrive (a consumer that has
//     just got two results back submits two new batches a fraction of a millisecond apart);
//   * only align tickets of the same kind (same calc flag, same raw format and preprocessing) are merged; training
//     tickets never are (their pooled statistics are per batch);
//   * a launch takes at most a QUARTER of the tickets the caller has had in flight at once (two at least), and at most
//     eight reads per wave slot: a caller that keeps 12 tickets in flight gets launches of three (four launches' worth
//     of tickets cover the one that runs, the one queued behind it, the one being unpacked and the one being collected
//     -- larger launches starve that pipeline: the CLI's run of 32 batches got 8 % slower with launches of five), one
//     that keeps 32 gets launches of eight, whose queue balances better (config 2, kernel time per batch: 42.0 ms at
//     three, 40.3 at eight, 39.2 without the certified reads).
// With nothing queued behind it a ticket starts at once, alone: latency is only ever added while the GPU is busy anyway.
constexpr size_t MERGE_MAX_TICKETS = 16;
constexpr uint64_t MERGE_READS_PER_SLOT = 8;

static uint64_t merge_reads_per_slot() {  // DYN_MERGE_READS_PER_SLOT: experiments
  static const uint64_t v = [] {
    const char* e = std::getenv("DYN_MERGE_READS_PER_SLOT");
    const long x = e ? std::atol(e) : 0;
    return x > 0 ? (uint64_t)x : MERGE_READS_PER_SLOT;
  }();
  return v;
}
static uint64_t merge_max_reads(const dyn_aligner* a) { return merge_reads_per_slot() * (uint64_t)a->n_cus * dynk::WAVES_PER_CU; }
static bool mergeable(const dyn_batch* x) {
  // batches that leave the waves underfilled (fewer than 1.5 reads per wave slot): a batch of four reads per slot balances
  // by itself, and merging those only makes the pipeline lumpy
  return x->job != DynJob::Train && x->n > 0 && 2 * x->n <= 3 * (uint64_t)x->a->n_cus * dynk::WAVES_PER_CU;
}
static bool same_kind(const dyn_batch* x, const dyn_batch* y) {
  if (x->job != y->job || x->has_raw != y->has_raw) return false;
  if (!x->has_raw) return true;
  const RawSource &p = x->raw_src, &q = y->raw_src;
  return p.vbz == q.vbz && p.dtype == q.dtype && p.window == q.window && p.n_sigmas == q.n_sigmas && p.compute_f32 == q.compute_f32;
}

BatchGroup::~BatchGroup() {
  if (g) dyn_batch_destroy(g);
}

// one batch out of several tickets: per-read metadata concatenated, bulk data left where it is (a pointer per read)
std::shared_ptr<BatchGroup> Pipeline::merge(const std::vector<dyn_batch*>& tickets) {
  auto grp = std::make_shared<BatchGroup>();
  dyn_batch* first = tickets[0];
  uint64_t n = 0;
  for (dyn_batch* t : tickets) n += t->n;
  grp->sig_offsets.assign(1, 0);
  grp->seq_offsets.assign(1, 0);
  const bool raw = first->has_raw, vbz = raw && first->raw_src.vbz;
  const size_t esz = raw ? first->raw_src.elem_size() : 8;
  if (vbz) grp->vbz_read_off.assign(1, 0);
  for (dyn_batch* t : tickets) {
    t->g_read0 = grp->sig_offsets.size() - 1;
    const RawSource& rs = t->raw_src;
    for (uint64_t i = 0; i < t->n; ++i) {
      const uint64_t len = t->in_sig_offsets[i + 1] - t->in_sig_offsets[i];
      grp->sig_offsets.push_back(grp->sig_offsets.back() + len);
      grp->seq_offsets.push_back(grp->seq_offsets.back() + (t->in_seq_offsets[i + 1] - t->in_seq_offsets[i]));
      if (!raw) {
        grp->ptrs.push_back(t->in_signals + t->in_sig_offsets[i]);
      } else if (vbz) {
        for (uint64_t c = rs.vbz_read_off[i]; c < rs.vbz_read_off[i + 1]; ++c) {
          grp->ptrs.push_back(rs.vbz_chunks[c]);
          grp->vbz_bytes.push_back(rs.vbz_bytes[c]);
          grp->vbz_samples.push_back(rs.vbz_samples[c]);
        }
        grp->vbz_read_off.push_back(grp->ptrs.size());
        grp->vbz_skip.push_back(rs.vbz_skip[i]);
      } else if (rs.scattered) {
        grp->ptrs.push_back(static_cast<const void* const*>(rs.raw)[i]);
      } else {
        grp->ptrs.push_back(static_cast<const char*>(rs.raw) + t->in_sig_offsets[i] * esz);
      }
      if (raw) {
        grp->shift.push_back(rs.shift[i]);
        grp->scale.push_back(rs.scale[i]);
        if (rs.dtype == 3) {
          grp->cal_offset.push_back(rs.cal_offset[i]);
          grp->cal_scale.push_back(rs.cal_scale[i]);
        }
      }
    }
    grp->seqs.append(t->in_seqs + t->in_seq_offsets[0], t->in_seq_offsets[t->n] - t->in_seq_offsets[0]);
  }
  dyn_batch* g = new dyn_batch();
  g->a = a;
  attach_cache(g);
  g->n = n;
  g->job = first->job;
  g->in_sig_offsets = grp->sig_offsets.data();
  g->in_seqs = grp->seqs.data();
  g->in_seq_offsets = grp->seq_offsets.data();
  if (!raw) {
    g->in_sig_ptrs = reinterpret_cast<const double* const*>(grp->ptrs.data());
  } else {
    g->has_raw = true;
    RawSource rs = first->raw_src;
    rs.raw = grp->ptrs.data();
    rs.scattered = !vbz;
    rs.shift = grp->shift.data();
    rs.scale = grp->scale.data();
    rs.cal_offset = rs.dtype == 3 ? grp->cal_offset.data() : nullptr;
    rs.cal_scale = rs.dtype == 3 ? grp->cal_scale.data() : nullptr;
    if (vbz) {
      rs.vbz_chunks = grp->ptrs.data();
      rs.vbz_bytes = grp->vbz_bytes.data();
      rs.vbz_samples = grp->vbz_samples.data();
      rs.vbz_read_off = grp->vbz_read_off.data();
      rs.vbz_skip = grp->vbz_skip.data();
    }
    g->raw_src = rs;
  }
  grp->g = g;
  grp->members = tickets;
  return grp;
}

void Pipeline::front_loop() {
  if (!a->host_only) (void)hipSetDevice(a->device);
  const bool merging = !a->host_only && std::getenv("DYN_NO_MERGE") == nullptr;
  for (;;) {
    std::vector<dyn_batch*> take;
    {
      std::unique_lock<std::mutex> lk(m);
      // A ticket for the resident read queue (session_candidate) is neither held back nor merged: it is published into the
      // running session as soon as its inputs are staged -- the waves take its reads when they get there.
      cv_front.wait(lk, [&] {
        return (stop || !q_front.empty()) && (launches_pending < 2 || (stop && q_front.empty()) || (!q_front.empty() && session_candidate(q_front.front())));
      });
      if (q_front.empty()) return;  // stop requested and nothing left
      const bool resident = session_candidate(q_front.front());
      const size_t max_tickets = std::min<size_t>(MERGE_MAX_TICKETS, std::max<size_t>(2, (size_t)(peak_in_flight / 4)));
      // the linger ends as soon as nothing more can join this launch, or the GPU has run dry (the back thread's notify)
      if (merging && !resident && launches_pending >= 1 && mergeable(q_front.front()) && q_front.size() < max_tickets)
        cv_front.wait_for(lk, std::chrono::milliseconds(2), [&] { return stop || launches_pending == 0 || q_front.size() >= max_tickets; });
      take.push_back(q_front.front());
      q_front.pop_front();
      uint64_t reads = take[0]->n;
      while (merging && !resident && launches_pending >= 1 && mergeable(take[0]) && !q_front.empty() && take.size() < max_tickets &&
             mergeable(q_front.front()) && same_kind(take[0], q_front.front()) && reads + q_front.front()->n <= merge_max_reads(a)) {
        reads += q_front.front()->n;
        take.push_back(q_front.front());
        q_front.pop_front();
      }
      ++launches_pending;
    }
    Work w;
    if (take.size() > 1) {
      w.grp = merge(take);
      w.b = w.grp->g;
      for (dyn_batch* t : take) t->group = w.grp;
    } else {
      w.b = take[0];
    }
    w.rc = front_stage(w.b);
    if (w.rc != DYN_OK && !a->host_only) {
      // whatever part of the batch was already enqueued must have left the GPU before the caller may tear it down
      for (hipStream_t st : {a->s_in, a->stream, a->s_out})
        if (st) (void)hipStreamSynchronize(st);
      (void)hipGetLastError();
    }
    {
      std::lock_guard<std::mutex> lk(m);
      q_back.push_back(w);  // failed batches pass through the back thread too: completion stays in order
    }
    cv_back.notify_one();
  }
}

void Pipeline::back_loop() {
  if (!a->host_only) (void)hipSetDevice(a->device);
  for (;;) {
    Work w;
    {
      std::unique_lock<std::mutex> lk(m);
      cv_back.wait(lk, [&] { return stop || !q_back.empty(); });
      if (q_back.empty()) return;
      w = q_back.front();
      q_back.pop_front();
    }
    int rc = w.rc;
    if (rc == DYN_OK) rc = back_stage(w.b, w.grp);
    if (rc != DYN_OK && w.b->in_session && !a->host_only) {
      // A session that lost a ticket takes no more of them -- and has LEFT before the ticket's buffers go back to the cache:
      // its waves may still be writing results of that ticket (they leave at the close mark, or when the idle watchdog fires)
      std::lock_guard<std::mutex> lk(a->mu);
      (void)hipSetDevice(a->device);
      (void)session_quiesce(a);
    }
    {
      std::lock_guard<std::mutex> lk(m);
      if (w.grp) {
        for (dyn_batch* t : w.grp->members) {
          t->rc = rc;
          if (rc != DYN_OK) t->error = w.b->error;
          t->done = true;
          --in_flight;
        }
        w.grp->members.clear();
      } else {
        w.b->rc = rc;
        w.b->done = true;
        --in_flight;
      }
      --launches_pending;
    }
    cv_done.notify_all();
    cv_front.notify_all();  // the gate on launches_pending
    close_idle_session();
  }
}

// Nothing left in the pipeline: the resident waves are told to leave (they would only poll). A ticket submitted a moment
// later opens the next session behind this one.
void Pipeline::close_idle_session() {
  if (a->host_only || !a->sess_open_hint.load()) return;
  {
    std::lock_guard<std::mutex> lk(m);
    if (in_flight != 0) return;
  }
  std::lock_guard<std::mutex> lk(a->mu);
  {
    std::lock_guard<std::mutex> lk2(m);
    if (in_flight != 0) return;  // (a ticket that arrives from here on finds the session closed and opens the next one)
  }
  (void)hipSetDevice(a->device);
  (void)session_close(a);
}

// Host prepare, H2D and every kernel launch of one batch; returns without waiting for the GPU.
int Pipeline::front_stage(dyn_batch* b) {
  const uint64_t n = b->n;
  const double t0 = now_ms();
  int rc = host_prepare(b, a->model, true, n, b->in_sig_offsets, b->in_seqs, b->in_seq_offsets, &helpers);
  const double t1 = now_ms();
  if (rc != DYN_OK) {
    b->error = last_error_of(a);
    return rc;
  }
  const uint64_t total_sig = n ? b->in_sig_offsets[n] - b->in_sig_offsets[0] : 0;
  std::lock_guard<std::mutex> lk(a->mu);
  const double t2 = now_ms();
  rc = alloc_batch_buffers(b, total_sig);
  if (rc != DYN_OK) {
    b->error = last_error_of(a);
    return rc;
  }
  const double t3 = now_ms();
  // ev_out is the one event a host thread waits on (the back thread, for the length of a launch): a blocking wait
  // leaves its core to the threads that decode, format and compress instead of spinning on it
  if (!b->ev_in) P_TRY(b, hipEventCreateWithFlags(&b->ev_in, hipEventDisableTiming));
  if (!b->ev_out) P_TRY(b, hipEventCreateWithFlags(&b->ev_out, hipEventDisableTiming | hipEventBlockingSync));
  // H2D on the copy-in stream. Pinned caller memory (dyn_host_alloc) is a true asynchronous DMA;
  // for pageable memory the runtime stages the copy and this thread blocks for its duration, which
  // is what the thread is for -- the compute stream keeps running the previous batch meanwhile.
  uint64_t raw_max_len = 0;
  if (total_sig && b->has_raw) {
    // RAW slices (segment.py:146-153 runs on the device): [offsets | shift | scale | samples] are gathered into ONE
    // pinned staging buffer by the helper threads and go up with one DMA; k_normalise / k_hampel then run on the
    // compute stream in front of the batch's read queue.
    const RawSource& rs = b->raw_src;
    const size_t esz = rs.elem_size();
    const uint64_t meta = (4 * n + 1) * 8;  // offsets, shift, scale, (cal_offset, cal_scale); 8-byte aligned
    P_TRY(b, b->h_sig.ensure(meta + total_sig * esz));
    uint64_t* h_offs = b->h_sig.as<uint64_t>();
    double* h_shift = reinterpret_cast<double*>(h_offs + n + 1);
    double* h_scale = h_shift + n;
    float* h_cal = reinterpret_cast<float*>(h_scale + n);  // [n] offsets, [n] scales
    char* h_raw = reinterpret_cast<char*>(h_cal + 2 * n);
    if (rs.dtype == 3) {
      std::memcpy(h_cal, rs.cal_offset, n * 4);
      std::memcpy(h_cal + n, rs.cal_scale, n * 4);
    }
    for (uint64_t i = 0; i <= n; ++i) h_offs[i] = b->in_sig_offsets[i] - b->in_sig_offsets[0];
    for (uint64_t i = 0; i < n; ++i) raw_max_len = std::max(raw_max_len, h_offs[i + 1] - h_offs[i]);
    std::memcpy(h_shift, rs.shift, n * 8);
    std::memcpy(h_scale, rs.scale, n * 8);
    const uint64_t bytes = total_sig * esz;
    if (rs.vbz) {
      // POD5 chunks, still compressed: every helper thread decodes whole reads (zstd + svb16 + zigzag + delta, the
      // decode ONT's pod5 library does inside record.signal) and copies the [start:end) slice into the staging buffer
      // A chunk that does not decode fails ITS read (status DYN_READ_BAD_SIGNAL, the slice zero-filled, no lattice work
      // for it) -- the reference's worker reports one line for such a read and goes on (segment.py:178-187).
      const int parts = std::max(1, std::min<int>(helpers.size() * 4, (int)(n / 4)));
      helpers.parallel_for(parts, [&](int t) {
        std::vector<uint8_t> tmp;
        std::vector<int16_t> whole;
        std::string err;
        for (uint64_t i = n * t / parts; i < n * (t + 1) / parts; ++i) {
          const uint64_t len = h_offs[i + 1] - h_offs[i];
          uint64_t total = 0;
          for (uint64_t c = rs.vbz_read_off[i]; c < rs.vbz_read_off[i + 1]; ++c) total += rs.vbz_samples[c];
          bool ok = rs.vbz_skip[i] + len <= total;
          if (ok && whole.size() < total) whole.resize(total);
          uint64_t pos = 0;
          for (uint64_t c = rs.vbz_read_off[i]; ok && c < rs.vbz_read_off[i + 1]; ++c) {
            ok = dynvbz::decode_chunk(rs.vbz_chunks[c], (size_t)rs.vbz_bytes[c], rs.vbz_samples[c], whole.data() + pos, tmp, err);
            pos += rs.vbz_samples[c];
          }
          if (ok) {
            std::memcpy(h_raw + h_offs[i] * 2, whole.data() + rs.vbz_skip[i], len * 2);
          } else {
            std::memset(h_raw + h_offs[i] * 2, 0, len * 2);
            if (b->reads[i].status == DYN_READ_OK) b->reads[i].status = DYN_READ_BAD_SIGNAL;  // (each read has one writer)
          }
        }
      });
    } else if (rs.scattered) {  // one pointer per read: the gather into the staging buffer IS the only host copy
      const void* const* slices = static_cast<const void* const*>(rs.raw);
      const int parts = std::max(1, std::min<int>(helpers.size() * 4, (int)(n / 8)));
      helpers.parallel_for(parts, [&](int t) {
        for (uint64_t i = n * t / parts; i < n * (t + 1) / parts; ++i)
          std::memcpy(h_raw + h_offs[i] * esz, slices[i], (h_offs[i + 1] - h_offs[i]) * esz);
      });
    } else {
      const char* src = static_cast<const char*>(rs.raw) + b->in_sig_offsets[0] * esz;
      const int parts = std::max(1, std::min<int>(helpers.size(), (int)(bytes >> 21)));
      helpers.parallel_for(parts, [&](int t) {
        const uint64_t lo = bytes * t / parts, hi = bytes * (t + 1) / parts;
        std::memcpy(h_raw + lo, src + lo, hi - lo);
      });
    }
    P_TRY(b, b->d_meta.ensure(meta + total_sig * esz));
    P_TRY(b, b->d_norm.ensure(total_sig * (rs.compute_f32 ? 4 : 8)));
    P_TRY(b, hipMemcpyAsync(b->d_meta.p, b->h_sig.p, meta + bytes, hipMemcpyHostToDevice, a->s_in));
  } else if (total_sig && b->in_sig_ptrs) {
    // a merged batch of float64 signals: one pointer per read, gathered into the pinned staging buffer by the helpers
    P_TRY(b, b->h_sig.ensure(total_sig * 8));
    double* dst = b->h_sig.as<double>();
    const int parts = std::max(1, std::min<int>(helpers.size() * 4, (int)(n / 8)));
    helpers.parallel_for(parts, [&](int t) {
      for (uint64_t i = n * t / parts; i < n * (t + 1) / parts; ++i)
        std::memcpy(dst + (b->in_sig_offsets[i] - b->in_sig_offsets[0]), b->in_sig_ptrs[i], (b->in_sig_offsets[i + 1] - b->in_sig_offsets[i]) * 8);
    });
    P_TRY(b, hipMemcpyAsync(b->d_sig.p, dst, total_sig * 8, hipMemcpyHostToDevice, a->s_in));
  } else if (total_sig) {
    const double* src = b->in_signals + b->in_sig_offsets[0];
    if (!is_pinned(src)) {
      // Pageable caller memory is staged HERE, by the helper threads, into a pinned buffer of the batch: measured,
      // hipMemcpyAsync from pageable memory blocks its caller for the length of the kernel that is running (~46 ms
      // instead of 3.4), which made this thread -- not the GPU -- the pipeline's bottleneck.
      P_TRY(b, b->h_sig.ensure(total_sig * 8));
      double* dst = b->h_sig.as<double>();
      const int parts = std::max(1, std::min<int>(helpers.size(), (int)(total_sig >> 18)));
      helpers.parallel_for(parts, [&](int t) {
        const uint64_t lo = total_sig * t / parts, hi = total_sig * (t + 1) / parts;
        std::memcpy(dst + lo, src + lo, (hi - lo) * 8);
      });
      src = dst;
    }
    P_TRY(b, hipMemcpyAsync(b->d_sig.p, src, total_sig * 8, hipMemcpyHostToDevice, a->s_in));
  }
  if (b->total_cols)
    P_TRY(b, hipMemcpyAsync(b->d_kmers.p, b->h_kmers.p, b->total_cols * 4, hipMemcpyHostToDevice, a->s_in));
  P_TRY(b, hipEventRecord(b->ev_in, a->s_in));
  const double t4 = now_ms();
  // The resident read queue (engine.hpp: Session)? Then the ticket's small kernels run on the copy-in stream, behind its
  // copies, and its record is published behind them; otherwise everything goes to the compute stream as one launch -- for
  // which the lattice pool must be free: an open session is closed and waited for first.
  bool resident = false;
  rc = session_plan(b, &resident);
  if (rc == DYN_OK && !resident) rc = session_quiesce(a);
  if (rc != DYN_OK) {
    b->error = last_error_of(a);
    return rc;
  }
  hipStream_t s_pre = resident ? a->s_in : a->stream;
  if (!resident) P_TRY(b, hipStreamWaitEvent(a->stream, b->ev_in, 0));
  if (total_sig && b->has_raw) {
    const RawSource& rs = b->raw_src;
    const uint64_t* d_offs = b->d_meta.as<uint64_t>();
    const double* d_shift = reinterpret_cast<const double*>(d_offs + n + 1);
    const float* d_cal = reinterpret_cast<const float*>(d_shift + 2 * n);
    dynk::launch_preprocess(reinterpret_cast<const char*>(d_cal + 2 * n), rs.dtype, rs.compute_f32, d_offs, d_shift, d_shift + n,
                            d_cal, d_cal + n, b->d_norm.p, b->d_sig.as<double>(), (int)n, raw_max_len, rs.window, rs.n_sigmas, s_pre);
    P_TRY(b, hipGetLastError());
  }
  if (b->total_cols) {
    dynk::launch_prep_params(b->d_kmers.as<int32_t>(), a->d_model.as<Emis>(), b->d_par.as<Emis>(), b->total_cols, (uint32_t)a->model.num_kmers, s_pre);
    P_TRY(b, hipGetLastError());
  }
  if (resident) {
    rc = session_publish(b);
    if (rc != DYN_OK) b->error = last_error_of(a);
    if (g_trace)
      std::fprintf(stderr, "[dyn] front %p: start %.2f prepare %.2f lock %.2f alloc %.2f h2d %.2f publish %.2f ms (resident)\n", (void*)b, t0, t1 - t0,
                   t2 - t1, t3 - t2, t4 - t3, now_ms() - t4);
    return rc;  // the copies out are enqueued by the back thread, once the ticket's completion word has been seen
  }
  rc = enqueue_job(b, b->job);  // records ev_done behind the batch's last kernel
  if (rc != DYN_OK) {
    b->error = last_error_of(a);
    return rc;  // front_loop drains the streams of a failed batch
  }
  P_TRY(b, hipStreamWaitEvent(a->s_out, b->ev_done, 0));
  // D2H into pinned per-batch buffers on the copy-out stream
  if (n) P_TRY(b, hipMemcpyAsync(b->h_state.p, b->d_state.p, n * sizeof(ReadState), hipMemcpyDeviceToHost, a->s_out));
  if (b->job == DynJob::AlignFull && b->capacity) {
    P_TRY(b, b->h_rows.ensure(b->capacity * sizeof(SegRow)));
    P_TRY(b, hipMemcpyAsync(b->h_rows.p, b->d_rows.p, b->capacity * sizeof(SegRow), hipMemcpyDeviceToHost, a->s_out));
  } else if (b->job == DynJob::Train) {
    // [colw | cols1 | cols2 | trans] in one pinned buffer; the per-column sums are only needed for the
    // per-read emission updates and the host-side pooled sum
    const dyn_train_out* ot = b->out_train;
    const bool need_cols = (ot->em_code && ot->em_mean && ot->em_stdev) || b->out_pooled;
    const uint64_t c = need_cols ? b->total_cols : 0;
    P_TRY(b, b->h_rows.ensure(std::max<uint64_t>(8, (3 * c + 2 * n) * 8)));
    double* h = b->h_rows.as<double>();
    if (c) {
      P_TRY(b, hipMemcpyAsync(h, b->d_colw.p, c * 8, hipMemcpyDeviceToHost, a->s_out));
      P_TRY(b, hipMemcpyAsync(h + c, b->d_cols1.p, c * 8, hipMemcpyDeviceToHost, a->s_out));
      P_TRY(b, hipMemcpyAsync(h + 2 * c, b->d_cols2.p, c * 8, hipMemcpyDeviceToHost, a->s_out));
    }
    if (n) P_TRY(b, hipMemcpyAsync(h + 3 * c, b->d_trans.p, n * 16, hipMemcpyDeviceToHost, a->s_out));
  }
  P_TRY(b, hipEventRecord(b->ev_out, a->s_out));
  if (g_trace)
    std::fprintf(stderr, "[dyn] front %p: start %.2f prepare %.2f lock %.2f alloc %.2f h2d %.2f enqueue+d2h %.2f ms\n", (void*)b,
                 t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3, now_ms() - t4);
  return DYN_OK;
}

// A ticket of the resident read queue is complete when the wave that finished its last read has raised the ticket's word
// in pinned host memory. Polled with short sleeps; every few milliseconds the ticket's own counter and the session's abort
// word are copied out as a second opinion (a word that never arrives must not hang the pipeline), and the wait is bounded.
constexpr int SESSION_LOST = -1000;  // wait_resident -> back_stage only

int Pipeline::wait_resident(dyn_batch* b) {
  const double t0 = now_ms();
  const double limit_ms = 1e3 * (std::getenv("DYN_SESSION_WAIT_S") ? std::atof(std::getenv("DYN_SESSION_WAIT_S")) : 300.0);
  uint32_t* h = b->h_stats.as<uint32_t>();  // pinned; the statistics copy overwrites it afterwards
  bool published = false;  // events[0]: the counter has been cleared on the device (session_publish)
  for (uint64_t spin = 0;; ++spin) {
    if (*b->sess_flag == b->sess_reads) return DYN_OK;
    if ((spin & 63) == 63 && !published) {
      const hipError_t q = hipEventQuery(b->events[0]);
      if (q == hipSuccess) published = true;
      else {
        (void)hipGetLastError();  // hipErrorNotReady is not an error of this thread's next launch
        if (q != hipErrorNotReady) P_TRY(b, q);
        if (now_ms() - t0 > limit_ms) {
          b->error = "the copy-in stream did not reach the ticket's record within DYN_SESSION_WAIT_S seconds";
          return DYN_ERR_DEVICE;
        }
      }
    }
    if ((spin & 63) == 63 && published) {
      const double c0 = now_ms();
      P_TRY(b, hipMemcpyAsync(h, b->d_tctl.p, 4, hipMemcpyDeviceToHost, a->s_out));
      P_TRY(b, hipMemcpyAsync(h + 1, a->sess_ctl[b->sess_blk].as<uint32_t>() + dynk::S_ABORT, 4, hipMemcpyDeviceToHost, a->s_out));
      const double c1 = now_ms();
      P_TRY(b, hipStreamSynchronize(a->s_out));
      if (g_trace && now_ms() - c0 > 20.0)
        std::fprintf(stderr, "[dyn] back  %p: second opinion took %.1f ms to enqueue, %.1f ms to complete (flag %u, counter %u of %u)\n", (void*)b, c1 - c0,
                     now_ms() - c1, *b->sess_flag, h[0], b->sess_reads);
      if (h[0] == b->sess_reads) return DYN_OK;  // the counter is there; the word is on its way
      // The session aborted (its waves found nothing to do for DYN_SESSION_IDLE_S seconds -- a slow front stage, a stopped
      // host -- and left), or its control block already serves a later session: the ticket was published to waves that are
      // gone, or to waves that stop taking reads. back_stage waits for that kernel's end and publishes again what is incomplete
      // then (session_recover: at most twice).
      const bool lost = h[1] != 0 || a->sess.blk_gen[b->sess_blk].load() != b->sess_gen;
      if (lost) return SESSION_LOST;
      if (now_ms() - t0 > limit_ms) {
        // what the queue looked like: reads claimed, tickets published, closed, abort -- and this ticket's own counter
        uint32_t* cw = h + 2;
        P_TRY(b, hipMemcpyAsync(cw, a->sess_ctl[b->sess_blk].p, 16, hipMemcpyDeviceToHost, a->s_out));
        P_TRY(b, hipStreamSynchronize(a->s_out));
        char what[256];
        std::snprintf(what, sizeof what, " (reads claimed %u, tickets published %u, closed %u, abort %u; this ticket: %u of %u reads done)", cw[0], cw[1],
                      cw[2], cw[3], h[0], b->sess_reads);
        b->error = std::string("the resident read queue did not finish a ticket within DYN_SESSION_WAIT_S seconds") + what;
        return DYN_ERR_DEVICE;
      }
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

// Wait for the batch's last D2H, then turn the device records into the caller's arrays.
int Pipeline::back_stage(dyn_batch* b, const std::shared_ptr<BatchGroup>& grp) {
  const double t0 = now_ms();
  int rc = DYN_OK;
  if (b->in_session) {
    while ((rc = wait_resident(b)) == SESSION_LOST) {
      std::lock_guard<std::mutex> lk(a->mu);
      P_TRY(b, hipSetDevice(a->device));
      if (g_trace) std::fprintf(stderr, "[dyn] back  %p: the ticket's session aborted before it took all of its reads; publishing it again\n", (void*)b);
      bool again = false;
      rc = session_recover(b, &again);
      if (rc != DYN_OK) {
        b->error = last_error_of(a);
        return rc;
      }
      if (!again) break;  // complete after all
    }
    if (rc != DYN_OK) return rc;
    // The last ticket in flight is complete and nothing waits behind it: the resident waves would only poll while this thread
    // runs the per-segment kernels and the copies out (config 3: ~100 ms for a batch's rows) -- they are told to leave NOW, not
    // when the pipeline has run dry (close_idle_session, which remains for the other ways a pipeline empties). A ticket that
    // arrives a moment later opens the next session, as it would have after the back stage.
    if (!std::getenv("DYN_SESSION_LATE_CLOSE")) {
      std::lock_guard<std::mutex> lk(a->mu);
      bool last;
      {
        std::lock_guard<std::mutex> lk2(m);
        last = in_flight == 1;
      }
      if (last && a->sess_open_hint.load()) {
        P_TRY(b, hipSetDevice(a->device));
        (void)session_close(a);
      }
    }
    // the per-segment kernels and the copies out, beside the resident waves
    const double f0 = now_ms();
    rc = session_finish_enqueue(b, a->s_out);
    if (g_trace) std::fprintf(stderr, "[dyn] back  %p: resident ticket complete after %.2f ms, finish enqueued in %.2f ms\n", (void*)b, f0 - t0, now_ms() - f0);
    if (rc != DYN_OK) {
      b->error = last_error_of(a);
      return rc;
    }
    const uint64_t n = b->n;
    if (n) P_TRY(b, hipMemcpyAsync(b->h_state.p, b->d_state.p, n * sizeof(ReadState), hipMemcpyDeviceToHost, a->s_out));
    if (b->capacity) {
      P_TRY(b, b->h_rows.ensure(b->capacity * sizeof(SegRow)));
      P_TRY(b, hipMemcpyAsync(b->h_rows.p, b->d_rows.p, b->capacity * sizeof(SegRow), hipMemcpyDeviceToHost, a->s_out));
    }
    P_TRY(b, hipEventRecord(b->ev_out, a->s_out));
  }
  P_TRY(b, hipEventSynchronize(b->ev_out));
  const double t1 = now_ms();
  rc = b->in_session ? session_collect_timing(b) : collect_timing(b);
  if (rc != DYN_OK) {
    b->error = last_error_of(a);
    return rc;
  }
  const ReadState* st = b->h_state.as<ReadState>();
  if (b->job == DynJob::Train) {
    const dyn_train_out* ot = b->out_train;
    const bool need_cols = (ot->em_code && ot->em_mean && ot->em_stdev) || b->out_pooled;
    const uint64_t c = need_cols ? b->total_cols : 0;
    const double* h = b->h_rows.as<double>();
    finalise_train(b, st, h, h + c, h + 2 * c, h + 3 * c, b->out_train, b->out_pooled);
  } else if (grp) {
    // a merged launch: every member gets its reads' results in its own arrays, the launch's timing, and its own counts
    // (dyn_timing.launch_share = its part of the launch, by lattice cells)
    for (dyn_batch* t : grp->members) {
      dyn_align_out* out = t->out_align;
      t->capacity = 0;
      for (uint64_t i = 0; i < t->n; ++i) t->capacity += b->reads[t->g_read0 + i].kc;
      t->g_seg0 = t->n ? b->reads[t->g_read0].seg_off : 0;
      const bool want_rows = b->job == DynJob::AlignFull && t->capacity &&
                             (out->sequence_positions || out->signal_positions || out->probabilities || out->states);
      unpack_align(b, st, want_rows ? b->h_rows.as<SegRow>() : nullptr, out, &helpers, t->g_read0, t->n, t->g_seg0);
      dyn_timing tm = b->timing;
      tm.cells = tm.samples = tm.reads_ok = 0;
      tm.reads_strict = 0;
      for (uint64_t i = 0; i < t->n; ++i) {
        const HostRead& r = b->reads[t->g_read0 + i];
        if (r.status != DYN_READ_OK || st[t->g_read0 + i].status == DYN_READ_TOO_LARGE) continue;  // never reached the device
        const uint64_t N = r.kc + 1, bw = std::min<uint64_t>(a->model.half_band, N / 2);
        tm.cells += (r.S + 1) * std::min<uint64_t>(2 * bw + 1, N);
        tm.samples += r.S;
        ++tm.reads_ok;
        if (t->g_read0 + i < b->strict_flag.size()) tm.reads_strict += b->strict_flag[t->g_read0 + i];
      }
      tm.launch_share = b->timing.cells ? (double)tm.cells / (double)b->timing.cells : 0.0;
      t->timing = tm;
      t->aligned = true;
      t->last_calc = b->last_calc;
    }
  } else {
    dyn_align_out* out = b->out_align;
    const bool want_rows = b->job == DynJob::AlignFull && b->capacity &&
                           (out->sequence_positions || out->signal_positions || out->probabilities || out->states);
    unpack_align(b, st, want_rows ? b->h_rows.as<SegRow>() : nullptr, out, &helpers);
  }
  if (g_trace) std::fprintf(stderr, "[dyn] back  %p: waited %.2f (until %.2f) unpack %.2f ms\n", (void*)b, t1 - t0, t1, now_ms() - t1);
  return DYN_OK;
}

}  // namespace dyneng

using namespace dyneng;

namespace {

int submit_common(dyn_aligner* a, uint64_t n_reads, const double* signals, const uint64_t* sig_offsets,
                  const char* seqs, const uint64_t* seq_offsets, DynJob job, dyn_align_out* oa,
                  dyn_train_out* ot, double* pooled, dyn_batch** ticket, const RawSource* rs = nullptr) {
  if (!a || !ticket || !sig_offsets || !seq_offsets || (n_reads && ((!signals && !rs) || !seqs))) return DYN_ERR_INVALID_ARGUMENT;
  RawSource rs_local;
  if (rs) {  // DYN_RAW_SCATTERED: `raw` is a table of n_reads pointers
    rs_local = *rs;
    rs_local.scattered = (rs->dtype & DYN_RAW_SCATTERED) != 0;
    rs_local.dtype = rs->dtype & ~DYN_RAW_SCATTERED;
    rs = &rs_local;
  }
  if (rs && (rs->dtype < 0 || rs->dtype > 3 || rs->window < 1 || rs->window > 16 || (n_reads && (!rs->raw || !rs->shift || !rs->scale)) ||
             (rs->dtype == 3 && n_reads && (!rs->cal_offset || !rs->cal_scale)))) {
    a->last_error = "raw batches: raw_dtype must be 0 (f32), 1 (i16), 2 (f64) or 3 (i16 + calibration arrays) and 1 <= window <= 16";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  *ticket = nullptr;
  {
    std::lock_guard<std::mutex> lk(a->mu);
    int rc = need_device(a);
    if (rc != DYN_OK) return rc;
    if (!a->pipe) a->pipe.reset(new Pipeline(a));
  }
  const uint64_t cap = dyn_segment_capacity(a, n_reads, seq_offsets);
  if (oa) {
    if (!oa->Z || !oa->status) return DYN_ERR_INVALID_ARGUMENT;
    const bool want_rows = job == DynJob::AlignFull && (oa->sequence_positions || oa->signal_positions || oa->probabilities || oa->states);
    if (want_rows && oa->capacity < cap) {
      a->last_error = "dyn_align_out.capacity is smaller than dyn_segment_capacity()";
      return DYN_ERR_INVALID_ARGUMENT;
    }
  }
  if (ot) {
    if (!ot->Z || !ot->status) return DYN_ERR_INVALID_ARGUMENT;
    if (ot->em_code && ot->em_mean && ot->em_stdev && ot->capacity < cap) {
      a->last_error = "dyn_train_out.capacity is smaller than dyn_segment_capacity()";
      return DYN_ERR_INVALID_ARGUMENT;
    }
  }
  dyn_batch* b = new dyn_batch();
  b->a = a;
  attach_cache(b);
  b->n = n_reads;
  b->async = true;
  b->job = job;
  b->in_signals = signals;
  if (rs) {
    b->has_raw = true;
    b->raw_src = *rs;
  }
  b->in_sig_offsets = sig_offsets;
  b->in_seqs = seqs;
  b->in_seq_offsets = seq_offsets;
  b->out_align = oa;
  b->out_train = ot;
  b->out_pooled = pooled;
  a->pipe->submit(b);
  *ticket = b;
  return DYN_OK;
}

}  // namespace

extern "C" {

int dyn_batch_align_async(dyn_aligner* a, uint64_t n_reads, const double* signals,
                          const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                          int calc_probabilities, dyn_align_out* out, dyn_batch** ticket) {
  if (!out) return DYN_ERR_INVALID_ARGUMENT;
  return submit_common(a, n_reads, signals, sig_offsets, seqs, seq_offsets,
                       calc_probabilities ? DynJob::AlignFull : DynJob::AlignZ, out, nullptr, nullptr, ticket);
}

int dyn_batch_train_async(dyn_aligner* a, uint64_t n_reads, const double* signals,
                          const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                          dyn_train_out* out, double* pooled3n, dyn_batch** ticket) {
  if (!out) return DYN_ERR_INVALID_ARGUMENT;
  return submit_common(a, n_reads, signals, sig_offsets, seqs, seq_offsets, DynJob::Train, nullptr, out, pooled3n, ticket);
}

int dyn_batch_align_raw_async(dyn_aligner* a, uint64_t n_reads, const void* raw, int raw_dtype,
                              const uint64_t* raw_offsets, const float* cal_offset, const float* cal_scale,
                              const double* shift, const double* scale, int hampel_window,
                              double hampel_n_sigmas, int compute_f32, const char* seqs, const uint64_t* seq_offsets,
                              int calc_probabilities, dyn_align_out* out, dyn_batch** ticket) {
  if (!out) return DYN_ERR_INVALID_ARGUMENT;
  RawSource rs;
  rs.raw = raw;
  rs.dtype = raw_dtype;
  rs.cal_offset = cal_offset;
  rs.cal_scale = cal_scale;
  rs.shift = shift;
  rs.scale = scale;
  rs.window = hampel_window;
  rs.n_sigmas = hampel_n_sigmas;
  rs.compute_f32 = compute_f32;
  return submit_common(a, n_reads, nullptr, raw_offsets, seqs, seq_offsets, calc_probabilities ? DynJob::AlignFull : DynJob::AlignZ,
                       out, nullptr, nullptr, ticket, &rs);
}

int dyn_batch_train_raw_async(dyn_aligner* a, uint64_t n_reads, const void* raw, int raw_dtype,
                              const uint64_t* raw_offsets, const float* cal_offset, const float* cal_scale,
                              const double* shift, const double* scale, int hampel_window,
                              double hampel_n_sigmas, int compute_f32, const char* seqs, const uint64_t* seq_offsets,
                              dyn_train_out* out, double* pooled3n, dyn_batch** ticket) {
  if (!out) return DYN_ERR_INVALID_ARGUMENT;
  RawSource rs;
  rs.raw = raw;
  rs.dtype = raw_dtype;
  rs.cal_offset = cal_offset;
  rs.cal_scale = cal_scale;
  rs.shift = shift;
  rs.scale = scale;
  rs.window = hampel_window;
  rs.n_sigmas = hampel_n_sigmas;
  rs.compute_f32 = compute_f32;
  return submit_common(a, n_reads, nullptr, raw_offsets, seqs, seq_offsets, DynJob::Train, nullptr, out, pooled3n, ticket, &rs);
}

int dyn_batch_align_vbz_async(dyn_aligner* a, uint64_t n_reads, const void* const* chunk_ptrs, const uint64_t* chunk_bytes,
                              const uint32_t* chunk_samples, const uint64_t* read_chunk_offsets, const uint64_t* slice_start,
                              const uint64_t* raw_offsets, const float* cal_offset, const float* cal_scale,
                              const double* shift, const double* scale, int hampel_window, double hampel_n_sigmas,
                              int compute_f32, const char* seqs, const uint64_t* seq_offsets, int calc_probabilities,
                              dyn_align_out* out, dyn_batch** ticket) {
  if (!out || !read_chunk_offsets || !slice_start || (n_reads && (!chunk_ptrs || !chunk_bytes || !chunk_samples)))
    return DYN_ERR_INVALID_ARGUMENT;
  RawSource rs;
  rs.raw = chunk_ptrs;  // not read as samples: vbz
  rs.dtype = cal_offset ? 3 : 1;
  rs.cal_offset = cal_offset;
  rs.cal_scale = cal_scale;
  rs.shift = shift;
  rs.scale = scale;
  rs.window = hampel_window;
  rs.n_sigmas = hampel_n_sigmas;
  rs.compute_f32 = compute_f32;
  rs.vbz = true;
  rs.vbz_chunks = chunk_ptrs;
  rs.vbz_bytes = chunk_bytes;
  rs.vbz_samples = chunk_samples;
  rs.vbz_read_off = read_chunk_offsets;
  rs.vbz_skip = slice_start;
  return submit_common(a, n_reads, nullptr, raw_offsets, seqs, seq_offsets, calc_probabilities ? DynJob::AlignFull : DynJob::AlignZ,
                       out, nullptr, nullptr, ticket, &rs);
}

int dyn_batch_wait(dyn_batch* b) {
  if (!b) return DYN_ERR_INVALID_ARGUMENT;
  if (!b->async) return DYN_OK;  // synchronous batches are complete when their call returns
  dyn_aligner* a = b->a;
  const int rc = a->pipe->wait(b);
  if (rc != DYN_OK) a->last_error = b->error;
  return rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Several GPUs behind ONE handle ("one handle may drive 1-8 GPUs"). Reads are
// independent (NTAligner::align keeps no cross-read state, NT_aligner_api.cpp:230-312), so a batch is
// cut into n_devices contiguous ranges of equal lattice work (sum of signal lengths); every range goes
// through its own device's asynchronous pipeline and writes straight into the caller's arrays. No
// device-to-device traffic: a C/C++ consumer of the library needs neither Python nor RCCL to shard.
// ---------------------------------------------------------------------------------------------
struct dyn_multi {
  std::vector<dyn_aligner*> dev;
  std::string last_error;
};

namespace {

// cut points: range d = [cut[d], cut[d+1])
std::vector<uint64_t> split_by_work(uint64_t n, const uint64_t* sig_offsets, size_t parts) {
  std::vector<uint64_t> cut(parts + 1, n);
  cut[0] = 0;
  const uint64_t total = n ? sig_offsets[n] - sig_offsets[0] : 0;
  uint64_t i = 0;
  for (size_t d = 1; d < parts; ++d) {
    const uint64_t want = sig_offsets[0] + total * d / parts;
    while (i < n && sig_offsets[i] < want) ++i;
    cut[d] = i;
  }
  return cut;
}

}  // namespace

extern "C" {

int dyn_multi_create(const char* model_path, int pore, const char* mode, int threads, uint64_t band,
                     const int* device_ids, int n_devices, dyn_multi** out, char* err, uint64_t errcap) {
  if (!out || !device_ids || n_devices < 1) return DYN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  dyn_multi* m = new dyn_multi();
  for (int d = 0; d < n_devices; ++d) {
    dyn_aligner* a = nullptr;
    const int rc = dyn_aligner_create(model_path, pore, mode, threads, band, device_ids[d], &a, err, errcap);
    if (rc != DYN_OK) {
      for (dyn_aligner* x : m->dev) dyn_aligner_destroy(x);
      delete m;
      return rc;
    }
    m->dev.push_back(a);
  }
  *out = m;
  return DYN_OK;
}

void dyn_multi_destroy(dyn_multi* m) {
  if (!m) return;
  for (dyn_aligner* a : m->dev) dyn_aligner_destroy(a);
  delete m;
}

int dyn_multi_device_count(const dyn_multi* m) { return m ? (int)m->dev.size() : 0; }

dyn_aligner* dyn_multi_handle(dyn_multi* m, int i) {
  return (m && i >= 0 && i < (int)m->dev.size()) ? m->dev[i] : nullptr;
}

const char* dyn_multi_last_error(const dyn_multi* m) { return m ? m->last_error.c_str() : ""; }

int dyn_multi_align_batch(dyn_multi* m, uint64_t n_reads, const double* signals, const uint64_t* sig_offsets,
                          const char* seqs, const uint64_t* seq_offsets, int calc_probabilities, dyn_align_out* out) {
  if (!m || !out || !out->Z || !out->status || !sig_offsets || !seq_offsets) return DYN_ERR_INVALID_ARGUMENT;
  const size_t nd = m->dev.size();
  const int k = m->dev[0]->model.k;
  // segment offsets of every read in the caller's arrays (prefix sums of max(0, len - k + 1))
  std::vector<uint64_t> seg(n_reads + 1, 0);
  for (uint64_t i = 0; i < n_reads; ++i) {
    const uint64_t L = seq_offsets[i + 1] - seq_offsets[i];
    seg[i + 1] = seg[i] + (L >= (uint64_t)k ? L - (uint64_t)k + 1 : 0);
  }
  const bool want_rows = calc_probabilities && (out->sequence_positions || out->signal_positions || out->probabilities || out->states);
  if (want_rows && out->capacity < seg[n_reads]) {
    m->last_error = "dyn_align_out.capacity is smaller than dyn_segment_capacity()";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  const std::vector<uint64_t> cut = split_by_work(n_reads, sig_offsets, nd);
  std::vector<dyn_align_out> sub(nd);
  std::vector<dyn_batch*> ticket(nd, nullptr);
  int rc = DYN_OK;
  for (size_t d = 0; d < nd; ++d) {
    const uint64_t lo = cut[d], hi = cut[d + 1], sb = seg[lo];
    dyn_align_out& o = sub[d];
    o = *out;
    o.Z += lo;
    o.status += lo;
    if (o.bad_char) o.bad_char += lo;
    o.seg_offsets = nullptr;  // rebuilt below: neighbouring ranges would both write the entry at their common border
    if (o.n_segments) o.n_segments += lo;
    if (o.sequence_positions) o.sequence_positions += sb;
    if (o.signal_positions) o.signal_positions += sb;
    if (o.probabilities) o.probabilities += sb;
    if (o.states) o.states += sb;
    o.capacity = seg[hi] - sb;
    const int r = dyn_batch_align_async(m->dev[d], hi - lo, signals, sig_offsets + lo, seqs, seq_offsets + lo,
                                        calc_probabilities, &o, &ticket[d]);
    if (r != DYN_OK) {
      rc = r;
      m->last_error = dyn_aligner_last_error(m->dev[d]);
    }
  }
  for (size_t d = 0; d < nd; ++d) {
    if (!ticket[d]) continue;
    const int r = dyn_batch_wait(ticket[d]);
    if (r != DYN_OK && rc == DYN_OK) {
      rc = r;
      m->last_error = dyn_aligner_last_error(m->dev[d]);
    }
    dyn_batch_destroy(ticket[d]);
  }
  if (out->seg_offsets) std::memcpy(out->seg_offsets, seg.data(), (n_reads + 1) * sizeof(uint64_t));
  return rc;
}

int dyn_multi_train_batch(dyn_multi* m, uint64_t n_reads, const double* signals, const uint64_t* sig_offsets,
                          const char* seqs, const uint64_t* seq_offsets, dyn_train_out* out, double* pooled3n) {
  if (!m || !out || !out->Z || !out->status || !sig_offsets || !seq_offsets) return DYN_ERR_INVALID_ARGUMENT;
  const size_t nd = m->dev.size();
  const int k = m->dev[0]->model.k;
  const uint64_t K = m->dev[0]->model.num_kmers;
  std::vector<uint64_t> seg(n_reads + 1, 0);
  for (uint64_t i = 0; i < n_reads; ++i) {
    const uint64_t L = seq_offsets[i + 1] - seq_offsets[i];
    seg[i + 1] = seg[i] + (L >= (uint64_t)k ? L - (uint64_t)k + 1 : 0);
  }
  const bool want_em = out->em_code && out->em_mean && out->em_stdev;
  if (want_em && out->capacity < seg[n_reads]) {
    m->last_error = "dyn_train_out.capacity is smaller than dyn_segment_capacity()";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  const std::vector<uint64_t> cut = split_by_work(n_reads, sig_offsets, nd);
  std::vector<dyn_train_out> sub(nd);
  std::vector<dyn_batch*> ticket(nd, nullptr);
  // every device sums into its own pooled array (the back threads run concurrently); added up at the end
  std::vector<std::vector<double>> pooled(pooled3n ? nd : 0, std::vector<double>(pooled3n ? 3 * K : 0, 0.0));
  int rc = DYN_OK;
  for (size_t d = 0; d < nd; ++d) {
    const uint64_t lo = cut[d], hi = cut[d + 1], sb = seg[lo];
    dyn_train_out& o = sub[d];
    o = *out;
    o.Z += lo;
    o.status += lo;
    if (o.bad_char) o.bad_char += lo;
    if (o.transitions) o.transitions += 3 * lo;
    o.em_offsets = nullptr;
    if (o.em_count) o.em_count += lo;
    if (o.em_code) o.em_code += sb;
    if (o.em_mean) o.em_mean += sb;
    if (o.em_stdev) o.em_stdev += sb;
    if (o.em_weight) o.em_weight += sb;
    if (o.em_sum) o.em_sum += sb;
    if (o.em_sumsq) o.em_sumsq += sb;
    if (o.trans_counts) o.trans_counts += 2 * lo;
    o.capacity = seg[hi] - sb;
    const int r = dyn_batch_train_async(m->dev[d], hi - lo, signals, sig_offsets + lo, seqs, seq_offsets + lo, &o,
                                        pooled3n ? pooled[d].data() : nullptr, &ticket[d]);
    if (r != DYN_OK) {
      rc = r;
      m->last_error = dyn_aligner_last_error(m->dev[d]);
    }
  }
  for (size_t d = 0; d < nd; ++d) {
    if (!ticket[d]) continue;
    const int r = dyn_batch_wait(ticket[d]);
    if (r != DYN_OK && rc == DYN_OK) {
      rc = r;
      m->last_error = dyn_aligner_last_error(m->dev[d]);
    }
    dyn_batch_destroy(ticket[d]);
  }
  if (out->em_offsets) std::memcpy(out->em_offsets, seg.data(), (n_reads + 1) * sizeof(uint64_t));
  if (pooled3n)
    for (size_t d = 0; d < nd; ++d)
      for (uint64_t j = 0; j < 3 * K; ++j) pooled3n[j] += pooled[d][j];
  return rc;
}

}  // extern "C"
