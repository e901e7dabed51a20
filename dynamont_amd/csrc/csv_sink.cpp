// csv_sink.cpp -- the output half of dynamont-resquiggle as native threads: wait for a batch of the asynchronous
// engine, format its rows, compress them into ONE zstd frame, write the file; `.errors` lines for failed reads.
//
// Reference being replaced: the listener process of src/dynamont/segmentation/segment.py:69-107 (a zstd level-3
// stream writer fed through a multiprocessing queue) and the per-read formatting of utils.py:193-232
// (csv_format.cpp). One GPU produces ~1 GB of CSV per second; a Python thread per stage loses most of that to the
// interpreter lock while the producer thread parses the next reads, so the whole back half lives here:
//
//   sink thread      dyn_batch_wait(ticket) -> dyn_format_csv (its own threads) -> dyn_csv_compact -> cut into jobs
//   compress threads one zstd context each; job k of the frame (see below)
//   writer thread    jobs in order -> fwrite
//
// One frame from independent jobs (what zstd's own multi-threaded mode does; the image's libzstd 1.4.8 is built
// without it, and ships no header -- the few entry points used are bound with dlopen): every job is compressed by
// a context of its own with the same parameters. Job 0 keeps its frame header; every later job flushes its header
// into the void (ZSTD_compressContinue with no input) and calls ZSTD_invalidateRepCodes so that its first block
// does not lean on repeat offsets the decoder will not have at that point; an empty last block closes the frame.
// Any zstd decoder sees one ordinary frame (tests/test_harness.py, tests/test_gpu_harness.py).
#include "../../include/dynamont_mi.h"
#include "zstd_dl.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <ctime>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr size_t kJobBytes = 4u << 20;
const char kHeader[] = "readid,signalid,start,end,basepos,base,motif,state,posterior_probability,polish\n";  // segment.py:80

struct Blob {  // the formatted rows of one batch; freed when its last job has been compressed
  std::unique_ptr<char[]> data;
  size_t cap = 0;
  // Every read is formatted into its own worst-case slot [begin, end) of `data` (dyn_format_csv); the rows as the file
  // holds them are those slots back to back, prefix[i] = bytes in front of read i. The jobs are cut in THAT coordinate
  // and each compress thread gathers its 4 MB out of the slots -- no serial compaction pass on the sink thread.
  std::vector<uint64_t> begin, end, prefix;
};

struct Job {
  uint64_t index = 0;
  std::shared_ptr<Blob> blob;  // keeps the source alive
  size_t off = 0, len = 0;  // [off, off + len) of the blob's rows, in the compacted coordinate
  bool last = false;  // the terminating job: no data, closes the frame
  std::vector<char> out;
  bool done = false;
};

struct Item {
  dyn_aligner* a;
  dyn_batch* ticket;
  const dyn_align_out* res;
  uint64_t n;
  const char* seqs;
  const uint64_t* seq_offsets;
  const char* const* readids;
  const char* const* signalids;
  const int64_t* sig_offsets;
  const uint64_t* signal_lengths;
  const uint32_t* stored_bases;  // bases of the basecall as stored (before the RNA pad), or nullptr
};

}  // namespace

struct dyn_csv_sink {
  dynzstd::Zstd z;
  int level = 3, threads = 4;
  FILE* out = nullptr;
  std::string errors_path, error;
  std::mutex m;
  std::condition_variable cv_items, cv_jobs, cv_write, cv_space;
  std::deque<Item> items;
  std::deque<std::shared_ptr<Job>> todo;             // jobs waiting for a compress thread
  std::map<uint64_t, std::shared_ptr<Job>> inflight;  // by index, until written
  uint64_t next_job = 0, next_write = 0;
  std::atomic<uint64_t> completed{0};
  std::condition_variable cv_done;  // completed has advanced, or the sink has failed
  uint64_t csv_bytes = 0, zst_bytes = 0;
  std::atomic<uint64_t> error_lines{0};
  bool closing = false, items_done = false, jobs_closed = false, failed = false;
  bool first_part = true, last_part = true;  // dyn_csv_sink_open_part: a part of a frame that several processes write
  std::thread t_sink, t_writer;
  std::vector<std::thread> t_comp;
  std::vector<std::shared_ptr<Blob>> spare;  // recycled row buffers (first-touch page faults cost more than formatting)
  std::vector<uint64_t> begin, end;
  std::vector<int64_t> last_index;

  void fail(const std::string& msg) {  // m held or single-threaded context
    if (!failed) error = msg;
    failed = true;
    cv_done.notify_all();
  }

  void add_job(std::shared_ptr<Blob> blob, size_t off, size_t len, bool last) {
    std::unique_lock<std::mutex> lk(m);
    cv_space.wait(lk, [&] { return inflight.size() < (size_t)(4 * threads) || failed; });
    auto j = std::make_shared<Job>();
    j->index = next_job++;
    j->blob = std::move(blob);
    j->off = off;
    j->len = len;
    j->last = last;
    inflight[j->index] = j;
    todo.push_back(j);
    cv_jobs.notify_one();
  }

  void append(std::shared_ptr<Blob> blob, size_t total) {
    csv_bytes += total;
    for (size_t off = 0; off < total; off += kJobBytes) add_job(blob, off, std::min(kJobBytes, total - off), false);
  }

  void compress_loop() {
    void* ctx = z.createCCtx();
    std::vector<char> local;
    for (;;) {
      std::shared_ptr<Job> j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_jobs.wait(lk, [&] { return !todo.empty() || jobs_closed; });
        if (todo.empty()) break;
        j = todo.front();
        todo.pop_front();
      }
      const double k0 = now_ms(), kc0 = thread_cpu_ms();
      const size_t cap = z.compressBound(j->len) + 64;
      j->out.resize(cap);
      us_resize.fetch_add((uint64_t)((now_ms() - k0) * 1e3));
      size_t rc = z.compressBegin(ctx, level);
      size_t pos = 0;
      if (!z.isError(rc) && (j->index != 0 || !first_part)) {
        rc = z.compressContinue(ctx, j->out.data(), cap, nullptr, 0);  // this context's frame header: dropped
        if (!z.isError(rc)) z.invalidateRepCodes(ctx);
      }
      const char* src = nullptr;
      if (j->len) {  // the job's rows, gathered out of the reads' slots
        const double p0 = now_ms();
        const Blob& b = *j->blob;
        local.resize(j->len);
        size_t r = (size_t)(std::upper_bound(b.prefix.begin(), b.prefix.end(), (uint64_t)j->off) - b.prefix.begin()) - 1;
        size_t at = j->off, filled = 0;
        while (filled < j->len) {
          const size_t in_read = at - b.prefix[r];
          const size_t take = std::min<size_t>(j->len - filled, (size_t)(b.end[r] - b.begin[r]) - in_read);
          std::memcpy(local.data() + filled, b.data.get() + b.begin[r] + in_read, take);
          filled += take;
          at += take;
          ++r;
        }
        us_gather.fetch_add((uint64_t)((now_ms() - p0) * 1e3));
        src = local.data();
      }
      if (!z.isError(rc))
        rc = j->last ? z.compressEnd(ctx, j->out.data(), cap, nullptr, 0) : z.compressContinue(ctx, j->out.data(), cap, src, j->len);
      if (!z.isError(rc)) pos = rc;
      us_compress.fetch_add((uint64_t)((now_ms() - k0) * 1e3));
      us_compress_cpu.fetch_add((uint64_t)((thread_cpu_ms() - kc0) * 1e3));
      {
        std::lock_guard<std::mutex> lk(m);
        if (z.isError(rc)) fail(std::string("zstd: ") + z.getErrorName(rc));
        j->out.resize(pos);
        j->blob.reset();
        j->done = true;
      }
      cv_write.notify_one();
    }
    z.freeCCtx(ctx);
  }

  void writer_loop() {
    for (;;) {
      std::shared_ptr<Job> j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_write.wait(lk, [&] {
          auto it = inflight.find(next_write);
          return (it != inflight.end() && it->second->done) || (jobs_closed && inflight.empty());
        });
        auto it = inflight.find(next_write);
        if (it == inflight.end()) break;
        j = it->second;
      }
      const double w0 = now_ms();
      const bool wrote = j->out.empty() || std::fwrite(j->out.data(), 1, j->out.size(), out) == j->out.size();
      us_write.fetch_add((uint64_t)((now_ms() - w0) * 1e3));
      if (!wrote) {
        std::lock_guard<std::mutex> lk(m);
        fail("write to the output file failed");
      }
      {
        std::lock_guard<std::mutex> lk(m);
        zst_bytes += j->out.size();
        inflight.erase(next_write);
        ++next_write;
      }
      cv_space.notify_all();
      cv_write.notify_one();
    }
  }

  void write_error_line(const std::string& line) {
    FILE* f = std::fopen(errors_path.c_str(), "a");  // created by the first error, like the reference's listener
    if (!f) {
      std::lock_guard<std::mutex> lk(m);
      fail("cannot open " + errors_path);
      return;
    }
    std::fwrite(line.data(), 1, line.size(), f);
    std::fputc('\n', f);
    std::fclose(f);
    ++error_lines;
  }

  // DYN_SINK_TRACE=1: where the sink thread's time goes (ms summed over the batches), printed at close
  double t_last_wait_done = 0;
  double t_wait = 0, t_format = 0, t_compact = 0, t_append = 0, t_errors = 0;
  std::atomic<uint64_t> us_compress{0}, us_write{0}, us_resize{0}, us_gather{0}, us_compress_cpu{0};
  static double thread_cpu_ms() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
  }
  static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  void consume(const Item& it) {
    const double c0 = now_ms();
    const int rc = dyn_batch_wait(it.ticket);
    const double c1 = now_ms();
    t_wait += c1 - c0;
    t_last_wait_done = c1;
    if (rc != DYN_OK) {
      std::lock_guard<std::mutex> lk(m);
      fail(std::string("batch failed: ") + dyn_aligner_last_error(it.a));
      return;
    }
    if (!it.n) return;
    last_index.resize(it.n);
    begin.resize(it.n);
    end.resize(it.n);
    for (uint64_t i = 0; i < it.n; ++i) last_index[i] = (int64_t)it.signal_lengths[i] + it.sig_offsets[i];
    const uint64_t bound = dyn_format_csv_bound(it.a, it.n, it.res, it.readids, it.signalids);
    std::shared_ptr<Blob> blob;
    {
      std::lock_guard<std::mutex> lk(m);
      for (size_t k = 0; k < spare.size(); ++k)
        if (spare[k].use_count() == 1 && spare[k]->cap >= bound) {
          blob = spare[k];
          break;
        }
    }
    if (!blob) {
      blob = std::make_shared<Blob>();
      blob->cap = std::max<size_t>(bound + bound / 8, 1);
      blob->data.reset(new char[blob->cap]);
      std::lock_guard<std::mutex> lk(m);
      if (spare.size() < 4) spare.push_back(blob);
    }
    const double c2 = now_ms();
    const int frc = dyn_format_csv(it.a, it.n, it.res, it.seqs, it.seq_offsets, it.readids, it.signalids, it.sig_offsets,
                                   last_index.data(), std::min(threads, 8), blob->data.get(), blob->cap, begin.data(), end.data());
    if (frc != DYN_OK) {
      std::lock_guard<std::mutex> lk(m);
      fail("dyn_format_csv failed");
      return;
    }
    const double c3 = now_ms();
    blob->begin.assign(begin.begin(), begin.begin() + it.n);
    blob->end.assign(end.begin(), end.begin() + it.n);
    blob->prefix.resize(it.n + 1);
    uint64_t total = 0;
    for (uint64_t i = 0; i < it.n; ++i) {
      blob->prefix[i] = total;
      total += end[i] - begin[i];
    }
    blob->prefix[it.n] = total;
    const double c4 = now_ms();
    if (total) append(blob, total);
    const double c5 = now_ms();
    t_format += c3 - c2;
    t_compact += c4 - c3;
    t_append += c5 - c4;
    t_errors += c2 - c1;
    for (uint64_t i = 0; i < it.n; ++i) {  // segment.py:172-176
      if (it.res->status[i] == DYN_READ_OK) continue;
      char msg[128];
      dyn_read_strerror(it.res->status[i], it.res->bad_char ? it.res->bad_char[i] : 0, msg, sizeof msg);
      // a read whose signal could not be read fails in the reference's WORKER, before the aligner (segment.py:178-187)
      const bool worker = it.res->status[i] == DYN_READ_BAD_SIGNAL;
      std::string line = std::string(worker ? "error: worker, " : "error: native, ") + msg +
                         (worker ? std::string() : "\tT: " + std::to_string(it.signal_lengths[i])) + "\tN: " +
                         // the reference's worker reports len(read) of the STORED basecall (segment.py:178-187), the aligner's
                         // failure the read in aligner orientation, RNA pad included (segment.py:172-176)
                         std::to_string(worker && it.stored_bases ? (uint64_t)it.stored_bases[i] : it.seq_offsets[i + 1] - it.seq_offsets[i]) +
                         "\tRid: " + it.readids[i] + "\tSid: " + it.signalids[i];
      write_error_line(line);
    }
  }

  void sink_loop() {
    if (first_part) {
      auto hdr = std::make_shared<Blob>();
      hdr->cap = sizeof kHeader;
      hdr->data.reset(new char[hdr->cap]);
      std::memcpy(hdr->data.get(), kHeader, sizeof kHeader - 1);
      hdr->begin = {0};
      hdr->end = {sizeof kHeader - 1};
      hdr->prefix = {0, sizeof kHeader - 1};
      append(hdr, sizeof kHeader - 1);
    }
    for (;;) {
      Item it;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_items.wait(lk, [&] { return !items.empty() || closing; });
        if (items.empty()) break;
        it = items.front();
        items.pop_front();
      }
      consume(it);
      {
        std::lock_guard<std::mutex> lk(m);  // (under the lock: dyn_csv_sink_wait must not miss the change between its test and its wait)
        completed.fetch_add(1);
      }
      cv_done.notify_all();
    }
    if (const char* e = std::getenv("DYN_SINK_TRACE"); e && *e == '1')
      std::fprintf(stderr, "[csv sink] batches %llu: waiting for the GPU %.1f ms, bound + buffer %.1f, format %.1f, prefix sums %.1f, handing jobs to the compressors %.1f; compress threads busy %.1f ms in sum, %.1f ms of CPU time (%d threads; of which sizing the output buffer %.1f, gathering the rows %.1f), writer in fwrite %.1f ms\n",
                   (unsigned long long)completed.load(), t_wait, t_errors, t_format, t_compact, t_append, us_compress.load() / 1e3, us_compress_cpu.load() / 1e3, threads, us_resize.load() / 1e3, us_gather.load() / 1e3, us_write.load() / 1e3);
    if (last_part) add_job(nullptr, 0, 0, true);  // the empty last block that closes the frame
    {
      std::lock_guard<std::mutex> lk(m);
      jobs_closed = true;
    }
    cv_jobs.notify_all();
    cv_write.notify_all();
  }
};

extern "C" {

int dyn_csv_sink_open(const char* csv_zst_path, const char* errors_path, int level, int threads, dyn_csv_sink** out,
                      char* err, uint64_t errcap) {
  return dyn_csv_sink_open_part(csv_zst_path, errors_path, level, threads, 1, 1, out, err, errcap);
}

int dyn_csv_sink_open_part(const char* csv_zst_path, const char* errors_path, int level, int threads, int first, int last,
                           dyn_csv_sink** out, char* err, uint64_t errcap) {
  auto put = [&](const std::string& s) {
    if (err && errcap) {
      std::snprintf(err, (size_t)errcap, "%s", s.c_str());
    }
  };
  if (!csv_zst_path || !errors_path || !out) return DYN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  std::unique_ptr<dyn_csv_sink> s(new dyn_csv_sink());
  std::string e;
  if (!s->z.load(e)) {
    put(e);
    return DYN_ERR_RUNTIME;
  }
  s->out = std::fopen(csv_zst_path, "wb");
  if (!s->out) {
    put(std::string("cannot open ") + csv_zst_path);
    return DYN_ERR_RUNTIME;
  }
  std::setvbuf(s->out, nullptr, _IOFBF, 1 << 20);
  s->errors_path = errors_path;
  s->level = level;
  s->threads = std::max(1, threads);
  s->first_part = first != 0;
  s->last_part = last != 0;
  dyn_csv_sink* p = s.release();
  for (int t = 0; t < p->threads; ++t) p->t_comp.emplace_back([p] { p->compress_loop(); });
  p->t_writer = std::thread([p] { p->writer_loop(); });
  p->t_sink = std::thread([p] { p->sink_loop(); });
  *out = p;
  return DYN_OK;
}

int dyn_csv_sink_submit(dyn_csv_sink* s, dyn_aligner* a, dyn_batch* ticket, const dyn_align_out* res, uint64_t n_reads,
                        const char* seqs, const uint64_t* seq_offsets, const char* const* readids,
                        const char* const* signalids, const int64_t* sig_offsets, const uint64_t* signal_lengths) {
  return dyn_csv_sink_submit_bases(s, a, ticket, res, n_reads, seqs, seq_offsets, readids, signalids, sig_offsets, signal_lengths, nullptr);
}

int dyn_csv_sink_submit_bases(dyn_csv_sink* s, dyn_aligner* a, dyn_batch* ticket, const dyn_align_out* res, uint64_t n_reads,
                              const char* seqs, const uint64_t* seq_offsets, const char* const* readids,
                              const char* const* signalids, const int64_t* sig_offsets, const uint64_t* signal_lengths,
                              const uint32_t* stored_bases) {
  if (!s || !a || !ticket || !res || !seq_offsets || (n_reads && (!seqs || !readids || !signalids || !sig_offsets || !signal_lengths)))
    return DYN_ERR_INVALID_ARGUMENT;
  {
    std::lock_guard<std::mutex> lk(s->m);
    if (s->closing) return DYN_ERR_INVALID_ARGUMENT;
    if (s->failed) return DYN_ERR_RUNTIME;  // a batch, the compressor or the file has failed: dyn_csv_sink_close has the message
    s->items.push_back(Item{a, ticket, res, n_reads, seqs, seq_offsets, readids, signalids, sig_offsets, signal_lengths, stored_bases});
  }
  s->cv_items.notify_one();
  return DYN_OK;
}

int dyn_csv_sink_error_line(dyn_csv_sink* s, const char* line) {
  if (!s || !line) return DYN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(s->m);  // serialises with nothing but other callers: the sink thread appends through its own handle
  FILE* f = std::fopen(s->errors_path.c_str(), "a");
  if (!f) return DYN_ERR_RUNTIME;
  std::fwrite(line, 1, std::strlen(line), f);
  std::fputc('\n', f);
  std::fclose(f);
  ++s->error_lines;
  return DYN_OK;
}

uint64_t dyn_csv_sink_completed(const dyn_csv_sink* s) { return s ? s->completed.load() : 0; }

uint64_t dyn_csv_sink_wait(dyn_csv_sink* s, uint64_t count, int timeout_ms) {
  if (!s) return 0;
  std::unique_lock<std::mutex> lk(s->m);
  auto ready = [&] { return s->completed.load() >= count || s->failed; };
  if (timeout_ms < 0)
    s->cv_done.wait(lk, ready);
  else
    s->cv_done.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready);
  return s->completed.load();
}

int dyn_csv_sink_failed(dyn_csv_sink* s) {
  if (!s) return 1;
  std::lock_guard<std::mutex> lk(s->m);
  return s->failed ? 1 : 0;
}

int dyn_csv_sink_close(dyn_csv_sink* s, uint64_t* csv_bytes, uint64_t* compressed_bytes, uint64_t* error_lines, char* err,
                       uint64_t errcap) {
  if (!s) return DYN_ERR_INVALID_ARGUMENT;
  {
    std::lock_guard<std::mutex> lk(s->m);
    s->closing = true;
  }
  s->cv_items.notify_all();
  const double q0 = dyn_csv_sink::now_ms();
  s->t_sink.join();
  const double q1 = dyn_csv_sink::now_ms();
  for (auto& t : s->t_comp) t.join();
  const double q2 = dyn_csv_sink::now_ms();
  s->t_writer.join();
  const double q3 = dyn_csv_sink::now_ms();
  if (std::fclose(s->out) != 0) s->fail("closing the output file failed");
  if (const char* e = std::getenv("DYN_SINK_TRACE"); e && *e == '1')
    std::fprintf(stderr, "[csv sink] close: the sink thread's last batches %.1f ms (of which behind the last ticket's results: %.1f), compressors %.1f, writer %.1f, fclose %.1f\n",
                 q1 - q0, q1 - s->t_last_wait_done, q2 - q1, q3 - q2, dyn_csv_sink::now_ms() - q3);
  if (csv_bytes) *csv_bytes = s->csv_bytes;
  if (compressed_bytes) *compressed_bytes = s->zst_bytes;
  if (error_lines) *error_lines = s->error_lines.load();
  const bool failed = s->failed;
  if (failed && err && errcap) std::snprintf(err, (size_t)errcap, "%s", s->error.c_str());
  // (releasing the row buffers -- ~0.5 GB of touched pages -- costs 55-60 ms; a detached thread for it moves the wait into the
  // caller's next release of memory: the address space is torn down under one lock. Packing the reads of a formatter thread
  // back to back instead of a worst-case slot per read changes nothing either: the pages are 4 KB, not huge. Measured, round 6)
  const bool trace_close = [] { const char* e = std::getenv("DYN_SINK_TRACE"); return e && *e == '1'; }();
  const double d0 = dyn_csv_sink::now_ms();
  size_t blob_bytes = 0;
  for (auto& b : s->spare) blob_bytes += b ? b->cap : 0;
  s->spare.clear();
  const double d1 = dyn_csv_sink::now_ms();
  delete s;
  if (trace_close)
    std::fprintf(stderr, "[csv sink] close: releasing the row buffers (%.0f MB reserved) %.1f ms, the rest of the sink %.1f ms\n", blob_bytes / 1e6, d1 - d0,
                 dyn_csv_sink::now_ms() - d1);
  return failed ? DYN_ERR_RUNTIME : DYN_OK;
}

}  // extern "C"
