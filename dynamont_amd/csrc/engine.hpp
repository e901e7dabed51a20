// engine.hpp -- host-side state shared by the C-ABI entry points (dynamont_mi.cpp) and the
// asynchronous batch pipeline (async_engine.cpp).
//
// The reference keeps its worker processes permanently fed (src/dynamont/segmentation/segment.py:
// 301-325: a multiprocessing pool over generate_jobs with a listener process draining results).
// The GPU counterpart of that is a three-stage pipeline per handle:
//   front thread : validateInput + sequenceToKmers of batch k+1, staging into pinned memory, H2D on
//                  the copy-in stream, kernel launches on the compute stream (no host sync)
//   GPU          : kernels of batch k (compute stream), D2H of batch k-1 (copy-out stream)
//   back thread  : unpack of batch k-1 into the caller's columns
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/dynamont_mi.h"
#include "nt_kernels.hpp"
#include "pore_model.hpp"

namespace dyneng {

// Recycles device and pinned-host allocations of finished batches: hipFree synchronises the whole
// device and hipHostMalloc of a 160 MB buffer costs tens of milliseconds, either of which would
// stall a pipeline that has other batches in flight. Sizes are rounded up to a granule so that a
// stream of batches of slightly different size keeps hitting the same buffers.
struct BufCache {
  std::mutex m;
  std::multimap<size_t, void*> dev, pin;
  static size_t round_up(size_t want);
  hipError_t take(bool pinned, size_t want, void** p, size_t* got);
  void give(bool pinned, void* p, size_t bytes);
  void purge();  // frees everything (device idle)
  // Handle destruction: what the cache holds is PARKED for the next handle on that device instead of freed (hipFree /
  // hipHostFree of a dozen batches' buffers: 0.2 s at the end of a run; allocating them again: as much at the start of the
  // next). Bounded; dyn_release_cached_memory() frees what is parked, DYN_NO_POOL_CACHE=1 switches parking off.
  void park(int device);
  int device = -1;  // set by the handle: take() looks at the device's parked buffers before it allocates
  // set by the handle: true while a resident session is open. purge() frees with hipFree, which waits for the whole device --
  // i.e. for resident waves that only leave when the handle (whose lock the caller of take() holds) closes their session. An
  // allocation that fails meanwhile is reported; alloc_batch_buffers quiesces the session, purges and tries again.
  const std::atomic<bool>* session_open = nullptr;
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  BufCache* cache = nullptr;  // nullptr: grow-only buffer owned by the handle (lattice pools)
  // grow-only. `headroom` > 1 over-allocates when the buffer has to grow: releasing and re-allocating
  // a 70 GB lattice pool costs 4-5 s on an MI355X (measured, tools/batch_latency.py), which a stream
  // of batches of slightly different size would otherwise pay every time a new maximum shows up.
  hipError_t ensure(size_t want, double headroom = 1.0);
  void release();
  template <class T> T* as() const { return static_cast<T*>(p); }
};

struct PinnedBuf {
  void* p = nullptr;
  size_t bytes = 0;
  BufCache* cache = nullptr;
  hipError_t ensure(size_t want);
  void release();
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// Small persistent worker pool for the host stages that are plain memory work (staging copies into
// pinned memory, rows -> columns unpack, k-mer coding).
class HelperPool {
 public:
  explicit HelperPool(int n_threads);
  ~HelperPool();
  // fn(task) for task in [0, n_tasks); returns when all are done. The calling thread takes part.
  void parallel_for(int n_tasks, const std::function<void(int)>& fn);
  int size() const { return (int)workers_.size() + 1; }

 private:
  void worker();
  std::vector<std::thread> workers_;
  std::mutex m_, call_m_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void(int)>* fn_ = nullptr;
  int next_ = 0, n_ = 0, active_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

struct Pipeline;

// ---- the resident read queue of a handle (nt_kernels.hpp: k_session) -------------------------------------------------
// A SESSION is one launch of resident waves on the handle's session stream (a CU-masked stream: it owns its hardware
// queue, so the small kernels and copies that feed the waves can never queue up behind them -- tools/ubench/resident_probe).
// The pipeline's front thread opens one when an align(calc=true) ticket arrives whose reads fit a static arena per wave,
// publishes that ticket and every later one that fits, and the session is closed when the pipeline has run dry (or a
// ticket of another kind must use the lattice pool). Two control blocks alternate, so that a new session can be set up
// while the last waves of the previous one are leaving.
constexpr uint32_t SESSION_RING = 1024;        // tickets per session (the ring is never reused within one)
constexpr uint64_t SESSION_MIN_READS = 512;    // a session is OPENED only for a ticket of at least this many reads
constexpr uint32_t SESSION_FLAGS = 4096;       // completion words in pinned host memory (a ring: far more than tickets in flight)
struct Session {
  bool open = false;
  bool mixed = false;           // the kernel variant that carries the certified sweeps
  int blk = 0;                  // control block / ticket ring in use
  uint32_t published = 0;       // tickets of the open session
  uint32_t next_base = 0;       // global read index of the next ticket's first read
  int log_r = 8;
  uint32_t arena_pages = 0;     // lattice pages per wave (layout 0); paged layouts: the most a read may need
  int layout = 0;               // 0: an arena per wave; 1: pages shared through the free list; 2: shared pages, posteriors in place
  uint32_t n_pages = 0;         // pages of the pool
  uint32_t n_waves = 0;
  uint64_t cells = 0, reads = 0, tickets = 0;   // of the open session
  bool pending[2] = {false, false};             // a session ran on this block and its statistics have not been collected
  uint64_t pend_cells[2] = {0, 0}, pend_reads[2] = {0, 0}, pend_tickets[2] = {0, 0};
  uint32_t pend_waves[2] = {0, 0};
  hipEvent_t ev_begin[2] = {nullptr, nullptr}, ev_end[2] = {nullptr, nullptr};
  uint64_t flag_seq = 0;        // next completion word
  // which session runs (or ran last) on a control block: a ticket remembers its session's number, and a block that has moved on
  // to a later session while the ticket is incomplete has LOST it (wait_resident -> session_recover)
  std::atomic<uint64_t> blk_gen[2] = {{0}, {0}};
  uint64_t gen = 0;
};

}  // namespace dyneng

struct dyn_aligner {
  dynhost::PoreModel model;
  int device = -1;
  bool host_only = false;
  int threads = 1;
  hipStream_t stream = nullptr;  // compute stream: every kernel of every batch of this handle, in submission order
  hipStream_t s_in = nullptr;    // H2D of the asynchronous pipeline
  hipStream_t s_out = nullptr;   // D2H of the asynchronous pipeline
  hipStream_t s_get = nullptr;   // the getters' copies (dyn_batch_fetch, _fetch_train, _signals): non-blocking, so that they do
                                 // not wait -- as a null-stream hipMemcpy would -- for a resident session that later tickets keep open
  dyneng::DevBuf d_model;
  dyneng::DevBuf d_sptab;  // softplus table (dp_math.hpp), staged into LDS by every DP workgroup
  uint64_t mem_budget = 0;
  int strict_mode = 1;  // dyn_aligner_set_strict: reads with a structural tie run bit for bit by default
  bool train_zcheck = false;  // dyn_aligner_set_train_zcheck
  bool ntk = false;     // created with mode "resquiggle" / "ntk"
  std::string last_error;
  // grow-only lattice workspace pool, reused across batches (only ever touched by work on `stream`,
  // whose order serialises the batches that share it)
  dyneng::DevBuf ws, lpe, bits;      // page pool of the read queue (nt_kernels.hpp, PagePool)
  dyneng::DevBuf free_list, ctl;     // its free-page stack and control words
  int n_cus = 256;                   // compute units: one persistent 4-wave workgroup each
  dyneng::PinnedBuf h_rows;  // staging of the synchronous dyn_batch_fetch
  dyneng::BufCache cache;
  // serialises GPU enqueue work on this handle between the caller's thread and the pipeline threads
  std::mutex mu;
  std::unique_ptr<dyneng::Pipeline> pipe;  // started by the first asynchronous submit
  // the resident read queue (guarded by mu)
  hipStream_t s_session = nullptr;   // CU-masked: a hardware queue of its own; nullptr = no sessions on this handle
  int sess_cus = 0;                  // CUs in its mask = workgroups of a session (n_cus minus what dyn_aligner_set_session_mode reserves)
  dyneng::DevBuf sess_ctl[2], sess_ring[2];
  dyneng::DevBuf sess_anchor;        // out_base of k_session: the address the ticket records' output offsets count from
  std::atomic<bool> sess_open_hint{false};  // mirrors sess.open for readers that do not hold mu
  std::atomic<bool> sess_enabled{false};    // mirrors s_session != nullptr likewise (the pipeline's front thread asks under ITS lock)
  std::mutex err_mu;                        // last_error is written by the caller's thread AND by the pipeline threads
  uint32_t* sess_flags = nullptr;    // [SESSION_FLAGS] pinned, coherent
  dyneng::PinnedBuf sess_hctl;       // D2H target of a control block
  dyneng::Session sess;
  dyn_session_stats sess_total{};    // closed and collected sessions
  uint64_t sess_idle_split[4] = {0, 0, 0, 0};  // wave-cycles: before a wave's first read, of which pages, last turn, longest last turn
  uint64_t sess_page_wait_cycles = 0;  // part of sess_total.wave_cycles_idle: waves of PAGED sessions getting their pages
  // the idle watchdog of the resident waves, seconds (DYN_SESSION_IDLE_S)
  double sess_idle_s = 20.0;
};

namespace dyneng {
// what a session is launched with (session_choose, session.cpp)
struct SessionGeom {
  bool ok = false;
  int layout = 0, log_r = 8;
  uint32_t arena_pages = 0;   // layout 0: per wave; paged: the most one read may need
  uint32_t n_pages = 0;       // of the pool
};
}  // namespace dyneng

struct HostRead {
  uint64_t S = 0, L = 0, kc = 0;
  uint64_t sig_off = 0, flat_off = 0 /* into kmers / per-column tables */, seg_off = 0;
  int32_t status = 0;
  char bad = 0;
  bool wide = false;  // half band above the register sweeps' 223: the read takes the generic kernel (wide_band.hip)
};

enum class DynJob { AlignZ, AlignFull, Train };

// P1/P2 on the device (dyn_batch_create_raw / dyn_batch_*_raw_async): the batch's samples arrive as RAW slices
struct RawSource {
  const void* raw = nullptr;   // concatenated [start:end) slices; scattered: a table of n_reads pointers, one per slice
  bool scattered = false;
  // vbz: the samples arrive as POD5 signal chunks still compressed (VBZ); read i = chunks [vbz_read_off[i], vbz_read_off[i+1])
  // decoded back to back, of which the slice [vbz_skip[i], vbz_skip[i] + length_i) is the read's signal
  bool vbz = false;
  const void* const* vbz_chunks = nullptr;
  const uint64_t* vbz_bytes = nullptr;
  const uint32_t* vbz_samples = nullptr;
  const uint64_t* vbz_read_off = nullptr;
  const uint64_t* vbz_skip = nullptr;
  int dtype = 0;               // 0 float32, 1 int16, 2 float64, 3 int16 ADC + per-read float32 calibration
  const float* cal_offset = nullptr;  // dtype 3: picoampere = (adc + cal_offset) * cal_scale, in float32
  const float* cal_scale = nullptr;
  const double* shift = nullptr;
  const double* scale = nullptr;
  int window = 3;
  double n_sigmas = 3.0;
  int compute_f32 = 0;
  size_t elem_size() const { return dtype == 0 ? 4 : (dtype == 1 || dtype == 3) ? 2 : 8; }
};

namespace dyneng {
struct BatchGroup;
}

struct dyn_batch {
  dyn_aligner* a = nullptr;
  uint64_t n = 0;
  std::vector<HostRead> reads;
  dyneng::PinnedBuf h_kmers;   // flat int32 codes, ok reads only
  uint64_t capacity = 0;       // sum of kc over ALL reads with L >= k (segment rows)
  uint64_t total_cols = 0;     // sum of kc over ok reads
  uint32_t max_T = 0, max_N = 0;
  dyneng::DevBuf d_sig, d_kmers, d_par, d_state, d_rows, d_segrow, d_medhi, d_medlo, d_descs;
  dyneng::DevBuf d_colw, d_cols1, d_cols2, d_trans, d_pooled, d_poolwork, d_pooltemp;
  dyneng::DevBuf d_pp, d_pathn;                // per-row path arrays (traceback -> k_median / k_final)
  dyneng::PinnedBuf h_descs, h_state, h_rows;  // h_state/h_rows: D2H targets of the asynchronous path
  dyneng::PinnedBuf h_stats;                   // wave-cycle statistics of the read-queue launch
  dyneng::PinnedBuf h_sig;                     // staging of pageable caller signals (asynchronous path)
  // raw asynchronous path: [offsets | shift | scale | raw samples] staged in h_sig; device-side scratch of the preprocessing
  dyneng::DevBuf d_norm, d_meta;
  dyneng::DevBuf d_wide;        // wide-band reads (wide_band.hip): queue head + one lattice arena per workgroup
  uint64_t n_wide = 0;          // reads of this batch that take the generic kernel
  dyneng::SessionGeom sess_geom;  // session_plan's decision for this ticket (ok = false: none taken)
  bool has_raw = false;
  RawSource raw_src;
  std::vector<hipEvent_t> events;              // before / after the read queue / after the per-segment kernels
  hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;  // ev_done: every kernel of the last job has finished
  uint32_t n_chunks = 0;                       // launches enqueued by the last job (0 or 1)
  std::vector<uint8_t> strict_flag;            // per read: took the certified sweeps in the last job
  dyn_timing timing{};
  bool aligned = false, trained = false;
  int last_calc = 0;
  // ---- asynchronous submission (async_engine.cpp) ----
  bool async = false;
  DynJob job = DynJob::AlignFull;
  const double* in_signals = nullptr;
  const uint64_t* in_sig_offsets = nullptr;
  const char* in_seqs = nullptr;
  const uint64_t* in_seq_offsets = nullptr;
  dyn_align_out* out_align = nullptr;
  dyn_train_out* out_train = nullptr;
  double* out_pooled = nullptr;
  // dyn_batch_device_pooled computes the device-resident pooled statistics on its first call: what it needs of the launch
  bool pooled_on_device = false;
  int pool_nr = 0;
  uint32_t pool_max_N = 0;
  int rc = DYN_OK;             // result of the pipeline stages
  std::string error;           // message for rc != DYN_OK (copied to the handle by dyn_batch_wait)
  bool done = false;
  // ---- merged launches (async_engine.cpp): tickets that waited together share ONE read-queue launch ----
  // A member ticket owns no device buffers of its own: its reads are reads [g_read0, g_read0 + n) and its segment rows
  // [g_seg0, g_seg0 + capacity) of the group's batch, which lives as long as any member does.
  std::shared_ptr<dyneng::BatchGroup> group;
  uint64_t g_read0 = 0, g_seg0 = 0;
  // merged batches (the group's own batch): no contiguous signal array -- one pointer per read (float64 samples)
  const double* const* in_sig_ptrs = nullptr;
  // ---- a ticket of the resident read queue (Session) ----
  bool in_session = false;
  uint32_t sess_reads = 0;             // reads published (the value its completion word reaches)
  uint32_t sess_waves = 0;
  int sess_blk = 0;                    // the session's control block
  uint64_t sess_gen = 0;               // ... and the session's number (Session::blk_gen)
  int sess_retries = 0;                // times the ticket was published again after its session had aborted
  volatile uint32_t* sess_flag = nullptr;
  dyneng::DevBuf d_tctl;               // the ticket's control block (reads done, wave-cycle statistics)
  uint32_t sess_max_N = 0;
  uint64_t sess_rows_total = 0;
  const int32_t* kmers() const { return h_kmers.as<int32_t>(); }
};

namespace dyneng {

// The batch a merged launch runs, and everything it borrows pointers into: the members' concatenated per-read metadata.
struct BatchGroup {
  dyn_batch* g = nullptr;               // the launch's batch (internal: never handed to a caller)
  std::vector<dyn_batch*> members;      // in submission order; cleared once they are complete
  std::vector<uint64_t> sig_offsets, seq_offsets, vbz_bytes, vbz_read_off, vbz_skip;
  std::vector<uint32_t> vbz_samples;
  std::vector<const void*> ptrs;        // one source pointer per read (raw slices, float64 signals) or per VBZ chunk
  std::vector<double> shift, scale;
  std::vector<float> cal_offset, cal_scale;
  std::string seqs;
  ~BatchGroup();
};

}  // namespace dyneng

namespace dyneng {

// ---- shared between the engine's translation units (dynamont_mi.cpp, launch.cpp, session.cpp, async_engine.cpp) ----
int need_device(dyn_aligner* a);
// CU-masked streams (every CU enabled: a hardware queue of their own) are parked per device and reused, never destroyed
// (dynamont_mi.cpp); nullptr when the runtime provides none. The session stream and a dyn_comm's stream come from here.
hipStream_t take_masked_stream(int device, int n_cus);
void park_masked_stream(int device, hipStream_t s);
// per-batch buffers come from / go back to the handle's BufCache
void attach_cache(dyn_batch* b);
// validateInput + sequenceToKmers of every read; fills b->reads / b->h_kmers / capacity / max_T / max_N
// (pinned: k-mer codes go to pinned memory for an asynchronous H2D; false = plain malloc, no HIP call)
int host_prepare(dyn_batch* b, const dynhost::PoreModel& m, bool pinned, uint64_t n_reads,
                 const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets, HelperPool* pool);
// allocate the per-batch device buffers (from the handle's cache)
int alloc_batch_buffers(dyn_batch* b, uint64_t total_sig);
// enqueue every kernel of `job` on the handle's compute stream without synchronising the host
int enqueue_job(dyn_batch* b, DynJob job);
// ---- resident read queue (session.cpp); all under a->mu ----
// could this ticket run in a session at all? (before host_prepare: kind and size only)
bool session_candidate(const dyn_batch* b);
// After host_prepare: will the ticket be published into a session (the open one, or one opened for it)? Closes an open
// session that cannot take it. *use = false: the classic launch (the caller quiesces the session first).
int session_plan(dyn_batch* b, bool* use);
// descriptors, per-read state, control block -> copy-in stream; the ticket's record published behind them (opens the
// session first if none is open). The ticket's inputs must already be enqueued on the copy-in stream.
int session_publish(dyn_batch* b);
// The ticket's session is gone (its waves left at the idle watchdog) and the ticket is incomplete: wait until that session's
// kernel has ended, then publish the ticket again -- into the session that is open now if it fits, else into a new one.
// All of its reads are redone (the results of those that had finished are the same). *republished = false: nothing was
// missing after all (the waves that were still busy finished the ticket before they left). Caller holds a->mu.
int session_recover(dyn_batch* b, bool* republished);
// the per-segment kernels and the statistics copy of a COMPLETED session ticket, on `s`
int session_finish_enqueue(dyn_batch* b, hipStream_t s);
int session_collect_timing(dyn_batch* b);
// no ticket will follow: the resident waves leave once what is published is done (returns at once)
int session_close(dyn_aligner* a);
// close, wait until the kernel has left, collect its statistics: the lattice pool is free for a classic launch afterwards
int session_quiesce(dyn_aligner* a);
// after the compute stream has passed the batch: read the event timings into b->timing
int collect_timing(dyn_batch* b);
// rows/state (host copies) -> the caller's columns; reads [read0, read0 + n) of b, whose segment rows start at seg0, land at
// index 0 of `out` (a member of a merged launch; the whole batch: read0 = seg0 = 0, n = b->n)
void unpack_align(const dyn_batch* b, const dynk::ReadState* st, const dynk::SegRow* rows, dyn_align_out* out,
                  HelperPool* pool, uint64_t read0 = 0, uint64_t n = ~0ull, uint64_t seg0 = 0);
// host finalisation of runTraining / trainTransition from host copies of the per-column sums
void finalise_train(const dyn_batch* b, const dynk::ReadState* st, const double* cw, const double* c1,
                    const double* c2, const double* tr, dyn_train_out* out, double* pooled3n);

struct Pipeline {
  explicit Pipeline(dyn_aligner* a);
  ~Pipeline();
  void submit(dyn_batch* b);
  int wait(dyn_batch* b);
  void drain();

  // one read-queue launch on its way through the stages: a ticket on its own, or a group of tickets merged into one batch
  struct Work {
    dyn_batch* b = nullptr;                  // the batch that runs (the ticket itself, or the group's batch)
    std::shared_ptr<BatchGroup> grp;         // set for merged launches
    int rc = DYN_OK;
  };

  dyn_aligner* a;
  HelperPool helpers;
  std::thread t_front, t_back;
  std::mutex m;
  std::condition_variable cv_front, cv_back, cv_done;
  std::deque<dyn_batch*> q_front;
  std::deque<Work> q_back;
  uint64_t in_flight = 0;        // tickets submitted and not yet done
  uint64_t peak_in_flight = 0;   // the most the caller has kept in flight: bounds the tickets one launch takes
  uint64_t launches_pending = 0; // launches handed to the GPU whose results have not been unpacked yet
  bool stop = false;

 private:
  void front_loop();
  void back_loop();
  int front_stage(dyn_batch* b);
  int back_stage(dyn_batch* b, const std::shared_ptr<BatchGroup>& grp);
  int wait_resident(dyn_batch* b);
  void close_idle_session();
  std::shared_ptr<BatchGroup> merge(const std::vector<dyn_batch*>& tickets);
};

}  // namespace dyneng
