// rccl_comm.cpp -- the two exchanges of a multi-GPU job, below Python: an RCCL gather of per-read segment rows to one
// rank (BASELINE.json config 4) and a sum all-reduce of the pooled Baum-Welch statistics (config 5), straight from the
// device buffers of a batch, over xGMI.
//
// The reference has no counterpart: it scales by forking worker processes on one host
// (src/dynamont/segmentation/segment.py:296-325) whose results travel through a multiprocessing queue. Reads are
// independent (NTAligner::align keeps no cross-read state, NT_aligner_api.cpp:230-312), so one process per GPU aligns
// its shard and these calls are the only traffic between them. RCCL is bound with dlopen at the first dyn_comm_create:
// processes that never call it (every single-GPU use, dyn_multi_*) neither need nor load librccl.
//
// Pattern chosen for point-to-point xGMI (7 links per GPU, no switch): the gather is one ncclSend per non-root rank and
// one ncclRecv per peer on the root inside a group -- every peer's rows cross its OWN link to the root once, no ring,
// no padding to a common size (counts are exchanged first with an 8-byte all-gather).
#include "engine.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;  // optional: a peer that has gone shows up here first
  std::string error;

  bool load() {
    if (lib) return true;
    // a process that already carries an RCCL (PyTorch bundles one) must not get a second copy: try the global scope first
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      if (lib) break;
    }
    if (!lib)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
      }
    if (!lib) {
      error = "librccl.so.1 not found";
      return false;
    }
#define DYN_R(field, name)                                              \
  field = reinterpret_cast<decltype(field)>(dlsym(lib, name));          \
  if (!field) {                                                         \
    error = std::string("librccl lacks ") + name;                       \
    lib = nullptr;                                                      \
    return false;                                                       \
  }
    DYN_R(GetUniqueId, "ncclGetUniqueId");
    DYN_R(CommInitRank, "ncclCommInitRank");
    DYN_R(CommDestroy, "ncclCommDestroy");
    DYN_R(CommAbort, "ncclCommAbort");
    DYN_R(GetErrorString, "ncclGetErrorString");
    DYN_R(GroupStart, "ncclGroupStart");
    DYN_R(GroupEnd, "ncclGroupEnd");
    DYN_R(Send, "ncclSend");
    DYN_R(Recv, "ncclRecv");
    DYN_R(AllReduce, "ncclAllReduce");
    DYN_R(AllGather, "ncclAllGather");
#undef DYN_R
    CommGetAsyncError = reinterpret_cast<decltype(CommGetAsyncError)>(dlsym(lib, "ncclCommGetAsyncError"));
    return true;
  }
};

Rccl g_rccl;

void put_err(char* err, uint64_t cap, const std::string& s) {
  if (err && cap) std::snprintf(err, (size_t)cap, "%s", s.c_str());
}

}  // namespace

struct dyn_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1, device = 0;
  hipStream_t stream = nullptr;
  bool stream_masked = false;    // a CU-masked stream (a hardware queue of its own), parked per device when the handle goes
  double timeout_s = 300.0;      // DYN_COMM_TIMEOUT_S: how long one exchange may take before the communicator is aborted
  uint64_t* d_counts = nullptr;  // [n_ranks] rows per rank (all-gather target)
  void* d_recv = nullptr;        // root: gathered rows
  size_t recv_bytes = 0;
  std::vector<void*> retired;    // receive buffers that were outgrown: freed with the communicator (hipFree waits for the whole
                                 // device, i.e. for the END of a resident read queue -- never in the middle of a run). Growth
                                 // is x1.5, so what is retired adds up to at most TWICE the live buffer
  void* d_stage = nullptr;       // dyn_comm_gather_bytes / dyn_comm_allreduce_f64: the caller's HOST payload on the device
  size_t stage_bytes = 0;
  uint64_t gathered_bytes = 0;   // root: what the last dyn_comm_gather_bytes left in d_recv
  std::string last_error;
  // rows every rank announced in the last dyn_comm_gather_counts (the exchange dyn_comm_gather_rows then performs)
  std::vector<uint64_t> counts;
  bool counts_valid = false;
  bool aborted = false;
};

// A failure BETWEEN the collectives of one exchange (an allocation on the root, an RCCL call inside the group) would leave
// the peers blocked in their half of it: the communicator is aborted instead, so that their pending operations return
// with an error. The handle is unusable afterwards (every later call fails).
static bool comm_trace() {
  static const bool on = std::getenv("DYN_COMM_TRACE") != nullptr;
  return on;
}
#define C_TRACE(c, ...)                                             \
  do {                                                              \
    if (comm_trace()) {                                             \
      std::fprintf(stderr, "[dyn_comm rank %d] ", (c)->rank);       \
      std::fprintf(stderr, __VA_ARGS__);                            \
      std::fprintf(stderr, "\n");                                   \
      std::fflush(stderr);                                          \
    }                                                               \
  } while (0)

static void abort_comm(dyn_comm* c) {
  C_TRACE(c, "abort: ncclCommAbort ...");
  if (c->comm && !c->aborted) (void)g_rccl.CommAbort(c->comm);
  C_TRACE(c, "abort: done");
  c->comm = nullptr;
  c->aborted = true;
}

#define C_TRY(c, expr)                                                                     \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      (c)->last_error = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr; \
      return DYN_ERR_DEVICE;                                                               \
    }                                                                                      \
  } while (0)
#define N_TRY(c, expr)                                                                          \
  do {                                                                                          \
    ncclResult_t _r = (expr);                                                                   \
    if (_r != ncclSuccess) {                                                                    \
      (c)->last_error = std::string("RCCL error: ") + g_rccl.GetErrorString(_r) + " at " #expr; \
      return DYN_ERR_DEVICE;                                                                    \
    }                                                                                           \
  } while (0)

extern "C" {

int dyn_comm_unique_id(uint8_t* id_out128, char* err, uint64_t errcap) {
  if (!id_out128) return DYN_ERR_INVALID_ARGUMENT;
  if (!g_rccl.load()) {
    put_err(err, errcap, g_rccl.error);
    return DYN_ERR_RUNTIME;
  }
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) {
    put_err(err, errcap, std::string("RCCL error: ") + g_rccl.GetErrorString(r));
    return DYN_ERR_DEVICE;
  }
  static_assert(sizeof id == DYN_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id_out128, &id, sizeof id);
  return DYN_OK;
}

int dyn_comm_create(const uint8_t* id128, int rank, int n_ranks, int device, dyn_comm** out, char* err, uint64_t errcap) {
  if (!id128 || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return DYN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!g_rccl.load()) {
    put_err(err, errcap, g_rccl.error);
    return DYN_ERR_RUNTIME;
  }
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    put_err(err, errcap, std::string("HIP error: ") + hipGetErrorString(e) + " at hipSetDevice");
    return DYN_ERR_DEVICE;
  }
  dyn_comm* c = new dyn_comm();
  c->rank = rank;
  c->n_ranks = n_ranks;
  c->device = device;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, n_ranks, id, rank);
  if (r != ncclSuccess) {
    put_err(err, errcap, std::string("RCCL error: ") + g_rccl.GetErrorString(r) + " at ncclCommInitRank");
    delete c;
    return DYN_ERR_DEVICE;
  }
  // A stream with a hardware queue of its own where the runtime provides one (a CU-masked stream, every CU enabled): an RCCL
  // kernel waits for its peers while it runs, and whatever shares its hardware queue waits with it -- the copies and small
  // kernels that feed a resident read queue must not (tools/ubench/resident_probe.hip). Such streams are parked per device
  // and reused (dyneng::take_masked_stream): a communicator created and destroyed in a loop costs no hardware queue per turn.
  {
    int n_cus = 0;
    if (hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cus > 0) {
      c->stream = dyneng::take_masked_stream(device, n_cus);
      c->stream_masked = c->stream != nullptr;
    }
  }
  if (const char* t = std::getenv("DYN_COMM_TIMEOUT_S")) c->timeout_s = std::max(0.05, std::atof(t));
  if ((!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) ||
      hipMalloc(reinterpret_cast<void**>(&c->d_counts), sizeof(uint64_t) * (size_t)n_ranks) != hipSuccess) {
    put_err(err, errcap, "HIP error while setting up the communicator's stream");
    dyn_comm_destroy(c);
    return DYN_ERR_DEVICE;
  }
  *out = c;
  return DYN_OK;
}

void dyn_comm_destroy(dyn_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && !c->aborted) (void)g_rccl.CommDestroy(c->comm);
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->d_stage) (void)hipFree(c->d_stage);
  for (void* p : c->retired) (void)hipFree(p);
  // (a CU-masked stream is parked for the next communicator or handle of the process: destroying a second one in a process
  // did not return on ROCm 7.2, dynamont_mi.cpp)
  if (c->stream && c->stream_masked) dyneng::park_masked_stream(c->device, c->stream);
  else if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* dyn_comm_last_error(const dyn_comm* c) { return c ? c->last_error.c_str() : ""; }

// hipError / ncclResult inside an exchange that has begun: abort the communicator, then fail
#define C_TRY_X(c, expr)                                                                   \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      (c)->last_error = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr; \
      abort_comm(c);                                                                       \
      return DYN_ERR_DEVICE;                                                               \
    }                                                                                      \
  } while (0)
#define N_TRY_X(c, expr)                                                                        \
  do {                                                                                          \
    ncclResult_t _r = (expr);                                                                   \
    if (_r != ncclSuccess) {                                                                    \
      (c)->last_error = std::string("RCCL error: ") + g_rccl.GetErrorString(_r) + " at " #expr; \
      abort_comm(c);                                                                            \
      return DYN_ERR_DEVICE;                                                                    \
    }                                                                                           \
  } while (0)

// The end of an exchange: the stream has passed the collective. RCCL's kernels wait for their peers ON THE DEVICE, so a peer
// that has gone (a crashed rank, a job that was cancelled on one GPU) would leave hipStreamSynchronize waiting for ever: the
// wait polls instead, asks RCCL for asynchronous errors, and after DYN_COMM_TIMEOUT_S (300 s) aborts the communicator --
// ncclCommAbort ends the local kernels -- and fails the call. The handle is unusable afterwards, like after any abort.
static int wait_exchange(dyn_comm* c, const char* what) {
  using clock = std::chrono::steady_clock;
  const auto t0 = clock::now();
  int spins = 0;
  for (;;) {
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipSuccess) return DYN_OK;
    if (q != hipErrorNotReady) {
      c->last_error = std::string("HIP error: ") + hipGetErrorString(q) + " while waiting for " + what;
      abort_comm(c);
      return DYN_ERR_DEVICE;
    }
    const double waited = std::chrono::duration<double>(clock::now() - t0).count();
    if ((++spins & 63) == 0 && g_rccl.CommGetAsyncError && c->comm) {
      ncclResult_t ar = ncclSuccess;
      if (g_rccl.CommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
        C_TRACE(c, "%s: asynchronous error %d after %.3f s", what, (int)ar, waited);
        c->last_error = std::string("RCCL error: ") + g_rccl.GetErrorString(ar) + " (asynchronous) during " + what +
                        ": the communicator was aborted";
        abort_comm(c);
        (void)hipStreamSynchronize(c->stream);
        return DYN_ERR_DEVICE;
      }
    }
    if (waited > c->timeout_s) {
      C_TRACE(c, "%s: timeout after %.1f s", what, waited);
      c->last_error = std::string(what) + " did not complete within " + std::to_string(c->timeout_s) +
                      " s (DYN_COMM_TIMEOUT_S): a peer has gone or never arrived; the communicator was aborted";
      abort_comm(c);
      (void)hipStreamSynchronize(c->stream);  // the aborted kernels leave
      return DYN_ERR_DEVICE;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.002 ? 20 : waited < 0.1 ? 100 : 1000));
  }
}

// rows this rank contributes: 0 when its batch failed (the error is kept; the rank still takes part in the collectives)
static int local_rows(dyn_comm* c, dyn_batch* b, void** d_rows, uint64_t* n_rows) {
  *d_rows = nullptr;
  *n_rows = 0;
  int rc = dyn_batch_wait(b);  // asynchronous tickets: the rows exist once the batch is complete
  if (rc != DYN_OK) {
    c->last_error = std::string("this rank's batch failed: ") + dyn_aligner_last_error(b->a);
    return rc;
  }
  rc = dyn_batch_device_results(b, d_rows, n_rows, nullptr);
  if (rc != DYN_OK) {
    c->last_error = "dyn_comm_gather: the batch has not been aligned";
    *d_rows = nullptr;
    *n_rows = 0;
  }
  return rc;
}

// ---- the variable-length gather both front ends share (segment rows of a batch; a rank's bytes) -----------------------
// counts: every rank's number of UNITS (8-byte all-gather over n_ranks entries) -> c->counts
static int exchange_counts(dyn_comm* c, uint64_t mine) {
  C_TRY_X(c, hipSetDevice(c->device));
  C_TRY_X(c, hipMemcpyAsync(c->d_counts + c->rank, &mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
  N_TRY_X(c, g_rccl.AllGather(c->d_counts + c->rank, c->d_counts, sizeof(uint64_t), ncclUint8, c->comm, c->stream));
  c->counts.assign((size_t)c->n_ranks, 0);
  // (the copies out come AFTER the bounded wait: a copy into pageable memory blocks its caller until the stream gets there,
  // i.e. for as long as the collective in front of it waits for a peer)
  if (int rc = wait_exchange(c, "the all-gather of counts")) return rc;
  C_TRY_X(c, hipMemcpyAsync(c->counts.data(), c->d_counts, sizeof(uint64_t) * c->counts.size(), hipMemcpyDeviceToHost, c->stream));
  C_TRY_X(c, hipStreamSynchronize(c->stream));
  return DYN_OK;
}

// payload: c->counts[r] units of `unit` bytes from every rank r -> root's d_recv, back to back in rank order. One ncclSend per
// non-root rank, one ncclRecv per peer on the root, in one group: each peer's payload crosses its own link to the root once.
static int exchange_payload(dyn_comm* c, const void* d_send, size_t unit, int root, const char* what) {
  const std::vector<uint64_t>& counts = c->counts;
  C_TRY_X(c, hipSetDevice(c->device));
  uint64_t total = 0;
  for (uint64_t n : counts) total += n;
  if (c->rank == root && c->recv_bytes < total * unit) {
    if (c->d_recv) c->retired.push_back(c->d_recv);  // (growth x1.5: the retired buffers add up to at most twice the live one)
    c->d_recv = nullptr;
    c->recv_bytes = 0;
    const size_t want = std::max<size_t>(total * unit + total * unit / 2, 16);
    C_TRY_X(c, hipMalloc(&c->d_recv, want));
    c->recv_bytes = want;
  }
  if (total == 0) return DYN_OK;  // (every rank knows: nobody posts anything)
  const uint64_t mine = counts[(size_t)c->rank];
  C_TRACE(c, "%s: ncclGroupEnd ...", what);
  N_TRY_X(c, g_rccl.GroupStart());
  if (c->rank == root) {
    uint64_t off = 0;
    for (int r = 0; r < c->n_ranks; ++r) {
      char* dst = static_cast<char*>(c->d_recv) + off * unit;
      if (r == root) {
        if (counts[(size_t)r]) C_TRY_X(c, hipMemcpyAsync(dst, d_send, counts[(size_t)r] * unit, hipMemcpyDeviceToDevice, c->stream));
      } else if (counts[(size_t)r]) {
        N_TRY_X(c, g_rccl.Recv(dst, counts[(size_t)r] * unit, ncclUint8, r, c->comm, c->stream));
      }
      off += counts[(size_t)r];
    }
  } else if (mine) {
    N_TRY_X(c, g_rccl.Send(d_send, mine * unit, ncclUint8, root, c->comm, c->stream));
  }
  N_TRY_X(c, g_rccl.GroupEnd());
  C_TRACE(c, "%s: group enqueued, waiting", what);
  return wait_exchange(c, what);
}

int dyn_comm_gather_counts(dyn_comm* c, dyn_batch* b, uint64_t* counts_out) {
  if (!c || !b) return DYN_ERR_INVALID_ARGUMENT;
  if (c->aborted) {
    c->last_error = "the communicator was aborted by an earlier failure";
    return DYN_ERR_DEVICE;
  }
  void* d_rows = nullptr;
  uint64_t mine = 0;
  const int local_rc = local_rows(c, b, &d_rows, &mine);  // a failed batch announces 0 rows: the peers must not hang
  if (int rc = exchange_counts(c, mine)) return rc;
  c->counts_valid = true;
  if (counts_out) std::memcpy(counts_out, c->counts.data(), sizeof(uint64_t) * c->counts.size());
  return local_rc;
}

int dyn_comm_gather_rows(dyn_comm* c, dyn_batch* b, int root, dyn_segment_row* rows_out, uint64_t rows_cap,
                         uint64_t* counts_out) {
  if (!c || !b || root < 0 || root >= c->n_ranks) return DYN_ERR_INVALID_ARGUMENT;
  if (c->aborted) {
    c->last_error = "the communicator was aborted by an earlier failure";
    return DYN_ERR_DEVICE;
  }
  int local_rc = DYN_OK;
  if (!c->counts_valid) {  // one-call form: the count exchange first (collective on every rank alike)
    local_rc = dyn_comm_gather_counts(c, b, nullptr);
    if (c->aborted) return DYN_ERR_DEVICE;
  }
  c->counts_valid = false;  // the counts are consumed by this exchange
  const std::vector<uint64_t>& counts = c->counts;
  if (counts_out) std::memcpy(counts_out, counts.data(), sizeof(uint64_t) * counts.size());
  void* d_rows = nullptr;
  uint64_t mine = 0;
  if (counts[(size_t)c->rank]) (void)local_rows(c, b, &d_rows, &mine);  // (complete already: no waiting here)
  uint64_t total = 0;
  for (uint64_t n : counts) total += n;
  constexpr size_t ROW = sizeof(dyn_segment_row);
  const bool too_small = c->rank == root && rows_out && rows_cap < total;
  // the batch's buffers may be released after this call
  if (int rc = exchange_payload(c, d_rows, ROW, root, "the gather of segment rows")) return rc;
  if (c->rank == root && rows_out && !too_small && total) {
    C_TRY_X(c, hipMemcpyAsync(rows_out, c->d_recv, total * ROW, hipMemcpyDeviceToHost, c->stream));
    C_TRY_X(c, hipStreamSynchronize(c->stream));
  }
  if (too_small) {  // the exchange itself is complete on every rank: nothing hangs, the root may call again with room
    c->last_error = "dyn_comm_gather_rows: rows_cap is smaller than the sum of all ranks' rows (dyn_comm_gather_counts tells it)";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  return local_rc;
}

// the caller's host payload on the device (grow-only; hipFree waits for the whole device: never in the middle of a run)
static int stage_payload(dyn_comm* c, const void* host, uint64_t n_bytes) {
  C_TRY(c, hipSetDevice(c->device));
  if (c->stage_bytes < n_bytes) {
    if (c->d_stage) c->retired.push_back(c->d_stage);
    c->d_stage = nullptr;
    c->stage_bytes = 0;
    const size_t want = std::max<size_t>(n_bytes + n_bytes / 2, 4096);
    C_TRY(c, hipMalloc(&c->d_stage, want));
    c->stage_bytes = want;
  }
  if (n_bytes) {
    C_TRY(c, hipMemcpyAsync(c->d_stage, host, n_bytes, hipMemcpyHostToDevice, c->stream));
    C_TRY(c, hipStreamSynchronize(c->stream));
  }
  return DYN_OK;
}

int dyn_comm_gather_bytes(dyn_comm* c, const void* bytes, uint64_t n_bytes, int root, uint64_t* counts_out) {
  if (!c || root < 0 || root >= c->n_ranks || (n_bytes && !bytes)) return DYN_ERR_INVALID_ARGUMENT;
  if (c->aborted) {
    c->last_error = "the communicator was aborted by an earlier failure";
    return DYN_ERR_DEVICE;
  }
  c->counts_valid = false;
  c->gathered_bytes = 0;
  // a rank that cannot stage its payload still takes part (with 0 bytes), then reports: its peers never block on it
  const int local_rc = stage_payload(c, bytes, n_bytes);
  if (int rc = exchange_counts(c, local_rc == DYN_OK ? n_bytes : 0)) return rc;
  if (counts_out) std::memcpy(counts_out, c->counts.data(), sizeof(uint64_t) * c->counts.size());
  if (int rc = exchange_payload(c, c->d_stage, 1, root, "the gather of bytes")) return rc;
  if (c->rank == root)
    for (uint64_t n : c->counts) c->gathered_bytes += n;
  return local_rc;
}

int dyn_comm_gathered_bytes(dyn_comm* c, void* out, uint64_t out_cap) {
  if (!c || (!out && c->gathered_bytes)) return DYN_ERR_INVALID_ARGUMENT;
  if (out_cap < c->gathered_bytes) {
    c->last_error = "dyn_comm_gathered_bytes: out_cap is smaller than the sum of the counts of the last dyn_comm_gather_bytes";
    return DYN_ERR_INVALID_ARGUMENT;
  }
  if (c->gathered_bytes) {
    C_TRY(c, hipSetDevice(c->device));
    C_TRY(c, hipMemcpyAsync(out, c->d_recv, c->gathered_bytes, hipMemcpyDeviceToHost, c->stream));
    C_TRY(c, hipStreamSynchronize(c->stream));
  }
  return DYN_OK;
}

int dyn_comm_allreduce_f64(dyn_comm* c, double* inout, uint64_t n, int op) {
  if (!c || (n && !inout) || op < 0 || op > 1) return DYN_ERR_INVALID_ARGUMENT;
  if (c->aborted) {
    c->last_error = "the communicator was aborted by an earlier failure";
    return DYN_ERR_DEVICE;
  }
  if (!n) return DYN_OK;
  if (int rc = stage_payload(c, inout, n * sizeof(double))) {
    abort_comm(c);  // the peers are (or will be) inside ncclAllReduce: fail them fast
    return rc;
  }
  N_TRY_X(c, g_rccl.AllReduce(c->d_stage, c->d_stage, n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, c->stream));
  if (int rc = wait_exchange(c, "the all-reduce")) return rc;
  C_TRY_X(c, hipMemcpyAsync(inout, c->d_stage, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  C_TRY_X(c, hipStreamSynchronize(c->stream));
  return DYN_OK;
}

int dyn_comm_allreduce_pooled(dyn_comm* c, dyn_batch* b, double* pooled3n) {
  if (!c || !b) return DYN_ERR_INVALID_ARGUMENT;
  if (c->aborted) {
    c->last_error = "the communicator was aborted by an earlier failure";
    return DYN_ERR_DEVICE;
  }
  void* d_pooled = nullptr;
  uint64_t count = 0;
  int rc = dyn_batch_wait(b);
  if (rc != DYN_OK) c->last_error = std::string("this rank's batch failed: ") + dyn_aligner_last_error(b->a);
  if (rc == DYN_OK) {
    rc = dyn_batch_device_pooled(b, &d_pooled, &count);
    if (rc != DYN_OK) c->last_error = "dyn_comm_allreduce_pooled: the batch has not been trained";
  }
  if (rc != DYN_OK) {
    // the peers are (or will be) inside ncclAllReduce with a buffer this rank cannot match: fail them fast
    abort_comm(c);
    return rc;
  }
  C_TRY_X(c, hipSetDevice(c->device));
  // linear-domain sums (w, s1, s2)[numKmers]: a plain sum all-reduce is exact up to fp64 association
  N_TRY_X(c, g_rccl.AllReduce(d_pooled, d_pooled, count, ncclDouble, ncclSum, c->comm, c->stream));
  if (int wrc = wait_exchange(c, "the all-reduce of pooled statistics")) return wrc;
  if (pooled3n) {
    C_TRY_X(c, hipMemcpyAsync(pooled3n, d_pooled, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    C_TRY_X(c, hipStreamSynchronize(c->stream));
  }
  return DYN_OK;
}

}  // extern "C"
