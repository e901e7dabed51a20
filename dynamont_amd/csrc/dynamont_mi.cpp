// dynamont_mi.cpp -- C-ABI entry points (include/dynamont_mi.h): handles, validation + k-mer coding per read, the staged and
// one-shot batch calls, result marshalling. The launches themselves: launch.cpp (one per batch), session.cpp (the resident read
// queue); buffers: buffers.cpp; the asynchronous pipeline: async_engine.cpp.
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
#include "engine_internal.hpp"
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


namespace {

void copy_msg(char* buf, uint64_t cap, const std::string& s) {
  if (!buf || !cap) return;
  const size_t n = std::min<size_t>(cap - 1, s.size());
  std::memcpy(buf, s.data(), n);
  buf[n] = 0;
}

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
