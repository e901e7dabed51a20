// launch.cpp -- one CLASSIC launch per (merged) batch: the planner of page-starved queues (plan_queue), the spread order of
// paged sessions, enqueue_job (read descriptors, lattice pool, k_read_queue, the generic wide-band kernel, the per-segment
// kernels -- everything enqueued on the compute stream without a host synchronisation) and collect_timing. Reference counterpart:
// the body of NTAligner::align / train (NT_aligner_api.cpp:230-312, 567-639) for one read at a time.
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
void plan_queue(std::vector<uint32_t>& order, const std::vector<uint32_t>& need, const std::vector<uint64_t>& rows,
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
void spread_order(std::vector<uint32_t>& order, int tail_div) {
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

}  // namespace dyneng
