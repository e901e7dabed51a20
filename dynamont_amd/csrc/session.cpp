// session.cpp -- host side of the RESIDENT read queue (engine.hpp: Session; nt_kernels.hpp: k_session; include/dynamont_mi.h:
// dyn_session_stats): choosing a session's geometry, opening it, publishing tickets into it while its kernel runs, recovering a
// ticket whose waves have left, closing, collecting statistics. Everything here runs under the handle's lock (a->mu) except where
// noted. Reference counterpart: none -- the reference keeps one Aligner per forked worker busy with one read at a time
// (src/dynamont/segmentation/segment.py:296-325); this is what keeps 1 024 waves busy across batches.
#include "engine_internal.hpp"
#include "dp_math_strict.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

using dynhost::PoreModel;
using dynk::ReadDesc;
using dynk::ReadState;
using dynk::SegRow;
using dynmath::Emis;
using namespace dyneng;

namespace dyneng {

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
  a->sess_idle_split[0] += st[6];
  a->sess_idle_split[1] += st[8];
  a->sess_idle_split[2] += st[7];
  a->sess_idle_split[3] += st[10];
  t.waves += ss.pend_waves[blk];
  // DYN_TRACE_HOST=1: where the waves' idle share of this session sat (shares of the summed wave lifetimes)
  static const bool trace = std::getenv("DYN_TRACE_HOST") != nullptr;
  if (trace && st[2]) {
    const double life = (double)st[2];
    std::fprintf(stderr, "[dyn] session: %.1f ms, %u tickets, waves busy %.4f; idle before a wave's first read %.4f (pages %.4f), between reads %.4f (pages %.4f), in its last turn %.4f (longest %.1f %% of the session, E[x^2]/E[x]^2 %.2f)\n",
                 ms, (unsigned)ss.pend_tickets[blk], st[0] / life, st[6] / life, st[8] / life, (double)(st[1] - st[6]) / life, (double)(st[5] - st[8]) / life, st[7] / life,
                 100.0 * st[10] * ss.pend_waves[blk] / life, st[7] ? (double)st[9] * 1048576.0 * ss.pend_waves[blk] / ((double)st[7] * (double)st[7]) : 0.0);
    std::fprintf(stderr, "[dyn] session: the read that ended last was number %llu of %llu in the queue; it had waited %.2f %% of the session for its pages\n",
                 (unsigned long long)(st[11] & 0xffffffu), (unsigned long long)ss.pend_reads[blk],
                 (st[11] >> 24) == (st[12] >> 24) ? 100.0 * (double)((st[12] & 0xffffffu) << 10) * ss.pend_waves[blk] / life : -1.0);
  }
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
      const char* lpt_env = std::getenv("DYN_SESSION_TAIL_LPT_READS");  // experiments: the length of that tail
      const size_t lpt_reads = std::min<size_t>(order.size(), lpt_env ? (size_t)std::atoll(lpt_env) : (size_t)ss.n_waves);
      if (!std::getenv("DYN_SESSION_NO_TAIL_LPT") && order.size() > ss.n_waves && lpt_reads > 1) {
        auto tail = order.end() - (ptrdiff_t)lpt_reads;
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

extern "C" int dyn_aligner_session_idle_split(dyn_aligner* a, uint64_t out4[4]) {
  if (!a || !out4) return DYN_ERR_INVALID_ARGUMENT;
  if (!a->host_only && a->s_session) {
    std::lock_guard<std::mutex> lk(a->mu);
    if (int rc = need_device(a)) return rc;
    if (int rc = session_quiesce(a)) return rc;
  }
  for (int k = 0; k < 4; ++k) out4[k] = a->sess_idle_split[k];
  return DYN_OK;
}

