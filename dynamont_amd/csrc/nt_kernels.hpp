// nt_kernels.hpp -- device-side data layout and launch interface of the NT hot path.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

#include "dp_math.hpp"

namespace dynk {

using dynmath::Emis;

// One 64-lane wavefront owns one read. A lattice row has at most 2*bw+1 in-band cells
// (NT_aligner_api.cpp:90-108); they live in P = 64*CPL "band slots", slot = n mod P, so a
// lattice column keeps its lane and register for as long as it is in the band and the band
// shift of the reference (mShift/eShift, NT_aligner_api.cpp:131-138) never moves data.
// P must exceed the band width by one so the out-of-band neighbour of an edge cell is a real
// (always -inf) slot and never aliases an in-band cell.
constexpr int CPL = 7;                       // cells per lane
constexpr int P = 64 * CPL;                  // 448 slots per row
constexpr int MAX_HALF_BAND = (P - 2) / 2;   // 2*bw+1 <= P-1  ->  bw <= 223 (band <= 447)

// The lattice of a read lives in PAGES of 2^log_rows rows taken from a pool that all reads of a
// launch share: a read holds ceil((T+1) / rows_per_page) pages from the moment a wave starts its
// backward sweep until its traceback is done. A wave keeps the page numbers of its current read in
// LDS (PT_MAX entries), so rows_per_page is chosen per launch such that the longest read fits.
constexpr int PT_MAX = 512;
// Waves per SIMD of the read queue (an experiment switch: 2 = eight waves per CU, a 2-deep row ring)
#ifndef DYN_WAVES_PER_SIMD
#define DYN_WAVES_PER_SIMD 1
#endif
constexpr int WAVES_PER_CU = 4 * DYN_WAVES_PER_SIMD;
constexpr uint32_t NO_PAGE = 0xffffffffu;

struct ReadDesc {
  uint64_t sig_off;   // first sample in the signal pool (doubles)
  uint64_t par_off;   // first entry in the per-column emission table; entry n-1 <-> column n
  uint64_t path_off;  // per-row path arrays [T]
  uint64_t seg_off;   // first output row of this read
  uint32_t T;         // signal length + 1   (NT_aligner_api.cpp:241)
  uint32_t N;         // k-mer count + 1     (NT_aligner_api.cpp:242)
  uint32_t bw;        // min(band/2, N/2)    (NT_aligner_api.cpp:243)
  uint32_t read;      // index into the per-read state arrays
  double ratio;       // double(N)/double(T) (NT_aligner_api.cpp:96)
  uint32_t first_page;  // pages first_page .. first_page+n_pages-1 were reserved by the host (first round of a
                        // launch), or NO_PAGE: the wave takes n_pages from the pool's free list
  uint32_t n_pages;     // lattice rows 0..T in pages; 0 for jobs without a stored lattice
  uint32_t flags;       // READ_STRICT / READ_STRICT_START: this read takes the bit-for-bit sweeps (dp_math_strict.hpp)
  uint32_t strict_rows; // READ_STRICT_START: forward rows 1 .. strict_rows in the strict arithmetic
};
constexpr uint32_t READ_STRICT = 1u;        // every row of both sweeps
constexpr uint32_t READ_STRICT_START = 2u;  // the backward sweep and the first strict_rows rows of the forward sweep

struct ReadState {
  double Zb;          // backwardE(0,0)          (NT_aligner_api.cpp:286)
  double Zf;          // forwardE(T-1, mid(T-1)) (NT_aligner_api.cpp:285)
  int32_t status;     // dyn_read_status
  uint32_t n_segments;
};

struct SegRow {
  uint32_t signal_pos;
  uint32_t sequence_pos;
  double probability;
};

struct TraceBuffers {
  double* pp;         // [path] posterior of the path cell of each row
  uint32_t* pathn;    // [path] lattice column of the path cell (bit 31: state M)
  uint32_t* segrow;   // [segments] row of each segment's M cell
  double* med_hi;     // [segments] upper middle order statistic
  double* med_lo;     // [segments] lower middle order statistic (even counts)
};

struct TrainBuffers {
  double* col_w;      // [par] expected count per lattice column
  double* col_s1;     // [par] sum gamma*x
  double* col_s2;     // [par] sum gamma*x^2
  double* trans;      // [2*reads] expected E->M and E->E transition counts (linear domain)
};

// Page pool + read queue of one launch (device pointers).
struct PagePool {
  double* ws;           // [n_pages][rows_per_page][P]   backward-E (8 B per slot); in place: (float LPM, float LPE)
  float* lpe;           // [n_pages][rows_per_page][P]   float LPE per slot (separate layout) or nullptr
  uint64_t* bits;       // [n_pages][rows_per_page][CPL] decision ballots
  uint32_t* free_list;  // stack of free page numbers
  // ctl (32-bit words): [0] lock of the free-page stack, [1] queue head (next read), [2] free pages on the
  // stack, [3] abort (a wave gave up waiting: the queue drains, the host reports the launch as failed),
  // [4] reads whose pages are in place, [5] waves waiting for pages.
  // ctl + QUEUE_STATS (64-bit words): wave-cycles spent in backward, forward, traceback, waiting for pages,
  // lifetime (all summed over the waves of the launch), longest lifetime; then the strict reads apart: their backward and
  // forward wave-cycles, registers recomputed by the certified logPlus' fallback, rows run in the certified arithmetic.
  uint32_t* ctl;
  int log_rows;         // rows per page = 1 << log_rows
  uint32_t n_pages;
  uint32_t reserve_after;  // waits (x ~30 us) after which a wave RESERVES its request for pages; 0 = never (one launch per batch:
                           // the queue is finite and its order planned). Paged sessions: 64
};
constexpr int QUEUE_CTL_WORDS = 32;   // 32-bit words reserved for ctl (stats start at word 8, 8-byte aligned)
constexpr int QUEUE_STATS = 8;        // first stats word (as uint32 index)
constexpr int QUEUE_N_STATS = 10;

enum QueueJob { JOB_Z = 0, JOB_ALIGN = 1, JOB_ALIGN_INPLACE = 2, JOB_TRAIN = 3, JOB_TRAIN_ZCHECK = 4 };

struct QueueArgs {
  const ReadDesc* descs;   // in processing order
  int n_reads;
  int n_static;            // reads [0, n_static) are pre-assigned: wave slot s starts with read s (no queue access)
  const double* sig;
  const Emis* par;
  PagePool pool;
  ReadState* st;
  TraceBuffers tb;
  TrainBuffers tr;
  double m1, e2;          // log transition probabilities (NT_aligner_api.cpp:84-86)
  const dynmath::SoftplusNode* sp_tab;
  int z_fail_status;
};

// ---- the RESIDENT read queue (round 5) ---------------------------------------------------------------------------
// k_read_queue drains ONE batch per launch: its waves finish up to one read apart, and the next launch -- queued on the same
// stream -- waits for the last of them (wave occupancy 0.89-0.91 at 2-3 batches per launch). k_session keeps the waves
// RESIDENT: the host publishes batch after batch ("tickets") into a ring while the kernel runs, a wave that has finished a
// read takes the next one of whatever ticket is next, and the kernel leaves only when the host closes the session (nothing
// left in its pipeline). Tickets complete one by one (a counter per ticket; the wave whose read completes it raises a word
// in pinned host memory), their per-segment kernels and copies run beside the resident waves on other streams.
//  * every wave owns a STATIC arena of `arena_pages` lattice pages (page numbers slot * arena_pages + k): no free list, no
//    lock; the host only opens a session when the pool holds an arena for every wave (separate LPE layout);
//  * inputs of a ticket (descriptors, samples, per-column parameters) were written by H2D copies and small kernels of other
//    streams WHILE this kernel runs, into buffers that earlier tickets used: a wave starts every read with s_dcache_inv and
//    an agent-scope acquire (this CU's L1), and it reads them through ONE `const __restrict__` kernel parameter (in_base +
//    a byte offset from the ticket record), so that wave-uniform loads stay scalar loads (see k_read_queue);
//  * what a wave wrote for a read (state, path arrays) is released (agent scope) before the ticket's counter moves;
//  * every wait is bounded: a wave that has found nothing to do for `idle_limit_ticks` raises the abort word and leaves.
// Validated in isolation by tools/ubench/resident_probe.hip (co-scheduling of the small kernels beside a resident kernel of
// this footprint, freshness of rewritten buffers, and that the resident kernel needs a hardware queue of its own).
struct SessionTicket {      // 128 bytes in device memory, written once per ticket by k_session_publish
  // Byte offsets, not pointers: an address the kernel forms from one of its own pointer PARAMETERS is known to be global
  // memory (global_load / global_store; a pointer loaded from memory is a generic one: flat_ instructions), and loads
  // through the `const __restrict__` in_base stay scalar loads where the address is wave-uniform.
  int64_t descs_off;        // from in_base: ReadDesc[n_reads] in processing order,
  int64_t sig_off;          //   the ticket's signal pool (ReadDesc::sig_off counts doubles from here),
  int64_t par_off;          //   its per-column emission table (ReadDesc::par_off counts entries from here)
  int64_t st_off;           // from out_base: ReadState[], then the five TraceBuffers arrays
  int64_t pp_off, pathn_off, segrow_off, medhi_off, medlo_off;
  int64_t tctl_off;         // the ticket's control block: [0] reads finished; 64-bit statistics from word SESSION_TSTATS on:
                            //   wave-cycles (shader clock) in backward / forward / traceback, [3] the reads' durations in 10 ns
                            //   ticks (s_memrealtime), [6..9] the certified sweeps apart (as QUEUE_STATS)
  int64_t flag_off;         // a word of pinned host memory: set to n_reads by the wave whose read completes the ticket
  uint32_t n_reads;
  uint32_t base;            // the ticket's reads are global indices [base, base + n_reads)
  int32_t z_fail_status;
  uint32_t pad_[7];
};
static_assert(sizeof(SessionTicket) == 128, "one cache line pair per record");
constexpr int SESSION_TSTATS = 2;        // first statistics word of a ticket's control block (as uint32 index, 8-byte aligned)
constexpr int SESSION_TCTL_WORDS = 32;
// session control words (device memory, 32-bit): next global read index, tickets published, closed, abort; 64-bit statistics
// from word SESSION_STATS on: wave-cycles busy (claim to release), idle, lifetime, longest lifetime, reads done, [5] the part
// of idle spent getting pages (paged sessions), [6] idle before a wave's first read (part of idle), [7] a wave's last turn (no read
// left to claim; NOT part of idle: lifetime - busy - idle), [8] the pages part of [6], [9] sum of ([7] >> 10)^2, [10] max of [7], [11] / [12] the read that ended last (k_session)
constexpr int S_HEAD = 0, S_TAIL = 1, S_CLOSED = 2, S_ABORT = 3, SESSION_STATS = 8, SESSION_CTL_WORDS = 40;

struct SessionArgs {
  const SessionTicket* ring;   // [ring_size]; slot i holds ticket i of the session (never reused within one)
  uint32_t ring_size;
  uint32_t arena_pages;        // pages per wave (layout 0)
  uint32_t give_always;        // paged layouts, experiments (DYN_SESSION_GIVE_ALWAYS): surplus pages go back at every read
  uint32_t* ctl;
  PagePool pool;               // ws, lpe, bits, log_rows; layouts 1 / 2 (paged): free_list and ctl as well (k_pool_init first)
  double m1, e2;
  uint64_t idle_limit_ticks;   // s_memrealtime ticks (100 MHz)
};

// n_cus workgroups of four waves on `s` -- which must own its hardware queue (hipExtStreamCreateWithCUMask)
// with_strict: the variant that carries the certified sweeps. layout: 0 = an arena per wave (separate LPE), 1 = pages shared
// through the pool's free list (separate LPE), 2 = shared pages, posteriors in place (page-starved batches)
void launch_session(bool with_strict, int layout, const SessionArgs& a, const void* in_base, void* out_base, const dynmath::SoftplusNode* sp_tab,
                    int n_cus, hipStream_t s);
// record `tk` as ticket `index` and make it visible (tail = index + 1); closed: no ticket will follow
void launch_session_publish(SessionTicket* ring, uint32_t* ctl, const SessionTicket& tk, uint32_t index, uint32_t ring_size, hipStream_t s);
void launch_session_close(uint32_t* ctl, hipStream_t s);

// ---- any band (wide_band.hip): reads whose half band exceeds MAX_HALF_BAND take a generic kernel, one workgroup per read ----
constexpr int WIDE_MAX_B = 4096;                          // band columns per row incl. the two guards (2 bw + 3), held in LDS
constexpr int WIDE_CPT = WIDE_MAX_B / 256;                // band columns per thread
constexpr int WIDE_MAX_HALF_BAND = (WIDE_MAX_B - 3) / 2;  // 2 046: band <= 4 093
struct WideArgs {
  const ReadDesc* descs;   // the wide reads of the batch
  int n_reads;
  const double* sig;
  const Emis* par;
  ReadState* st;
  TraceBuffers tb;
  TrainBuffers tr;
  char* arena;             // n_groups arenas of arena_bytes: the lattice of the read a workgroup is working on
  uint64_t arena_bytes;    // >= wide_arena_bytes of the largest wide read
  uint32_t* head;          // queue head (cleared by launch_wide_reads)
  const uint64_t* exp_tab; // dynmath::strict_exp_table on the device
  double m1, e2;
  int z_fail_status;
};
uint64_t wide_arena_bytes(uint64_t T, uint64_t bw, bool calc);
// job: 0 = Z only, 1 = align(calc_probabilities = true) up to the per-row path arrays (launch_segments follows), 2 = train
void launch_wide_reads(int job, const WideArgs& a, int n_groups, hipStream_t s);

// P1/P2 on the device: out[i] = hampel((REAL(raw[i]) - shift) / scale); REAL = float when compute_f32.
// raw_dtype: 0 float32, 1 int16, 2 float64, 3 int16 ADC with per-read float32 calibration (pA = (adc + cal_offset) *
// cal_scale in float32; cal_* may be null otherwise). norm_tmp: scratch of total samples * sizeof(REAL).
void launch_preprocess(const void* raw, int raw_dtype, int compute_f32, const uint64_t* offs,
                       const double* shift, const double* scale, const float* cal_offset, const float* cal_scale,
                       void* norm_tmp, double* out, int n_reads, uint64_t max_len, int W, double n_sigmas, hipStream_t s);
void launch_prep_params(const int32_t* kmers, const Emis* model, Emis* par, uint64_t total, uint32_t num_kmers,
                        hipStream_t s);
// free list = pages [first_free, n_pages); queue head = n_static; statistics cleared
void launch_pool_init(const PagePool& pool, uint32_t first_free, int n_static, hipStream_t s);
// The whole per-read pipeline as ONE launch of persistent waves (see nt_kernels.hip): each wave takes
// reads off the queue until it is empty; per read backward -> forward (+ posterior, Viterbi fill,
// decision bits | training statistics) -> Z check -> traceback -> segment-start posteriors.
// n_cus: compute units of the device (one 4-wave workgroup per CU).
// with_strict: the launch holds reads flagged READ_STRICT (JOB_ALIGN / JOB_ALIGN_INPLACE only): the kernel variant
// that carries both arithmetic flavours and branches per read
void launch_read_queue(QueueJob job, bool with_strict, const QueueArgs& q, int n_cus, hipStream_t s);
// per-segment median posterior + output rows for all reads of descs (after launch_read_queue)
void launch_segments(const ReadDesc* descs, int n_reads, uint64_t rows_total, uint32_t max_N, const ReadState* st,
                     TraceBuffers tb, SegRow* rows, int kmer_size, hipStream_t s);
// pooled[3*num_kmers] (zeroed by the caller) = per-k-mer (w, s1, s2) of the ok reads in descs, summed in a FIXED order:
// per read over its columns ascending, then over the reads in input order -- bit for bit the host's sum (pool_stats.hip).
// work: pool_stats_work_bytes(total_cols) device bytes; temp: pool_stats_temp_bytes(...) (rocprim's radix sort).
size_t pool_stats_temp_bytes(uint64_t total_cols, uint64_t num_kmers);
size_t pool_stats_work_bytes(uint64_t total_cols);
hipError_t launch_pool_stats(const ReadDesc* descs, int n_reads, uint32_t max_N, const ReadState* st, const int32_t* kmers,
                             TrainBuffers tb, double* pooled, uint64_t num_kmers, uint64_t total_cols, void* work, void* temp,
                             size_t temp_bytes, hipStream_t s);

}  // namespace dynk
