/*
 * dynamont_mi.h -- C ABI of the MI355X-native NT ("basic" mode) resquiggling core.
 *
 * This is the drop-in boundary for the one hot path of rnajena/dynamont that this repository
 * accelerates. The reference has no C ABI of its own: its boundary is the pybind11 class
 * `_dynamont.Aligner` (src/cpp/aligner_bindings.cpp:180-219) over the virtual C++ interface
 * dynamont::Aligner::{align,train} (include/dynamont/aligner.hpp:72-81). Every entry point
 * below names the reference interface it replaces. All citations are relative to the
 * reference checkout.
 *
 * Conventions: plain pointers and sizes only; no exceptions cross the ABI; functions return
 * a dyn_status; message texts equal the reference's exception texts (callers print them
 * into the `.errors` file, segment.py:172-176, so they are observable output).
 * A handle is bound to ONE GPU (one process per GPU; shard reads across handles/ranks, or use
 * dyn_multi_* below, which owns one handle per device). GPU work of one handle is serialised
 * internally, so calls on DIFFERENT batches of one handle may come from different threads; a
 * single batch object must not be used from two threads at once.
 */
#ifndef DYNAMONT_MI_H
#define DYNAMONT_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYN_ABI_VERSION 8 /* 8: dyn_aligner_session_idle_split (additive). 7: dyn_aligner_session_page_wait, dyn_comm_gather_bytes / _gathered_bytes / _allreduce_f64 (additive). 7: half bands 224 .. 2 046 are computed (wide_band.hip). 5: any band constructs (DYN_READ_BAND_TOO_WIDE per read); dyn_bam_*, dyn_csv_sink_wait / _open_part */

/* device argument of dyn_aligner_create: bind no GPU. Such a handle serves the host-side
 * contract only (model loading, dyn_aligner_info/_model, dyn_validate_batch); every compute
 * entry point fails with DYN_ERR_DEVICE -- there is no CPU compute path in this library. */
#define DYN_DEVICE_HOST_ONLY (-2)

/* dynamont::PoreType (include/dynamont/aligner.hpp:26-33), same numbering as the pybind enum
 * (aligner_bindings.cpp:184-189). */
enum dyn_pore {
  DYN_PORE_RNA002 = 0,
  DYN_PORE_RNA004 = 1,
  DYN_PORE_DNA_R9 = 2,
  DYN_PORE_DNA_R10_260 = 3,
  DYN_PORE_DNA_R10_400 = 4
};

/* Call-level status. INVALID_ARGUMENT corresponds to std::invalid_argument (-> Python
 * ValueError), RUNTIME to std::runtime_error (-> Python RuntimeError) in the reference. */
enum dyn_status {
  DYN_OK = 0,
  DYN_ERR_INVALID_ARGUMENT = 1,
  DYN_ERR_RUNTIME = 2,
  DYN_ERR_DEVICE = 3,      /* HIP runtime failure / no GPU / kernel image missing */
  DYN_ERR_OUT_OF_MEMORY = 4
};

/* Per-read status: a failing read never aborts its batch (segment.py:160-187). */
enum dyn_read_status {
  DYN_READ_OK = 0,
  DYN_READ_SIGNAL_EMPTY = 1,     /* "Signal is empty"                        aligner.cpp:149-152 */
  DYN_READ_SEQ_SHORT = 2,        /* "Sequence shorter than model kmer size"  aligner.cpp:154-157 */
  DYN_READ_SIGNAL_SHORT = 3,     /* "Signal too short compared to sequence"  aligner.cpp:159-163 */
  DYN_READ_INVALID_NT = 4,       /* "Invalid nucleotide: X"                  aligner.cpp:180-196 */
  DYN_READ_Z_MISMATCH = 5,       /* "Alignment failed: alignment scores do not match"  NT_aligner_api.cpp:288-291 */
  DYN_READ_TRAIN_Z_MISMATCH = 6, /* "Training failed: alignment scores do not match"   NT_aligner_api.cpp:622-625.
                                  * train() here has ONE Z (the backward sweep's); what it checks instead of |Zf - Zb| is
                                  * that Z is finite and that the posterior chain delivers all its mass to the end cell
                                  * and weight 1 per sample. Same outcome for inf / NaN samples; a sample ~1e6 model
                                  * standard deviations out (|Z| ~ 1e12), where the reference's two roundings of Z
                                  * disagree and it refuses the read, trains here. DESIGN.md section 3. */
  DYN_READ_INTERNAL = 7,         /* traceback left the lattice (cannot happen once the Z check passed) */
  DYN_READ_NTK_MISMATCH = 9,     /* "NTK alignment failed: alignment scores do not match"  NTK_aligner_api.cpp:911-917:
                                    what every read that passes validation gets from a handle created with mode
                                    "resquiggle"/"ntk" -- the reference's NTKAligner fails that check on every read in this
                                    snapshot (observed with the compiled reference, tests/golden/g11_ntk_messages.json) */
  DYN_READ_TOO_LARGE = 8,        /* "Read too large for the device memory budget": this read's lattice alone exceeds
                                    the HBM budget (or 2^31 rows); the reference would raise std::bad_alloc for that
                                    read only (segment.py:172-176), so it is a per-read status, not a batch error */
  DYN_READ_BAND_TOO_WIDE = 11,   /* "Band wider than this build's 4096 band columns for a read of this length": the handle was
                                    created with band > 4 093 and the read has more than 4 093 lattice columns, so that its
                                    half band min(band / 2, columns / 2) exceeds 2 046 -- what the generic kernel's rows hold
                                    (wide_band.hip). Every other read is computed as the reference computes it: half bands up
                                    to 223 by the register sweeps, 224 .. 2 046 by the generic kernel. */
  DYN_READ_BAD_SIGNAL = 10       /* "Signal could not be decoded" (dyn_batch_align_vbz_async): a POD5 chunk of this read is
                                    corrupt, truncated or shorter than the read's [start:end) slice. In the reference the
                                    pod5 reader raises inside the worker and the listener gets ONE line for that read,
                                    "error: worker, <message>\tN: ..\tRid: ..\tSid: .." (segment.py:178-187); the CSV sink
                                    writes that form for this status. The rest of the batch is unaffected. */
};

/* or-ed into raw_dtype of the *_raw_async calls: `raw` is not one concatenated array but a table of n_reads pointers
 * (const void* const*), one per read's slice; raw_offsets still hold the prefix sums of the slice lengths. The library
 * gathers the slices into its pinned staging buffer on its helper threads -- no copy on the caller's side. */
#define DYN_RAW_SCATTERED 0x100

typedef struct dyn_aligner dyn_aligner;
typedef struct dyn_batch dyn_batch;
typedef struct dyn_multi dyn_multi;

typedef struct dyn_info {
  int32_t abi_version;
  int32_t pore;
  int32_t rna;             /* 1 for RNA pores (aligner.cpp:62-86) */
  int32_t kmer_size;
  int32_t alphabet_size;   /* inferred from the model file (aligner.cpp:118) */
  int32_t device;
  uint64_t num_kmers;      /* alphabet_size ** kmer_size */
  uint64_t half_band;      /* band / 2 (aligner.cpp:21) */
  double log_m1, log_e1, log_e2; /* NT_aligner_api.cpp:84-86 */
  uint64_t max_half_band;  /* largest half band this build computes (2 046: the generic kernel's; the tuned sweeps': 223) */
} dyn_info;

/* One output row per segment (= one CSV line of segmentation_to_string, utils.py:193-232).
 * state is always 'M' on the NT path (NT_aligner_api.cpp:424-430) and is not stored per row. */
typedef struct dyn_segment_row {
  uint32_t signal_pos;   /* Segment::signalPosition   */
  uint32_t sequence_pos; /* Segment::sequencePosition */
  double probability;    /* Segment::probability      */
} dyn_segment_row;

/* Caller-allocated result arrays for dyn_align_batch / dyn_batch_fetch.
 * Segment arrays must hold dyn_segment_capacity() entries; read i's segments start at
 * seg_offsets[i] and there are n_segments[i] of them (0 when calc_probabilities == 0 or the
 * read failed). Any pointer except Z/status may be NULL to skip that column. */
typedef struct dyn_align_out {
  double* Z;                  /* [n_reads]  Result::Z (= backward score)        */
  int32_t* status;            /* [n_reads]  dyn_read_status                      */
  char* bad_char;             /* [n_reads]  offending base for DYN_READ_INVALID_NT, else 0 */
  uint64_t* seg_offsets;      /* [n_reads+1]                                     */
  uint64_t* n_segments;       /* [n_reads]                                       */
  uint64_t* sequence_positions; /* [capacity]  aligner_bindings.cpp:59,72        */
  uint64_t* signal_positions;   /* [capacity]  aligner_bindings.cpp:60,73        */
  double* probabilities;        /* [capacity]  aligner_bindings.cpp:61,74        */
  uint8_t* states;              /* [capacity]  'M'                               */
  uint64_t capacity;
} dyn_align_out;

/* Per-read training results (replaces dynamont::TrainingResult, aligner.hpp:48-53, whose
 * pybind form is a list of num_kmers dicts per read, aligner_bindings.cpp:86-107). The emission
 * update is returned SPARSE: only k-mers with weight > 0 (all others keep the loaded model,
 * NT_aligner_api.cpp:531-534). Read i's entries start at em_offsets[i]; there are em_count[i]. */
typedef struct dyn_train_out {
  double* Z;              /* [n_reads] */
  int32_t* status;        /* [n_reads] */
  char* bad_char;         /* [n_reads] or NULL */
  double* transitions;    /* [3*n_reads] m1, e1, e2 as probabilities (NT_aligner_api.cpp:703-722) */
  uint64_t* em_offsets;   /* [n_reads+1] */
  uint64_t* em_count;     /* [n_reads] */
  int32_t* em_code;       /* [capacity] k-mer code */
  double* em_mean;        /* [capacity] */
  double* em_stdev;       /* [capacity] */
  double* em_weight;      /* [capacity] or NULL: expected count w (sufficient statistic)  */
  double* em_sum;         /* [capacity] or NULL: sum of gamma*x                            */
  double* em_sumsq;       /* [capacity] or NULL: sum of gamma*x*x                          */
  double* trans_counts;   /* [2*n_reads] or NULL: expected #(E->M) and #(E->E) transitions (linear) */
  uint64_t capacity;      /* >= dyn_segment_capacity() */
} dyn_train_out;

/* Timings of the last dyn_batch_align / dyn_batch_train on a batch, measured with HIP events on the
 * stream the kernels were launched on. All reads of a batch run in ONE launch of persistent waves
 * (k_read_queue: per read backward -> forward -> Z check -> traceback), so the per-phase figures are
 * that launch's duration split by the share of wave time each phase took (device cycle counters). */
typedef struct dyn_timing {
  double ms_total;       /* read-queue launch + per-segment kernels */
  double ms_dp;          /* the read-queue launch alone */
  double ms_backward;    /* ms_dp x share of wave time in backward sweeps */
  double ms_forward;     /* ms_dp x share in forward sweeps (forward + posterior + posterior-Viterbi | training sums) */
  double ms_trace;       /* ms_dp x share in traceback / write-back, + k_median + k_final (+ k_pool_stats) */
  double wave_wait_share;  /* share of wave time spent waiting for lattice pages or the queue lock */
  double wave_occupancy;   /* sum of wave lifetimes / (waves x longest lifetime): 1 = every wave busy to the end */
  uint64_t cells;        /* in-band lattice cells processed: sum over ok reads of T*min(2bw+1,N) */
  uint64_t samples;      /* sum of signal lengths over ok reads */
  uint64_t reads_ok;
  uint32_t launches;     /* read-queue launches (1; 0 for a batch without an ok read): strict and plain reads share one launch */
  uint32_t lp_inplace;   /* 1: the page pool could not hold a separate-layout lattice (12 B per band slot) for every
                            wave, the posteriors overwrote the backward rows in place (8 B per slot, slower sweep) */
  uint32_t pool_pages;   /* pages in the lattice pool */
  uint32_t page_rows;    /* lattice rows per page */
  uint32_t n_static;     /* reads whose pages were reserved by the host (first round) */
  uint32_t n_waves;      /* persistent waves launched */
  uint32_t reads_strict; /* reads that went through the strict kernels (dyn_aligner_set_strict) */
  uint32_t reserved;
  /* (ABI 4) the strict reads of the launch apart: their share of ms_backward / ms_forward, and how often a register of 64
   * sums had to be recomputed with the restated glibc because a certificate could not decide it */
  double ms_backward_strict;
  double ms_forward_strict;
  uint64_t cert_fallbacks;
  uint64_t cert_rows;    /* lattice rows computed in the certified arithmetic (both sweeps) */
  /* (ABI 4) Asynchronous tickets that were waiting together while the GPU was busy are MERGED into one read-queue launch
   * (a launch of fewer reads than waves cannot balance). Such a ticket reports the timing of the launch that carried it
   * (ms_*, wave_*, cert_*) and its OWN reads, cells, samples and strict reads; launch_share is its part of that launch
   * (its cells / the launch's cells; 1 for a ticket that ran alone): sum ms_dp x launch_share over tickets = GPU time,
   * sum launch_share = number of launches. */
  double launch_share;
} dyn_timing;

/* (ABI 6, additive) The RESIDENT read queue. Asynchronous align(calc_probabilities=1) tickets of at least 512 reads whose
 * lattices fit one static arena per wave do not get a kernel launch each: the handle keeps ONE launch of resident waves
 * (a "session") on a stream that owns its hardware queue, publishes ticket after ticket into it while it runs, and closes
 * it when its pipeline has run dry -- no wave idles between batches (a launch per batch, or per two or three, ends with
 * its waves up to one read apart; k_session, dynamont_amd/csrc/nt_kernels.hip). Results are those of separate launches bit
 * for bit. Such a ticket's dyn_timing reports its own wave time: ms_dp = (wave-cycles its reads took) / waves, launches =
 * 0, launch_share = 0; the sessions themselves are accounted for here -- ms is the sum of the session kernels' durations
 * (HIP events on the session stream), which is what the roofline of a run must be taken over. Everything else (training,
 * Z-only jobs, small batches with no session open, the synchronous calls) runs as one launch per batch as before; a handle
 * whose session stream cannot be created, or with DYN_NO_SESSION=1 in the environment, never opens a session.
 * Resident waves that find no work for DYN_SESSION_IDLE_S seconds (default 20) while the session is held open leave on
 * their own (`aborted`); a ticket that was published after that is published again into the next session, its results
 * unchanged (`republished`). */
typedef struct dyn_session_stats {
  uint64_t sessions;       /* closed sessions */
  uint64_t tickets, reads, cells;
  double ms;               /* sum of their kernels' durations */
  uint64_t wave_cycles_busy, wave_cycles_idle, wave_cycles_life;   /* shader-clock cycles, summed over waves and sessions */
  uint64_t waves;          /* sum over sessions of waves launched */
  uint64_t aborted;        /* sessions whose waves raised the abort word (idle watchdog) */
  uint64_t republished;    /* tickets published again because their session had aborted before it took them */
} dyn_session_stats;

/* aligner_bindings.cpp:18-32 poreTypeFromString. Unknown -> DYN_ERR_INVALID_ARGUMENT,
 * message "Unknown pore type: <s>". */
int dyn_pore_from_string(const char* s, int* pore_out, char* err, uint64_t errcap);

/* PyAligner ctor -> makeAligner -> NTAligner ctor (aligner_bindings.cpp:34-51,112-130;
 * NT_aligner_api.cpp:11-19; aligner.cpp:13-36,88-143). mode "basic" or "nt": the NT path this library accelerates.
 * mode "resquiggle" or "ntk" (-> NTKAligner, aligner_bindings.cpp:46-49) is accepted like the reference accepts it and
 * behaves like the reference's does in this snapshot: reads fail validation with the usual messages or get
 * DYN_READ_NTK_MISMATCH, training returns DYN_ERR_RUNTIME "Training is not implemented for this aligner" (aligner.cpp:
 * 38-44); no kernel runs. Any other value -> DYN_ERR_INVALID_ARGUMENT "Unknown aligner mode: <m>".
 * band: any value, as in the reference (aligner.cpp:21). The tuned sweeps hold 448 band slots per lattice row (half bands
 * up to 223: every caller in the reference fixes band = 400, segment.py:45, utils.py:161). A read whose half band
 * min(band / 2, columns / 2) exceeds that -- band > 447 AND more than 447 lattice columns -- takes a generic kernel in the
 * same batch (wide_band.hip: one workgroup per read, rows of up to 4 096 band columns in LDS, the reference's own
 * arithmetic in every cell: Z bit for bit; align, the Z-only call and train alike; a fraction of the tuned sweeps' rate).
 * Only a half band above 2 046 is refused, per read: DYN_READ_BAND_TOO_WIDE. device < 0 -> current HIP device. */
int dyn_aligner_create(const char* model_path, int pore, const char* mode, int threads,
                       uint64_t band, int device, dyn_aligner** out, char* err, uint64_t errcap);
/* The lattice pool of a destroyed handle (up to ~100 GB; allocating or freeing that much takes seconds) is PARKED per
 * device and taken over by the next handle created on that device -- the reference's training loop builds a new
 * Aligner for every batch (train.py:179,227). dyn_release_cached_memory() frees whatever is parked (all devices);
 * DYN_NO_POOL_CACHE=1 in the environment switches the parking off. */
void dyn_aligner_destroy(dyn_aligner* a);
void dyn_release_cached_memory(void);
int dyn_aligner_info(const dyn_aligner* a, dyn_info* info);
/* Totals over the handle's CLOSED sessions (dyn_session_stats above). Closes an open session first and waits until its
 * waves have left -- call it when the tickets of interest have been waited for. */
int dyn_aligner_session_stats(dyn_aligner* a, dyn_session_stats* out);
/* (ABI 7, additive) The part of dyn_session_stats.wave_cycles_idle that waves of PAGED sessions (page-starved batches: reads of
 * 100 k samples, BASELINE configs[2]) spent getting their lattice pages from the shared pool -- giving back what they held,
 * waiting for the free list to hold their request, taking it. Same units and the same closing behaviour as above. No
 * counterpart in the reference (it allocates eight T x B matrices per read, NT_aligner_api.cpp:249-262). */
int dyn_aligner_session_page_wait(dyn_aligner* a, uint64_t* wave_cycles_waiting_for_pages);
/* (ABI 8, additive) Where the resident waves' idle share sat, counted by the waves themselves (wave-cycles, summed over the
 * handle's sessions like dyn_session_stats; closes an open session first): out4[0] before a wave's first read (part of
 * wave_cycles_idle), out4[1] the part of out4[0] spent getting pages, out4[2] in a wave's LAST turn -- no read left to claim,
 * polling for the session's close; part of wave_cycles_life - busy - idle --, out4[3] the sum over the sessions of the longest
 * last turn of any wave. A finite run pays out4[2] once per session ("Paged sessions"); a stream does not.
 * No counterpart in the reference. */
int dyn_aligner_session_idle_split(dyn_aligner* a, uint64_t out4[4]);
/* enabled = 0: no resident read queue on this handle (one launch per batch, as DYN_NO_SESSION=1 does for a process).
 * enabled = 1: sessions of n_cus - reserved_cus workgroups, one per compute unit. A resident session leaves 9.5 KB of LDS and 152
 * registers per lane free on every CU it occupies: the library's own small kernels run beside it. A kernel whose workgroups need
 * more (RCCL's: 37 KB of LDS, 248-256 registers) starts beside a session iff it has NO MORE workgroups than reserved_cus
 * (reserve multiples of 8: one per XCD); one workgroup more and it waits until the session has ended (measured:
 * tools/ubench/resident_probe.hip) -- a process that runs such kernels WHILE tickets are in flight
 * reserves CUs and caps their grid (RCCL: one workgroup per channel, NCCL_MAX_NCHANNELS), accepts the wait (sessions end when
 * the handle's pipeline runs dry), or switches the queue off; bench.py --gpus N does the first and falls back to the last
 * after probing. Closes an open session first. Default: enabled, nothing reserved (DYN_SESSION_RESERVE_CUS in the
 * environment changes the default). */
int dyn_aligner_set_session_mode(dyn_aligner* a, int enabled, int reserved_cus);
/* Dense model table in k-mer-code order, (mean, stdev) interleaved, 2*num_kmers doubles. */
int dyn_aligner_model(const dyn_aligner* a, double* out2n);
/* Replace the model table (same layout as dyn_aligner_model: k-mer-code order, (mean, stdev) interleaved). The handle
 * then behaves like one created from a model file holding these values -- the reference's training loop writes such a
 * file after every batch and builds a new Aligner from it (train.py:179,221-227); a float64 survives that round trip
 * through its shortest decimal representation unchanged. Waits for the handle's device to be idle; no asynchronous
 * ticket of the handle may be pending. */
int dyn_aligner_set_model(dyn_aligner* a, const double* in2n);
/* Upper limit on HBM used for lattice workspaces (bytes; 0 = 90 % of free memory). */
int dyn_aligner_set_mem_budget(dyn_aligner* a, uint64_t bytes);
/* Strict mode (affects align with calc_probabilities only). The reference's traceback takes exact fp64 comparisons
 * (NT_aligner_api.cpp:445-448). Where two neighbouring lattice columns carry the same emission parameters (every RNA
 * read that starts with the polyA pad + A, any homopolymer of k+1 bases, distinct k-mers with coinciding table entries)
 * moving the border between their segments leaves the exact score unchanged: such a comparison is a tie in exact
 * arithmetic and the reference's choice rests on the last bits of glibc's exp/log1p inside logPlus (aligner.cpp:
 * 276-285). The plain kernels use a table softplus that is <= 1 ulp away from those and reproduce 3 397 of the 3 400
 * such reads of tests/golden/g10_ties.npz. The strict kernels reproduce the reference's sums bit for bit at 1.3-1.4x
 * the price of a row ("certified arithmetic", dynamont_amd/csrc/dp_math_strict.hpp): the emission's quotient formed
 * exactly from stdev and 1/stdev, each logPlus from the table softplus plus a certificate that its rounded sum cannot
 * depend on the last bits of the softplus, and glibc 2.35's x86-64 exp (FMA variant) / fdlibm log1p restated operation
 * by operation for the ~1e-4 of the sums the certificate cannot decide.
 *   mode 0: off
 *   mode 1: (DEFAULT) every read with such a pair of columns (dyn_tie_rows != 0) takes the strict backward sweep (Z
 *           becomes bit-identical) and the strict arithmetic for the forward rows up to the one in which the last tied
 *           pair has left the band -- the Viterbi values of a row depend on forward values of earlier rows only, so every
 *           decision up to there is the reference's own, and no decision is taken on a column outside the band
 *   mode 2: every read, every row of both sweeps
 * Returns DYN_ERR_INVALID_ARGUMENT for an unknown mode, and for modes 1/2 on a model holding a stdev whose significand
 * is all ones (no division-free exact quotient exists for that one divisor; no decimal parses to it). */
int dyn_aligner_set_strict(dyn_aligner* a, int mode);
/* train(): which reads are refused. Off (default): the posterior chain's own rule -- Z finite, all mass delivered to the
 * end cell, weight 1 per sample (see DYN_READ_TRAIN_Z_MISMATCH). On: additionally the reference's rule,
 * |Zf - Zb| / (T x B) > 1e-8 -> "Training failed: alignment scores do not match" (NT_aligner_api.cpp:619-625), with Zf
 * from one more (Z-only, no lattice traffic) forward sweep per read: `.errors` then lists the pathological reads the
 * reference lists (a sample ~1e6 model standard deviations out), at ~25 % more time per train() launch. */
int dyn_aligner_set_train_zcheck(dyn_aligner* a, int on);
/* The rule of mode 1 for one read, given its k-mer codes (dyn_validate_batch) and signal length: 0 = no structural tie;
 * otherwise the number of forward rows that run in the strict arithmetic (UINT32_MAX: all of them). Host only. */
uint32_t dyn_tie_rows(const dyn_aligner* a, const int32_t* kmers, uint64_t n_kmers, uint64_t signal_len);
/* Message of the last failing call on this handle (thread-unsafe like the handle itself). */
const char* dyn_aligner_last_error(const dyn_aligner* a);
/* Reference exception text for a per-read status (bad_char fills "Invalid nucleotide: X"). */
int dyn_read_strerror(int read_status, char bad_char, char* buf, uint64_t cap);

/* Number of segment rows to allocate: sum over reads of max(0, len(seq_i) - k + 1). */
uint64_t dyn_segment_capacity(const dyn_aligner* a, uint64_t n_reads, const uint64_t* seq_offsets);

/* Host front half of align()/train() only: validateInput (aligner.cpp:145-164) then
 * sequenceToKmers (aligner.cpp:166-205) per read. status/bad_char: [n_reads]. kmers_out
 * (optional): k-mer codes of read i at kmers_out[seg_offset_i ...] in the dyn_segment_capacity()
 * layout (prefix sums of max(0, len_i - k + 1)). Needs no GPU. */
int dyn_validate_batch(const dyn_aligner* a, uint64_t n_reads, const uint64_t* sig_offsets,
                       const char* seqs, const uint64_t* seq_offsets, int32_t* status,
                       char* bad_char, int32_t* kmers_out, uint64_t kmers_cap);

/* NTAligner::align for a batch of reads held in HOST memory (NT_aligner_api.cpp:230-312):
 * signals = concatenated fp64 samples, read i = [sig_offsets[i], sig_offsets[i+1]);
 * seqs = concatenated bases (aligner orientation), read i = [seq_offsets[i], seq_offsets[i+1]).
 * Equivalent to create + align + fetch + destroy of a dyn_batch. */
int dyn_align_batch(dyn_aligner* a, uint64_t n_reads, const double* signals,
                    const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                    int calc_probabilities, dyn_align_out* out);

/* NTAligner::train for a batch (NT_aligner_api.cpp:567-639, 462-561, 641-725).
 * pooled3n (optional, 3*num_kmers doubles, caller-zeroed or accumulated across calls) receives
 * the batch-pooled sufficient statistics sum_i (w, s1, s2) in k-mer-code order -- the quantity
 * a multi-GPU job all-reduces (BASELINE.json config 5). */
int dyn_train_batch(dyn_aligner* a, uint64_t n_reads, const double* signals,
                    const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                    dyn_train_out* out, double* pooled3n);

/* segmentation_to_string (src/dynamont/segmentation/utils.py:193-232) for a whole batch, multi-
 * threaded on the host. `res` is a filled dyn_align_out; seqs/seq_offsets are the aligner-orientation
 * sequences the batch was aligned with; sig_offsets[i] = sigOffset, last_index[i] = lastIndex of the
 * reference call. Read i's rows are written to out[row_begin[i] .. row_end[i]) (each read formats
 * into its own worst-case slot, so the byte ranges are not contiguous; failed reads get an empty
 * range and the caller writes their .errors line). out_cap must be >= dyn_format_csv_bound().
 * Bytes are identical to the reference's Python formatting, including f"{p:.6f}". */
uint64_t dyn_format_csv_bound(const dyn_aligner* a, uint64_t n_reads, const dyn_align_out* res,
                              const char* const* readids, const char* const* signalids);
int dyn_format_csv(const dyn_aligner* a, uint64_t n_reads, const dyn_align_out* res,
                   const char* seqs, const uint64_t* seq_offsets, const char* const* readids,
                   const char* const* signalids, const int64_t* sig_offsets,
                   const int64_t* last_index, int threads, char* out, uint64_t out_cap,
                   uint64_t* row_begin, uint64_t* row_end);

/* Closes the gaps between the per-read ranges dyn_format_csv produced: all rows become one contiguous run at the
 * front of `out`, in read order; row_begin/row_end are updated. Returns the total byte count. */
uint64_t dyn_csv_compact(char* out, uint64_t n_reads, uint64_t* row_begin, uint64_t* row_end);

/* The text of a k-mer model file as write_kmer_model produces it (src/dynamont/segmentation/utils.py:136-152: header
 * "kmer\tlevel_mean\tlevel_stdv", then f"{kmer}\t{mean}\t{stdev}\n" per row; dynamont-train writes one per batch,
 * train.py:221-224): kmers = n rows of k characters, file order; a float64 prints as Python's repr(). Returns the
 * byte count; with out == NULL or cap too small nothing is written and the size to provide is returned. */
uint64_t dyn_format_model(const char* kmers, int k, const double* mean, const double* stdev, uint64_t n,
                          char* out, uint64_t cap);

/* ---- the output half of dynamont-resquiggle (src/dynamont/segmentation/segment.py:69-107, the listener) ----
 *
 * A sink owns `<out>.csv.zst` (ONE zstd frame at `level`, like the reference's stream writer; header line written at
 * open) and appends to `<out>.errors`. dyn_csv_sink_submit hands it a ticket of dyn_batch_align[_raw]_async and
 * returns at once: a sink thread waits for the batch, formats its rows (dyn_format_csv, bytes ==
 * utils.segmentation_to_string), compresses them on `threads` threads and writes them in submission order; reads
 * whose status is not DYN_READ_OK get the reference's line
 * "error: native, <message>\tT: <samples>\tN: <bases>\tRid: <readid>\tSid: <signalid>" (segment.py:172-176).
 * sig_offsets[i] = sigOffset of segmentation_to_string (the read's first sample in the raw signal), signal_lengths[i]
 * = its samples (lastIndex = sum). Everything passed to submit (and the ticket) must stay valid until
 * dyn_csv_sink_completed() has counted the batch; the sink never destroys a ticket. */
typedef struct dyn_csv_sink dyn_csv_sink;
int dyn_csv_sink_open(const char* csv_zst_path, const char* errors_path, int level, int threads, dyn_csv_sink** out,
                      char* err, uint64_t errcap);
/* The same sink writing a PART of a frame that several processes write (one process per GPU: every rank compresses its
 * own rows, the root only concatenates bytes -- compressing is the largest host cost of the output by far). A part with
 * `first` carries the CSV header line and the zstd frame header, a part with `last` the frame's closing block;
 * first = last = 1 is dyn_csv_sink_open. Parts are sequences of complete zstd blocks: [first part][any other parts, in any
 * order][DYN_ZSTD_FRAME_END] is ONE valid frame. */
int dyn_csv_sink_open_part(const char* csv_zst_path, const char* errors_path, int level, int threads, int first, int last,
                           dyn_csv_sink** out, char* err, uint64_t errcap);
#define DYN_ZSTD_FRAME_END "\x01\x00\x00" /* an empty last block (raw, size 0): 3 bytes; the frames carry no checksum */
int dyn_csv_sink_submit(dyn_csv_sink* s, dyn_aligner* a, dyn_batch* ticket, const dyn_align_out* res, uint64_t n_reads,
                        const char* seqs, const uint64_t* seq_offsets, const char* const* readids,
                        const char* const* signalids, const int64_t* sig_offsets, const uint64_t* signal_lengths);
/* The same with stored_bases[i] = the bases of read i's basecall AS STORED (before the RNA reversal and polyA pad), or
 * NULL: the N of a read that fails like the reference's WORKER does -- its signal cannot be read, DYN_READ_BAD_SIGNAL,
 * "error: worker, <message>\tN: <len(read)>\tRid: ...\tSid: ..." (segment.py:178-187) -- where the aligner's own failures
 * report the read in aligner orientation, pad included (segment.py:172-176). Additive: dyn_csv_sink_submit = NULL. */
int dyn_csv_sink_submit_bases(dyn_csv_sink* s, dyn_aligner* a, dyn_batch* ticket, const dyn_align_out* res, uint64_t n_reads,
                              const char* seqs, const uint64_t* seq_offsets, const char* const* readids,
                              const char* const* signalids, const int64_t* sig_offsets, const uint64_t* signal_lengths,
                              const uint32_t* stored_bases);
/* one line for `.errors` from the caller (reads that failed before they reached the aligner, segment.py:178-187) */
int dyn_csv_sink_error_line(dyn_csv_sink* s, const char* line);
/* 1 once the sink has failed (a batch error, the compressor, the output file): later submits return DYN_ERR_RUNTIME, and
 * dyn_csv_sink_close reports the first message. Lets a producer stop instead of aligning the rest of its input. */
int dyn_csv_sink_failed(dyn_csv_sink* s);
/* batches fully consumed so far */
uint64_t dyn_csv_sink_completed(const dyn_csv_sink* s);
/* blocks until `count` batches have been consumed, the sink has failed, or timeout_ms have passed (< 0: no timeout);
 * returns the batches consumed so far. What a producer that keeps a bounded number of batches in flight waits on. */
uint64_t dyn_csv_sink_wait(dyn_csv_sink* s, uint64_t count, int timeout_ms);
/* drains, closes the frame and the file, frees the sink; DYN_ERR_RUNTIME + message if any batch, compression or write
 * failed */
int dyn_csv_sink_close(dyn_csv_sink* s, uint64_t* csv_bytes, uint64_t* compressed_bytes, uint64_t* error_lines, char* err,
                       uint64_t errcap);

/* ---- the input half of dynamont-resquiggle: the basecalls, in batches (src/dynamont/segmentation/segment.py:189-258
 * generate_jobs over pysam's fetch(until_eof=True); segment.py:141-158 for the RNA orientation) ----
 *
 * A reader walks an (unaligned) BAM file -- BGZF blocks inflated a window ahead on `threads` threads, CRC-checked --
 * and dyn_bam_next returns COLUMNS for the next (up to) max_reads jobs, in file order:
 *   names / signal_ids     NUL-terminated strings back to back; *_off[i] = start of entry i, *_off[n] = total bytes
 *                          (signal id = the `pi` tag when present, else the read name)
 *   signal_uuid[16 i ..]   the signal id as the 16 bytes of its UUID (what a POD5 reads table is keyed by),
 *                          signal_uuid_ok[i] = 0 when the text is not 32 hex digits (hyphens ignored)
 *   seqs / seq_off         the basecalled sequences back to back, no separators, ready for dyn_batch_align_*:
 *                          as stored, or with DYN_JOBS_RNA reversed and with `rna_pad` in front unless the reversed read
 *                          starts with it
 *   shift, scale           the `sm`, `sd` tags (a BAM `f` value widened to double, as pysam hands it out)
 *   start, end             `sp + ts`, `sp + ns` (`sp` = 0 when absent): the slice of the raw signal
 *   file_id / files        index into the batch's distinct raw-file names (`fn`, else `f5`)
 *   bases                  bases of the record as stored (the N of the reference's error lines)
 * The arrays belong to the reader and stay valid until its next call. Reads with `qs` < min_qual (min_qual != 0) are
 * counted in dyn_bam_skipped and left out; of the reads that remain, those with index % world == rank are returned
 * (every rank of a multi-GPU run walks the file and keeps its share). n = 0: end of file. A tag generate_jobs reads
 * without asking (qs, ns, ts, fn|f5, sm, sd) and that is absent fails the call with DYN_ERR_INVALID_ARGUMENT "tag 'xx'
 * not present" (pysam: KeyError, same text); a damaged file with DYN_ERR_RUNTIME. */
typedef struct dyn_bam_reader dyn_bam_reader;
typedef struct dyn_job_batch {
  uint64_t n;
  const char* names;
  const uint64_t* name_off;
  uint64_t names_bytes;
  const char* signal_ids;
  const uint64_t* signal_id_off;
  uint64_t signal_ids_bytes;
  const uint8_t* signal_uuid;
  const uint8_t* signal_uuid_ok;
  const char* seqs;
  const uint64_t* seq_off;
  uint64_t seqs_bytes;
  const double* shift;
  const double* scale;
  const int64_t* start;
  const int64_t* end;
  uint64_t n_files;
  const char* files;
  const uint64_t* file_off;
  uint64_t files_bytes;
  const uint32_t* file_id;
  const uint32_t* bases;
} dyn_job_batch;
#define DYN_JOBS_RNA 1u
int dyn_bam_open(const char* path, int threads, const char* rna_pad, dyn_bam_reader** out, char* err, uint64_t errcap);
int dyn_bam_next(dyn_bam_reader* r, uint64_t max_reads, uint32_t flags, double min_qual, uint32_t rank, uint32_t world,
                 dyn_job_batch* out, char* err, uint64_t errcap);
uint64_t dyn_bam_skipped(const dyn_bam_reader* r);
void dyn_bam_close(dyn_bam_reader* r);

/* ---- staged form: inputs resident in HBM before the timed region (bench.py, pipelining) ---- */

/* Validate (aligner.cpp:145-164), k-mer-code (aligner.cpp:166-205), and upload one batch. */
int dyn_batch_create(dyn_aligner* a, uint64_t n_reads, const double* signals,
                     const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                     dyn_batch** out);
/* The same with P1/P2 done on the device (src/dynamont/segmentation/segment.py:146-153,
 * train.py:163-170, utils.py:16-43): raw = the concatenated raw[start:end) slices of every read
 * (raw_dtype 0 = float32 picoampere `signal_pa`, 1 = int16 ADC `signal`, 2 = float64), read i =
 * [raw_offsets[i], raw_offsets[i+1]); per read x = REAL(raw); x -= shift[i]; x /= scale[i]; then the
 * Hampel filter (window, n_sigmas) -- (3, 3.0) for dynamont-resquiggle, (7, 5.0) for dynamont-train.
 * compute_f32 = 1 performs the arithmetic in float32, as NumPy does in train.py where the signal stays
 * float32; 0 = float64 as in segment.py. Results are bit-identical to the NumPy code. */
int dyn_batch_create_raw(dyn_aligner* a, uint64_t n_reads, const void* raw, int raw_dtype,
                         const uint64_t* raw_offsets, const double* shift, const double* scale,
                         int hampel_window, double hampel_n_sigmas, int compute_f32, const char* seqs,
                         const uint64_t* seq_offsets, dyn_batch** out);
/* Device-resident preprocessed signals of a batch copied back to the host (count doubles). */
int dyn_batch_signals(dyn_batch* b, double* out, uint64_t count);
void dyn_batch_destroy(dyn_batch* b);
/* Run the kernels (stream-ordered, returns after the stream is idle). Batches made by dyn_batch_create[_raw] only: a
 * ticket of the asynchronous API is a one-shot submission (it may have shared its launch, and with it every device
 * buffer, with other tickets) and gets DYN_ERR_INVALID_ARGUMENT here. dyn_batch_fetch, dyn_batch_signals,
 * dyn_batch_device_results and dyn_batch_timing serve a completed ticket whether or not its launch was shared. */
int dyn_batch_align(dyn_batch* b, int calc_probabilities);
int dyn_batch_train(dyn_batch* b);
/* Copy results of the last dyn_batch_align to the host. */
int dyn_batch_fetch(dyn_batch* b, dyn_align_out* out);
int dyn_batch_fetch_train(dyn_batch* b, dyn_train_out* out, double* pooled3n);
/* Device-resident results of the last dyn_batch_align, for an RCCL gather without a host hop:
 * rows = dyn_segment_row[capacity] (read i at seg_offsets[i], as in dyn_align_out);
 * z_status = per read {double Z; int32 status; uint32 n_segments}. Pointers stay valid until the
 * next call on the batch. */
int dyn_batch_device_results(dyn_batch* b, void** d_rows, uint64_t* capacity, void** d_z_status);
/* Device-resident pooled sufficient statistics (3*num_kmers doubles) of the last
 * dyn_batch_train, for an RCCL all-reduce. Computed by the first call (a sort by k-mer and a fixed-order sum of the
 * launch's per-column sums, on the handle's compute stream) and waited for: training that never asks does not pay for it. */
int dyn_batch_device_pooled(dyn_batch* b, void** d_pooled3n, uint64_t* count);
int dyn_batch_timing(const dyn_batch* b, dyn_timing* t);

/* Planning diagnostic, needs no GPU. A batch runs as ONE launch of persistent waves that take reads off a
 * queue; the lattice of a read lives in pages of a pool. When the pool cannot hold a lattice for every wave
 * (long reads: BASELINE config 3), the engine plans the queue order by replaying the launch on the host. This
 * entry exposes that planner: pages[i] / rows[i] = lattice pages and rows of read i, LONGEST FIRST; n_slots =
 * persistent waves (4 per CU); pool_pages = pages in the pool. order_out[k] = index of the read at queue position k;
 * the makespans are in lattice rows per wave (every wave sweeps rows at the same rate). */
int dyn_plan_queue(uint64_t n_reads, const uint32_t* pages, const uint64_t* rows, uint64_t n_slots,
                   uint64_t pool_pages, uint32_t* order_out, uint64_t* makespan_longest_first,
                   uint64_t* makespan_planned);
/* (ABI 6, additive) Planning diagnostic, needs no GPU: the order in which a PAGED session of the resident read queue takes
 * the reads of a page-starved ticket. Input: n_reads reads ranked LONGEST FIRST (rank 0 = longest); order_out[k] = rank of
 * the read at queue position k. The longer 7/8 are dealt out in a low-discrepancy order (any window of consecutive positions
 * holds ranks spread evenly over the whole range: the demand for lattice pages stays near its average), the shortest eighth
 * follows, longest first. */
int dyn_session_order(uint64_t n_reads, uint32_t* order_out);

/* ---- asynchronous form: a stream of batches with H2D, kernels, D2H and host marshalling of
 * neighbouring batches overlapped (the reference keeps its worker pool permanently fed,
 * segment.py:301-325) ----
 *
 * dyn_batch_align_async == dyn_align_batch, but returns at once with a ticket: validateInput /
 * sequenceToKmers, the H2D copy, every kernel, the D2H copy and the unpacking into `out` run on the
 * handle's pipeline threads and streams. Batches of one handle complete in submission order.
 * EVERYTHING the call was given (signals, offsets, seqs, `out` and the arrays it points to) must stay
 * valid and untouched until dyn_batch_wait(ticket) has returned. After the wait the ticket behaves
 * like an aligned batch: dyn_batch_timing and dyn_batch_device_results work on it. Release it with
 * dyn_batch_destroy (which waits first if need be). Inputs allocated with dyn_host_alloc are copied
 * by DMA without staging.
 * Merged launches: a read-queue launch of fewer reads than the device has waves (1 024) cannot balance -- every wave
 * holds one read and the launch lasts as long as its slowest. Align tickets of one kind that are WAITING while the GPU
 * still has a launch queued are therefore run as one launch (at most a quarter of the tickets the caller has had in
 * flight at once -- 12 in flight: launches of three batches -- and eight reads per wave): same results, same
 * completion order, and each ticket still reports the launch that carried it (dyn_timing, launch_share) and serves its
 * own slice of the device rows (dyn_batch_device_results). A ticket that finds the GPU idle starts at once, alone;
 * training tickets are never merged. Environment variable DYN_NO_MERGE=1 switches the merging off. */
int dyn_batch_align_async(dyn_aligner* a, uint64_t n_reads, const double* signals,
                          const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                          int calc_probabilities, dyn_align_out* out, dyn_batch** ticket);
/* dyn_train_batch in the same form (pooled3n as there; it is accumulated into when the batch completes). */
int dyn_batch_train_async(dyn_aligner* a, uint64_t n_reads, const double* signals,
                          const uint64_t* sig_offsets, const char* seqs, const uint64_t* seq_offsets,
                          dyn_train_out* out, double* pooled3n, dyn_batch** ticket);
/* The asynchronous calls on RAW slices: dyn_batch_create_raw's preprocessing (P1/P2: segment.py:146-153,
 * train.py:163-170, utils.py:16-43) as the first stage of the same pipeline -- the helper threads gather
 * [offsets | shift | scale | samples] into pinned staging, one DMA carries them up, k_normalise / k_hampel run on the
 * compute stream in front of the batch's read queue. This is what dynamont-resquiggle / dynamont-train drive: the
 * reference keeps its workers fed the same way (segment.py:296-325). raw_dtype 3 (these two calls only): int16 ADC
 * counts with the read's pod5 calibration, picoampere = (float(adc) + cal_offset[i]) * cal_scale[i] in float32 -- the
 * value `signal_pa` (src/dynamont/pod5_io.py:6-16) hands the reference, formed on the device instead (half the bytes
 * over PCIe, no per-read NumPy pass); cal_offset / cal_scale may be NULL for the other dtypes. raw, raw_offsets,
 * cal_*, shift, scale, seqs, seq_offsets and `out` must stay valid until dyn_batch_wait(ticket) has returned. */
int dyn_batch_align_raw_async(dyn_aligner* a, uint64_t n_reads, const void* raw, int raw_dtype,
                              const uint64_t* raw_offsets, const float* cal_offset, const float* cal_scale,
                              const double* shift, const double* scale,
                              int hampel_window, double hampel_n_sigmas, int compute_f32, const char* seqs,
                              const uint64_t* seq_offsets, int calc_probabilities, dyn_align_out* out,
                              dyn_batch** ticket);
int dyn_batch_train_raw_async(dyn_aligner* a, uint64_t n_reads, const void* raw, int raw_dtype,
                              const uint64_t* raw_offsets, const float* cal_offset, const float* cal_scale,
                              const double* shift, const double* scale,
                              int hampel_window, double hampel_n_sigmas, int compute_f32, const char* seqs,
                              const uint64_t* seq_offsets, dyn_train_out* out, double* pooled3n,
                              dyn_batch** ticket);
/* The same for a batch whose signal is still in POD5 form: VBZ-compressed chunks (POD5 format specification: zstd around
 * StreamVByte-16 of the zigzag-coded sample deltas) -- the decode ONT's pod5 library performs inside `record.signal`,
 * which the reference calls per read (src/dynamont/pod5_io.py:6-16). Read i's signal is the concatenation of the chunks
 * [read_chunk_offsets[i], read_chunk_offsets[i+1]) (chunk c: chunk_bytes[c] compressed bytes at chunk_ptrs[c] holding
 * chunk_samples[c] samples), of which the slice [slice_start[i], slice_start[i] + (raw_offsets[i+1] - raw_offsets[i]))
 * is aligned (segment.py:147: signal[start:end]). The pipeline's helper threads decode whole reads straight into the
 * pinned staging buffer. cal_offset / cal_scale as for raw_dtype 3, or both NULL for plain ADC counts (raw_dtype 1).
 * A chunk that does not decode fails the batch with DYN_ERR_RUNTIME ("VBZ: ..."). */
int dyn_batch_align_vbz_async(dyn_aligner* a, uint64_t n_reads, const void* const* chunk_ptrs, const uint64_t* chunk_bytes,
                              const uint32_t* chunk_samples, const uint64_t* read_chunk_offsets, const uint64_t* slice_start,
                              const uint64_t* raw_offsets, const float* cal_offset, const float* cal_scale,
                              const double* shift, const double* scale, int hampel_window, double hampel_n_sigmas,
                              int compute_f32, const char* seqs, const uint64_t* seq_offsets, int calc_probabilities,
                              dyn_align_out* out, dyn_batch** ticket);
/* One VBZ chunk -> `samples` int16 values (host only; tests, other readers). */
int dyn_vbz_decode(const void* blob, uint64_t blob_bytes, uint32_t samples, int16_t* out, char* err, uint64_t errcap);
/* Block until the batch behind the ticket is complete; returns its status code, with the message in
 * dyn_aligner_last_error. Returns DYN_OK at once for batches of the synchronous calls. */
int dyn_batch_wait(dyn_batch* ticket);
/* Page-locked host memory for inputs/outputs of the asynchronous calls (NULL on failure). */
void* dyn_host_alloc(uint64_t bytes);
void dyn_host_free(void* p);

/* ---- several GPUs behind one handle (reads are independent, NT_aligner_api.cpp:230-312) ----
 *
 * dyn_multi owns one dyn_aligner per entry of device_ids (an ordinal may appear more than once). A batch is cut
 * into n_devices contiguous ranges of equal lattice work (sum of signal lengths); every range runs through its
 * device's asynchronous pipeline and the results land directly in the caller's arrays, in the layout of
 * dyn_align_batch / dyn_train_batch. No device-to-device traffic and no Python: this is how a C/C++ host shards
 * a batch over the GPUs of a node (one PROCESS per GPU with an RCCL gather is the other form: bench.py --gpus N). */
int dyn_multi_create(const char* model_path, int pore, const char* mode, int threads, uint64_t band,
                     const int* device_ids, int n_devices, dyn_multi** out, char* err, uint64_t errcap);
void dyn_multi_destroy(dyn_multi* m);
int dyn_multi_device_count(const dyn_multi* m);
/* the per-device handle (dyn_aligner_info, dyn_aligner_model, dyn_aligner_set_mem_budget, ...) */
dyn_aligner* dyn_multi_handle(dyn_multi* m, int i);
const char* dyn_multi_last_error(const dyn_multi* m);
int dyn_multi_align_batch(dyn_multi* m, uint64_t n_reads, const double* signals, const uint64_t* sig_offsets,
                          const char* seqs, const uint64_t* seq_offsets, int calc_probabilities, dyn_align_out* out);
/* pooled3n as in dyn_train_batch: the sum over all devices is ADDED to it */
int dyn_multi_train_batch(dyn_multi* m, uint64_t n_reads, const double* signals, const uint64_t* sig_offsets,
                          const char* seqs, const uint64_t* seq_offsets, dyn_train_out* out, double* pooled3n);

/* ---- one PROCESS per GPU: the two exchanges of a multi-GPU job over RCCL / xGMI ----
 *
 * The reference scales by forking worker processes on one host (segment.py:296-325); reads are independent
 * (NT_aligner_api.cpp:230-312), so a multi-GPU job shards the reads over one process per GPU and needs nothing but
 *   config 4: a gather of the per-read segment rows to one rank           -> dyn_comm_gather_rows
 *   config 5: a sum all-reduce of the pooled statistics (w, s1, s2)[4^k]  -> dyn_comm_allreduce_pooled
 * both straight from the device buffers of a batch. librccl is bound with dlopen at the first call below; nothing else
 * in this library needs it. Rank 0 creates the 128-byte id and hands it to the other ranks by the host's own means
 * (a file, MPI, a socket); dyn_comm_create is collective (ncclCommInitRank). One communicator per process and GPU. */
#define DYN_COMM_ID_BYTES 128
typedef struct dyn_comm dyn_comm;
int dyn_comm_unique_id(uint8_t* id_out128, char* err, uint64_t errcap);
int dyn_comm_create(const uint8_t* id128, int rank, int n_ranks, int device, dyn_comm** out, char* err, uint64_t errcap);
void dyn_comm_destroy(dyn_comm* c);
const char* dyn_comm_last_error(const dyn_comm* c);
/* The gather of config 4 is two collectives: the 8-byte count exchange, then the rows.
 * dyn_comm_gather_counts: collective. `b` = an aligned batch or a ticket of dyn_batch_align[_raw]_async (waited for
 * here). counts_out[r] (n_ranks entries, may be NULL) = the dyn_segment_row records rank r will send
 * (dyn_segment_capacity() of its batch). A rank whose batch FAILED announces 0 rows, takes part in both collectives all
 * the same -- its peers never block on it -- and gets its batch's error code back from both calls.
 * dyn_comm_gather_rows: collective. The rows travel device to device to `root`: one ncclSend per peer, each over its own
 * xGMI link, no padding. On root rows_out (host, may be NULL; rows_cap records) receives them back to back in rank order
 * (read i of a rank at seg_offsets[i] as in dyn_align_out). Called without a preceding dyn_comm_gather_counts it performs
 * the count exchange itself; a root that cannot know the total in advance calls dyn_comm_gather_counts first and
 * allocates sum(counts). rows_cap < total on root: the exchange completes on every rank (nothing hangs), the rows are
 * dropped and root gets DYN_ERR_INVALID_ARGUMENT.
 * A HIP / RCCL failure in the middle of an exchange aborts the communicator (ncclCommAbort), and so does an exchange that is
 * not complete after DYN_COMM_TIMEOUT_S seconds (default 300; a peer that died or never arrived -- RCCL's kernels wait for
 * their peers on the device, so the wait is a polled one): the call fails with DYN_ERR_DEVICE, every later call on the
 * handle returns DYN_ERR_DEVICE at once. */
int dyn_comm_gather_counts(dyn_comm* c, dyn_batch* b, uint64_t* counts_out);
int dyn_comm_gather_rows(dyn_comm* c, dyn_batch* b, int root, dyn_segment_row* rows_out, uint64_t rows_cap,
                         uint64_t* counts_out);
/* Collective. `b` = a trained batch or a ticket of dyn_batch_train[_raw]_async. Sum over ranks of the device-resident
 * pooled statistics (dyn_batch_device_pooled; reduced in place), copied to pooled3n (3 * num_kmers doubles, may be NULL). */
int dyn_comm_allreduce_pooled(dyn_comm* c, dyn_batch* b, double* pooled3n);
/* (ABI 7, additive) The same exchange for payloads that live in HOST memory -- what the multi-rank CLIs move: a rank's part
 * of the compressed output frame and its `.errors` lines to rank 0 (the reference's listener collects every worker's rows,
 * src/dynamont/segmentation/segment.py:296-325,69-107), the training loop's per-batch sums (train.py:195-220 over all ranks).
 * dyn_comm_gather_bytes: collective. n_bytes of `bytes` from every rank to `root` through the code path of
 * dyn_comm_gather_rows (8-byte count all-gather, then one ncclSend per peer / ncclRecv per peer on the root in one group;
 * the payload is staged on the device first). counts_out[r] (n_ranks entries, may be NULL) = rank r's n_bytes, on every rank.
 * On root the gathered bytes stay in the communicator's device buffer until the next gather:
 * dyn_comm_gathered_bytes copies them out, back to back in rank order (out_cap >= sum of the counts; not a collective).
 * dyn_comm_allreduce_f64: collective. Elementwise over ranks, in place on the caller's n doubles; op 0 = sum, 1 = max.
 * Bounded like every exchange here: DYN_COMM_TIMEOUT_S (default 300) seconds or an asynchronous RCCL error abort the
 * communicator and fail the call -- a peer that has gone costs its peers an error, not a hang. */
int dyn_comm_gather_bytes(dyn_comm* c, const void* bytes, uint64_t n_bytes, int root, uint64_t* counts_out);
int dyn_comm_gathered_bytes(dyn_comm* c, void* out, uint64_t out_cap);
int dyn_comm_allreduce_f64(dyn_comm* c, double* inout, uint64_t n, int op);

#ifdef __cplusplus
}
#endif
#endif /* DYNAMONT_MI_H */
