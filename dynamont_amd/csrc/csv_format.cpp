// csv_format.cpp -- native counterpart of segmentation_to_string (reference:
// src/dynamont/segmentation/utils.py:193-232) for a whole batch, multi-threaded.
//
// Row: readid,signalid,start,end,basepos,base,motif,state,posterior_probability,polish\n
//   start   = signal_pos[i] + sig_offset            end = signal_pos[i+1] + sig_offset (last: last_index)
//   basepos = sequence_pos[i]   (RNA: len(read) - basepos - 1, AFTER base/motif were taken)
//   base    = read[basepos]     motif = read[max(0,bp-k/2) : min(len,bp+k/2+1)]  (reversed for RNA)
//   state   = 'M'               posterior = f"{p:.6f}"      polish = "NA"
// The bytes must equal Python's: "%.6f" of a double is the correctly rounded decimal in both glibc
// and CPython; the fast path (scaled integer) is used only when the value is provably not within
// 1e-6 of a rounding boundary, otherwise snprintf decides.
#include "../../include/dynamont_mi.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// two digits at a time
const char kDigits[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
    "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
    "8081828384858687888990919293949596979899";

inline char* put_uint(char* p, uint64_t v) {
  char buf[24];
  int n = 0;
  while (v >= 100) {
    const unsigned r = (unsigned)(v % 100);
    v /= 100;
    buf[n++] = kDigits[2 * r + 1];
    buf[n++] = kDigits[2 * r];
  }
  if (v >= 10) {
    buf[n++] = kDigits[2 * v + 1];
    buf[n++] = kDigits[2 * v];
  } else {
    buf[n++] = (char)('0' + v);
  }
  while (n) *p++ = buf[--n];
  return p;
}

inline char* put_int(char* p, int64_t v) {
  if (v < 0) {
    *p++ = '-';
    return put_uint(p, (uint64_t)(-v));
  }
  return put_uint(p, (uint64_t)v);
}

inline char* put_prob6(char* p, double x) {
  if (x >= 0.0 && x < 4.0e9) {
    const double scaled = x * 1e6;
    const double fl = std::floor(scaled);
    const double frac = scaled - fl;
    if (std::fabs(frac - 0.5) > 1e-6) {  // far from a tie: the scaled product rounds like the exact value
      const uint64_t q = (uint64_t)fl + (frac > 0.5 ? 1 : 0);
      p = put_uint(p, q / 1000000);
      *p++ = '.';
      unsigned f = (unsigned)(q % 1000000);
      const unsigned a = f / 10000, b = (f / 100) % 100, c = f % 100;
      *p++ = kDigits[2 * a]; *p++ = kDigits[2 * a + 1];
      *p++ = kDigits[2 * b]; *p++ = kDigits[2 * b + 1];
      *p++ = kDigits[2 * c]; *p++ = kDigits[2 * c + 1];
      return p;
    }
  }
  // Ties of the scaled product, and values no posterior can take (negative, non-finite, >= 4e9: the caller's arrays, not the
  // device's). "%.6f" of a double can be 317 characters long; a row's slot holds 48 (kRowBound), and snprintf RETURNS the
  // untruncated length -- found by tools/sanitize/fuzz_host.cpp (round 5). What does not fit is written as "%.17g" instead
  // (<= 24 characters, the value still exact): outside the reference's range of values there is no reference text to match.
  char buf[352];
  int n = std::snprintf(buf, sizeof buf, "%.6f", x);
  if (n < 0 || n > 47) n = std::snprintf(buf, sizeof buf, "%.17g", x);
  std::memcpy(p, buf, (size_t)n);
  return p + n;
}

struct Args {
  int k, rna;
  const dyn_align_out* res;
  const char* seqs;
  const uint64_t* seq_offsets;
  const char* const* readids;
  const char* const* signalids;
  const int64_t* sig_offsets;
  const int64_t* last_index;
};

// worst-case bytes of one row, excluding the two ids: 3 ints (<= 20 chars each), base, motif (<= 32),
// state, "%.6f" (<= 48 by the snprintf bound above), "NA", 9 commas, newline
constexpr uint64_t kRowBound = 3 * 20 + 1 + 32 + 1 + 48 + 2 + 9 + 1;

uint64_t read_bound(const Args& a, uint64_t i) {
  if (a.res->status[i] != DYN_READ_OK) return 0;
  return a.res->n_segments[i] * (std::strlen(a.readids[i]) + std::strlen(a.signalids[i]) + kRowBound);
}

char* format_read(const Args& a, uint64_t i, char* p) {
  const dyn_align_out& r = *a.res;
  if (r.status[i] != DYN_READ_OK) return p;
  const uint64_t n = r.n_segments[i];
  const uint64_t o = r.seg_offsets[i];
  const char* read = a.seqs + a.seq_offsets[i];
  const int64_t L = (int64_t)(a.seq_offsets[i + 1] - a.seq_offsets[i]);
  const int64_t half = a.k / 2;
  const size_t rid_len = std::strlen(a.readids[i]), sid_len = std::strlen(a.signalids[i]);
  for (uint64_t s = 0; s < n; ++s) {
    const int64_t bp = (int64_t)r.sequence_positions[o + s];
    const int64_t start = (int64_t)r.signal_positions[o + s] + a.sig_offsets[i];
    const int64_t end = s + 1 < n ? (int64_t)r.signal_positions[o + s + 1] + a.sig_offsets[i] : a.last_index[i];
    const int64_t m0 = std::max<int64_t>(0, bp - half), m1 = std::min<int64_t>(L, bp + half + 1);
    std::memcpy(p, a.readids[i], rid_len); p += rid_len;
    *p++ = ',';
    std::memcpy(p, a.signalids[i], sid_len); p += sid_len;
    *p++ = ',';
    p = put_int(p, start);
    *p++ = ',';
    p = put_int(p, end);
    *p++ = ',';
    p = put_int(p, a.rna ? L - bp - 1 : bp);
    *p++ = ',';
    *p++ = read[bp];
    *p++ = ',';
    if (a.rna) for (int64_t q = m1 - 1; q >= m0; --q) *p++ = read[q];
    else { std::memcpy(p, read + m0, (size_t)(m1 - m0)); p += m1 - m0; }
    *p++ = ',';
    *p++ = (char)(r.states ? r.states[o + s] : 'M');
    *p++ = ',';
    p = put_prob6(p, r.probabilities[o + s]);
    *p++ = ','; *p++ = 'N'; *p++ = 'A'; *p++ = '\n';
  }
  return p;
}

}  // namespace

extern "C" uint64_t dyn_format_csv_bound(const dyn_aligner* a, uint64_t n_reads, const dyn_align_out* res,
                                         const char* const* readids, const char* const* signalids) {
  if (!a || !res) return 0;
  Args args{0, 0, res, nullptr, nullptr, readids, signalids, nullptr, nullptr};
  uint64_t total = 0;
  for (uint64_t i = 0; i < n_reads; ++i) total += read_bound(args, i);
  return total;
}

extern "C" int dyn_format_csv(const dyn_aligner* a, uint64_t n_reads, const dyn_align_out* res,
                              const char* seqs, const uint64_t* seq_offsets,
                              const char* const* readids, const char* const* signalids,
                              const int64_t* sig_offsets, const int64_t* last_index, int threads,
                              char* out, uint64_t out_cap, uint64_t* row_begin, uint64_t* row_end) {
  if (!a || !res || !res->status || !res->n_segments || !res->seg_offsets || !res->sequence_positions ||
      !res->signal_positions || !res->probabilities || !out || !row_begin || !row_end)
    return DYN_ERR_INVALID_ARGUMENT;
  dyn_info info;
  dyn_aligner_info(a, &info);
  if (info.kmer_size > 31) return DYN_ERR_INVALID_ARGUMENT;
  Args args{info.kmer_size, info.rna, res, seqs, seq_offsets, readids, signalids, sig_offsets, last_index};
  // every read formats straight into its own worst-case slot of the caller's buffer: no allocation,
  // no shared cache lines between threads
  uint64_t pos = 0;
  for (uint64_t i = 0; i < n_reads; ++i) {
    row_begin[i] = pos;
    pos += read_bound(args, i);
  }
  if (pos > out_cap) return DYN_ERR_INVALID_ARGUMENT;
  const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, threads), std::max<uint64_t>(1, n_reads / 8)));
  auto work = [&](int tid) {
    const uint64_t lo = n_reads * (uint64_t)tid / (uint64_t)nt, hi = n_reads * (uint64_t)(tid + 1) / (uint64_t)nt;
    for (uint64_t i = lo; i < hi; ++i) row_end[i] = (uint64_t)(format_read(args, i, out + row_begin[i]) - out);
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
  }
  return DYN_OK;
}

// Close the gaps between the per-read slots dyn_format_csv wrote: the rows of all reads become ONE contiguous
// run at the front of `out`, in read order (what the writer thread hands to the compressor as a single blob).
// row_begin/row_end are updated; returns the total number of bytes.
extern "C" uint64_t dyn_csv_compact(char* out, uint64_t n_reads, uint64_t* row_begin, uint64_t* row_end) {
  if (!out || !row_begin || !row_end) return 0;
  uint64_t pos = 0;
  for (uint64_t i = 0; i < n_reads; ++i) {
    const uint64_t len = row_end[i] - row_begin[i];
    if (len && row_begin[i] != pos) std::memmove(out + pos, out + row_begin[i], len);
    row_begin[i] = pos;
    pos += len;
    row_end[i] = pos;
  }
  return pos;
}
