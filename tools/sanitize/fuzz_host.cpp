// Byte-level fuzz of the host-side parsers of untrusted files (tools/sanitize/Makefile links it against the sanitizer build
// of the library): BGZF framing + BAM records + aux tags (bam_reader.cpp, the reference's generate_jobs over htslib,
// segment.py:189-258), VBZ signal chunks (vbz_decode.cpp: zstd + StreamVByte-16 + zigzag + delta, pod5_io.py:6-16), the
// k-mer model TSV (pore_model.cpp, aligner.cpp:88-143) and the probability formatter of the CSV rows (csv_format.cpp,
// utils.py:193-232). Contract under test: a malformed input is a per-file or per-read ERROR -- never a crash, never a
// sanitizer report, never an exception across the C boundary (the reference isolates failures per read, segment.py:160-187).
//
//   fuzz_host <corpus dir> <mutations per target> [seed]
//
// Deterministic (xorshift64*, seed on the command line). Mutations: byte flips, runs of 0x00 / 0xff, truncation, block
// duplication, little-endian length fields set to boundary values. BAM is fuzzed twice: the raw file (BGZF framing, CRC,
// ISIZE) and the INFLATED payload, re-deflated into valid BGZF blocks so that the damage reaches the record parser.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../dynamont_amd/csrc/pore_model.hpp"
#include "../../include/dynamont_mi.h"

namespace {

struct Rng {
  uint64_t s;
  uint64_t next() {
    s ^= s >> 12;
    s ^= s << 25;
    s ^= s >> 27;
    return s * 0x2545F4914F6CDD1Dull;
  }
  uint64_t below(uint64_t n) { return n ? next() % n : 0; }
};

std::vector<uint8_t> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "fuzz_host: cannot read %s\n", p.c_str());
    std::exit(2);
  }
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

void spit(const std::string& p, const std::vector<uint8_t>& d) {
  FILE* f = std::fopen(p.c_str(), "wb");
  if (!f || (d.size() && std::fwrite(d.data(), 1, d.size(), f) != d.size())) {
    std::fprintf(stderr, "fuzz_host: cannot write %s\n", p.c_str());
    std::exit(2);
  }
  std::fclose(f);
}

void mutate(std::vector<uint8_t>& d, Rng& r) {
  if (d.empty()) return;
  const int edits = 1 + (int)r.below(6);
  for (int e = 0; e < edits; ++e) {
    const size_t n = d.size();
    if (!n) return;
    const size_t at = (size_t)r.below(n);
    switch (r.below(9)) {
      case 0: d[at] ^= (uint8_t)(1u << r.below(8)); break;
      case 1: d[at] = (uint8_t)r.next(); break;
      case 2: {  // a run of one value
        const size_t len = std::min<size_t>(n - at, 1 + r.below(64));
        std::memset(d.data() + at, r.below(2) ? 0xff : 0x00, len);
        break;
      }
      case 3: d.resize(at); break;  // truncation
      case 4: {  // a 32-bit little-endian field set to a boundary value
        static const uint32_t vals[] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0xfffffff0u, 65535u, 65536u, 0x01000000u};
        const uint32_t v = vals[r.below(sizeof vals / sizeof vals[0])];
        if (at + 4 <= n) std::memcpy(d.data() + at, &v, 4);
        break;
      }
      case 5: {  // duplicate a block
        const size_t len = std::min<size_t>(n - at, 1 + r.below(512));
        std::vector<uint8_t> blk(d.begin() + (ptrdiff_t)at, d.begin() + (ptrdiff_t)(at + len));
        d.insert(d.begin() + (ptrdiff_t)r.below(n), blk.begin(), blk.end());
        break;
      }
      case 6: {  // delete a block
        const size_t len = std::min<size_t>(n - at, 1 + r.below(256));
        d.erase(d.begin() + (ptrdiff_t)at, d.begin() + (ptrdiff_t)(at + len));
        break;
      }
      case 7: {  // a 16-bit field
        const uint16_t v = (uint16_t)(r.below(3) == 0 ? 0xffff : r.next());
        if (at + 2 <= n) std::memcpy(d.data() + at, &v, 2);
        break;
      }
      default: std::swap(d[at], d[(size_t)r.below(n)]); break;
    }
  }
}

// ---- BGZF (SAM specification 4.1), for the payload fuzz --------------------------------------------------------------
std::vector<uint8_t> bgzf_inflate_all(const std::vector<uint8_t>& file) {
  std::vector<uint8_t> out;
  size_t pos = 0;
  while (pos + 18 <= file.size()) {
    const size_t xlen = file[pos + 10] | (file[pos + 11] << 8);
    size_t bsize = 0;
    for (size_t x = 0; x + 4 <= xlen;) {
      const uint8_t* f = file.data() + pos + 12 + x;
      const size_t slen = f[2] | (f[3] << 8);
      if (f[0] == 'B' && f[1] == 'C' && slen == 2) bsize = (size_t)(f[4] | (f[5] << 8)) + 1;
      x += 4 + slen;
    }
    if (!bsize || pos + bsize > file.size()) break;
    uint32_t isize;
    std::memcpy(&isize, file.data() + pos + bsize - 4, 4);
    const size_t o = out.size();
    out.resize(o + isize);
    z_stream z{};
    inflateInit2(&z, -15);
    z.next_in = const_cast<Bytef*>(file.data() + pos + 12 + xlen);
    z.avail_in = (uInt)(bsize - 12 - xlen - 8);
    z.next_out = out.data() + o;
    z.avail_out = isize;
    inflate(&z, Z_FINISH);
    inflateEnd(&z);
    pos += bsize;
  }
  return out;
}

void bgzf_block(std::vector<uint8_t>& file, const uint8_t* p, size_t n) {
  std::vector<uint8_t> c(compressBound((uLong)n) + 64);
  z_stream z{};
  deflateInit2(&z, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
  z.next_in = const_cast<Bytef*>(p);
  z.avail_in = (uInt)n;
  z.next_out = c.data();
  z.avail_out = (uInt)c.size();
  deflate(&z, Z_FINISH);
  const size_t clen = c.size() - z.avail_out;
  deflateEnd(&z);
  const size_t total = 12 + 6 + clen + 8;
  const uint8_t head[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
  file.insert(file.end(), head, head + 12);
  const uint16_t bs = (uint16_t)(total - 1);
  const uint8_t extra[6] = {'B', 'C', 2, 0, (uint8_t)(bs & 255), (uint8_t)(bs >> 8)};
  file.insert(file.end(), extra, extra + 6);
  file.insert(file.end(), c.begin(), c.begin() + (ptrdiff_t)clen);
  const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n), isz = (uint32_t)n;
  uint8_t tail[8];
  std::memcpy(tail, &crc, 4);
  std::memcpy(tail + 4, &isz, 4);
  file.insert(file.end(), tail, tail + 8);
}

std::vector<uint8_t> bgzf_deflate_all(const std::vector<uint8_t>& payload, Rng& r) {
  std::vector<uint8_t> file;
  size_t pos = 0;
  while (pos < payload.size()) {
    const size_t n = std::min<size_t>(payload.size() - pos, 1 + r.below(30000));  // records straddle block borders
    bgzf_block(file, payload.data() + pos, n);
    pos += n;
  }
  bgzf_block(file, nullptr, 0);  // the EOF marker
  return file;
}

struct Tally {
  uint64_t runs = 0, ok = 0, errors = 0, reads = 0;
};

void run_bam(const std::string& path, bool rna, Tally& t) {
  dyn_bam_reader* rd = nullptr;
  char err[512];
  err[0] = 0;
  ++t.runs;
  if (dyn_bam_open(path.c_str(), 3, rna ? "AAAAAAAAA" : "", &rd, err, sizeof err) != DYN_OK) {
    if (rd) {
      std::fprintf(stderr, "fuzz_host: dyn_bam_open failed but left a reader behind\n");
      std::abort();
    }
    ++t.errors;
    return;
  }
  for (int rounds = 0; rounds < 64; ++rounds) {
    dyn_job_batch jb{};
    const int rc = dyn_bam_next(rd, 7, rna ? 1u : 0u, rounds & 1 ? 9.5 : 0.0, 0, 1, &jb, err, sizeof err);
    if (rc != DYN_OK) {
      ++t.errors;
      break;
    }
    if (!jb.n) {
      ++t.ok;
      break;
    }
    // touch what the caller would read: a column that points outside its buffer is what ASan is here to see
    uint64_t sum = 0;
    for (uint64_t i = 0; i < jb.n; ++i) {
      sum += jb.seq_off[i + 1] - jb.seq_off[i];
      sum += (uint64_t)jb.seqs[jb.seq_off[i]] + (uint64_t)jb.names[jb.name_off[i]] + (uint64_t)jb.signal_ids[jb.signal_id_off[i]];
      sum += jb.file_id[i] < jb.n_files ? (uint64_t)jb.files[jb.file_off[jb.file_id[i]]] : 0;
      sum += (uint64_t)jb.bases[i] + (uint64_t)jb.start[i] + (uint64_t)jb.end[i] + (std::isnan(jb.shift[i] + jb.scale[i]) ? 1 : 0);
    }
    t.reads += jb.n + (sum & 0);
  }
  dyn_bam_close(rd);
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: fuzz_host <corpus dir> <mutations per target> [seed]\n");
    return 2;
  }
  const std::string corpus = argv[1];
  const uint64_t iters = std::strtoull(argv[2], nullptr, 10);
  Rng rng{argc > 3 ? std::strtoull(argv[3], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1 : 0x1234567ull};
  const std::string tmp = std::string("/dev/shm/dyn_fuzz_") + std::to_string((long)getpid());
  uint64_t total = 0;

  // ---- 1. BAM: raw bytes, then the inflated payload --------------------------------------------------------------
  {
    const std::vector<uint8_t> seed = slurp(corpus + "/spec.bam");
    const std::vector<uint8_t> payload = bgzf_inflate_all(seed);
    Tally raw, pay;
    spit(tmp + ".bam", seed);
    run_bam(tmp + ".bam", true, raw);  // the seed itself must read
    if (raw.ok != 1 || raw.reads == 0) {
      std::fprintf(stderr, "fuzz_host: the seed BAM does not read (ok %llu, reads %llu)\n", (unsigned long long)raw.ok, (unsigned long long)raw.reads);
      return 1;
    }
    raw = Tally();
    for (uint64_t i = 0; i < iters; ++i) {
      std::vector<uint8_t> d = seed;
      mutate(d, rng);
      spit(tmp + ".bam", d);
      run_bam(tmp + ".bam", (i & 1) != 0, raw);
    }
    for (uint64_t i = 0; i < iters; ++i) {
      std::vector<uint8_t> d = payload;
      mutate(d, rng);
      spit(tmp + ".bam", bgzf_deflate_all(d, rng));
      run_bam(tmp + ".bam", (i & 1) != 0, pay);
    }
    std::printf("bam raw bytes      : %llu mutations, %llu files read to their end, %llu refused with an error, %llu reads handed out\n",
                (unsigned long long)raw.runs, (unsigned long long)raw.ok, (unsigned long long)raw.errors, (unsigned long long)raw.reads);
    std::printf("bam inflated payload: %llu mutations, %llu files read to their end, %llu refused with an error, %llu reads handed out\n",
                (unsigned long long)pay.runs, (unsigned long long)pay.ok, (unsigned long long)pay.errors, (unsigned long long)pay.reads);
    total += raw.runs + pay.runs;
    std::remove((tmp + ".bam").c_str());
  }

  // ---- 2. VBZ chunks ------------------------------------------------------------------------------------------------
  {
    const std::vector<uint8_t> seed = slurp(corpus + "/chunk.vbz");
    const std::vector<uint8_t> want_raw = slurp(corpus + "/chunk.i16");
    const uint32_t n = (uint32_t)(want_raw.size() / 2);
    std::vector<int16_t> out((size_t)n + 70000);
    char err[256];
    if (dyn_vbz_decode(seed.data(), seed.size(), n, out.data(), err, sizeof err) != DYN_OK || std::memcmp(out.data(), want_raw.data(), want_raw.size()) != 0) {
      std::fprintf(stderr, "fuzz_host: the seed VBZ chunk does not decode to chunk.i16 (%s)\n", err);
      return 1;
    }
    uint64_t ok = 0, bad = 0;
    for (uint64_t i = 0; i < iters; ++i) {
      std::vector<uint8_t> d = seed;
      mutate(d, rng);
      uint32_t samples = n;
      const uint64_t how = rng.below(8);
      if (how == 0) samples = (uint32_t)rng.below(70000);
      else if (how == 1) samples = n + 1;
      else if (how == 2) samples = 0;
      std::vector<int16_t> o2((size_t)samples + 1, 0x5555);
      // exact-size heap buffer: a decoder that writes one sample too many is caught by ASan
      const int rc = dyn_vbz_decode(d.data(), d.size(), samples, o2.data(), err, sizeof err);
      if (rc == DYN_OK) ++ok;
      else ++bad;
      if (o2[samples] != 0x5555) {
        std::fprintf(stderr, "fuzz_host: dyn_vbz_decode wrote past its output\n");
        std::abort();
      }
    }
    std::printf("vbz chunks          : %llu mutations, %llu decoded, %llu refused with an error\n", (unsigned long long)iters, (unsigned long long)ok,
                (unsigned long long)bad);
    total += iters;
  }

  // ---- 3. model TSV -------------------------------------------------------------------------------------------------
  {
    const std::vector<uint8_t> seed = slurp(corpus + "/model5.tsv");
    uint64_t ok = 0, bad = 0;
    const uint64_t n_it = std::max<uint64_t>(1, iters / 8);  // (a model load is ~0.3 ms)
    for (uint64_t i = 0; i < n_it; ++i) {
      std::vector<uint8_t> d = seed;
      mutate(d, rng);
      spit(tmp + ".model", d);
      try {
        dynhost::PoreModel m;
        m.load(tmp + ".model", (int)rng.below(5), 400);
        // the table is what every later stage indexes: its size must be what the header fields say
        if (m.table.size() != m.num_kmers || m.mean.size() != m.num_kmers || m.stdev.size() != m.num_kmers) std::abort();
        ++ok;
      } catch (const std::exception&) {
        ++bad;  // "Inconsistent kmer size in model", stod failures, ...: the reference throws too (aligner.cpp:88-143)
      }
      // the C entry point on the same file: an error code and a message, never an exception
      dyn_aligner* a = nullptr;
      char err[512];
      const int rc = dyn_aligner_create((tmp + ".model").c_str(), (int)rng.below(5), "basic", 1, 400, DYN_DEVICE_HOST_ONLY, &a, err, sizeof err);
      if (rc == DYN_OK) dyn_aligner_destroy(a);
    }
    std::printf("model TSV           : %llu mutations, %llu loaded, %llu refused with the reference's exceptions\n", (unsigned long long)n_it,
                (unsigned long long)ok, (unsigned long long)bad);
    total += n_it;
    std::remove((tmp + ".model").c_str());
  }

  // ---- 4. CSV rows: every double the device can hand over formats into its slot -----------------------------------
  {
    spit(tmp + ".model", slurp(corpus + "/model5.tsv"));
    dyn_aligner* a = nullptr;
    char err[512];
    if (dyn_aligner_create((tmp + ".model").c_str(), DYN_PORE_RNA002, "basic", 1, 400, DYN_DEVICE_HOST_ONLY, &a, err, sizeof err) != DYN_OK) {
      std::fprintf(stderr, "fuzz_host: %s\n", err);
      return 1;
    }
    uint64_t rows = 0;
    for (uint64_t i = 0; i < std::max<uint64_t>(1, iters / 16); ++i) {
      const uint64_t n_reads = 1 + rng.below(4);
      std::vector<uint64_t> seq_off(n_reads + 1, 0), seg_off(n_reads + 1, 0), nseg(n_reads), sig_len(n_reads);
      std::vector<int32_t> status(n_reads, 0);
      std::string seqs;
      for (uint64_t k = 0; k < n_reads; ++k) {
        const uint64_t L = 5 + rng.below(60);
        for (uint64_t j = 0; j < L; ++j) seqs.push_back("ACGT"[rng.below(4)]);
        seq_off[k + 1] = seqs.size();
        nseg[k] = L - 4;
        seg_off[k + 1] = seg_off[k] + nseg[k];
        sig_len[k] = 2 * nseg[k] + rng.below(500);
        status[k] = rng.below(7) == 0 ? DYN_READ_Z_MISMATCH : DYN_READ_OK;
      }
      const uint64_t cap = seg_off[n_reads];
      std::vector<uint64_t> sp(cap), gp(cap);
      std::vector<double> pr(cap), Z(n_reads, -1.0);
      std::vector<uint8_t> st(cap, 'M');
      for (uint64_t k = 0; k < n_reads; ++k)
        for (uint64_t j = 0; j < nseg[k]; ++j) {
          sp[seg_off[k] + j] = j + 2;
          gp[seg_off[k] + j] = std::min<uint64_t>(sig_len[k] - 1, 2 * j + rng.below(2));
          double p;
          switch (rng.below(8)) {
            case 0: p = 0.0; break;
            case 1: p = 1.0; break;
            case 2: p = 4.9e-324; break;
            case 3: p = 0.9999995; break;
            case 4: p = std::nan(""); break;
            case 5: p = 1e300; break;
            default: {
              const uint64_t bits = rng.next() & 0x3fffffffffffffffull;  // any non-negative double below 2
              std::memcpy(&p, &bits, 8);
            }
          }
          pr[seg_off[k] + j] = p;
        }
      dyn_align_out res{};
      res.Z = Z.data();
      res.status = status.data();
      res.seg_offsets = seg_off.data();
      res.n_segments = nseg.data();
      res.sequence_positions = sp.data();
      res.signal_positions = gp.data();
      res.probabilities = pr.data();
      res.states = st.data();
      res.capacity = cap;
      std::vector<std::string> ids(n_reads, "read-0000");
      std::vector<const char*> idp(n_reads);
      for (uint64_t k = 0; k < n_reads; ++k) idp[k] = ids[k].c_str();
      std::vector<int64_t> so(n_reads, 17), li(n_reads);
      for (uint64_t k = 0; k < n_reads; ++k) li[k] = 17 + (int64_t)sig_len[k];
      const uint64_t bound = dyn_format_csv_bound(a, n_reads, &res, idp.data(), idp.data());
      std::vector<char> out(bound + 1, 0x55);
      std::vector<uint64_t> rb(n_reads), re(n_reads);
      const int rc = dyn_format_csv(a, n_reads, &res, seqs.data(), seq_off.data(), idp.data(), idp.data(), so.data(), li.data(), 1 + (int)rng.below(3),
                                    out.data(), bound, rb.data(), re.data());
      if (rc != DYN_OK || out[bound] != 0x55) {
        std::fprintf(stderr, "fuzz_host: dyn_format_csv rc %d (or wrote past its bound)\n", rc);
        std::abort();
      }
      rows += cap;
    }
    dyn_aligner_destroy(a);
    std::printf("csv rows            : %llu rows formatted inside their bounds (NaN, inf-like, denormal and boundary probabilities included)\n",
                (unsigned long long)rows);
    std::remove((tmp + ".model").c_str());
  }
  // ---- 5. the sink's threads (csv_sink.cpp) without a GPU: header frame, error lines from several producers, close -------
  {
    uint64_t lines = 0;
    for (int round = 0; round < 8; ++round) {
      dyn_csv_sink* sink = nullptr;
      char err[512];
      const std::string out = tmp + ".csv.zst", errs = tmp + ".errors";
      if (dyn_csv_sink_open(out.c_str(), errs.c_str(), 3, 1 + round % 4, &sink, err, sizeof err) != DYN_OK) {
        std::fprintf(stderr, "fuzz_host: dyn_csv_sink_open: %s\n", err);
        return 1;
      }
      std::vector<std::thread> producers;
      for (int t = 0; t < 4; ++t)
        producers.emplace_back([&, t] {
          for (int k = 0; k < 200; ++k) {
            const std::string line = "error: worker, boom\tN: " + std::to_string(k) + "\tRid: r" + std::to_string(t) + "\tSid: s";
            if (dyn_csv_sink_error_line(sink, line.c_str()) != DYN_OK) std::abort();
            (void)dyn_csv_sink_completed(sink);
            (void)dyn_csv_sink_failed(sink);
          }
        });
      for (auto& th : producers) th.join();
      uint64_t csv = 0, zst = 0, nerr = 0;
      if (dyn_csv_sink_close(sink, &csv, &zst, &nerr, err, sizeof err) != DYN_OK || nerr != 800) {
        std::fprintf(stderr, "fuzz_host: dyn_csv_sink_close: %s (%llu error lines)\n", err, (unsigned long long)nerr);
        return 1;
      }
      lines += nerr;
      std::remove(out.c_str());
      std::remove(errs.c_str());
    }
    std::printf("csv sink            : 8 sinks opened, fed by four threads each and closed; %llu error lines written\n", (unsigned long long)lines);
  }
  std::printf("TOTAL %llu mutated inputs, no crash, no sanitizer report\n", (unsigned long long)total);
  return 0;
}
