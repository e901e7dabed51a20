// bam_reader.cpp -- the basecalls of a run, read in batches: the native counterpart of the reference's generate_jobs.
//
// The reference walks the basecalls with pysam (htslib) and yields one job per read
// (src/dynamont/segmentation/segment.py:189-258): the `qs` filter, `pi` parent id, `start = sp + ts`, `end = sp + ns`,
// the raw file `fn` or `f5`, the normalisation tags `sm` / `sd`, and, in the worker (segment.py:141-158), the RNA
// orientation (basecall reversed, polyA pad in front unless present). pysam is native code; the vendor-free Python
// parser of this package (bam_io.py) costs 24 us per read, which with the POD5 lookup made the front end of the CLI
// as slow as the GPU. This file does the same walk in C++: BGZF blocks (concatenated gzip members, SAM spec 4.1) are
// inflated a window at a time on a few threads, one window ahead of the parser, and a call returns COLUMNS for up to
// `max_reads` jobs -- packed names, sequences in aligner orientation, numeric tags as arrays, the signal id as the 16
// bytes of its UUID -- so that the caller's per-read work disappears.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <future>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/dynamont_mi.h"

namespace {

void set_err(char* err, uint64_t errlen, const std::string& msg) {
  if (err && errlen) {
    std::snprintf(err, errlen, "%s", msg.c_str());
  }
}

struct Window {
  std::vector<uint8_t> data;  // inflated bytes of a run of BGZF blocks
  std::string error;
};

inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct Block {
  size_t src, csize;  // deflate payload within the file
  size_t dst;         // offset in the window
  uint32_t isize, crc;
};

}  // namespace

struct dyn_bam_reader {
  std::string path;
  int fd = -1;
  const uint8_t* map = nullptr;
  size_t size = 0;
  int threads = 4;
  std::string pad;  // polyA pad of the RNA orientation (segment.py:151-153)

  std::vector<uint8_t> buf;  // inflated stream not yet consumed
  size_t bpos = 0;
  bool header_done = false;
  static constexpr size_t WINDOW_BLOCKS = 128;  // <= 8 MiB inflated per window: the first batch does not wait for more
  static constexpr size_t WINDOWS_AHEAD = 4;
  size_t scan_pos = 0;
  bool scan_eof = false;
  std::deque<std::future<Window>> ahead;

  uint64_t kept = 0;     // jobs that passed the quality filter so far (the index the rank sharding counts)
  uint64_t skipped = 0;  // "Skipped reads due to low quality"

  // columns of the last batch
  std::string names, seqs, sids, files;
  std::vector<uint64_t> name_off, seq_off, sid_off, file_off;
  std::vector<double> sm, sd;
  std::vector<int64_t> start, end;
  std::vector<uint32_t> file_id, bases;
  std::vector<uint8_t> sid16, sid_ok;
  std::unordered_map<std::string, uint32_t> file_ids;
  std::string tmp;

  ~dyn_bam_reader() {
    for (auto& f : ahead) f.wait();
    if (map) munmap(const_cast<uint8_t*>(map), size);
    if (fd >= 0) ::close(fd);
  }

  // Headers of the next run of BGZF blocks, walked serially (cheap): where each payload lies and where it inflates to.
  struct Run {
    std::vector<Block> blocks;
    size_t total = 0;
    std::string error;
  };
  Run scan_run() {
    Run run;
    size_t pos = scan_pos;
    while (run.blocks.size() < WINDOW_BLOCKS && pos < size) {
      if (size - pos < 18) {
        run.error = path + ": truncated BGZF block header";
        break;
      }
      const uint8_t* h = map + pos;
      if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) {
        run.error = path + ": not a BGZF block (gzip member without the BC extra field)";
        break;
      }
      const size_t xlen = le16(h + 10);
      if (size - pos < 12 + xlen) {
        run.error = path + ": truncated BGZF block header";
        break;
      }
      size_t bsize = 0;
      for (size_t x = 0; x + 4 <= xlen;) {
        const uint8_t* f = h + 12 + x;
        const size_t slen = le16(f + 2);
        if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (size_t)le16(f + 4) + 1;
        x += 4 + slen;
      }
      if (bsize < 12 + xlen + 8 || size - pos < bsize) {
        run.error = path + ": truncated BGZF block";
        break;
      }
      Block b;
      b.src = pos + 12 + xlen;
      b.csize = bsize - 12 - xlen - 8;
      b.crc = le32(map + pos + bsize - 8);
      b.isize = le32(map + pos + bsize - 4);
      b.dst = run.total;
      // SAM specification 4.1 caps a block's data at 64 KiB; a gzip member that inflates to more is still read, within what
      // deflate can make of a member of at most 64 KiB (1 032 : 1) -- and a window never asks for more than 256 MiB at once,
      // so that a damaged ISIZE field cannot demand gigabytes (128 blocks x 1 GiB before round 5)
      if (b.isize > (68u << 20)) {
        run.error = path + ": implausible BGZF block size";
        break;
      }
      if (!run.blocks.empty() && run.total + b.isize > ((size_t)256 << 20)) break;  // the next window starts with this block
      run.total += b.isize;
      run.blocks.push_back(b);
      pos += bsize;
    }
    scan_pos = pos;
    if (pos >= size || !run.error.empty()) scan_eof = true;
    return run;
  }

  // The payloads of a run, inflated on up to `threads` threads, each block CRC-checked.
  Window inflate_run(const Run& run) const {
    Window w;
    if (!run.error.empty()) {
      w.error = run.error;
      return w;
    }
    const std::vector<Block>& blocks = run.blocks;
    w.data.resize(run.total);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    auto work = [&]() {
      z_stream z;
      std::memset(&z, 0, sizeof z);
      if (inflateInit2(&z, -15) != Z_OK) {
        bad = true;
        return;
      }
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= blocks.size() || bad.load()) break;
        const Block& b = blocks[i];
        if (b.isize == 0) continue;  // the EOF marker, or an empty block
        inflateReset(&z);
        z.next_in = const_cast<Bytef*>(map + b.src);
        z.avail_in = (uInt)b.csize;
        z.next_out = w.data.data() + b.dst;
        z.avail_out = b.isize;
        const int rc = inflate(&z, Z_FINISH);
        if (rc != Z_STREAM_END || z.avail_out != 0 || crc32(crc32(0L, Z_NULL, 0), w.data.data() + b.dst, b.isize) != b.crc) bad = true;
      }
      inflateEnd(&z);
    };
    const int nt = (int)std::min<size_t>((size_t)std::max(1, threads / 2), std::max<size_t>(1, blocks.size() / 8));  // several windows are in flight
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (bad) w.error = path + ": corrupt BGZF block (inflate or CRC)";
    return w;
  }

  // keep WINDOWS_AHEAD runs inflating behind the parser's back
  void top_up() {
    while (ahead.size() < WINDOWS_AHEAD && !scan_eof) {
      auto run = std::make_shared<Run>(scan_run());
      ahead.push_back(std::async(std::launch::async, [this, run]() { return inflate_run(*run); }));
    }
  }

  // Make at least `need` unconsumed bytes available; false at the end of the stream (or on error: `err` set).
  bool fill(size_t need, std::string& err) {
    while (buf.size() - bpos < need) {
      top_up();
      if (ahead.empty()) return false;
      Window w = ahead.front().get();
      ahead.pop_front();
      if (!w.error.empty()) {
        err = w.error;
        scan_eof = true;
        for (auto& f : ahead) f.wait();
        ahead.clear();
        return false;
      }
      top_up();
      if (bpos == buf.size()) {
        buf.swap(w.data);
        bpos = 0;
      } else {
        buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)bpos);
        bpos = 0;
        buf.insert(buf.end(), w.data.begin(), w.data.end());
      }
    }
    return true;
  }

  bool read_header(std::string& err) {
    if (!fill(12, err)) {
      if (err.empty()) err = path + ": not a BAM file";
      return false;
    }
    if (std::memcmp(buf.data() + bpos, "BAM\1", 4) != 0) {
      err = path + ": not a BAM file";
      return false;
    }
    const size_t l_text = le32(buf.data() + bpos + 4);
    if (!fill(12 + l_text, err)) {
      if (err.empty()) err = path + ": truncated BAM header";
      return false;
    }
    bpos += 8 + l_text;
    const uint32_t n_ref = le32(buf.data() + bpos);
    bpos += 4;
    for (uint32_t r = 0; r < n_ref; ++r) {
      if (!fill(4, err)) {
        if (err.empty()) err = path + ": truncated BAM header";
        return false;
      }
      const size_t l_name = le32(buf.data() + bpos);
      if (!fill(8 + l_name, err)) {
        if (err.empty()) err = path + ": truncated BAM header";
        return false;
      }
      bpos += 8 + l_name;
    }
    header_done = true;
    return true;
  }
};

namespace {

struct Tags {
  bool has_qs = false, has_pi = false, has_ns = false, has_ts = false, has_sp = false, has_fn = false, has_f5 = false, has_sm = false, has_sd = false;
  double qs = 0, sm = 0, sd = 0;
  int64_t ns = 0, ts = 0, sp = 0;
  const char *pi = nullptr, *fn = nullptr, *f5 = nullptr;
};

inline int hexval(unsigned char c) {
  if (c >= '0' && c <= '9') return c - '0';
  if (c >= 'a' && c <= 'f') return c - 'a' + 10;
  if (c >= 'A' && c <= 'F') return c - 'A' + 10;
  return -1;
}

// 32 hex digits with hyphens anywhere between them (what uuid.UUID() makes of a plain id) -> 16 bytes
bool parse_uuid(const char* s, size_t n, uint8_t* out) {
  int nib = 0;
  uint8_t cur = 0;
  for (size_t i = 0; i < n; ++i) {
    if (s[i] == '-') continue;
    const int v = hexval((unsigned char)s[i]);
    if (v < 0 || nib >= 32) return false;
    cur = (uint8_t)((cur << 4) | v);
    if (nib & 1) out[nib >> 1] = cur;
    ++nib;
  }
  return nib == 32;
}

bool numeric_tag(char typ, const uint8_t* p, double& d, int64_t& i) {
  switch (typ) {
    case 'c': i = (int8_t)p[0]; d = (double)i; return true;
    case 'C': i = p[0]; d = (double)i; return true;
    case 's': i = (int16_t)le16(p); d = (double)i; return true;
    case 'S': i = le16(p); d = (double)i; return true;
    case 'i': i = (int32_t)le32(p); d = (double)i; return true;
    case 'I': i = le32(p); d = (double)i; return true;
    case 'f': {
      const uint32_t u = le32(p);
      float f;
      std::memcpy(&f, &u, 4);
      d = (double)f;  // the float32-rounded value, as pysam hands it out
      // an integer view only where one exists: the conversion of inf, NaN or |f| >= 2^63 is undefined behaviour
      i = (f == f && f > -9.2e18f && f < 9.2e18f) ? (int64_t)f : 0;
      return true;
    }
    default: return false;
  }
}

int tag_size(char t) {
  switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    default: return 0;
  }
}

struct Pairs {
  char t[256][2];
  Pairs() {
    const char* d = "=ACMGRSVTWYHKDBN";
    for (int b = 0; b < 256; ++b) {
      t[b][0] = d[b >> 4];
      t[b][1] = d[b & 15];
    }
  }
};
const Pairs PAIRS;

}  // namespace

extern "C" {

static int bam_open_impl(const char* path, int threads, const char* rna_pad, dyn_bam_reader** out, char* err, uint64_t errlen) {
  if (!path || !out) {
    set_err(err, errlen, "dyn_bam_open: null argument");
    return DYN_ERR_INVALID_ARGUMENT;
  }
  std::unique_ptr<dyn_bam_reader> holder(new dyn_bam_reader());  // freed on every early return and on a throw
  dyn_bam_reader* r = holder.get();
  r->path = path;
  r->threads = threads > 0 ? threads : 4;
  r->pad = rna_pad ? rna_pad : "";
  r->fd = ::open(path, O_RDONLY);
  struct stat st;
  if (r->fd < 0 || fstat(r->fd, &st) != 0) {
    set_err(err, errlen, std::string(path) + ": cannot open");
    return DYN_ERR_RUNTIME;
  }
  r->size = (size_t)st.st_size;
  if (r->size) {
    void* m = mmap(nullptr, r->size, PROT_READ, MAP_PRIVATE, r->fd, 0);
    if (m == MAP_FAILED) {
      set_err(err, errlen, std::string(path) + ": cannot map");
      return DYN_ERR_RUNTIME;
    }
    r->map = static_cast<const uint8_t*>(m);
    madvise(m, r->size, MADV_SEQUENTIAL);
  }
  r->top_up();
  std::string e;
  if (!r->read_header(e)) {
    set_err(err, errlen, e);
    return DYN_ERR_RUNTIME;
  }
  *out = holder.release();
  return DYN_OK;
}

void dyn_bam_close(dyn_bam_reader* r) { delete r; }

uint64_t dyn_bam_skipped(const dyn_bam_reader* r) { return r ? r->skipped : 0; }

static int bam_next_impl(dyn_bam_reader* r, uint64_t max_reads, uint32_t flags, double min_qual, uint32_t rank, uint32_t world, dyn_job_batch* out, char* err,
                         uint64_t errlen) {
  if (!r || !out) {
    set_err(err, errlen, "dyn_bam_next: null argument");
    return DYN_ERR_INVALID_ARGUMENT;
  }
  if (world == 0) world = 1;
  const bool rna = flags & DYN_JOBS_RNA;
  r->names.clear(); r->seqs.clear(); r->sids.clear(); r->files.clear();
  r->name_off.assign(1, 0); r->seq_off.assign(1, 0); r->sid_off.assign(1, 0); r->file_off.assign(1, 0);
  r->sm.clear(); r->sd.clear(); r->start.clear(); r->end.clear(); r->file_id.clear(); r->bases.clear(); r->sid16.clear(); r->sid_ok.clear();
  r->file_ids.clear();
  uint64_t n = 0;
  std::string e;
  while (n < max_reads) {
    if (!r->fill(4, e)) {
      if (!e.empty()) {
        set_err(err, errlen, e);
        return DYN_ERR_RUNTIME;
      }
      if (r->buf.size() != r->bpos) {
        set_err(err, errlen, r->path + ": truncated BAM record");
        return DYN_ERR_RUNTIME;
      }
      break;  // end of file
    }
    const size_t block_size = le32(r->buf.data() + r->bpos);
    if (block_size < 32 || !r->fill(4 + block_size, e)) {
      set_err(err, errlen, e.empty() ? r->path + ": truncated BAM record" : e);
      return DYN_ERR_RUNTIME;
    }
    const uint8_t* d = r->buf.data() + r->bpos + 4;
    const uint8_t* dend = d + block_size;
    r->bpos += 4 + block_size;
    const size_t l_read_name = d[8];
    const size_t n_cigar = le16(d + 12);
    const size_t l_seq = le32(d + 16);
    const uint8_t* q = d + 32;
    const size_t nb = (l_seq + 1) / 2;
    if (l_read_name == 0 || (size_t)(dend - q) < l_read_name + 4 * n_cigar + nb + l_seq) {
      set_err(err, errlen, r->path + ": malformed BAM record");
      return DYN_ERR_RUNTIME;
    }
    const char* name = reinterpret_cast<const char*>(q);
    const size_t name_len = strnlen(name, l_read_name - 1);
    q += l_read_name + 4 * n_cigar;
    const uint8_t* seq = q;
    q += nb + l_seq;
    Tags t;
    while (q + 3 <= dend) {
      const char a = (char)q[0], b = (char)q[1], typ = (char)q[2];
      q += 3;
      const uint8_t* val = q;
      if (const int s = tag_size(typ)) {
        if (q + s > dend) { q = dend + 1; break; }
        q += s;
      } else if (typ == 'Z' || typ == 'H') {
        const void* z = std::memchr(q, 0, (size_t)(dend - q));
        if (!z) { q = dend + 1; break; }
        q = static_cast<const uint8_t*>(z) + 1;
      } else if (typ == 'B') {
        if (q + 5 > dend) { q = dend + 1; break; }
        const int s = tag_size((char)q[0]);
        const size_t cnt = le32(q + 1);
        if (!s || (size_t)(dend - q - 5) < (size_t)s * cnt) { q = dend + 1; break; }
        q += 5 + (size_t)s * cnt;
      } else {
        set_err(err, errlen, r->path + ": unknown BAM tag type '" + std::string(1, typ) + "'");
        return DYN_ERR_RUNTIME;
      }
      double dv;
      int64_t iv;
#define NUM(A, B, HAS, FIELD_D, FIELD_I)                                  \
  if (a == A && b == B && numeric_tag(typ, val, dv, iv)) {               \
    t.HAS = true;                                                         \
    FIELD_D;                                                              \
    FIELD_I;                                                              \
    continue;                                                             \
  }
      NUM('q', 's', has_qs, t.qs = dv, (void)0)
      NUM('s', 'm', has_sm, t.sm = dv, (void)0)
      NUM('s', 'd', has_sd, t.sd = dv, (void)0)
      NUM('n', 's', has_ns, (void)0, t.ns = iv)
      NUM('t', 's', has_ts, (void)0, t.ts = iv)
      NUM('s', 'p', has_sp, (void)0, t.sp = iv)
#undef NUM
      if (typ == 'Z') {
        if (a == 'p' && b == 'i') { t.has_pi = true; t.pi = reinterpret_cast<const char*>(val); }
        else if (a == 'f' && b == 'n') { t.has_fn = true; t.fn = reinterpret_cast<const char*>(val); }
        else if (a == 'f' && b == '5') { t.has_f5 = true; t.f5 = reinterpret_cast<const char*>(val); }
      }
    }
    if (q != dend) {
      set_err(err, errlen, r->path + ": malformed tags in BAM record '" + std::string(name, name_len) + "'");
      return DYN_ERR_RUNTIME;
    }
    // generate_jobs (segment.py:222-245): a tag the reference reads without asking is a KeyError there
    const char* missing = !t.has_qs ? "qs" : !t.has_ns ? "ns" : !t.has_ts ? "ts" : !(t.has_fn || t.has_f5) ? "f5" : !t.has_sm ? "sm" : !t.has_sd ? "sd" : nullptr;
    if (!t.has_qs) {
      set_err(err, errlen, std::string("tag 'qs' not present"));
      return DYN_ERR_INVALID_ARGUMENT;
    }
    if (min_qual != 0.0 && t.qs < min_qual) {
      ++r->skipped;
      continue;
    }
    if (missing) {
      set_err(err, errlen, std::string("tag '") + missing + "' not present");
      return DYN_ERR_INVALID_ARGUMENT;
    }
    const uint64_t idx = r->kept++;
    if (idx % world != rank) continue;
    // columns
    r->names.append(name, name_len);
    r->names.push_back('\0');
    r->name_off.push_back(r->names.size());
    const char* sid = t.has_pi ? t.pi : name;
    const size_t sid_len = t.has_pi ? std::strlen(t.pi) : name_len;
    r->sids.append(sid, sid_len);
    r->sids.push_back('\0');
    r->sid_off.push_back(r->sids.size());
    r->sid16.resize(r->sid16.size() + 16, 0);
    r->sid_ok.push_back(parse_uuid(sid, sid_len, r->sid16.data() + r->sid16.size() - 16) ? 1 : 0);
    const char* fname = t.has_fn ? t.fn : t.f5;
    auto it = r->file_ids.find(fname);
    if (it == r->file_ids.end()) {
      it = r->file_ids.emplace(fname, (uint32_t)r->file_ids.size()).first;
      r->files.append(fname);
      r->files.push_back('\0');
      r->file_off.push_back(r->files.size());
    }
    r->file_id.push_back(it->second);
    r->sm.push_back(t.sm);
    r->sd.push_back(t.sd);
    r->start.push_back(t.sp + t.ts);
    r->end.push_back(t.sp + t.ns);
    r->bases.push_back((uint32_t)l_seq);
    // sequence: two bases per byte, high nibble first (a 256-entry table of base pairs); RNA: reversed, the pad in front
    // unless the read brings it
    r->tmp.resize(2 * nb + 2);
    for (size_t k = 0; k < nb; ++k) std::memcpy(&r->tmp[2 * k], PAIRS.t[seq[k]], 2);
    if (rna) {
      bool padded = l_seq >= r->pad.size();
      for (size_t k = 0; padded && k < r->pad.size(); ++k) padded = r->tmp[l_seq - 1 - k] == r->pad[k];
      if (!padded) r->seqs.append(r->pad);
      const size_t o = r->seqs.size();
      r->seqs.resize(o + l_seq);
      std::reverse_copy(r->tmp.begin(), r->tmp.begin() + (ptrdiff_t)l_seq, r->seqs.begin() + (ptrdiff_t)o);
    } else {
      r->seqs.append(r->tmp.data(), l_seq);
    }
    r->seq_off.push_back(r->seqs.size());
    ++n;
  }
  out->n = n;
  out->names = r->names.data();
  out->name_off = r->name_off.data();
  out->seqs = r->seqs.data();
  out->seq_off = r->seq_off.data();
  out->signal_ids = r->sids.data();
  out->signal_id_off = r->sid_off.data();
  out->signal_uuid = r->sid16.data();
  out->signal_uuid_ok = r->sid_ok.data();
  out->n_files = r->file_ids.size();
  out->files = r->files.data();
  out->file_off = r->file_off.data();
  out->file_id = r->file_id.data();
  out->bases = r->bases.data();
  out->shift = r->sm.data();
  out->scale = r->sd.data();
  out->start = r->start.data();
  out->end = r->end.data();
  out->names_bytes = r->names.size();
  out->seqs_bytes = r->seqs.size();
  out->signal_ids_bytes = r->sids.size();
  out->files_bytes = r->files.size();
  return DYN_OK;
}

// No exception crosses the C boundary: a damaged file can make a window's allocation or a worker thread's start fail
// (std::bad_alloc, std::system_error rethrown by future::get()); the caller gets DYN_ERR_RUNTIME and a message.
int dyn_bam_open(const char* path, int threads, const char* rna_pad, dyn_bam_reader** out, char* err, uint64_t errlen) {
  try {
    return bam_open_impl(path, threads, rna_pad, out, err, errlen);
  } catch (const std::exception& e) {
    set_err(err, errlen, std::string(path ? path : "?") + ": " + e.what());
  } catch (...) {
    set_err(err, errlen, std::string(path ? path : "?") + ": unknown failure while opening");
  }
  if (out) *out = nullptr;
  return DYN_ERR_RUNTIME;
}

int dyn_bam_next(dyn_bam_reader* r, uint64_t max_reads, uint32_t flags, double min_qual, uint32_t rank, uint32_t world, dyn_job_batch* out, char* err,
                 uint64_t errlen) {
  try {
    return bam_next_impl(r, max_reads, flags, min_qual, rank, world, out, err, errlen);
  } catch (const std::exception& e) {
    set_err(err, errlen, std::string(r ? r->path : "?") + ": " + e.what());
  } catch (...) {
    set_err(err, errlen, std::string(r ? r->path : "?") + ": unknown failure while reading");
  }
  return DYN_ERR_RUNTIME;
}

}  // extern "C"
