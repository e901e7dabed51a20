// pore_model.cpp -- see pore_model.hpp for the reference lines each routine follows.
#include "pore_model.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cstdint>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../include/dynamont_mi.h"

namespace dynhost {

int pore_from_string(const std::string& s) {
  static const struct { const char* name; int id; } kPores[] = {
      {"rna002", DYN_PORE_RNA002},          {"rna004", DYN_PORE_RNA004},
      {"dna_r9", DYN_PORE_DNA_R9},          {"dna_r10_260bps", DYN_PORE_DNA_R10_260},
      {"dna_r10_400bps", DYN_PORE_DNA_R10_400}};
  for (const auto& p : kPores)
    if (s == p.name) return p.id;
  throw std::invalid_argument("Unknown pore type: " + s);
}

namespace {

struct PoreDefaults { bool rna; int k; double m1, e2; };

// (rna, k): aligner.cpp:62-86; (m1, e2): NT_aligner_api.cpp:36-82; e1 = 1 always.
const PoreDefaults& defaults_for(int pore) {
  static const PoreDefaults kTable[5] = {
      {true, 5, 0.019889650396799997, 0.9801103496029998},
      {true, 9, 0.031111753637096777, 0.9688882463622581},
      {false, 5, 1.0, 1.0},
      {false, 9, 0.031111753637096777, 0.9688882463622581},
      {false, 9, 0.031111753637096777, 0.9688882463622581}};
  if (pore < 0 || pore > 4) throw std::runtime_error("Unknown pore type");
  return kTable[pore];
}

}  // namespace

void PoreModel::load(const std::string& path, int pore_id, uint64_t band) {
  const PoreDefaults& d = defaults_for(pore_id);
  pore = pore_id;
  rna = d.rna;
  k = d.k;
  half_band = band / 2;
  log_m1 = std::log(d.m1);
  log_e1 = std::log(1.0);
  log_e2 = std::log(d.e2);

  std::memset(base_digit, -1, sizeof base_digit);
  const char* letters = "AaCcGgTtUuNn";
  const int digit_of[] = {0, 0, 1, 1, 2, 2, 3, 3, 3, 3, 4, 4};
  for (int i = 0; letters[i]; ++i) base_digit[(unsigned char)letters[i]] = (int8_t)digit_of[i];

  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("Could not open model file, please prove a valid model path " + path);
  // The whole file in one read (a 9-mer table is 262 145 lines, 12 MB): line by line through an ifstream, with a
  // std::string per field, the load took 0.3 s -- a sixth of a 32 768-read dynamont-resquiggle run.
  std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  in.close();
  // line table; the header line is skipped (aligner.cpp:97-99), a last line without '\n' counts, "\r\n" is tolerated
  // the way std::getline + substr tolerate it (the '\r' stays with the last field, where strtod ignores it)
  struct Line { const char* b; const char* e; };
  std::vector<Line> lines;
  {
    const char* p = text.data();
    const char* const end = p + text.size();
    bool header = true;
    while (p < end) {
      const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(end - p)));
      const char* le = nl ? nl : end;
      if (!header) lines.push_back(Line{p, le});
      header = false;
      p = nl ? nl + 1 : end;
    }
  }
  // Pass 1, in file order, as the reference's first pass (aligner.cpp:100-119): every k-mer has length k -- this check
  // takes precedence over every other complaint about the file -- and the alphabet is what the k-mers are made of.
  bool seen[256] = {false};
  for (const Line& ln : lines) {
    const char* tab = static_cast<const char*>(std::memchr(ln.b, '\t', (size_t)(ln.e - ln.b)));
    const size_t klen = (size_t)((tab ? tab : ln.e) - ln.b);
    if (klen != (size_t)k) throw std::runtime_error("Inconsistent kmer size in model");
    for (size_t i = 0; i < klen; ++i) seen[(unsigned char)ln.b[i]] = true;
  }
  alphabet = 0;
  for (bool sn : seen) alphabet += sn ? 1 : 0;
  num_kmers = (uint64_t)std::pow((double)alphabet, (double)k);
  mean.assign(num_kmers, 0.0);
  stdev.assign(num_kmers, 0.0);
  // Pass 2 on several threads: k-mer code (reversed for RNA, aligner.cpp:136-139) and the two numbers of every line. The
  // first offending line IN FILE ORDER decides which error is raised, as in the reference's serial loop; values are
  // scattered into the table serially afterwards, so that a k-mer listed twice keeps its LAST line, as there.
  const size_t n_lines = lines.size();
  std::vector<uint64_t> codes(n_lines);
  std::vector<double> means(n_lines), sds(n_lines);
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)(hw ? hw : 1), n_lines / 4096 + 1}));
  std::vector<size_t> bad_line(n_threads, SIZE_MAX);
  std::vector<int> bad_kind(n_threads, 0);  // 1: invalid nucleotide, 2: a field that is not a number
  std::vector<std::string> bad_text(n_threads);
  auto parse_range = [&](size_t t) {
    const size_t lo = n_lines * t / n_threads, hi = n_lines * (t + 1) / n_threads;
    char buf[64];
    auto number = [&](const char* fb, const char* fe, double* out) {  // strtod on a bounded copy (fields are not terminated)
      const size_t len = std::min<size_t>((size_t)(fe - fb), sizeof buf - 1);
      std::memcpy(buf, fb, len);
      buf[len] = 0;
      char* endp = nullptr;
      *out = std::strtod(buf, &endp);
      return endp != buf;
    };
    for (size_t i = lo; i < hi; ++i) {
      const Line& ln = lines[i];
      const char* t1 = static_cast<const char*>(std::memchr(ln.b, '\t', (size_t)(ln.e - ln.b)));
      uint64_t code = 0;
      bool ok = true;
      for (int j = 0; j < k && ok; ++j) {
        const char c = rna ? ln.b[k - 1 - j] : ln.b[j];  // 5'->3' file, 3'->5' sequencing
        const int dgt = base_digit[(unsigned char)c];
        if (dgt < 0 || dgt >= alphabet) ok = false;
        else code = code * (uint64_t)alphabet + (uint64_t)dgt;
      }
      if (!ok) {
        bad_line[t] = i;
        bad_kind[t] = 1;
        bad_text[t].assign(ln.b, (size_t)k);
        if (rna) bad_text[t].assign(bad_text[t].rbegin(), bad_text[t].rend());
        return;
      }
      // fields 2 and 3; a missing field is an empty string, which strtod rejects like the reference's std::stod
      const char* f2b = t1 ? t1 + 1 : ln.e;
      const char* t2 = t1 ? static_cast<const char*>(std::memchr(f2b, '\t', (size_t)(ln.e - f2b))) : nullptr;
      const char* f2e = t2 ? t2 : ln.e;
      const char* f3b = t2 ? t2 + 1 : ln.e;
      const char* t3 = t2 ? static_cast<const char*>(std::memchr(f3b, '\t', (size_t)(ln.e - f3b))) : nullptr;
      const char* f3e = t3 ? t3 : ln.e;
      if (!number(f2b, f2e, &means[i]) || !number(f3b, f3e, &sds[i])) {
        bad_line[t] = i;
        bad_kind[t] = 2;
        return;
      }
      codes[i] = code;
    }
  };
  {
    std::vector<std::thread> workers;
    for (size_t t = 1; t < n_threads; ++t) workers.emplace_back(parse_range, t);
    parse_range(0);
    for (std::thread& w : workers) w.join();
  }
  for (size_t t = 0; t < n_threads; ++t) {  // ranges are in file order: the first one with a complaint holds the first bad line
    if (bad_line[t] == SIZE_MAX) continue;
    if (bad_kind[t] == 1) throw std::runtime_error("Invalid nucleotide in k-mer: " + bad_text[t]);
    throw std::invalid_argument("stod");
  }
  for (size_t i = 0; i < n_lines; ++i) {
    mean[codes[i]] = means[i];
    stdev[codes[i]] = sds[i];
  }
  highest_power = 1;
  for (int i = 1; i < k; ++i) highest_power *= (uint64_t)alphabet;

  table.resize(num_kmers);
  {
    // std::log: the same libm call the reference makes per cell (aligner.cpp:291)
    auto fill = [&](size_t t) {
      for (uint64_t i = num_kmers * t / n_threads; i < num_kmers * (t + 1) / n_threads; ++i)
        table[i] = dynmath::make_emis(mean[i], stdev[i], std::log(stdev[i]));
    };
    std::vector<std::thread> workers;
    for (size_t t = 1; t < n_threads; ++t) workers.emplace_back(fill, t);
    fill(0);
    for (std::thread& w : workers) w.join();
  }
}

int PoreModel::validate(uint64_t signal_len, uint64_t seq_len) const {
  if (signal_len < 1) return DYN_READ_SIGNAL_EMPTY;
  if (seq_len < (uint64_t)k) return DYN_READ_SEQ_SHORT;
  const uint64_t kc = seq_len - (uint64_t)k + 1;
  if (signal_len < 2 * kc) return DYN_READ_SIGNAL_SHORT;
  return DYN_READ_OK;
}

int PoreModel::encode(const char* seq, uint64_t len, int32_t* out, char* bad) const {
  // Rolling base-`alphabet` code with the reference's scan order, so the FIRST offending base
  // reported is the same one (first k bases left to right, then each newly entering base).
  const uint64_t kc = len - (uint64_t)k + 1;
  int32_t value = 0;
  for (int i = 0; i < k; ++i) {
    const int dgt = base_digit[(unsigned char)seq[i]];
    if (dgt < 0 || dgt >= alphabet) { *bad = seq[i]; return DYN_READ_INVALID_NT; }
    value = value * alphabet + dgt;
  }
  out[0] = value;
  for (uint64_t i = 1; i < kc; ++i) {
    const int left = base_digit[(unsigned char)seq[i - 1]];
    const int right = base_digit[(unsigned char)seq[i + k - 1]];
    if (right < 0 || right >= alphabet) { *bad = seq[i + k - 1]; return DYN_READ_INVALID_NT; }
    value -= (int32_t)((uint64_t)left * highest_power);
    value = value * alphabet + right;
    out[i] = value;
  }
  return DYN_READ_OK;
}

}  // namespace dynhost
