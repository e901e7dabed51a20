// pore_model.cpp -- see pore_model.hpp for the reference lines each routine follows.
#include "pore_model.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

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

double parse_field(const std::string& s) {
  const char* b = s.c_str();
  char* end = nullptr;
  const double v = std::strtod(b, &end);
  if (end == b) throw std::invalid_argument("stod");
  return v;
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

  std::ifstream in(path);
  if (!in) throw std::runtime_error("Could not open model file, please prove a valid model path " + path);

  // Read every data line once; the reference makes two passes over the file, the observable
  // behaviour (error precedence: k-mer length check over the whole file first) is kept.
  struct Row { std::string kmer, mean, stdev; };
  std::vector<Row> rows;
  std::string line;
  std::getline(in, line);  // header
  bool seen[256] = {false};
  while (std::getline(in, line)) {
    Row r;
    size_t a = line.find('\t');
    r.kmer = line.substr(0, a);
    if (r.kmer.size() != (size_t)k) throw std::runtime_error("Inconsistent kmer size in model");
    for (char c : r.kmer) seen[(unsigned char)c] = true;
    if (a != std::string::npos) {
      size_t b = line.find('\t', a + 1);
      r.mean = line.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1);
      if (b != std::string::npos) {
        size_t c = line.find('\t', b + 1);
        r.stdev = line.substr(b + 1, c == std::string::npos ? std::string::npos : c - b - 1);
      }
    }
    rows.push_back(std::move(r));
  }
  alphabet = 0;
  for (bool s : seen) alphabet += s ? 1 : 0;
  num_kmers = (uint64_t)std::pow((double)alphabet, (double)k);
  mean.assign(num_kmers, 0.0);
  stdev.assign(num_kmers, 0.0);
  for (Row& r : rows) {
    std::string key = r.kmer;
    if (rna) key.assign(r.kmer.rbegin(), r.kmer.rend());  // 5'->3' file, 3'->5' sequencing
    uint64_t code = 0;
    for (char c : key) {
      const int dgt = base_digit[(unsigned char)c];
      if (dgt < 0 || dgt >= alphabet) throw std::runtime_error("Invalid nucleotide in k-mer: " + key);
      code = code * (uint64_t)alphabet + (uint64_t)dgt;
    }
    mean[code] = parse_field(r.mean);
    stdev[code] = parse_field(r.stdev);
  }
  highest_power = 1;
  for (int i = 1; i < k; ++i) highest_power *= (uint64_t)alphabet;

  table.resize(num_kmers);
  for (uint64_t i = 0; i < num_kmers; ++i) {
    // std::log: the same libm call the reference makes per cell (aligner.cpp:291)
    table[i] = dynmath::make_emis(mean[i], stdev[i], std::log(stdev[i]));
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
