// pore_model.hpp -- pore table, k-mer model loader and read validation (host side).
//
// Behavioural contract (messages included) follows the reference:
//   pore -> (rna, k)                 src/cpp/aligner.cpp:62-86
//   default log transitions          src/cpp/NT_aligner_api.cpp:21-87
//   model TSV loader                 src/cpp/aligner.cpp:88-143
//   validateInput                    src/cpp/aligner.cpp:145-164
//   sequenceToKmers                  src/cpp/aligner.cpp:166-205
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "dp_math.hpp"

namespace dynhost {

struct PoreModel {
  int pore = 0;
  bool rna = false;
  int k = 0;
  int alphabet = 0;
  uint64_t num_kmers = 0;
  uint64_t highest_power = 1;
  uint64_t half_band = 0;
  double log_m1 = 0, log_e1 = 0, log_e2 = 0;
  std::vector<double> mean, stdev;      // k-mer-code order
  std::vector<dynmath::Emis> table;     // device image of the same
  int8_t base_digit[256];

  // Throws std::runtime_error / std::invalid_argument with the reference's texts.
  void load(const std::string& path, int pore_id, uint64_t band);

  // 0 = ok, else dyn_read_status; bad receives the offending base for "Invalid nucleotide".
  int validate(uint64_t signal_len, uint64_t seq_len) const;
  int encode(const char* seq, uint64_t len, int32_t* out, char* bad) const;
};

int pore_from_string(const std::string& s);  // throws std::invalid_argument("Unknown pore type: ...")

}  // namespace dynhost
