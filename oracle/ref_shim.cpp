// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// C-ABI shim over the *unmodified* reference classes so that the compiled
// reference (built by oracle/Makefile from the sources where they lie under
// /root/reference, never copied) can be driven through ctypes to
//   (a) generate the golden vectors in tests/golden/ (tests/golden/make_golden.py)
//   (b) validate oracle/nt_oracle.c
//   (c) serve as bench.py's cpu_baseline (kind "reference").
// It only calls the reference's public interface:
//   dynamont::NTAligner::NTAligner  (include/dynamont/NT_aligner_api.hpp:21-25)
//   dynamont::NTAligner::align      (include/dynamont/NT_aligner_api.hpp:27-31)
//   dynamont::NTAligner::train      (include/dynamont/NT_aligner_api.hpp:33-36)
// Exceptions are translated to a message buffer (as pybind11 does for Python,
// src/cpp/aligner_bindings.cpp:132-165).
#include <cstdint>
#include <cstring>
#include <exception>
#include <string>

#include "dynamont/NT_aligner_api.hpp"

namespace {
void set_err(char* buf, size_t cap, const char* msg) {
  if (!buf || !cap) return;
  std::strncpy(buf, msg, cap - 1);
  buf[cap - 1] = 0;
}
}  // namespace

extern "C" {

// pore: 0 RNA002, 1 RNA004, 2 DNA_R9, 3 DNA_R10_260, 4 DNA_R10_400 (aligner.hpp:26-33)
void* ref_create(const char* model, int pore, uint64_t band, char* err, uint64_t errcap) {
  try {
    return new dynamont::NTAligner(model, static_cast<dynamont::PoreType>(pore), 1, band);
  } catch (const std::exception& e) {
    set_err(err, errcap, e.what());
    return nullptr;
  }
}

void ref_destroy(void* h) { delete static_cast<dynamont::NTAligner*>(h); }

// Returns number of segments (>=0) or -1 on exception (message in err).
// Output arrays must hold at least len(seq) entries.
int64_t ref_align(void* h, const double* sig, uint64_t S, const char* seq, int calc, double* Z,
                  uint64_t* seqpos, uint64_t* sigpos, double* prob, char* state, char* err,
                  uint64_t errcap) {
  try {
    auto* a = static_cast<dynamont::NTAligner*>(h);
    dynamont::Result r = a->align(sig, S, std::string(seq), calc != 0);
    *Z = r.Z;
    for (size_t i = 0; i < r.segments.size(); ++i) {
      seqpos[i] = r.segments[i].sequencePosition;
      sigpos[i] = r.segments[i].signalPosition;
      prob[i] = r.segments[i].probability;
      state[i] = r.segments[i].state;
    }
    return static_cast<int64_t>(r.segments.size());
  } catch (const std::exception& e) {
    set_err(err, errcap, e.what());
    return -1;
  }
}

// emission: 2*numKmers doubles (mean, stdev interleaved), in k-mer-code order.
// Returns numKmers or -1 on exception. If emission==nullptr only the count, Z and
// transitions are produced.
int64_t ref_train(void* h, const double* sig, uint64_t S, const char* seq, double* Z,
                  double* trans3 /* m1,e1,e2 */, double* emission, uint64_t emcap, char* err,
                  uint64_t errcap) {
  try {
    auto* a = static_cast<dynamont::NTAligner*>(h);
    dynamont::TrainingResult r = a->train(sig, S, std::string(seq));
    *Z = r.Z;
    trans3[0] = r.transitions.m1;
    trans3[1] = r.transitions.e1;
    trans3[2] = r.transitions.e2;
    if (emission) {
      for (size_t k = 0; k < r.emissionModel.size() && 2 * k + 1 < emcap; ++k) {
        emission[2 * k] = r.emissionModel[k].mean;
        emission[2 * k + 1] = r.emissionModel[k].stdev;
      }
    }
    return static_cast<int64_t>(r.emissionModel.size());
  } catch (const std::exception& e) {
    set_err(err, errcap, e.what());
    return -1;
  }
}

}  // extern "C"
