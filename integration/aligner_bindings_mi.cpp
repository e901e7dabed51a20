// aligner_bindings_mi.cpp -- the reference-side binding over the C ABI of the MI355X core.
//
// This is the file a maintainer of rnajena/dynamont would compile INSTEAD of src/cpp/aligner_bindings.cpp
// (+ aligner.cpp, NT_aligner_api.cpp, NTK_aligner_api.cpp): the same pybind11 module `_dynamont` with the
// same surface (aligner_bindings.cpp:180-219) --
//     PoreType{RNA002, RNA004, DNA_R9, DNA_R10_260, DNA_R10_400}, pore_type(str),
//     Aligner(model_file, pore: PoreType|str, mode="basic", threads=1, band=400)
//         .align(signal, sequence, calc_probabilities=False) -> dict
//         .train(signal, sequence) -> dict
// -- implemented on include/dynamont_mi.h and linked with -ldynamont_mi. Exceptions translate as in the
// reference (pybind11 defaults): std::invalid_argument -> ValueError, std::runtime_error -> RuntimeError,
// with the reference's message texts. Added for the GPU: align_batch(signals, sequences, calc) -> list of
// dicts (a single read cannot fill an MI355X).
//
// Build (what tests/test_pybind_stub.py does):
//   g++ -O2 -std=c++17 -shared -fPIC $(python3 -m pybind11 --includes) -Iinclude integration/aligner_bindings_mi.cpp \
//       -Ldynamont_amd -ldynamont_mi -Wl,-rpath,$PWD/dynamont_amd -o _dynamont$(python3-config --extension-suffix)
// Device: environment variable DYNAMONT_MI_DEVICE = HIP ordinal (default: current device), or "host" for a
// handle that only serves the host-side contract (model loading, validation; compute calls raise).
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "dynamont_mi.h"

namespace py = pybind11;

namespace {

[[noreturn]] void raise(int rc, const std::string& msg) {  // pybind11 default exception translation
  if (rc == DYN_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

int pore_from_string(const std::string& s) {  // aligner_bindings.cpp:18-32
  int pore = 0;
  char err[512];
  const int rc = dyn_pore_from_string(s.c_str(), &pore, err, sizeof err);
  if (rc != DYN_OK) raise(rc, err);
  return pore;
}

int device_from_env() {
  const char* d = std::getenv("DYNAMONT_MI_DEVICE");
  if (!d || !*d) return -1;
  if (std::strcmp(d, "host") == 0) return DYN_DEVICE_HOST_ONLY;
  return std::atoi(d);
}

using Signal = py::array_t<double, py::array::c_style | py::array::forcecast>;

class PyAligner {  // aligner_bindings.cpp:109-167
 public:
  PyAligner(const std::string& model, int pore, const std::string& mode, int threads, std::size_t band) {
    char err[4096];
    const int rc = dyn_aligner_create(model.c_str(), pore, mode.c_str(), threads, band, device_from_env(), &h_, err, sizeof err);
    if (rc != DYN_OK) raise(rc, err);
    dyn_aligner_info(h_, &info_);
  }
  PyAligner(const std::string& model, const std::string& pore, const std::string& mode, int threads, std::size_t band)
      : PyAligner(model, pore_from_string(pore), mode, threads, band) {}
  PyAligner(const PyAligner&) = delete;
  PyAligner& operator=(const PyAligner&) = delete;
  ~PyAligner() { dyn_aligner_destroy(h_); }

  py::list align_batch(const std::vector<Signal>& signals, const std::vector<std::string>& seqs, bool calc) {
    const uint64_t n = signals.size();
    if (n != seqs.size()) throw std::invalid_argument("signals and sequences differ in length");
    std::vector<uint64_t> so(n + 1, 0), qo(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) {
      if (signals[i].ndim() != 1) throw std::invalid_argument("Signal must be a one-dimensional array");  // :137-138
      so[i + 1] = so[i] + (uint64_t)signals[i].shape(0);
      qo[i + 1] = qo[i] + seqs[i].size();
    }
    std::vector<double> sig(so[n] ? so[n] : 1);
    std::string flat;
    flat.reserve(qo[n]);
    for (uint64_t i = 0; i < n; ++i) {
      if (so[i + 1] > so[i]) std::memcpy(sig.data() + so[i], signals[i].data(), (so[i + 1] - so[i]) * sizeof(double));
      flat += seqs[i];
    }
    const uint64_t cap = dyn_segment_capacity(h_, n, qo.data());
    std::vector<double> Z(n), pr(cap ? cap : 1);
    std::vector<int32_t> status(n);
    std::vector<char> bad(n);
    std::vector<uint64_t> off(n + 1), nseg(n), sp(cap ? cap : 1), gp(cap ? cap : 1);
    dyn_align_out out{Z.data(), status.data(), bad.data(), off.data(), nseg.data(), sp.data(), gp.data(), pr.data(), nullptr, cap};
    int rc;
    {
      py::gil_scoped_release nogil;  // aligner_bindings.cpp:143
      rc = dyn_align_batch(h_, n, sig.data(), so.data(), flat.data(), qo.data(), calc ? 1 : 0, &out);
    }
    if (rc != DYN_OK) raise(rc, dyn_aligner_last_error(h_));
    py::list results;
    for (uint64_t i = 0; i < n; ++i) {
      if (status[i] != DYN_READ_OK) {  // a failing read raises for single-read calls, is reported in place for batches
        char m[256];
        dyn_read_strerror(status[i], bad[i], m, sizeof m);
        if (n == 1) throw std::runtime_error(m);
        py::dict d;
        d["error"] = std::string(m);
        results.append(d);
        continue;
      }
      py::dict d;  // resultToPython, aligner_bindings.cpp:53-84
      d["Z"] = Z[i];
      d["sequence_positions"] = py::array_t<std::size_t>((py::ssize_t)nseg[i], reinterpret_cast<const std::size_t*>(sp.data() + off[i]));
      d["signal_positions"] = py::array_t<std::size_t>((py::ssize_t)nseg[i], reinterpret_cast<const std::size_t*>(gp.data() + off[i]));
      d["probabilities"] = py::array_t<double>((py::ssize_t)nseg[i], pr.data() + off[i]);
      py::list states, polishes;
      for (uint64_t s = 0; s < nseg[i]; ++s) {
        states.append(py::str("M"));  // NT path: every segment starts in M (NT_aligner_api.cpp:424-430)
        polishes.append(py::str(""));
      }
      d["states"] = states;
      d["polishes"] = polishes;
      results.append(d);
    }
    return results;
  }

  py::dict align(const Signal& signal, const std::string& seq, bool calc) {  // :132-147
    return align_batch(std::vector<Signal>{signal}, std::vector<std::string>{seq}, calc)[0].cast<py::dict>();
  }

  py::dict train(const Signal& signal, const std::string& seq) {  // :149-163
    if (signal.ndim() != 1) throw std::invalid_argument("Signal must be a one-dimensional array");
    const uint64_t so[2] = {0, (uint64_t)signal.shape(0)}, qo[2] = {0, seq.size()};
    const uint64_t cap = dyn_segment_capacity(h_, 1, qo);
    double Z = 0, trans[3] = {0, 0, 0};
    int32_t status = 0;
    char bad = 0;
    uint64_t off[2] = {0, 0}, cnt = 0;
    std::vector<int32_t> code(cap ? cap : 1);
    std::vector<double> mean(cap ? cap : 1), sd(cap ? cap : 1);
    dyn_train_out out{};
    out.Z = &Z;
    out.status = &status;
    out.bad_char = &bad;
    out.transitions = trans;
    out.em_offsets = off;
    out.em_count = &cnt;
    out.em_code = code.data();
    out.em_mean = mean.data();
    out.em_stdev = sd.data();
    out.capacity = cap;
    int rc;
    {
      py::gil_scoped_release nogil;  // aligner_bindings.cpp:159
      rc = dyn_train_batch(h_, 1, signal.data(), so, seq.data(), qo, &out, nullptr);
    }
    if (rc != DYN_OK) raise(rc, dyn_aligner_last_error(h_));
    if (status != DYN_READ_OK) {
      char m[256];
      dyn_read_strerror(status, bad, m, sizeof m);
      throw std::runtime_error(m);
    }
    // trainingResultToPython, aligner_bindings.cpp:86-107: the DENSE model in k-mer-code order; k-mers without
    // observations keep the loaded model (NT_aligner_api.cpp:531-534)
    std::vector<double> model(2 * info_.num_kmers);
    dyn_aligner_model(h_, model.data());
    for (uint64_t j = 0; j < cnt; ++j) {
      model[2 * (uint64_t)code[j]] = mean[j];
      model[2 * (uint64_t)code[j] + 1] = sd[j];
    }
    py::dict transitions;
    transitions["m1"] = trans[0];
    transitions["e1"] = trans[1];
    transitions["e2"] = trans[2];
    py::list emission;
    for (uint64_t k = 0; k < info_.num_kmers; ++k) {
      py::dict item;
      item["mean"] = model[2 * k];
      item["stdev"] = model[2 * k + 1];
      emission.append(std::move(item));
    }
    py::dict d;
    d["Z"] = Z;
    d["transition_params"] = std::move(transitions);
    d["emission_model"] = std::move(emission);
    return d;
  }

 private:
  dyn_aligner* h_ = nullptr;
  dyn_info info_{};
};

}  // namespace

PYBIND11_MODULE(_dynamont, m) {  // aligner_bindings.cpp:180-219
  m.doc() = "dynamont aligner bindings over the MI355X C ABI (libdynamont_mi)";
  py::enum_<dyn_pore>(m, "PoreType")
      .value("RNA002", DYN_PORE_RNA002)
      .value("RNA004", DYN_PORE_RNA004)
      .value("DNA_R9", DYN_PORE_DNA_R9)
      .value("DNA_R10_260", DYN_PORE_DNA_R10_260)
      .value("DNA_R10_400", DYN_PORE_DNA_R10_400);
  py::class_<PyAligner>(m, "Aligner")
      .def(py::init([](const std::string& model, dyn_pore pore, const std::string& mode, int threads, std::size_t band) {
             return new PyAligner(model, (int)pore, mode, threads, band);
           }),
           py::arg("model_file"), py::arg("pore"), py::arg("mode") = "basic", py::arg("threads") = 1, py::arg("band") = 400)
      .def(py::init<const std::string&, const std::string&, const std::string&, int, std::size_t>(), py::arg("model_file"),
           py::arg("pore"), py::arg("mode") = "basic", py::arg("threads") = 1, py::arg("band") = 400)
      .def("align", &PyAligner::align, py::arg("signal"), py::arg("sequence"), py::arg("calc_probabilities") = false)
      .def("train", &PyAligner::train, py::arg("signal"), py::arg("sequence"))
      .def("align_batch", &PyAligner::align_batch, py::arg("signals"), py::arg("sequences"), py::arg("calc_probabilities") = true);
  m.def("pore_type", [](const std::string& s) { return (dyn_pore)pore_from_string(s); }, py::arg("pore"));
}
