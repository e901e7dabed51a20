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

struct ReadDesc {
  uint64_t sig_off;   // first sample in the signal pool (doubles)
  uint64_t par_off;   // first entry in the per-column emission table; entry n-1 <-> column n
  uint64_t ws_off;    // lattice workspace [T+1][P] (8 B per slot; row T = -inf)
  uint64_t bits_off;  // decision bits [T][CPL] (uint64)
  uint64_t path_off;  // per-row path arrays [T]
  uint64_t seg_off;   // first output row of this read
  uint32_t T;         // signal length + 1   (NT_aligner_api.cpp:241)
  uint32_t N;         // k-mer count + 1     (NT_aligner_api.cpp:242)
  uint32_t bw;        // min(band/2, N/2)    (NT_aligner_api.cpp:243)
  uint32_t read;      // index into the per-read state arrays
  double ratio;       // double(N)/double(T) (NT_aligner_api.cpp:96)
};

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

// P1/P2 on the device: out[i] = hampel((REAL(raw[i]) - shift) / scale); REAL = float when compute_f32.
// raw_dtype: 0 float32, 1 int16, 2 float64. norm_tmp: scratch of total samples * sizeof(REAL).
void launch_preprocess(const void* raw, int raw_dtype, int compute_f32, const uint64_t* offs,
                       const double* shift, const double* scale, void* norm_tmp, double* out,
                       int n_reads, uint64_t max_len, int W, double n_sigmas, hipStream_t s);
void launch_prep_params(const int32_t* kmers, const Emis* model, Emis* par, uint64_t total,
                        hipStream_t s);
// sp_tab: device copy of dynmath::softplus_build_table (SP_NODES entries)
void launch_backward(const ReadDesc* descs, int n_reads, const double* sig, const Emis* par,
                     double* ws, ReadState* st, double m1, double e2, bool store,
                     const dynmath::SoftplusNode* sp_tab, hipStream_t s);
// lpe != nullptr: float [rows][P] log-posterior of state E per slot, indexed like ws (ws stays intact);
// lpe == nullptr (with post): (float LPM, float LPE) overwrite the 8-byte slots of ws in place
void launch_forward(const ReadDesc* descs, int n_reads, const double* sig, const Emis* par,
                    double* ws, float* lpe, uint64_t* bits, ReadState* st, double m1, double e2, bool post,
                    const dynmath::SoftplusNode* sp_tab, hipStream_t s);
void launch_forward_train(const ReadDesc* descs, int n_reads, const double* sig, const Emis* par,
                          const double* ws, ReadState* st, TrainBuffers tb, double m1, double e2,
                          const dynmath::SoftplusNode* sp_tab, hipStream_t s);
void launch_trace(const ReadDesc* descs, int n_reads, uint32_t max_T, uint32_t max_N,
                  const double* ws, const float* lpe, const uint64_t* bits, const double* sig, const Emis* par,
                  ReadState* st, TraceBuffers tb, SegRow* rows, int kmer_size, double m1, int z_fail_status,
                  hipStream_t s);
void launch_zcheck(const ReadDesc* descs, int n_reads, ReadState* st, int z_fail_status,
                   hipStream_t s);
// pooled[3*num_kmers] += per-k-mer (w, s1, s2) of the reads in descs (fp64 atomics)
void launch_pool_stats(const ReadDesc* descs, int n_reads, uint32_t max_N, const ReadState* st,
                       const int32_t* kmers, TrainBuffers tb, double* pooled, uint64_t num_kmers,
                       hipStream_t s);

}  // namespace dynk
