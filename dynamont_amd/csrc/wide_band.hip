// wide_band.hip -- any band: the reads whose half band min(band / 2, N / 2) exceeds what the register sweeps hold.
//
// The reference constructs with ANY band (src/cpp/aligner.cpp:21, aligner_bindings.cpp:191-199) and walks rows of
// 2 * bw + 3 band columns whatever bw is (NT_aligner_api.cpp:243-244). The tuned sweeps of nt_kernels.hip keep a lattice
// row in 64 lanes x 7 registers: half bands up to 223, which covers every caller of the reference (band = 400,
// segment.py:45). Reads beyond that -- a handle built with band > 447 AND more than 447 lattice columns -- take THIS
// kernel: the reference's algorithm in its own band-column addressing,
//   computeBounds          NT_aligner_api.cpp:90-108     cell (t, n) <-> band column c = n - start_t + 1, 1 <= c <= 2 bw + 1
//   backward               NT_aligner_api.cpp:158-207    stores bE and bM rows (16 B per cell)
//   forward + posterior    NT_aligner_api.cpp:110-152, 213-224   )
//   posterior-Viterbi      NT_aligner_api.cpp:338-363            )  one ascending sweep, previous rows in LDS
//   decision bit           NT_aligner_api.cpp:448                )
//   decodeMAP              NT_aligner_api.cpp:383-456    one lane walks the bits
//   runTraining            NT_aligner_api.cpp:462-561    per-column (w, s1, s2), flushed when the band moves
// One 256-thread workgroup per read, thread i owns band columns i, i + 256, ...; a row is one pass + one barrier. The
// arithmetic is the reference's OWN, bit for bit, in every cell (dp_math_strict.hpp: the IEEE quotient, glibc's exp and
// fdlibm's log1p restated) -- no certificate, no table softplus: Z, the borders and the decision bits are the reference's
// whether or not the read carries a structural tie. It is the rare path and runs at a fraction of the tuned sweeps' rate
// (~25 B of HBM and ~300 instructions per cell); what it buys is that no band the reference accepts is refused.
// Per-row outputs (pp, pathn, segrow) are those of nt_kernels.hip's traceback: k_median / k_final follow unchanged.
#include "nt_kernels.hpp"

#include "dp_math_strict.hpp"

namespace dynk {

using dynmath::NEG_INF;

namespace {

constexpr int WIDE_THREADS = 256;

__device__ __forceinline__ int wide_mid(int t, double ratio) { return (int)__dmul_rn((double)t, ratio); }  // size_t(t * RATIO), :100

// TRAIN: the statistics pass instead of posterior-Viterbi + traceback. CALC: align(calc_probabilities = true).
template <bool CALC, bool TRAIN>
__device__ void wide_read(const ReadDesc& rd, const WideArgs& a, double* __restrict__ bE, double* __restrict__ bM,
                          float* __restrict__ lp, uint8_t* __restrict__ bit, double (*s_row)[WIDE_MAX_B], const uint64_t* s_exp) {
  const int tid = threadIdx.x;
  const int T = (int)rd.T, N = (int)rd.N, bw = (int)rd.bw, B = 2 * bw + 3;
  const double ratio = rd.ratio, m1 = a.m1, e2 = a.e2;
  const double* __restrict__ sg = a.sig + rd.sig_off;
  const Emis* __restrict__ pr = a.par + rd.par_off;  // entry n - 1 <-> lattice column n
  double* nxE = s_row[0];  // backward: row t + 1;  forward: fE of row t - 1
  double* nxM = s_row[1];  //                         forward: fM of row t - 1
  double* pvE = s_row[2];  // forward: vE of row t - 1
  double* pvM = s_row[3];  // forward: vM of row t - 1
  __shared__ int s_bad;
  if (tid == 0) s_bad = 0;
  auto at = [&](const double* row, int c) { return (c >= 0 && c < B) ? row[c] : NEG_INF; };  // guard columns: -inf (:129-130)

  // ---- backward (NT_aligner_api.cpp:158-207): t = T-2 .. 0, rows bE / bM stored ----
  for (int c = tid; c < B; c += WIDE_THREADS) {
    const double v = (c == bw + 1) ? 0.0 : NEG_INF;  // E[(T-1) * B + bw + 1] = 0 (:170)
    nxE[c] = v;
    nxM[c] = NEG_INF;
    bE[(size_t)(T - 1) * B + c] = v;
    bM[(size_t)(T - 1) * B + c] = NEG_INF;
  }
  __syncthreads();
  int bad = 0;
  for (int t = T - 2; t >= 0; --t) {
    const int mid = wide_mid(t, ratio), start = mid - bw;
    const int n_lo = start > 0 ? start : 0, n_hi = (mid + bw + 1 < N) ? mid + bw + 1 : N;
    const int shift = (wide_mid(t + 1, ratio) - bw) - start;  // start_{t+1} - start_t: 0 or 1 (:177-184)
    const double x = sg[t];
    bad |= !(__builtin_fabs(x) <= 1.7976931348623157e308);
    double oE[WIDE_CPT], oM[WIDE_CPT];
#pragma unroll 1
    for (int k = 0, c = tid; c < B; c += WIDE_THREADS, ++k) {
      const int n = start + c - 1;
      double ext = NEG_INF, bm = NEG_INF;
      if (c >= 1 && c <= 2 * bw + 1 && n >= n_lo && n < n_hi) {
        const int cn = c - shift;  // band column of lattice column n in row t + 1
        if (n + 1 < N) ext = (at(nxM, cn + 1) + dynmath::log_normal_pdf_strict(x, pr[n])) + m1;  // :192-195
        if (n > 0) {
          const double score = dynmath::log_normal_pdf_strict(x, pr[n - 1]);
          const double e_next = at(nxE, cn);
          bm = e_next + score;                                                            // :200
          ext = dynmath::log_plus_strict(ext, (e_next + score) + e2, s_exp);               // :201
        }
      }
      oE[k] = ext;
      oM[k] = bm;
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 0, c = tid; c < B; c += WIDE_THREADS, ++k) {
      nxE[c] = oE[k];
      nxM[c] = oM[k];
      bE[(size_t)t * B + c] = oE[k];
      bM[(size_t)t * B + c] = oM[k];
    }
    __syncthreads();
  }
  if (bad) s_bad = 1;
  const double Zb_raw = nxE[bw + 1];  // backwardE[bw + 1] (:286)
  __syncthreads();
  const double Zb = s_bad ? NEG_INF : Zb_raw;  // an infinite / NaN sample: "alignment scores do not match", as the tuned sweeps report it

  // ---- forward (:110-152) + posterior (:213-224) + posterior-Viterbi (:338-363) + decision bit (:448), t = 1 .. T-1 ----
  for (int c = tid; c < B; c += WIDE_THREADS) {
    const double v = (c == bw + 1) ? 0.0 : NEG_INF;  // E[bw + 1] = 0 (:120, :336)
    nxE[c] = v;
    nxM[c] = NEG_INF;
    pvE[c] = v;
    pvM[c] = NEG_INF;
  }
  __syncthreads();
  // TRAIN: this thread's running sums for the lattice column its band column currently covers (flushed when the band moves)
  double aw[WIDE_CPT], a1[WIDE_CPT], a2[WIDE_CPT];
  if (TRAIN) {
#pragma unroll 1
    for (int k = 0; k < WIDE_CPT; ++k) aw[k] = a1[k] = a2[k] = 0.0;
  }
  double* __restrict__ cw = TRAIN ? a.tr.col_w + rd.par_off : nullptr;
  double* __restrict__ cs1 = TRAIN ? a.tr.col_s1 + rd.par_off : nullptr;
  double* __restrict__ cs2 = TRAIN ? a.tr.col_s2 + rd.par_off : nullptr;
  if (TRAIN) {
    for (int n = tid; n < N - 1; n += WIDE_THREADS) cw[n] = cs1[n] = cs2[n] = 0.0;  // the read's columns: sums start at 0
    __syncthreads();
  }
  for (int t = 1; t < T; ++t) {
    const int mid = wide_mid(t, ratio), start = mid - bw;
    const int n_lo = start > 1 ? start : 1, n_hi = (mid + bw + 1 < N) ? mid + bw + 1 : N;  // forward never fills n = 0 (:126-127)
    const int shift = start - (wide_mid(t - 1, ratio) - bw);  // 0 or 1 (:131-138)
    const double x = sg[t - 1];
    if (TRAIN && shift) {
      // the band moves up by one column: band column c covered lattice column start_{t-1} + c - 1 until now
#pragma unroll 1
      for (int k = 0, c = tid; c < B; c += WIDE_THREADS, ++k) {
        const int n_old = start - 1 + c - 1;
        if (n_old >= 1 && n_old < N && aw[k] != 0.0) {
          cw[n_old - 1] += aw[k];
          cs1[n_old - 1] += a1[k];
          cs2[n_old - 1] += a2[k];
        }
        aw[k] = a1[k] = a2[k] = 0.0;
      }
      __syncthreads();  // the next owner of a column adds to the same words later
    }
    double ofE[WIDE_CPT], ofM[WIDE_CPT], ovE[WIDE_CPT], ovM[WIDE_CPT];
#pragma unroll 1
    for (int k = 0, c = tid; c < B; c += WIDE_THREADS, ++k) {
      const int n = start + c - 1;
      double fM = NEG_INF, fE = NEG_INF, vM = NEG_INF, vE = NEG_INF;
      if (c >= 1 && c <= 2 * bw + 1 && n >= n_lo && n < n_hi) {
        const int cp = c + shift;  // band column of lattice column n in row t - 1
        const double score = dynmath::log_normal_pdf_strict(x, pr[n - 1]);               // :144
        fM = (at(nxE, cp - 1) + score) + m1;                                              // :146
        fE = dynmath::log_plus_strict((at(nxM, cp) + score) + 0.0, (at(nxE, cp) + score) + e2, s_exp);  // :147-149, e1 = log 1
        if (CALC || TRAIN) {
          const size_t cell = (size_t)t * B + c;
          const double LPM = (fM + bM[cell]) - Zb, LPE = (fE + bE[cell]) - Zb;            // :222
          if (TRAIN) {
            const double g = exp(LPM) + exp(LPE);                                         // :505-512
            aw[k] += g;
            a1[k] += g * x;
            a2[k] += (g * x) * x;
          } else {
            vM = at(pvE, cp - 1) + LPM;                                                   // :360
            const double um = at(pvM, cp), ue = at(pvE, cp);
            vE = (um < ue ? ue : um) + LPE;                                               // :361 (std::max)
            lp[2 * cell] = (float)LPM;
            lp[2 * cell + 1] = (float)LPE;
            bit[cell] = (vE == um + LPE) ? 1 : 0;                                         // :448
          }
        }
      }
      ofE[k] = fE;
      ofM[k] = fM;
      ovE[k] = vE;
      ovM[k] = vM;
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 0, c = tid; c < B; c += WIDE_THREADS, ++k) {
      nxE[c] = ofE[k];
      nxM[c] = ofM[k];
      if (CALC && !TRAIN) {
        pvE[c] = ovE[k];
        pvM[c] = ovM[k];
      }
    }
    __syncthreads();
  }
  if (TRAIN) {
    const int start = wide_mid(T - 1, ratio) - bw;
#pragma unroll 1
    for (int k = 0, c = tid; c < B; c += WIDE_THREADS, ++k) {
      const int n_old = start + c - 1;
      if (n_old >= 1 && n_old < N && aw[k] != 0.0) {
        cw[n_old - 1] += aw[k];
        cs1[n_old - 1] += a1[k];
        cs2[n_old - 1] += a2[k];
      }
    }
  }
  const double Zf = nxE[bw + 1];  // forwardE[T * B - bw - 2] (:285)
  // Z check (:285-291 / :619-625)
  const double size = (double)((uint64_t)T * (uint64_t)B);
  const bool ok = !(isinf(Zf) || isinf(Zb)) && !(__builtin_fabs(Zf - Zb) / size > 1e-8);
  int status = ok ? 0 : a.z_fail_status;
  uint32_t n_seg = 0;
  __syncthreads();

  // ---- decodeMAP (:383-456): one lane walks the decision bits from (T-1, N-1) in state E ----
  if (CALC && !TRAIN && ok) {
    if (tid == 0) {
      double* __restrict__ pp = a.tb.pp + rd.path_off;
      uint32_t* __restrict__ pathn = a.tb.pathn + rd.path_off;
      uint32_t* __restrict__ segrow = a.tb.segrow + rd.seg_off;
      int t = T - 1, n = N - 1;
      bool isM = false;
      while (t > 0 && n > 0) {
        const int c = n - (wide_mid(t, ratio) - bw) + 1;
        const size_t cell = (size_t)t * B + c;
        if (isM) {
          pp[t] = exp((double)lp[2 * cell]);
          pathn[t] = (uint32_t)n | 0x80000000u;
          segrow[n - 1] = (uint32_t)t;
          --t;
          --n;
          isM = false;
        } else {
          pp[t] = exp((double)lp[2 * cell + 1]);
          pathn[t] = (uint32_t)n;
          isM = bit[cell] != 0;
          --t;
        }
      }
      s_bad = (t == 0 && n == 0) ? 0 : 2;
    }
    __syncthreads();
    if (s_bad == 2) status = 7;  // DYN_READ_INTERNAL
    else n_seg = rd.N - 1;
  }
  if (TRAIN && tid == 0 && ok) {
    // expected transition counts (:641-725): every path makes N - 1 moves and T - 1 - 2 (N - 1) extensions
    a.tr.trans[2 * rd.read] = (double)(N - 1);
    a.tr.trans[2 * rd.read + 1] = (double)(T - 1 - 2 * (N - 1));
  }
  if (tid == 0) {
    ReadState s;
    s.Zb = Zb;
    s.Zf = Zf;
    s.status = status;
    s.n_segments = n_seg;
    a.st[rd.read] = s;
  }
  __syncthreads();
}

}  // namespace

// one workgroup takes wide reads off a queue until it is empty; its lattice arena holds one read at a time
template <bool CALC, bool TRAIN>
__global__ __launch_bounds__(WIDE_THREADS) void k_wide_reads(const WideArgs a) {
  __shared__ __attribute__((aligned(16))) double s_row[4][WIDE_MAX_B];
  __shared__ uint64_t s_exp[dynmath::STRICT_EXP_WORDS];
  __shared__ int s_next;
  for (int i = threadIdx.x; i < dynmath::STRICT_EXP_WORDS; i += WIDE_THREADS) s_exp[i] = a.exp_tab[i];
  char* arena = a.arena + (size_t)blockIdx.x * a.arena_bytes;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_next = (int)atomicAdd(a.head, 1u);
    __syncthreads();
    const int k = s_next;
    if (k >= a.n_reads) break;
    const ReadDesc rd = a.descs[k];
    const size_t cells = (size_t)rd.T * (size_t)(2 * rd.bw + 3);
    double* bE = reinterpret_cast<double*>(arena);
    double* bM = bE + cells;
    float* lp = reinterpret_cast<float*>(bM + cells);
    uint8_t* bit = reinterpret_cast<uint8_t*>(lp + 2 * cells);
    wide_read<CALC, TRAIN>(rd, a, bE, bM, lp, bit, s_row, s_exp);
  }
}

uint64_t wide_arena_bytes(uint64_t T, uint64_t bw, bool calc) {
  const uint64_t cells = T * (2 * bw + 3);
  return (cells * (16 + (calc ? 9 : 0)) + 255) & ~255ull;  // bE, bM doubles; (float LPM, float LPE); one byte per decision
}

void launch_wide_reads(int job, const WideArgs& a, int n_groups, hipStream_t s) {
  if (a.n_reads <= 0 || n_groups <= 0) return;
  (void)hipMemsetAsync(a.head, 0, 4, s);
  if (job == 2) hipLaunchKernelGGL((k_wide_reads<false, true>), dim3(n_groups), dim3(WIDE_THREADS), 0, s, a);
  else if (job == 1) hipLaunchKernelGGL((k_wide_reads<true, false>), dim3(n_groups), dim3(WIDE_THREADS), 0, s, a);
  else hipLaunchKernelGGL((k_wide_reads<false, false>), dim3(n_groups), dim3(WIDE_THREADS), 0, s, a);
}

}  // namespace dynk
