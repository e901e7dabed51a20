// nt_kernels.hip -- hand-written gfx950 kernels of the NT resquiggling hot path.
//
// Reference being reproduced (paths relative to the reference checkout):
//   computeBounds          src/cpp/NT_aligner_api.cpp:90-108
//   forward                src/cpp/NT_aligner_api.cpp:110-152
//   backward               src/cpp/NT_aligner_api.cpp:158-207
//   calculatePosterior     src/cpp/NT_aligner_api.cpp:213-224
//   calculateSegments      src/cpp/NT_aligner_api.cpp:314-377   (posterior-Viterbi fill)
//   decodeMAP              src/cpp/NT_aligner_api.cpp:383-456   (traceback)
//   formattedMedian        src/cpp/aligner.cpp:247-263
//   runTraining            src/cpp/NT_aligner_api.cpp:462-561
//   trainTransition        src/cpp/NT_aligner_api.cpp:641-725
//
// Design (not a translation): the reference streams eight T x B fp64 matrices per read. Here
//   K_bwd  walks t = T-2..0 and stores ONLY backward-E (bM(t,n) = bE(t+1,n) + e(t+1,n) is one
//          add away, NT_aligner_api.cpp:200);
//   K_fwd  walks t = 1..T-1 and fuses forward, posterior (needs Zb, known after K_bwd),
//          the posterior-Viterbi fill and the traceback decision
//          (vE == vM_prev + LPE, NT_aligner_api.cpp:448); forward / Viterbi rows never leave
//          registers; per cell it writes one float, LPE, for the path-probability lookup and
//          1 decision bit (footprint-limited batches: (float LPM, float LPE) over the bE slot);
//   K_trace walks the decision bits back one SEGMENT per step (ballot + find-first-set inside
//          64-row LDS blocks), K_mpost rebuilds LPM for the segment starts, K_median/K_final
//          produce the per-segment median posterior.
// HBM traffic is 20.1 B per in-band cell instead of the 64.1 B of the three-pass formulation
// (SURVEY.md §8d), and the per-read footprint is 12 B per slot instead of 64 B per cell.
//
// Mapping: one 64-lane wave owns one read, four reads share a 256-thread workgroup (one wave per
// SIMD). Band slot s = n mod P lives in lane s / CPL, register s % CPL: a lane owns CPL CONSECUTIVE
// slots, so the cross-lane neighbour (n-1 forward, n+1 backward) of all but one of its cells is its
// own next register and one DPP wave rotate per exchange serves the remaining cell -- no barrier
// anywhere in the DP loops. In HBM a row stores slot s at position row_pos(s) = (s % CPL)*64 + s / CPL,
// i.e. register j of all lanes is one contiguous 512-byte run: every row access is CPL fully
// coalesced operations (a 56-byte lane stride, the "natural" placement of this slot numbering, cost
// K_bwd 20 % in partial-line writes). The CU's LDS holds the softplus table shared by the four waves
// (dp_math.hpp) and, in K_fwd, a 4-row-deep ring per wave that is filled straight from HBM by
// global_load_lds_dwordx4 (see ring_dma_row).
#include "nt_kernels.hpp"

#include <algorithm>

// Placement: the SPI packs single-wave workgroups onto one SIMD for as long as its registers
// allow (measured: a 184-VGPR build of K_bwd ran 1 024 reads as 2 waves on each of 512 SIMDs and
// took 2x the time of 512 reads; tools/ubench + DESIGN.md). The DP waves are pure fp64 issue
// streams with 7-way ILP that saturate a SIMD on their own, so they are compiled for exactly one
// wave per SIMD: the dispatcher must then spread the reads over all 1 024 SIMDs, and the whole
// 512-entry register file is available to keep the interleaved chains out of AGPR spills.
#define DYN_ONE_WAVE_PER_SIMD __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))

// The DP kernels run FOUR reads per 256-thread workgroup (one wave each, one per SIMD of a CU):
// the four waves never synchronise after the prologue, they only share the 82 KB softplus table
// that the workgroup stages into the CU's LDS once (dp_math.hpp, softplus_table_vec).
#define DYN_READS_PER_GROUP 4

namespace dynk {

using dynmath::NEG_INF;
using dynmath::EmisV;
using dynmath::log_normal_pdf_vec;
using dynmath::SoftplusNode;
using dynmath::SoftplusLookup;
using dynmath::log_plus_issue;
using dynmath::log_plus_finish;
using dynmath::SP_NODES;

namespace {

__device__ __forceinline__ int pmod(int a) {
  int r = a % P;
  return r < 0 ? r + P : r;
}

// position of band slot s inside a stored row (see "Mapping" above)
__device__ __forceinline__ int row_pos(int s) { return (s % CPL) * 64 + s / CPL; }

// size_t(t * RATIO): one IEEE fp64 multiply, then truncation (NT_aligner_api.cpp:100).
__device__ __forceinline__ int band_mid(int t, double ratio) {
  return (int)__dmul_rn((double)t, ratio);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// lane i receives lane i-1 (lane 0 receives lane 63)
__device__ __forceinline__ double wave_ror1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  // mov_dpp: no "old" operand tied to the source, so a still-live source needs no copy first
  const int rlo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xf, 0xf, false);
  const int rhi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xf, 0xf, false);
  return __hiloint2double(rhi, rlo);
}

// lane i receives lane i+1 (lane 63 receives lane 0)
__device__ __forceinline__ double wave_rol1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  // mov_dpp: no "old" operand tied to the source, so a still-live source needs no copy first
  const int rlo = __builtin_amdgcn_mov_dpp(lo, 0x134, 0xf, 0xf, false);
  const int rhi = __builtin_amdgcn_mov_dpp(hi, 0x134, 0xf, 0xf, false);
  return __hiloint2double(rhi, rlo);
}

// Blocked slot layout: lane l owns the CPL consecutive slots l*CPL .. l*CPL+CPL-1, so the left
// neighbour of its cells 1..CPL-1 is its own previous register and only cell 0 needs the last cell
// of lane l-1: ONE wave rotate per exchange (the cyclic layout slot = j*64+lane needed CPL rotates
// plus CPL lane-0 fix-ups). The rotate wraps lane 63 -> lane 0, which is exactly the slot P-1 -> 0
// wrap of the band ring.
__device__ __forceinline__ void from_left(const double (&x)[CPL], double (&out)[CPL]) {
  out[0] = wave_ror1(x[CPL - 1]);
#pragma unroll
  for (int j = 1; j < CPL; ++j) out[j] = x[j - 1];
}

// out[slot] = x[slot+1]
__device__ __forceinline__ void from_right(const double (&x)[CPL], double (&out)[CPL]) {
#pragma unroll
  for (int j = 0; j < CPL - 1; ++j) out[j] = x[j + 1];
  out[CPL - 1] = wave_rol1(x[0]);
}

// n is wave-uniform at the re-assignment sites: the load then becomes an s_load (lgkmcnt), which
// keeps it out of the vector-memory counter the DMA ring waits on.
__device__ __forceinline__ Emis load_emis(const Emis* __restrict__ pr, int n, int N) {
  Emis e;
  if (n >= 1 && n < N) {
    e = pr[n - 1];
  } else {  // column without a k-mer (n <= 0 or n >= N): log-density -inf (log_norm = -inf)
    e.mean = 0.0;
    e.inv_stdev = 1.0;
    e.log_norm = NEG_INF;
    e.stdev = 1.0;
  }
  return e;
}

// ---- LDS-DMA row ring (k_forward) -------------------------------------------------------------
// K_fwd must read one [448]-double bE row per lattice row while it overwrites an older row. With
// the rows prefetched into VGPRs the single wave of a SIMD stalled on s_waitcnt for 43 % of its
// life (compiler-placed vmcnt(0) behind the previous row's stores, prefetch registers spilled to
// AGPRs mid-row). The rows now go HBM -> LDS directly (global_load_lds_dwordx4: no VGPRs, 3 full +
// 1 half-wave instruction per 3584-byte row), RING_D rows ahead, and are picked up with ds_read_b64
// behind a hand-counted s_waitcnt. Validated in isolation by tools/ubench/lds_dma_test.hip.
constexpr int RING_D = 4;  // 2..5 rows deep measure the same: the kernel is bandwidth-, not latency-bound
constexpr int ROW_BYTES = P * 8;
// vmcnt(N) lets the N youngest vector-memory operations stay in flight. Only the DMA instructions
// themselves are counted (4 per row, RING_D-1 younger rows => 12): loads retire in issue order, so
// the wait is correct whatever the compiler does with the stores, sample and parameter loads
// that share the counter (they can only make it stricter).
constexpr int RING_WAIT = 4 * (RING_D - 1);

// row_lane_ptr = &row[lane*2]; `offset:` advances both the global and the LDS address.
// Each row is consumed exactly once: the DMA carries the non-temporal hint (-1 % on the kernel).
#define DYN_DMA_MOD " nt"
__device__ __forceinline__ void ring_dma_row(const double* row_lane_ptr, unsigned lds_slot_addr) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off" DYN_DMA_MOD "\n\t"
      "global_load_lds_dwordx4 %0, off offset:1024" DYN_DMA_MOD "\n\t"
      "global_load_lds_dwordx4 %0, off offset:2048" DYN_DMA_MOD "\n\t"
      "s_mov_b32 exec_hi, 0\n\t"
      "global_load_lds_dwordx4 %0, off offset:3072" DYN_DMA_MOD "\n\t"
      "s_mov_b32 exec_hi, -1\n\t"
      ::"v"(row_lane_ptr), "s"(__builtin_amdgcn_readfirstlane(lds_slot_addr))
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the wave's copy of one row: cell (lane, j) sits at row_pos*8 = (j*64 + lane)*8
__device__ __forceinline__ void ring_read_row(unsigned lds_lane_addr, double (&b)[CPL]) {
  static_assert(CPL == 7, "ring_read_row is written for 7 cells per lane");
  asm volatile(
      "ds_read_b64 %0, %7\n\t"
      "ds_read_b64 %1, %7 offset:512\n\t"
      "ds_read_b64 %2, %7 offset:1024\n\t"
      "ds_read_b64 %3, %7 offset:1536\n\t"
      "ds_read_b64 %4, %7 offset:2048\n\t"
      "ds_read_b64 %5, %7 offset:2560\n\t"
      "ds_read_b64 %6, %7 offset:3072\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6])
      : "v"(lds_lane_addr)
      : "memory");
}

// Stage the softplus table into LDS (all 256 threads), then tell the caller which read this
// wave owns (-1: none; the wave must still have taken part in the barrier).
__device__ __forceinline__ int stage_table_and_pick_read(SoftplusNode* s_tab, const SoftplusNode* __restrict__ tab,
                                                         int n_reads) {
  for (int i = threadIdx.x; i < SP_NODES; i += 256) s_tab[i] = tab[i];
  __syncthreads();
  // readfirstlane: the wave index is uniform, and the compiler must know it -- otherwise the read
  // descriptor, every pointer and loop bound derived from it live in VGPRs and every table /
  // parameter access becomes a vector load.
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = blockIdx.x * DYN_READS_PER_GROUP + wave;
  return r < n_reads ? r : -1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// per-column emission table: par[i] = model[kmers[i]]  (aligner.cpp:241-245 scoreKmer lookup)
// ---------------------------------------------------------------------------------------------
__global__ void k_prep_params(const int32_t* __restrict__ kmers, const Emis* __restrict__ model,
                              Emis* __restrict__ par, uint64_t total) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) par[i] = model[kmers[i]];
}

// ---------------------------------------------------------------------------------------------
// K_bwd: backward recursion, t = T-2 .. 0 (NT_aligner_api.cpp:158-207)
//   e(t+1,n)  = logN(sig[t]; kmers[n-1])                      emission of lattice cell (t+1,n)
//   bM(t,n)   = bE(t+1,n) + e(t+1,n)                          (n > 0)            :197-200
//   bE(t,n)   = logPlus( (bM(t+1,n+1) + e(t+1,n+1)) + m1 ,    (n+1 < N)          :192-195
//                        (bE(t+1,n)   + e(t+1,n))   + e2 )    (n > 0)            :201
// Columns without a k-mer (n <= 0, n >= N) carry emission -inf, which realises the n > 0 and
// n+1 < N guards arithmetically; the upper band edge is handled in the window-move block.
// ---------------------------------------------------------------------------------------------
template <bool STORE>
__global__ DYN_ONE_WAVE_PER_SIMD void k_backward(const ReadDesc* __restrict__ descs,
                                                  const double* __restrict__ sig,
                                                  const Emis* __restrict__ par,
                                                  double* __restrict__ ws,
                                                  ReadState* __restrict__ st, double m1,
                                                  double e2, const SoftplusNode* __restrict__ sp_tab,
                                                  int n_reads) {
  __shared__ __attribute__((aligned(16))) SoftplusNode s_tab[SP_NODES];
  const int ridx = stage_table_and_pick_read(s_tab, sp_tab, n_reads);
  if (ridx < 0) return;
  const ReadDesc rd = descs[ridx];
  const int lane = threadIdx.x & 63;
  const int T = (int)rd.T, N = (int)rd.N, bw = (int)rd.bw, W = 2 * bw + 1;
  const double ratio = rd.ratio;
  const double* __restrict__ sg = sig + rd.sig_off;
  const Emis* __restrict__ pr = par + rd.par_off;
  double* __restrict__ out = ws + rd.ws_off + lane;

  bool bad_sample = false;
  int lo = band_mid(T - 1, ratio) - bw;
  const int n_init = lo + bw;  // band column bw+1 of row T-1 (NT_aligner_api.cpp:170)
  int n[CPL];
  double bE[CPL], bM[CPL], e[CPL];
  EmisV<CPL> p;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int slot = lane * CPL + j;
    n[j] = lo + pmod(slot - lo);
    p.set(j, load_emis(pr, n[j], (n[j] <= lo + W - 1) ? N : 0));  // columns above the band: no k-mer yet
    bE[j] = (n[j] == n_init) ? 0.0 : NEG_INF;
    bM[j] = NEG_INF;
    if (STORE) {
      out[(size_t)(T - 1) * P + j * 64] = bE[j];
      // one row past the lattice, all -inf: bM(T-1, .) = bE(T, .) + e has no successor. It lets
      // k_forward stream row t+1 for every t without a last-row special case in its row loop.
      out[(size_t)T * P + j * 64] = NEG_INF;
    }
  }

  for (int thi = T - 2; thi >= 0; thi -= 64) {
    const int base = thi - 63;
    const int idx = base + lane;
    const double xs = (idx >= 0) ? sg[idx] : 0.0;
    bad_sample |= !(fabs(xs) <= 1.7976931348623157e308);  // inf or NaN
    const int ilo = base < 0 ? -base : 0;
    log_normal_pdf_vec<CPL>(readlane_f64(xs, 63), p, e);  // e(thi+1, n) from sig[thi]
#pragma unroll 1
    for (int i = 63; i >= ilo; --i) {
      const int t = base + i;
      // e[] = e(t+1, n) was computed during the previous row's table lookups (software pipeline)
      double Y[CPL], A[CPL], Yr[CPL], x1[CPL], x2[CPL], ne[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) Y[j] = bM[j] + e[j];
#pragma unroll
      for (int j = 0; j < CPL; ++j) A[j] = bE[j] + e[j];
      from_right(Y, Yr);
      const int new_lo = band_mid(t, ratio) - bw;
      if (new_lo != lo) {  // wave-uniform: the window moved down by one column
        const int leaving = lo + P - 1;
        const int top = lo + W - 1;  // last band column of row t+1: above the band of row t
        // uniform address, outside the per-lane branch: a scalar load (lgkmcnt). A vector load here
        // makes hipcc guard every later read of p with s_waitcnt vmcnt(0) in EVERY row.
        const Emis fresh = load_emis(pr, new_lo, N);
        const Emis none = load_emis(pr, 0, 0);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          if (n[j] == leaving) {
            n[j] = new_lo;
            p.set(j, fresh);
          }
          // Upper band edge: bM(t, top) = A must not see the in-band cell (t+1, top); Y keeps it for
          // the diagonal into (t, top-1). From row t on the slot carries the "no k-mer" parameters, so
          // e = -inf and with it A = Y = -inf for every column above the band without a per-row
          // select -- including the slot of column lo+P-1, whose bE picks up a finite x1 from its
          // ring neighbour, band column lo, in every row (the slot ring wraps) and is emptied by
          // e = -inf before anything reads it.
          if (n[j] == top) {
            A[j] = NEG_INF;
            p.set(j, none);
          }
        }
        lo = new_lo;
      }
#pragma unroll
      for (int j = 0; j < CPL; ++j) x1[j] = Yr[j] + m1;
#pragma unroll
      for (int j = 0; j < CPL; ++j) x2[j] = A[j] + e2;
      SoftplusLookup<CPL> L;
      log_plus_issue<CPL>(x1, x2, L, s_tab);
      // while the LDS lookups are in flight: emission of the NEXT row, e(t, n) = logN(sig[t-1]; .)
      // (rows are consumed top-down; at i == 0 the next block's sample is not loaded yet: that one
      //  emission is computed after the block switch below)
      if (i > 0) {
        const double xnext = readlane_f64(xs, i - 1);
        log_normal_pdf_vec<CPL>(xnext, p, e);
      }
      log_plus_finish<CPL>(L, ne);
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        bE[j] = ne[j];
        bM[j] = A[j];
        // streamed once, read back by k_forward ~70 GB later: non-temporal (-2 % on the kernel)
        if (STORE) __builtin_nontemporal_store(ne[j], &out[(size_t)t * P + j * 64]);
      }
    }
  }
  // An infinite sample gives every cell of its row the score -inf in the reference (aligner.cpp:
  // 287-292), hence Z = -inf and "alignment scores do not match" (NT_aligner_api.cpp:288-291). The
  // residual-corrected quotient of log_normal_pdf_vec would turn it into NaN instead, so non-finite
  // samples are flagged here and reported through the same Z check (NaN samples, for which the
  // reference's behaviour is undefined, fail the same way).
  const bool any_bad = __any(bad_sample);
  // lattice column 0 sits in slot 0 at row 0 (lo = -bw)
  if (lane == 0) st[rd.read].Zb = any_bad ? NEG_INF : bE[0];
}

// ---------------------------------------------------------------------------------------------
// K_fwd: forward (NT_aligner_api.cpp:110-152) fused with posterior (:213-224,297-300) and the
// posterior-Viterbi fill (:338-363), t = 1 .. T-1:
//   e(t,n)   = logN(sig[t-1]; kmers[n-1])
//   fM(t,n)  = (fE(t-1,n-1) + e) + m1                                             :146
//   fE(t,n)  = logPlus( (fM(t-1,n) + e) + e1 , (fE(t-1,n) + e) + e2 ), e1 = log 1 = 0   :147-149
//   LPM      = (fM + bM) - Zb,  LPE = (fE + bE) - Zb                              :222
//   vM(t,n)  = vE(t-1,n-1) + LPM ;  vE(t,n) = max(vM(t-1,n), vE(t-1,n)) + LPE     :360-361
//   bit(t,n) = ( vE(t,n) == vM(t-1,n) + LPE )                                     :448
// Masks: none per row. Emission -inf for k-mer-less columns covers n < 1 and n >= N, and a slot
// carries its column's k-mer parameters only while the column is inside the band (hand-over in the
// window-move block that looks one row ahead), so e = -inf empties every out-of-band slot.
// ---------------------------------------------------------------------------------------------
// INPLACE (only with POST): see the comment at lat_lp below
template <bool POST, bool INPLACE>
__global__ DYN_ONE_WAVE_PER_SIMD void k_forward(const ReadDesc* __restrict__ descs,
                                                 const double* __restrict__ sig,
                                                 const Emis* __restrict__ par,
                                                 const double* __restrict__ ws_rd,
                                                 float* __restrict__ lp_out,
                                                 uint64_t* __restrict__ bits,
                                                 ReadState* __restrict__ st, double m1,
                                                 double e2, const SoftplusNode* __restrict__ sp_tab,
                                                 int n_reads) {
  __shared__ __attribute__((aligned(16))) SoftplusNode s_tab[SP_NODES];
  const int ridx = stage_table_and_pick_read(s_tab, sp_tab, n_reads);
  if (ridx < 0) return;
  const ReadDesc rd = descs[ridx];
  const int lane = threadIdx.x & 63;
  const int T = (int)rd.T, N = (int)rd.N, bw = (int)rd.bw, W = 2 * bw + 1;
  const double ratio = rd.ratio;
  const double* __restrict__ sg = sig + rd.sig_off;
  const Emis* __restrict__ pr = par + rd.par_off;
  // Two layouts for the log-posteriors k_trace needs (one cell per row):
  //  * separate (default): the bE rows stay intact and ONE float per slot, LPE, goes to its own array
  //    (same [row][row_pos] indexing); the ~10 % M cells of the path (segment starts) get LPM rebuilt
  //    from LPE of the diagonal predecessor and two bE cells (k_mpost). 12 B per slot per row here.
  //  * INPLACE: (float LPM, float LPE) overwrite the bE slot of row t while rows t+1.. are still
  //    being read (lp_out then aliases ws_rd; no address is read after it has been written within one
  //    launch). 16 B per slot per row make this kernel HBM-bound, but the footprint is 8 instead of
  //    12 B per slot: the host picks it when the separate layout would not fit the batch in one launch.
  // ws_rd and lp_out are separate __restrict__ parameters on purpose: with a pointer derived from
  // the load pointer hipcc orders every prefetch behind the previous row's stores (s_waitcnt vmcnt(0)
  // at the top of each row = 43 % of the wave's lifetime spent waiting).
  const double* __restrict__ lat = ws_rd + rd.ws_off + lane;
  float* __restrict__ lat_lp = lp_out + (INPLACE ? 2 : 1) * (rd.ws_off + lane);
  uint64_t* __restrict__ bt = bits + rd.bits_off;
  const double Z = POST ? st[rd.read].Zb : 0.0;

  // Band edges without per-row masks. Lower edge: the slot of a column that leaves the band is
  // handed to column lo+P. Upper edge: a slot carries the k-mer parameters of its column only from
  // the row in which the column ENTERS the band; before that it carries the "no k-mer" parameters
  // (log_norm = -inf), hence e = -inf, hence fM = fE = LPM = LPE = vM = vE = -inf in that slot with
  // no select in the row loop. Both hand-overs happen in the rare block that looks one row ahead.
  int lo = band_mid(1, ratio) - bw;  // band of row 1 (column 0 of row 0 is inside: band_mid(1) <= 1 <= bw)
  int n[CPL];
  double fM[CPL], fE[CPL], e[CPL];
  double vM[CPL], vE[CPL], bcur[CPL], bnext[CPL];
  EmisV<CPL> p;
  const double x0 = sg[0];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int slot = lane * CPL + j;
    n[j] = lo + pmod(slot - lo);
    p.set(j, load_emis(pr, n[j], (n[j] <= lo + W - 1) ? N : 0));
    fE[j] = (n[j] == 0) ? 0.0 : NEG_INF;  // E[bandwidth+1] = 0 (NT_aligner_api.cpp:120)
    fM[j] = NEG_INF;
    if (POST) {
      vE[j] = fE[j];                      // E[bandwidth+1] = 0 (NT_aligner_api.cpp:336)
      vM[j] = NEG_INF;
      bcur[j] = lat[(size_t)1 * P + j * 64];
      bnext[j] = NEG_INF;
    }
  }
  log_normal_pdf_vec<CPL>(x0, p, e);  // e(1, n)
  // ring prologue: rows 2 .. RING_D+1 (row r lives in ring slot r % RING_D)
  __shared__ __attribute__((aligned(16))) double s_ring[DYN_READS_PER_GROUP][RING_D][P];
  const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)&s_ring[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))][0][0];
  const unsigned ring_lane = ring_base + lane * 8;
  const double* __restrict__ dma_src = ws_rd + rd.ws_off + lane * 2;
  if (POST) {  // rows past T repeat the all -inf row T (k_backward): RING_D rows are always in flight
    for (int r = 2; r <= RING_D + 1; ++r) ring_dma_row(dma_src + (size_t)min(r, T) * P, ring_base + (r % RING_D) * ROW_BYTES);
  }

  for (int tb = 1; tb < T; tb += 64) {
    const int idx = tb + lane;  // sig[t] is the sample of lattice row t+1
    const double xs = (idx < T - 1) ? sg[idx] : 0.0;
    const int iend = min(64, T - tb);
#pragma unroll 1
    for (int i = 0; i < iend; ++i) {
      const int t = tb + i;
      const double xn = readlane_f64(xs, i);
      double fEl[CPL], vEl[CPL];
      if (POST) {
        // bE(t+1, .) from the ring; its slot is then refilled with row t+1+RING_D (clamped to the
        // -inf row T, so that the same number of memory operations is in flight in every row and
        // one hand-counted s_waitcnt serves the whole loop, tail included)
        wait_vmcnt<RING_WAIT>();
        ring_read_row(ring_lane + ((t + 1) % RING_D) * ROW_BYTES, bnext);
        ring_dma_row(dma_src + (size_t)min(t + 1 + RING_D, T) * P, ring_base + ((t + 1) % RING_D) * ROW_BYTES);
      }
      from_left(fE, fEl);
      if (POST) from_left(vE, vEl);
      const int next_lo = band_mid(t + 1, ratio) - bw;
      if (next_lo != lo) {  // wave-uniform: the window moves up by one column between rows t and t+1
        // uniform addresses -> scalar loads (see k_backward)
        const Emis none = load_emis(pr, 0, 0);
        const Emis entering = load_emis(pr, lo + W, N);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          // Column lo is in the band for the last time in this row; its slot becomes column lo+P. n and
          // p are only read by the emission of row t+1 below, which must already be -inf here; e(t, lo)
          // and fE/vE(t-1, lo) stay where this row and the right neighbour (fEl/vEl) still read them.
          const bool leaves = n[j] == lo;
          n[j] = leaves ? lo + P : n[j];
          if (leaves) p.set(j, none);
          if (n[j] == lo + W) p.set(j, entering);  // first band row of this column is t+1
        }
        lo = next_lo;
      }
      double a1[CPL], a2[CPL], fMn[CPL], fEn[CPL], en[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) fMn[j] = (fEl[j] + e[j]) + m1;
#pragma unroll
      for (int j = 0; j < CPL; ++j) a1[j] = fM[j] + e[j];
#pragma unroll
      for (int j = 0; j < CPL; ++j) a2[j] = (fE[j] + e[j]) + e2;
      SoftplusLookup<CPL> L;
      log_plus_issue<CPL>(a1, a2, L, s_tab);
      log_normal_pdf_vec<CPL>(xn, p, en);  // e(t+1, n): independent work under the LDS latency
      log_plus_finish<CPL>(L, fEn);
      if (POST) {
        double LPM[CPL], LPE[CPL], vMn[CPL], vEn[CPL], alt[CPL];
        uint64_t bj[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) LPM[j] = (fMn[j] + (bnext[j] + en[j])) - Z;  // bM(t,n) = bE(t+1,n) + e(t+1,n)
#pragma unroll
        for (int j = 0; j < CPL; ++j) LPE[j] = (fEn[j] + bcur[j]) - Z;
#pragma unroll
        for (int j = 0; j < CPL; ++j) vMn[j] = vEl[j] + LPM[j];
#pragma unroll
        for (int j = 0; j < CPL; ++j) vEn[j] = fmax(vM[j], vE[j]) + LPE[j];
#pragma unroll
        for (int j = 0; j < CPL; ++j) alt[j] = vM[j] + LPE[j];
#pragma unroll
        for (int j = 0; j < CPL; ++j) bj[j] = __ballot(vEn[j] == alt[j]);
        // lane j keeps ballot j: two v_writelane per word (a select chain costs twice as much). All 14
        // sit in ONE asm block behind an s_nop: the ballots are SGPR pairs written by v_cmp (VALU), and
        // a v_writelane that reads such an SGPR too soon after the v_cmp picks up stale data -- the
        // compiler's hazard recogniser does not look inside inline asm (observed: a build that happened
        // to schedule v_cmp two instructions ahead of the v_writelane produced garbage decision bits).
        unsigned wlo, whi;
        static_assert(CPL == 7, "one v_writelane pair per cell register");
#define DYN_WL(J, LO, HI) "v_writelane_b32 %0, %" #LO ", " #J "\n\tv_writelane_b32 %1, %" #HI ", " #J "\n\t"
        asm("s_nop 4\n\t"
            DYN_WL(0, 2, 3) DYN_WL(1, 4, 5) DYN_WL(2, 6, 7) DYN_WL(3, 8, 9) DYN_WL(4, 10, 11) DYN_WL(5, 12, 13) DYN_WL(6, 14, 15)
            : "=&v"(wlo), "=&v"(whi)
            : "s"((unsigned)bj[0]), "s"((unsigned)(bj[0] >> 32)), "s"((unsigned)bj[1]), "s"((unsigned)(bj[1] >> 32)),
              "s"((unsigned)bj[2]), "s"((unsigned)(bj[2] >> 32)), "s"((unsigned)bj[3]), "s"((unsigned)(bj[3] >> 32)),
              "s"((unsigned)bj[4]), "s"((unsigned)(bj[4] >> 32)), "s"((unsigned)bj[5]), "s"((unsigned)(bj[5] >> 32)),
              "s"((unsigned)bj[6]), "s"((unsigned)(bj[6] >> 32)));
#undef DYN_WL
        const uint64_t mybits = ((uint64_t)whi << 32) | wlo;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {  // read again only by k_trace, one cell per row: non-temporal
          if (INPLACE) {
            typedef float dyn_f2 __attribute__((ext_vector_type(2)));
            dyn_f2 v2;
            v2.x = (float)LPM[j];
            v2.y = (float)LPE[j];
            __builtin_nontemporal_store(v2, reinterpret_cast<dyn_f2*>(&lat_lp[2 * ((size_t)t * P + j * 64)]));
          } else {
            __builtin_nontemporal_store((float)LPE[j], &lat_lp[(size_t)t * P + j * 64]);
          }
        }
        if (lane < CPL) bt[(size_t)t * CPL + lane] = mybits;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          vM[j] = vMn[j];
          vE[j] = vEn[j];
          bcur[j] = bnext[j];
        }
      }
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        fM[j] = fMn[j];
        fE[j] = fEn[j];
        e[j] = en[j];
      }
    }
  }
  if (POST) wait_vmcnt<0>();  // the clamped tail DMAs still target this wave's LDS ring
  // Zf = forwardE[T*B - bandwidth - 2] = fE(T-1, mid(T-1))  (NT_aligner_api.cpp:285)
  const int nf = band_mid(T - 1, ratio);
  const int sf = pmod(nf);
#pragma unroll
  for (int j = 0; j < CPL; ++j)
    if (j == sf % CPL && lane == sf / CPL) st[rd.read].Zf = fE[j];
}

// ---------------------------------------------------------------------------------------------
// Z check of align()/train() (NT_aligner_api.cpp:285-291 / 619-625) for the calc=false path.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool z_ok(const ReadDesc& rd, const ReadState& s) {
  const double size = (double)((uint64_t)rd.T * (uint64_t)(2 * rd.bw + 3));
  if (isinf(s.Zf) || isinf(s.Zb)) return false;
  return !(fabs(s.Zf - s.Zb) / size > 1e-8);
}

__global__ void k_zcheck(const ReadDesc* __restrict__ descs, int n_reads,
                         ReadState* __restrict__ st, int fail_status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_reads) return;
  const ReadDesc rd = descs[i];
  ReadState s = st[rd.read];
  if (s.status != 0) return;
  if (!z_ok(rd, s)) st[rd.read].status = fail_status;
  st[rd.read].n_segments = 0;
}

// ---------------------------------------------------------------------------------------------
// K_trace: decodeMAP (NT_aligner_api.cpp:383-456). Start in state E at (T-1, N-1); walk the
// decision bits. Each lattice row 1..T-1 holds exactly one path cell, so the walk is recorded
// as pathn[row] (column, bit31 = state M) and pp[row] = exp(LP of that cell); a segment is the
// run of rows that share a column, its M cell is the lowest row (segrow).
// 64 rows of bits are staged in LDS per step so the serial walk pays LDS, not HBM, latency.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_trace(const ReadDesc* __restrict__ descs,
                                               const float* __restrict__ lpe,
                                               const uint64_t* __restrict__ bits,
                                               ReadState* __restrict__ st, TraceBuffers tb,
                                               int fail_status, int inplace) {
  __shared__ uint64_t sb[64 * CPL];
  const ReadDesc rd = descs[blockIdx.x];
  const int lane = threadIdx.x;
  ReadState s = st[rd.read];
  if (s.status != 0) return;
  if (!z_ok(rd, s)) {
    if (lane == 0) {
      st[rd.read].status = fail_status;
      st[rd.read].n_segments = 0;
    }
    return;
  }
  const int T = (int)rd.T, N = (int)rd.N;
  // separate layout: float LPE rows [T][P]; in-place layout: (float LPM, float LPE) in the 8-byte bE slots
  const float* __restrict__ lp = lpe + (inplace ? 2 : 1) * rd.ws_off;
  const uint64_t* __restrict__ bt = bits + rd.bits_off;
  double* __restrict__ pp = tb.pp + rd.path_off;
  uint32_t* __restrict__ pathn = tb.pathn + rd.path_off;
  uint32_t* __restrict__ segrow = tb.segrow + rd.seg_off;

  // Segment-at-a-time walk. Lane i owns row base+i of the current 64-row block. While the path
  // stays in column n every row's decision is bit(row, slot(n)), which all 64 lanes test at once;
  // the highest row <= t whose bit is set is where the path turns (E with bit 1 -> M one row
  // below -> E in column n-1 two rows below), found with one ballot + find-first-set. A block
  // costs one step per segment that crosses it (~6) instead of one per row (64).
  int t = T - 1, n = N - 1;
  int slot = n % P;
  int stM = 0;  // the cell at row t is an M cell (carried across blocks)
  while (t > 0 && n > 0) {
    const int base = t - 63;
    const int row = base + lane;
    if (row >= 1) {
#pragma unroll
      for (int j = 0; j < CPL; ++j) sb[lane * CPL + j] = bt[(size_t)row * CPL + j];
    }
    __syncthreads();
    int my_n = 0, my_slot = 0, my_st = -1;
    const int block_lo = base < 1 ? 1 : base;  // lowest row of this block that exists
    while (t >= block_lo && n > 0) {
      if (stM) {  // M(t,n): segment start; continue with E(t-1, n-1)
        if (row == t) {
          my_n = n;
          my_slot = slot;
          my_st = 1;
        }
        --n;
        slot = slot ? slot - 1 : P - 1;
        stM = 0;
        --t;
        continue;
      }
      // state E in column n at row t: every lane tests its own row for this column
      const uint64_t w = sb[lane * CPL + (slot % CPL)];  // ballot word of cell index j = slot % CPL
      const bool bit = (row >= block_lo) && (row <= t) && ((w >> (slot / CPL)) & 1);  // lane = slot / CPL
      const uint64_t m = __ballot(bit);
      const int r = m ? base + (63 - __builtin_clzll(m)) : block_lo - 1;  // highest turning row, or none
      const int e_lo = m ? r : block_lo;  // rows e_lo..t are E cells of column n
      if (row >= e_lo && row <= t) {
        my_n = n;
        my_slot = slot;
        my_st = 0;
      }
      if (m) {
        t = r - 1;  // M cell of column n (may lie in the next block: stM carries over)
        stM = 1;
      } else {
        t = block_lo - 1;
      }
    }
    if (my_st >= 0) {
      // E cell of the path: k_forward stored its log-posterior. M cells (segment starts, ~1 row in 10)
      // get theirs from k_mpost, one thread per segment, instead of ~6 dependent loads in this walk.
      const size_t cell = (size_t)row * P + row_pos(my_slot);
      if (inplace) pp[row] = exp((double)lp[2 * cell + (my_st ? 0 : 1)]);
      else if (my_st == 0) pp[row] = exp((double)lp[cell]);
      pathn[row] = (uint32_t)my_n | (my_st ? 0x80000000u : 0u);
      if (my_st) segrow[my_n - 1] = (uint32_t)row;
    }
    __syncthreads();
  }
  if (lane == 0) {
    const bool complete = (t == 0 && n == 0);
    st[rd.read].status = complete ? 0 : 7;
    st[rd.read].n_segments = complete ? (uint32_t)(N - 1) : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// K_mpost: posterior of the M cell (segment start) of every segment. k_forward stores only LPE; the
// M cell (row, n) is preceded on the path by the E cell (row-1, n-1), so
//   fE(row-1, n-1) = LPE(row-1, n-1) - bE(row-1, n-1) + Zb      (NT_aligner_api.cpp:222 solved for fE)
//   fM(row, n)     = (fE(row-1, n-1) + e(row, n)) + m1            (:146)
//   bM(row, n)     = bE(row+1, n) + e(row+1, n)                   (:200; row T of the workspace is -inf)
//   LPM(row, n)    = (fM + bM) - Zb                               (:222)
// The float LPE costs <= 6e-8 * |LPE| here; every other term is fp64. One thread per segment.
// ---------------------------------------------------------------------------------------------
__global__ void k_mpost(const ReadDesc* __restrict__ descs, const double* __restrict__ ws,
                        const float* __restrict__ lpe, const double* __restrict__ sig,
                        const Emis* __restrict__ par, const ReadState* __restrict__ st, TraceBuffers tb,
                        double m1) {
  const ReadDesc rd = descs[blockIdx.y];
  const ReadState s = st[rd.read];
  if (s.status != 0) return;
  const int T = (int)rd.T, N = (int)rd.N;
  const int n = blockIdx.x * blockDim.x + threadIdx.x + 1;  // lattice column of the segment, 1 .. N-1
  if (n >= N) return;
  const double* __restrict__ bE = ws + rd.ws_off;
  const float* __restrict__ lp = lpe + rd.ws_off;
  const double* __restrict__ sg = sig + rd.sig_off;
  const int row = (int)tb.segrow[rd.seg_off + n - 1];
  const int slot = n % P, pslot = (n - 1) % P;
  const size_t pcell = (size_t)(row - 1) * P + row_pos(pslot);
  const double fE_prev = row == 1 ? (n == 1 ? 0.0 : NEG_INF)  // fE(0, 0) = 0, nothing else in row 0 (:120)
                                  : ((double)lp[pcell] - bE[pcell]) + s.Zb;
  const Emis em = par[rd.par_off + n - 1];
  const double e_here = dynmath::log_normal_pdf(sg[row - 1], em);
  const double e_next = row + 1 < T ? dynmath::log_normal_pdf(sg[row], em) : NEG_INF;
  const double fM = (fE_prev + e_here) + m1;
  const double bM = bE[(size_t)(row + 1) * P + row_pos(slot)] + e_next;
  tb.pp[rd.path_off + row] = exp((fM + bM) - s.Zb);
}

// ---------------------------------------------------------------------------------------------
// K_median: formattedMedian (aligner.cpp:247-263) by rank counting. One thread per path row;
// the segment of column n spans rows [segrow[n-1], segrow[n]) (last column: up to T-1).
// Ties are broken by row so ranks are a permutation.
// ---------------------------------------------------------------------------------------------
__global__ void k_median(const ReadDesc* __restrict__ descs, const ReadState* __restrict__ st,
                         TraceBuffers tb) {
  const ReadDesc rd = descs[blockIdx.y];
  if (st[rd.read].status != 0) return;
  const int T = (int)rd.T, N = (int)rd.N;
  const int t = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (t >= T) return;
  const double* __restrict__ pp = tb.pp + rd.path_off;
  const uint32_t* __restrict__ segrow = tb.segrow + rd.seg_off;
  const int n = (int)(tb.pathn[rd.path_off + t] & 0x7fffffffu);
  const int a = (int)segrow[n - 1];
  const int b = (n < N - 1) ? (int)segrow[n] : T;
  const int L = b - a;
  const double x = pp[t];
  int rank = 0;
  for (int u = a; u < b; ++u) {
    const double y = pp[u];
    rank += (y < x) || (y == x && u < t);
  }
  const int mid = L >> 1;
  if (rank == mid) tb.med_hi[rd.seg_off + n - 1] = x;
  if (!(L & 1) && rank == mid - 1) tb.med_lo[rd.seg_off + n - 1] = x;
}

// K_final: one output row per segment (NT_aligner_api.cpp:420-430).
__global__ void k_final(const ReadDesc* __restrict__ descs, const ReadState* __restrict__ st,
                        TraceBuffers tb, SegRow* __restrict__ rows, int kmer_size) {
  const ReadDesc rd = descs[blockIdx.y];
  if (st[rd.read].status != 0) return;
  const int T = (int)rd.T, N = (int)rd.N;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // segment index = column - 1
  if (i >= N - 1) return;
  const uint32_t* __restrict__ segrow = tb.segrow + rd.seg_off;
  const int a = (int)segrow[i];
  const int b = (i + 1 < N - 1) ? (int)segrow[i + 1] : T;
  const int L = b - a;
  const double hi = tb.med_hi[rd.seg_off + i];
  SegRow r;
  r.signal_pos = (uint32_t)(a - 1);
  r.sequence_pos = (uint32_t)(i + kmer_size / 2);
  r.probability = (L & 1) ? hi : (tb.med_lo[rd.seg_off + i] + hi) / 2.0;
  rows[rd.seg_off + i] = r;
}

// ---------------------------------------------------------------------------------------------
// K_fwd_train: forward fused with the Baum-Welch statistics of runTraining / trainTransition.
// Per lattice cell (t >= 1, n >= 1), with gamma_M = exp(LPM), gamma_E = exp(LPE):
//   w[kmer] += gamma_M + gamma_E ; s1 += gamma*x ; s2 += gamma*x*x      NT_aligner_api.cpp:505-512
// Transition expectations (:683,:690) are rewritten through the recursions they sum over:
//   fE(t,n)+m1+e(t+1,n+1)+bM(t+1,n+1) = fM(t+1,n+1)+bM(t+1,n+1)  ->  sum exp(LPM) over cells
//   fE(t,n)+e2+e(t+1,n)+bE(t+1,n)     = second logPlus operand of fE(t+1,n) + bE(t+1,n)
// so both are plain sums of per-cell posteriors (linear domain, relative to Zb).
// Column sums stay in registers while the column is in the band and are flushed once per column.
// ---------------------------------------------------------------------------------------------
__global__ DYN_ONE_WAVE_PER_SIMD void k_forward_train(const ReadDesc* __restrict__ descs,
                                                       const double* __restrict__ sig,
                                                       const Emis* __restrict__ par,
                                                       const double* __restrict__ ws,
                                                       ReadState* __restrict__ st,
                                                       TrainBuffers tb, double m1, double e2,
                                                       const SoftplusNode* __restrict__ sp_tab, int n_reads) {
  __shared__ __attribute__((aligned(16))) SoftplusNode s_tab[SP_NODES];
  const int ridx = stage_table_and_pick_read(s_tab, sp_tab, n_reads);
  if (ridx < 0) return;
  const ReadDesc rd = descs[ridx];
  const int lane = threadIdx.x & 63;
  const int T = (int)rd.T, N = (int)rd.N, bw = (int)rd.bw, W = 2 * bw + 1;
  const double ratio = rd.ratio;
  const double* __restrict__ sg = sig + rd.sig_off;
  const Emis* __restrict__ pr = par + rd.par_off;
  const double* __restrict__ lat = ws + rd.ws_off + lane;
  double* __restrict__ cw = tb.col_w + rd.par_off;
  double* __restrict__ cs1 = tb.col_s1 + rd.par_off;
  double* __restrict__ cs2 = tb.col_s2 + rd.par_off;
  const double Z = st[rd.read].Zb;

  int lo = band_mid(0, ratio) - bw;
  int n[CPL];
  double fM[CPL], fE[CPL], e[CPL], bcur[CPL], bnext[CPL];
  double aw[CPL], a1[CPL], a2[CPL];
  EmisV<CPL> p;
  double sumM = 0.0, sumE2 = 0.0;
  const double x0 = sg[0];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int slot = lane * CPL + j;
    n[j] = lo + pmod(slot - lo);
    p.set(j, load_emis(pr, n[j], N));
    fE[j] = (n[j] == 0) ? 0.0 : NEG_INF;
    fM[j] = NEG_INF;
    bcur[j] = lat[(size_t)1 * P + j * 64];
    bnext[j] = (T > 2) ? lat[(size_t)2 * P + j * 64] : NEG_INF;
    aw[j] = a1[j] = a2[j] = 0.0;
  }
  log_normal_pdf_vec<CPL>(x0, p, e);

  double xcur = x0;  // sample of row t
  for (int tb0 = 1; tb0 < T; tb0 += 64) {
    const int idx = tb0 + lane;
    const double xs = (idx < T - 1) ? sg[idx] : 0.0;
    const int iend = min(64, T - tb0);
#pragma unroll 1
    for (int i = 0; i < iend; ++i) {
      const int t = tb0 + i;
      const double xn = readlane_f64(xs, i);
      double fEl[CPL], bnn[CPL];
      from_left(fE, fEl);
      const bool have = (t + 2 < T);
#pragma unroll
      for (int j = 0; j < CPL; ++j) bnn[j] = have ? lat[(size_t)(t + 2) * P + j * 64] : NEG_INF;
      const int new_lo = band_mid(t, ratio) - bw;
      if (new_lo != lo) {
        const Emis fresh = load_emis(pr, lo + P, N);  // uniform address -> scalar load
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          if (n[j] == lo) {
            if (n[j] >= 1 && n[j] < N) {  // column leaves the band for good: flush its sums
              cw[n[j] - 1] = aw[j];
              cs1[n[j] - 1] = a1[j];
              cs2[n[j] - 1] = a2[j];
            }
            aw[j] = a1[j] = a2[j] = 0.0;
            n[j] = lo + P;
            p.set(j, fresh);
            e[j] = NEG_INF;  // empties the slot (see k_forward)
          }
        }
        lo = new_lo;
      }
      const int hi_n = lo + W - 1;
      double x1[CPL], op2[CPL], fMn[CPL], fEn[CPL], en[CPL], aM[CPL], aE[CPL], aT[CPL], gM[CPL], gE[CPL], gT[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) fEl[j] = (n[j] <= hi_n) ? fEl[j] : NEG_INF;
#pragma unroll
      for (int j = 0; j < CPL; ++j) fMn[j] = (fEl[j] + e[j]) + m1;
#pragma unroll
      for (int j = 0; j < CPL; ++j) x1[j] = fM[j] + e[j];
#pragma unroll
      for (int j = 0; j < CPL; ++j) op2[j] = (fE[j] + e[j]) + e2;
      SoftplusLookup<CPL> L;
      log_plus_issue<CPL>(x1, op2, L, s_tab);
      log_normal_pdf_vec<CPL>(xn, p, en);
      log_plus_finish<CPL>(L, fEn);
      // posteriors: state M, state E, and the E->E transition into (t,n)  (see header comment)
#pragma unroll
      for (int j = 0; j < CPL; ++j) aM[j] = (fMn[j] + (bnext[j] + en[j])) - Z;
#pragma unroll
      for (int j = 0; j < CPL; ++j) aE[j] = (fEn[j] + bcur[j]) - Z;
#pragma unroll
      for (int j = 0; j < CPL; ++j) aT[j] = (op2[j] + bcur[j]) - Z;
      dynmath::exp_vec<CPL>(aM, gM);
      dynmath::exp_vec<CPL>(aE, gE);
      dynmath::exp_vec<CPL>(aT, gT);
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const double g = gM[j] + gE[j];
        aw[j] += g;
        a1[j] += g * xcur;
        a2[j] += g * xcur * xcur;
        sumM += gM[j];
        sumE2 += gT[j];
        fM[j] = fMn[j];
        fE[j] = fEn[j];
        e[j] = en[j];
        bcur[j] = bnext[j];
        bnext[j] = bnn[j];
      }
      xcur = xn;
    }
  }
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    if (n[j] >= 1 && n[j] < N) {
      cw[n[j] - 1] = aw[j];
      cs1[n[j] - 1] = a1[j];
      cs2[n[j] - 1] = a2[j];
    }
  }
  // wave reduction of the two transition sums
  for (int off = 32; off >= 1; off >>= 1) {
    sumM += __shfl_xor(sumM, off);
    sumE2 += __shfl_xor(sumE2, off);
  }
  const int nf = band_mid(T - 1, ratio);
  const int sf = pmod(nf);
#pragma unroll
  for (int j = 0; j < CPL; ++j)
    if (j == sf % CPL && lane == sf / CPL) st[rd.read].Zf = fE[j];
  if (lane == 0) {
    tb.trans[2 * rd.read] = sumM;
    tb.trans[2 * rd.read + 1] = sumE2;
  }
}

// ---------------------------------------------------------------------------------------------
// P1/P2 on the device: the per-read preprocessing of segment.py:146-153 / train.py:163-170.
//   x = REAL(raw[i]); x -= shift; x /= scale            (REAL = double for dynamont-resquiggle,
//   hampel(x, W, n_sigmas)   (utils.py:16-43)             float for dynamont-train, whose signal
// Windows are taken from the UNFILTERED normalised signal; centres W/2 .. W/2 + nwin - 1 with        stays float32)
// nwin = S - W - (W even); a centre is replaced by the window median when
// |x - med| > n_sigmas * (1.4826 * MAD). Every operation is a single IEEE op in the reference's
// order, so the result is bit-identical to the NumPy code (tests/test_gpu_preprocess.py).
// ---------------------------------------------------------------------------------------------
template <class REAL, class RAW>
__global__ void k_normalise(const RAW* __restrict__ raw, const uint64_t* __restrict__ offs,
                            const double* __restrict__ shift, const double* __restrict__ scale,
                            REAL* __restrict__ norm, int n_reads) {
  const int r = blockIdx.y;
  if (r >= n_reads) return;
  const uint64_t a = offs[r], b = offs[r + 1];
  const REAL sh = (REAL)shift[r], sc = (REAL)scale[r];
  for (uint64_t i = a + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < b; i += (uint64_t)gridDim.x * blockDim.x) {
    REAL x = (REAL)raw[i];
    x = x - sh;
    norm[i] = x / sc;
  }
}

template <class REAL, int MAXW>
__device__ __forceinline__ REAL window_median(REAL (&w)[MAXW], int W) {
  for (int i = 1; i < W; ++i) {  // insertion sort of <= MAXW values
    const REAL v = w[i];
    int j = i - 1;
    while (j >= 0 && w[j] > v) {
      w[j + 1] = w[j];
      --j;
    }
    w[j + 1] = v;
  }
  return (W & 1) ? w[W / 2] : (w[W / 2 - 1] + w[W / 2]) / (REAL)2;
}

template <class REAL>
__global__ void k_hampel(const REAL* __restrict__ norm, const uint64_t* __restrict__ offs,
                         double* __restrict__ out, int n_reads, int W, double n_sigmas) {
  constexpr int MAXW = 16;
  const int r = blockIdx.y;
  if (r >= n_reads) return;
  const uint64_t a = offs[r], S = offs[r + 1] - offs[r];
  const long nwin = (S <= (uint64_t)W) ? 0 : (long)S - W - ((W & 1) ? 0 : 1);
  const int half = W / 2;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (uint64_t)gridDim.x * blockDim.x) {
    REAL x = norm[a + i];
    const long wi = (long)i - half;  // window index of this centre
    if (wi >= 0 && wi < nwin) {
      REAL w[MAXW], d[MAXW];
      for (int q = 0; q < W; ++q) w[q] = norm[a + wi + q];
      for (int q = 0; q < W; ++q) d[q] = w[q];
      const REAL med = window_median<REAL, MAXW>(d, W);
      for (int q = 0; q < W; ++q) d[q] = fabs(w[q] - med);
      const REAL mad = window_median<REAL, MAXW>(d, W);
      const REAL sigma = (REAL)1.4826 * mad;
      if (fabs(x - med) > (REAL)n_sigmas * sigma) x = med;
    }
    out[a + i] = (double)x;
  }
}

template <class REAL, class RAW>
static void preprocess_t(const RAW* raw, const uint64_t* offs, const double* shift, const double* scale,
                         void* norm_tmp, double* out, int n_reads, uint64_t max_len, int W, double ns,
                         hipStream_t s) {
  const int bx = (int)std::min<uint64_t>(1024, (max_len + 255) / 256);
  hipLaunchKernelGGL((k_normalise<REAL, RAW>), dim3(bx ? bx : 1, n_reads), dim3(256), 0, s, raw, offs, shift, scale,
                     (REAL*)norm_tmp, n_reads);
  hipLaunchKernelGGL((k_hampel<REAL>), dim3(bx ? bx : 1, n_reads), dim3(256), 0, s, (const REAL*)norm_tmp, offs, out,
                     n_reads, W, ns);
}

void launch_preprocess(const void* raw, int raw_dtype, int compute_f32, const uint64_t* offs,
                       const double* shift, const double* scale, void* norm_tmp, double* out,
                       int n_reads, uint64_t max_len, int W, double n_sigmas, hipStream_t s) {
  if (n_reads <= 0) return;
  if (compute_f32) {
    if (raw_dtype == 0) preprocess_t<float, float>((const float*)raw, offs, shift, scale, norm_tmp, out, n_reads, max_len, W, n_sigmas, s);
    else if (raw_dtype == 1) preprocess_t<float, int16_t>((const int16_t*)raw, offs, shift, scale, norm_tmp, out, n_reads, max_len, W, n_sigmas, s);
    else preprocess_t<float, double>((const double*)raw, offs, shift, scale, norm_tmp, out, n_reads, max_len, W, n_sigmas, s);
  } else {
    if (raw_dtype == 0) preprocess_t<double, float>((const float*)raw, offs, shift, scale, norm_tmp, out, n_reads, max_len, W, n_sigmas, s);
    else if (raw_dtype == 1) preprocess_t<double, int16_t>((const int16_t*)raw, offs, shift, scale, norm_tmp, out, n_reads, max_len, W, n_sigmas, s);
    else preprocess_t<double, double>((const double*)raw, offs, shift, scale, norm_tmp, out, n_reads, max_len, W, n_sigmas, s);
  }
}

// ---------------------------------------------------------------------------------------------
// Pooled sufficient statistics for the multi-GPU M-step (BASELINE.json config 5): every lattice
// column of every successfully trained read adds its (w, s1, s2) to its k-mer's bins. One fp64
// atomic per statistic per column; ~2 000 columns per read spread over 4^k bins, so contention
// is negligible. (The per-read results of dyn_batch_fetch_train are summed on the host in column
// order instead and are bitwise reproducible; this pooled form is for the all-reduce.)
// ---------------------------------------------------------------------------------------------
__global__ void k_pool_stats(const ReadDesc* __restrict__ descs, const ReadState* __restrict__ st,
                             const int32_t* __restrict__ kmers, TrainBuffers tb,
                             double* __restrict__ pooled, uint64_t num_kmers) {
  const ReadDesc rd = descs[blockIdx.y];
  if (st[rd.read].status != 0) return;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;  // column index - 1
  if (c >= (int)rd.N - 1) return;
  const uint64_t i = rd.par_off + c;
  const int32_t code = kmers[i];
  const double w = tb.col_w[i];
  if (w > 0.0) {
    atomicAdd(&pooled[code], w);
    atomicAdd(&pooled[num_kmers + code], tb.col_s1[i]);
    atomicAdd(&pooled[2 * num_kmers + code], tb.col_s2[i]);
  }
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
void launch_prep_params(const int32_t* kmers, const Emis* model, Emis* par, uint64_t total,
                        hipStream_t s) {
  if (!total) return;
  const int block = 256;
  const uint64_t want = (total + block - 1) / block;
  const int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(k_prep_params, dim3(grid), dim3(block), 0, s, kmers, model, par, total);
}

void launch_backward(const ReadDesc* descs, int n_reads, const double* sig, const Emis* par,
                     double* ws, ReadState* st, double m1, double e2, bool store,
                     const SoftplusNode* sp_tab, hipStream_t s) {
  if (n_reads <= 0) return;
  const dim3 grid((n_reads + DYN_READS_PER_GROUP - 1) / DYN_READS_PER_GROUP), block(256);
  if (store)
    hipLaunchKernelGGL(k_backward<true>, grid, block, 0, s, descs, sig, par, ws, st, m1, e2, sp_tab, n_reads);
  else
    hipLaunchKernelGGL(k_backward<false>, grid, block, 0, s, descs, sig, par, ws, st, m1, e2, sp_tab, n_reads);
}

void launch_forward(const ReadDesc* descs, int n_reads, const double* sig, const Emis* par,
                    double* ws, float* lpe, uint64_t* bits, ReadState* st, double m1, double e2, bool post,
                    const SoftplusNode* sp_tab, hipStream_t s) {
  if (n_reads <= 0) return;
  const dim3 grid((n_reads + DYN_READS_PER_GROUP - 1) / DYN_READS_PER_GROUP), block(256);
  const double* rd = ws;
  if (!post)
    hipLaunchKernelGGL((k_forward<false, false>), grid, block, 0, s, descs, sig, par, rd, (float*)nullptr, bits, st, m1, e2, sp_tab, n_reads);
  else if (lpe)
    hipLaunchKernelGGL((k_forward<true, false>), grid, block, 0, s, descs, sig, par, rd, lpe, bits, st, m1, e2, sp_tab, n_reads);
  else  // in place: the log-posteriors overwrite the bE rows
    hipLaunchKernelGGL((k_forward<true, true>), grid, block, 0, s, descs, sig, par, rd, reinterpret_cast<float*>(ws), bits, st, m1, e2, sp_tab, n_reads);
}

void launch_forward_train(const ReadDesc* descs, int n_reads, const double* sig, const Emis* par,
                          const double* ws, ReadState* st, TrainBuffers tb, double m1, double e2,
                          const SoftplusNode* sp_tab, hipStream_t s) {
  if (n_reads <= 0) return;
  const dim3 grid((n_reads + DYN_READS_PER_GROUP - 1) / DYN_READS_PER_GROUP), block(256);
  hipLaunchKernelGGL(k_forward_train, grid, block, 0, s, descs, sig, par, ws, st, tb, m1, e2, sp_tab, n_reads);
}

void launch_pool_stats(const ReadDesc* descs, int n_reads, uint32_t max_N, const ReadState* st,
                       const int32_t* kmers, TrainBuffers tb, double* pooled, uint64_t num_kmers,
                       hipStream_t s) {
  if (n_reads <= 0) return;
  hipLaunchKernelGGL(k_pool_stats, dim3((max_N + 255) / 256, n_reads), dim3(256), 0, s, descs, st, kmers, tb,
                     pooled, num_kmers);
}

void launch_zcheck(const ReadDesc* descs, int n_reads, ReadState* st, int z_fail_status,
                   hipStream_t s) {
  if (n_reads <= 0) return;
  hipLaunchKernelGGL(k_zcheck, dim3((n_reads + 255) / 256), dim3(256), 0, s, descs, n_reads, st,
                     z_fail_status);
}

void launch_trace(const ReadDesc* descs, int n_reads, uint32_t max_T, uint32_t max_N,
                  const double* ws, const float* lpe, const uint64_t* bits, const double* sig, const Emis* par,
                  ReadState* st, TraceBuffers tb, SegRow* rows, int kmer_size, double m1, int z_fail_status,
                  hipStream_t s) {
  if (n_reads <= 0) return;
  if (lpe) {
    hipLaunchKernelGGL(k_trace, dim3(n_reads), dim3(64), 0, s, descs, lpe, bits, st, tb, z_fail_status, 0);
    hipLaunchKernelGGL(k_mpost, dim3((max_N + 255) / 256, n_reads), dim3(256), 0, s, descs, ws, lpe, sig, par, st, tb, m1);
  } else {
    hipLaunchKernelGGL(k_trace, dim3(n_reads), dim3(64), 0, s, descs, reinterpret_cast<const float*>(ws), bits, st, tb, z_fail_status, 1);
  }
  hipLaunchKernelGGL(k_median, dim3((max_T + 255) / 256, n_reads), dim3(256), 0, s, descs, st, tb);
  hipLaunchKernelGGL(k_final, dim3((max_N + 255) / 256, n_reads), dim3(256), 0, s, descs, st, tb, rows,
                     kmer_size);
}

}  // namespace dynk
