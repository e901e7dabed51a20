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
// Design (not a translation): the reference streams eight T x B fp64 matrices per read. Here, per read,
//   backward  walks t = T-2..0 and stores ONLY backward-E (bM(t,n) = bE(t+1,n) + e(t+1,n) is one
//             add away, NT_aligner_api.cpp:200);
//   forward   walks t = 1..T-1 and fuses forward, posterior (needs Zb, known after the backward sweep),
//             the posterior-Viterbi fill and the traceback decision
//             (vE == vM_prev + LPE, NT_aligner_api.cpp:448); forward / Viterbi rows never leave
//             registers; per cell it writes one float, LPE, for the path-probability lookup and
//             1 decision bit (footprint-limited launches: (float LPM, float LPE) over the bE slot);
//   traceback walks the decision bits back one SEGMENT per step (ballot + find-first-set inside
//             64-row LDS blocks) and rebuilds LPM for the segment starts;
//   k_median / k_final produce the per-segment median posterior and the output rows.
// HBM traffic is 20.1 B per in-band cell instead of the 64.1 B of the three-pass formulation
// (SURVEY.md §8d).
//
// Execution: ONE launch of persistent waves per batch (k_read_queue). One 64-lane wave owns one read at
// a time and runs its whole pipeline -- backward, forward, Z check, traceback -- then takes the next read
// off a queue until the batch is drained. Four waves share a 256-thread workgroup (one per SIMD of a CU)
// and nothing but the CU's LDS. Consequences:
//   * no SIMD idles behind the longest read of a launch (reads of 10 k .. 100 k samples in one batch);
//   * the lattice of a read exists only while a wave works on it: it lives in PAGES drawn from a pool
//     sized for the reads in flight (<= 1 024), so a batch of any size runs as one launch, and
//   * write-only backward sweeps and read-mostly forward sweeps of different reads overlap in time.
//
// Mapping: band slot s = n mod P lives in lane s / CPL, register s % CPL: a lane owns CPL CONSECUTIVE
// slots, so the cross-lane neighbour (n-1 forward, n+1 backward) of all but one of its cells is its
// own next register and one DPP wave rotate per exchange serves the remaining cell -- no barrier
// anywhere in the DP loops. In HBM a row stores the register PAIRS (0,1), (2,3), (4,5) of all lanes as three
// contiguous 1 024-byte runs (16 bytes per lane) and register 6 as a 512-byte run (row_pos): every row access
// is four fully coalesced operations (a 56-byte lane stride, the "natural" placement of this slot numbering,
// cost the backward sweep 20 % in partial-line writes). The CU's LDS holds the softplus table shared by the
// four waves (dp_math.hpp), a 4-row-deep ring per wave that the forward sweep fills straight from HBM
// with global_load_lds_dwordx4 (see ring_dma_row), and each wave's page table.
#include "nt_kernels.hpp"

#include <algorithm>

#include "dp_math_strict.hpp"

// Placement: the SPI packs single-wave workgroups onto one SIMD for as long as its registers
// allow (measured: a 184-VGPR build of the backward sweep ran 1 024 reads as 2 waves on each of 512
// SIMDs and took 2x the time of 512 reads; tools/ubench + DESIGN.md). The DP waves are pure fp64 issue
// streams with 7-way ILP that saturate a SIMD on their own, so they are compiled for exactly one
// wave per SIMD: the dispatcher must then spread the waves over all 1 024 SIMDs, and the whole
// 512-entry register file is available to keep the interleaved chains out of AGPR spills.
#define DYN_ONE_WAVE_PER_SIMD \
  __launch_bounds__(256 * DYN_WAVES_PER_SIMD) __attribute__((amdgpu_waves_per_eu(DYN_WAVES_PER_SIMD, DYN_WAVES_PER_SIMD)))

// FOUR waves per 256-thread workgroup (one per SIMD of a CU): they never synchronise after the
// prologue, they only share the 82 KB softplus table that the workgroup stages into the CU's LDS once
// (dp_math.hpp, softplus_table_vec).
#define DYN_WAVES_PER_GROUP (4 * DYN_WAVES_PER_SIMD)

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

typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint64_t lds_u64_t;

__device__ __forceinline__ int pmod(int a) {
  int r = a % P;
  return r < 0 ? r + P : r;
}

// position of band slot s = lane*CPL + j inside a stored row: registers (0,1), (2,3), (4,5) of a lane sit side by
// side -- 16 bytes per lane, so that a row is written with three 16-byte-per-lane stores and one 8-byte one (and
// picked up from the LDS ring with three ds_read_b128 and one ds_read_b64) instead of seven 8-byte operations.
__device__ __forceinline__ int pos_of(int lane, int j) { return j < 6 ? (j >> 1) * 128 + lane * 2 + (j & 1) : 384 + lane; }
__device__ __forceinline__ int row_pos(int s) { return pos_of(s / CPL, s % CPL); }

typedef double dyn_d2 __attribute__((ext_vector_type(2)));
typedef float dyn_f2 __attribute__((ext_vector_type(2)));
typedef float dyn_f4 __attribute__((ext_vector_type(4)));

// one row of 8-byte slots: (x[0],x[1]) (x[2],x[3]) (x[4],x[5]) as 16-byte stores, x[6] as an 8-byte store
template <bool NT>
__device__ __forceinline__ void store_row_f64(double* __restrict__ row, int lane, const double (&x)[CPL]) {
  static_assert(CPL == 7, "three register pairs and a single");
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    dyn_d2 v;
    v.x = x[2 * q];
    v.y = x[2 * q + 1];
    dyn_d2* dst = reinterpret_cast<dyn_d2*>(row + q * 128 + lane * 2);
    if (NT) __builtin_nontemporal_store(v, dst);
    else *dst = v;
  }
  if (NT) __builtin_nontemporal_store(x[6], row + 384 + lane);
  else row[384 + lane] = x[6];
}

// ---- out-of-band lanes (experiment DYN_SKIP_OOB) -------------------------------------------------
// A row holds P = 448 slots for the 2 bw + 1 <= 401 cells of the band: the other >= 47 slots -- 6.7 lanes of the blocked
// layout, more than half the wave for reads shorter than the band is wide -- carry "no k-mer" cells whose values nobody
// needs. The lanes that hold NOTHING of a window of the slot ring skip their row stores and their part of the ring DMA.
// lanes_of_window: the lanes that hold at least one slot of the lattice columns [c0, c0 + len).
__device__ __forceinline__ uint64_t lanes_of_window(int c0, int len) {
  const int s0 = pmod(c0);
  const int l0 = s0 / CPL;
  const int n = (s0 + len - 1) / CPL - l0 + 1;   // lanes touched (slots s and s - P share a lane: P = 64 CPL)
  if (n >= 64) return ~0ull;
  const uint64_t m = (1ull << n) - 1ull;
  return l0 ? ((m << l0) | (m >> (64 - l0))) : m;
}
// a wave-uniform 64-bit value the compiler holds in vector registers (everything derived from band_mid is): into scalar ones
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi32 << 32) | lo32;
}
// the half-wave instruction of a row (register 6 of lanes 2l, 2l + 1 comes from lane l < 32): lane l takes part when either does
__device__ __forceinline__ uint64_t half_mask_of(uint64_t live) {
  uint64_t x = (live | (live >> 1)) & 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
  x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
  x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
  x = (x | (x >> 16)) & 0x00000000ffffffffull;
  return x;
}
#ifdef DYN_SKIP_OOB
constexpr bool SKIP_OOB = true;
#else
constexpr bool SKIP_OOB = false;
#endif
// Rows are fetched RING_D + 1 rows ahead of the row that decides the forward sweep's mask, and the band moves at most one
// column per row: the backward sweep stores the lanes of the band widened by OOB_SLACK columns on either side, the forward
// sweep fetches those of [lo, lo + W + OOB_SLACK) -- always inside what was stored, always containing the fetched row's band.
constexpr int OOB_SLACK = 5;

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
    e.neg_log_stdev = NEG_INF;
    e.stdev = 1.0;
  }
  return e;
}

// Arithmetic flavours of the sweeps.
//  ARITH_DEFAULT  dp_math.hpp: 5-operation emission, table softplus (<= 1 ulp from glibc)
//  ARITH_STRICT   dp_math_strict.hpp, "certified arithmetic": the reference's bits. The emission's quotient is formed exactly
//                 from 1/stdev and stdev (one multiplication, four FMAs, no division); a logPlus is the table softplus plus
//                 a rounding certificate (5 operations), and only the registers that hold an AMBIGUOUS sum -- one whose
//                 rounding could depend on the last bits of glibc's log1p(exp()) -- are recomputed with the restated glibc
//  ARITH_FOLDED   train() only (its backward sweep; the forward sweep is the posterior chain): emission ln K - u^2 in two
//                 fused operations, softplus polynomial of degree 3. No integer decision hangs on the last bits there,
//                 unlike in the align sweeps: train()'s outputs are sums of posteriors
constexpr int ARITH_DEFAULT = 0, ARITH_STRICT = 1, ARITH_FOLDED = 2;

template <int ARITH>
__device__ __forceinline__ void set_emis(EmisV<CPL>& p, int j, const Emis& e) {
  p.set(j, e);
  if (ARITH == ARITH_FOLDED) {  // train(): ln P = ln K - u^2, u = x c - mu c, c = 1/(stdev sqrt 2) -- two fused operations
    const double c = e.inv_stdev * 0x1.6a09e667f3bcdp-1;
    p.inv_stdev[j] = c;
    p.mean[j] = e.mean * c;
    p.neg_log_stdev[j] = e.neg_log_stdev - dynmath::HALF_LOG_2PI;
  }
}

// stdev: ARITH_STRICT only (the divisor of the reference's quotient, beside its reciprocal in p)
// y_lo: ARITH_STRICT only (the low part of 1/stdev: the exact quotient in four operations, dp_math_strict.hpp; -DDYN_QUOT5 keeps
// round 4's five)
template <int ARITH, int NS>
__device__ __forceinline__ void emission_vec(double x, const EmisV<CPL>& p, const double (&stdev)[NS], const double (&y_lo)[NS], double (&out)[CPL]) {
  if constexpr (ARITH == ARITH_STRICT) {
    static_assert(NS == CPL, "strict emission needs every cell's stdev");
#ifdef DYN_QUOT5
    dynmath::log_normal_pdf_cert_vec<CPL>(x, p, stdev, out);
#else
    dynmath::log_normal_pdf_cert4_vec<CPL>(x, p, stdev, y_lo, out);
#endif
  } else if constexpr (ARITH == ARITH_FOLDED) {
    double u[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) u[j] = dynmath::fma_(x, p.inv_stdev[j], -p.mean[j]);
#pragma unroll
    for (int j = 0; j < CPL; ++j) out[j] = dynmath::fma_(-u[j], u[j], p.neg_log_stdev[j]);
  } else {
    log_normal_pdf_vec<CPL>(x, p, out);
  }
}

// Second half of a certified logPlus (dp_math_strict.hpp): out = the reference's sum bit for bit. The registers in which
// some lane's certificate failed (a rounding boundary inside the interval: ~1e-4 of the cells of a 20 k-sample read,
// profiles/r04/cert_ambiguity.json) are recomputed with the restated glibc -- for all 64 lanes of that register: where the
// certificate held, the restated value IS the certified one.

__device__ __forceinline__ void log_plus_finish_certified(const SoftplusLookup<CPL>& L, double (&out)[CPL], const uint64_t* exp_tab,
                                                          uint32_t& fallbacks) {
  double hi[CPL];
  dynmath::log_plus_finish_cert<CPL>(L, out, hi);
  // One branch per row, not one test per register: the seven comparison masks are OR-ed (7 v_cmp + 6 s_or_b64, where the
  // per-register form took a compare, a select and an OR of SALU each), and which registers hold the ambiguous sums is only
  // worked out on the rare path.
#ifdef DYN_EXP_NO_CERT_BRANCH  // development (WRONG results): what the ambiguity test and its branch cost the certified rows
  (void)hi; (void)exp_tab; (void)fallbacks;
  return;
#endif
  uint64_t any_amb = 0;
#pragma unroll
  for (int j = 0; j < CPL; ++j) any_amb |= __ballot(out[j] != hi[j]);
  if (__builtin_expect(any_amb != 0, 0)) {
#ifdef DYN_EXP_NO_CERT_FALLBACK  // development (WRONG results): the test and the branch, but no recomputation
    ++fallbacks;
    return;
#endif
    unsigned amb = 0;  // wave-uniform: bit j = some lane's certificate failed in register j
#pragma unroll
    for (int j = 0; j < CPL; ++j) amb |= __any(out[j] != hi[j]) ? 1u << j : 0u;
    // ONE copy of the restated glibc (M = 1) for whichever registers need it: the operands are picked by the uniform
    // register number (a select chain: ~30 instructions per pass, next to ~130 of the restatement itself). Unrolled by
    // register, the seven copies cost the hot loop 100 spilled VGPRs.
#pragma unroll 1
    for (int j = 0; j < CPL; ++j) {
      if (!((amb >> j) & 1u)) continue;
      ++fallbacks;
      double hj = L.hi[0], dj = L.diff[0];
#pragma unroll
      for (int k = 1; k < CPL; ++k) {
        hj = (j == k) ? L.hi[k] : hj;
        dj = (j == k) ? L.diff[k] : dj;
      }
      const double v = dynmath::log_plus_strict_from(hj, dj, exp_tab);
#pragma unroll
      for (int k = 0; k < CPL; ++k) out[k] = (j == k) ? v : out[k];
    }
  }
}

__device__ __forceinline__ const uint64_t* strict_tab(const SoftplusNode* s_tab) {
  return reinterpret_cast<const uint64_t*>(s_tab + SP_NODES + dynmath::EXP128_NODES);
}
__device__ __forceinline__ const double* exp128_tab(const SoftplusNode* s_tab) {
  return reinterpret_cast<const double*>(s_tab + SP_NODES);
}

// ---- paged lattice rows --------------------------------------------------------------------------
// Lattice row t of the wave's current read is row (pt[t >> log_r] << log_r) + (t mod 2^log_r) of the
// pool arrays. The sweeps keep the page base in a scalar register and look the table (LDS) up only
// when a row crosses into another page.
struct WaveCtx {
  int lane;
  lds_u32_t* pt;  // this wave's page table
  int log_r;
};

struct RowCursor {
  int page = -1;
  uint32_t base = 0;
  // t is wave-uniform
  __device__ __forceinline__ uint32_t at(const WaveCtx& w, int t) {
    const int pg = t >> w.log_r;
    if (pg != page) {
      page = pg;
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.pt[pg]) << w.log_r;
    }
    return base + ((uint32_t)t & ((1u << w.log_r) - 1u));
  }
};

// per-lane row (traceback, segment-start posteriors)
__device__ __forceinline__ uint32_t pool_row(const WaveCtx& w, int t) {
  return (w.pt[t >> w.log_r] << w.log_r) + ((uint32_t)t & ((1u << w.log_r) - 1u));
}

// LDS written by some lanes of this wave, read by others: the wave's LDS operations execute in order,
// the fence keeps the compiler from moving them and waits for the writes.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
// The same for LDS alone: the fence above also waits for every global load and store in flight (s_waitcnt vmcnt(0)),
// which is what a loop that prefetches through global memory must not do.
__device__ __forceinline__ void wave_lds_sync_local() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
}

// ---- LDS-DMA row ring (forward sweep) ---------------------------------------------------------------
// The forward sweep must read one [448]-double bE row per lattice row while it overwrites an older row.
// With the rows prefetched into VGPRs the single wave of a SIMD stalled on s_waitcnt for 43 % of its
// life (compiler-placed vmcnt(0) behind the previous row's stores, prefetch registers spilled to
// AGPRs mid-row). The rows now go HBM -> LDS directly (global_load_lds_dwordx4: no VGPRs, 3 full +
// 1 half-wave instruction per 3584-byte row), RING_D rows ahead, and are picked up with ds_read_b64
// behind a hand-counted s_waitcnt. Validated in isolation by tools/ubench/lds_dma_test.hip.
constexpr int RING_D = DYN_WAVES_PER_SIMD == 1 ? 4 : 2;  // 2..5 rows deep measure the same: the sweep is bandwidth-, not latency-bound
constexpr int ROW_BYTES = P * 8;
// vmcnt(N) lets the N youngest vector-memory operations stay in flight. Only the DMA instructions
// themselves are counted (4 per row, RING_D-1 younger rows => 12): loads retire in issue order, so
// the wait is correct whatever the compiler does with the stores, sample and parameter loads
// that share the counter (they can only make it stricter).
constexpr int RING_WAIT = 4 * (RING_D - 1);

// row_lane_ptr = &row[lane*2]; `offset:` advances both the global and the LDS address.
// Each row is consumed exactly once: the DMA carries the non-temporal hint (-1 % on the sweep).
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

// the same for the lanes of `live` only (`half` = half_mask_of(live)); the others' ring slots keep what they held
__device__ __forceinline__ void ring_dma_row_masked(const double* row_lane_ptr, unsigned lds_slot_addr, uint64_t live, uint64_t half) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_mov_b64 exec, %2\n\t"
      "global_load_lds_dwordx4 %0, off" DYN_DMA_MOD "\n\t"
      "global_load_lds_dwordx4 %0, off offset:1024" DYN_DMA_MOD "\n\t"
      "global_load_lds_dwordx4 %0, off offset:2048" DYN_DMA_MOD "\n\t"
      "s_mov_b64 exec, %3\n\t"
      "global_load_lds_dwordx4 %0, off offset:3072" DYN_DMA_MOD "\n\t"
      "s_mov_b64 exec, -1\n\t"
      ::"v"(row_lane_ptr), "s"(__builtin_amdgcn_readfirstlane(lds_slot_addr)), "s"(live), "s"(half)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the wave's copy of one row: cell (lane, j) sits at pos_of(lane, j) * 8 behind the ring slot's LDS address
__device__ __forceinline__ void ring_read_row(unsigned slot_addr, int lane, double (&b)[CPL]) {
  static_assert(CPL == 7, "ring_read_row is written for 7 cells per lane");
  dyn_d2 p0, p1, p2;
  double last;
  asm volatile(
      "ds_read_b128 %0, %4\n\t"
      "ds_read_b128 %1, %4 offset:1024\n\t"
      "ds_read_b128 %2, %4 offset:2048\n\t"
      "ds_read_b64 %3, %5 offset:3072\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(last)
      : "v"(slot_addr + lane * 16), "v"(slot_addr + lane * 8)
      : "memory");
  b[0] = p0.x;
  b[1] = p0.y;
  b[2] = p1.x;
  b[3] = p1.y;
  b[4] = p2.x;
  b[5] = p2.y;
  b[6] = last;
}

// ---------------------------------------------------------------------------------------------
// backward recursion, t = T-2 .. 0 (NT_aligner_api.cpp:158-207)
//   e(t+1,n)  = logN(sig[t]; kmers[n-1])                      emission of lattice cell (t+1,n)
//   bM(t,n)   = bE(t+1,n) + e(t+1,n)                          (n > 0)            :197-200
//   bE(t,n)   = logPlus( (bM(t+1,n+1) + e(t+1,n+1)) + m1 ,    (n+1 < N)          :192-195
//                        (bE(t+1,n)   + e(t+1,n))   + e2 )    (n > 0)            :201
// Columns without a k-mer (n <= 0, n >= N) carry emission -inf, which realises the n > 0 and
// n+1 < N guards arithmetically; the upper band edge is handled in the window-move block.
// Returns Zb = bE(0,0) (-inf when the signal holds a non-finite sample).
// ---------------------------------------------------------------------------------------------
template <bool STORE, int ARITH>
__device__ __forceinline__ double backward_sweep(const ReadDesc& rd, const WaveCtx& w,
                                                 const double* __restrict__ sig,
                                                 const Emis* __restrict__ par, double* __restrict__ ws,
                                                 double m1, double e2, const SoftplusNode* s_tab, uint32_t* fallbacks = nullptr) {
  const int lane = w.lane;
  uint32_t nfb = 0;
  const int T = (int)rd.T, N = (int)rd.N, bw = (int)rd.bw, W = 2 * bw + 1;
  const double ratio = rd.ratio;
  const double* __restrict__ sg = sig + rd.sig_off;
  const Emis* __restrict__ pr = par + rd.par_off;
  double* __restrict__ out = ws;
  RowCursor cur;

  bool bad_sample = false;
  int lo = band_mid(T - 1, ratio) - bw;
  // DYN_SKIP_OOB: the lanes whose row stores count (align jobs: the forward sweep fetches exactly these or fewer)
  constexpr bool MASKED = SKIP_OOB && STORE && ARITH != ARITH_FOLDED;
  uint64_t store_mask = MASKED ? uniform_u64(lanes_of_window(lo - OOB_SLACK, W + 2 * OOB_SLACK)) : ~0ull;
  const int n_init = lo + bw;  // band column bw+1 of row T-1 (NT_aligner_api.cpp:170)
  int n[CPL];
  double bE[CPL], bM[CPL], bE2[CPL], bM2[CPL], e[CPL];
  EmisV<CPL> p;
  double p_stdev[ARITH == ARITH_STRICT ? CPL : 1];  // certified arithmetic: the divisor itself, beside its reciprocal
  double p_ylo[ARITH == ARITH_STRICT ? CPL : 1];    // ... and the low part of the reciprocal (two operations per hand-over)
  auto set_p = [&](int j, const Emis& em) {
    set_emis<ARITH>(p, j, em);
    if constexpr (ARITH == ARITH_STRICT) {
      p_stdev[j] = em.stdev;
      p_ylo[j] = dynmath::recip_lo(em.stdev, em.inv_stdev);
    }
  };
  // the exact quotient of the certified emission must not overflow (dp_math_strict.hpp); such a sample makes the reference's
  // z*z infinite, every cell of its row -inf and Z = -inf: the same verdict as an infinite sample
  constexpr double SAMPLE_MAX = ARITH == ARITH_STRICT ? 1e300 : 1.7976931348623157e308;
  {
    // one row past the lattice, all -inf: bM(T-1, .) = bE(T, .) + e has no successor. It lets the
    // forward sweep stream row t+1 for every t without a last-row special case in its row loop.
    const size_t rT = STORE ? (size_t)cur.at(w, T) * P : 0, rT1 = STORE ? (size_t)cur.at(w, T - 1) * P : 0;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int slot = lane * CPL + j;
      n[j] = lo + pmod(slot - lo);
      set_p(j, load_emis(pr, n[j], (n[j] <= lo + W - 1) ? N : 0));  // columns above the band: no k-mer yet
      bE[j] = (n[j] == n_init) ? 0.0 : NEG_INF;
      bM[j] = NEG_INF;
    }
    if (STORE) {
      store_row_f64<false>(out + rT1, lane, bE);
      store_row_f64<false>(out + rT, lane, bM);  // all -inf
    }
  }

  for (int thi = T - 2; thi >= 0; thi -= 64) {
    const int base = thi - 63;
    const int idx = base + lane;
    const double xs = (idx >= 0) ? sg[idx] : 0.0;
    bad_sample |= !(fabs(xs) <= SAMPLE_MAX);  // inf or NaN
    const int ilo = base < 0 ? -base : 0;
    emission_vec<ARITH>(readlane_f64(xs, 63), p, p_stdev, p_ylo, e);  // e(thi+1, n) from sig[thi]
    // one lattice row: reads (bE_in, bM_in) = row t+1, writes (bE_out, bM_out) = row t; the loop is unrolled by two and
    // ping-pongs between the two pairs (see forward_sweep: no register moves at the loop's back edge)
    auto row = [&](int i, const double (&bE_in)[CPL], const double (&bM_in)[CPL], double (&bE_out)[CPL], double (&bM_out)[CPL]) {
      const int t = base + i;
      // e[] = e(t+1, n) was computed during the previous row's table lookups (software pipeline)
      double Y[CPL], Yr[CPL], x1[CPL], x2[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) Y[j] = bM_in[j] + e[j];
#pragma unroll
      for (int j = 0; j < CPL; ++j) bM_out[j] = bE_in[j] + e[j];  // bM(t, n) ("A" below)
      from_right(Y, Yr);
      const int new_lo = band_mid(t, ratio) - bw;
      if (__builtin_expect(new_lo != lo, 0)) {  // wave-uniform: the window moved down by one column
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
            set_p(j, fresh);
          }
          // Upper band edge: bM(t, top) = A must not see the in-band cell (t+1, top); Y keeps it for
          // the diagonal into (t, top-1). From row t on the slot carries the "no k-mer" parameters, so
          // e = -inf and with it A = Y = -inf for every column above the band without a per-row
          // select -- including the slot of column lo+P-1, whose bE picks up a finite x1 from its
          // ring neighbour, band column lo, in every row (the slot ring wraps) and is emptied by
          // e = -inf before anything reads it.
          if (n[j] == top) {
            bM_out[j] = NEG_INF;
            set_p(j, none);
          }
        }
        lo = new_lo;
        if constexpr (MASKED) store_mask = uniform_u64(lanes_of_window(lo - OOB_SLACK, W + 2 * OOB_SLACK));
      }
#pragma unroll
      for (int j = 0; j < CPL; ++j) x1[j] = Yr[j] + m1;
#pragma unroll
      for (int j = 0; j < CPL; ++j) x2[j] = bM_out[j] + e2;
      SoftplusLookup<CPL> L;
      log_plus_issue<CPL>(x1, x2, L, s_tab);
      // while the LDS lookups are in flight: emission of the NEXT row, e(t, n) = logN(sig[t-1]; .)
      // (rows are consumed top-down; at i == 0 the next block's sample is not loaded yet: the value computed
      //  here is thrown away and that one emission is computed after the block switch above. Unconditional on
      //  purpose: behind an `if (i > 0)` the seven results are merged with the old ones by a register move each,
      //  in every row)
      {
        const double xnext = readlane_f64(xs, i > 0 ? i - 1 : 0);
        emission_vec<ARITH>(xnext, p, p_stdev, p_ylo, e);
      }
      if constexpr (ARITH == ARITH_FOLDED) dynmath::log_plus_finish3<CPL>(L, bE_out);  // train(): no decision hangs on it
      else if constexpr (ARITH == ARITH_STRICT) log_plus_finish_certified(L, bE_out, strict_tab(s_tab), nfb);
      else log_plus_finish<CPL>(L, bE_out);
      const size_t rt = STORE ? (size_t)cur.at(w, t) * P : 0;
      // streamed once, read back by the forward sweep ~100 MB later: non-temporal (-2 %)
      if constexpr (STORE && ARITH == ARITH_FOLDED) {
        // train(): what its forward sweep (the posterior chain) needs of row t is ln P(stay | E(t,n)) = the stay operand
        // of the logPlus minus its result -- stored instead of bE, so that the chain reads a row and takes its
        // exponential ((-inf) - (-inf) = NaN for a cell nothing can reach: exponential 0)
        double ls[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) ls[j] = x2[j] - bE_out[j];
        store_row_f64<true>(out + rt, lane, ls);
      } else if (STORE) {
        if constexpr (MASKED) {
          if (__builtin_amdgcn_inverse_ballot_w64(store_mask)) store_row_f64<true>(out + rt, lane, bE_out);
        } else {
          store_row_f64<true>(out + rt, lane, bE_out);
        }
      }
    };
    int i = 63;
#pragma unroll 1
    for (; i - 1 >= ilo; i -= 2) {
      row(i, bE, bM, bE2, bM2);
      row(i - 1, bE2, bM2, bE, bM);
    }
    if (i >= ilo) {  // odd number of rows (first block of a read only)
      row(i, bE, bM, bE2, bM2);
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        bE[j] = bE2[j];
        bM[j] = bM2[j];
      }
    }
  }
  // An infinite sample gives every cell of its row the score -inf in the reference (aligner.cpp:
  // 287-292), hence Z = -inf and "alignment scores do not match" (NT_aligner_api.cpp:288-291). The
  // residual-corrected quotient of log_normal_pdf_vec would turn it into NaN instead, so non-finite
  // samples are flagged here and reported through the same Z check (NaN samples, for which the
  // reference's behaviour is undefined, fail the same way).
  const bool any_bad = __any(bad_sample);
  if (ARITH == ARITH_STRICT && fallbacks) *fallbacks += nfb;
  // lattice column 0 sits in slot 0 at row 0 (lo = -bw)
  return any_bad ? NEG_INF : readlane_f64(bE[0], 0);
}

// ---------------------------------------------------------------------------------------------
// forward (NT_aligner_api.cpp:110-152) fused with posterior (:213-224,297-300) and the
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
// Returns Zf = forwardE[T*B - bandwidth - 2] = fE(T-1, mid(T-1))  (NT_aligner_api.cpp:285).
// ---------------------------------------------------------------------------------------------
// INPLACE (only with POST): see the comment at lat_lp below
// max of two values the compiler cannot prove canonical (loop-carried): fmax() makes hipcc quiet each operand with
// a v_max_f64 v, v, v first -- three fp64 instructions instead of one. vM and vE are sums or -inf, never signalling.
__device__ __forceinline__ double max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <bool POST, bool INPLACE, bool STRICT>
__device__ __forceinline__ double forward_sweep(const ReadDesc& rd, const WaveCtx& w,
                                                const double* __restrict__ sig, const Emis* __restrict__ par,
                                                const double* __restrict__ ws_rd, float* __restrict__ lp_out,
                                                uint64_t* __restrict__ bits, double Z, double m1, double e2,
                                                const SoftplusNode* s_tab, unsigned ring_base, int strict_rows = 0,
                                                uint32_t* fallbacks = nullptr) {
  uint32_t nfb = 0;
  // STRICT instantiation: rows t <= strict_rows are computed with the certified (bit-for-bit) arithmetic, later rows with the
  // default one (strict_rows = INT_MAX: the whole sweep). The Viterbi values of a row depend on forward values of rows <= t only,
  // so with a strict backward sweep every decision up to row strict_rows is the reference's own.
  const int lane = w.lane;
  const int T = (int)rd.T, N = (int)rd.N, bw = (int)rd.bw, W = 2 * bw + 1;
  const double ratio = rd.ratio;
  const double* __restrict__ sg = sig + rd.sig_off;
  const Emis* __restrict__ pr = par + rd.par_off;
  // Two layouts for the log-posteriors the traceback needs (one cell per row):
  //  * separate (default): the bE rows stay intact and ONE float per slot, LPE, goes to its own array
  //    (same [row][row_pos] indexing); the ~10 % M cells of the path (segment starts) get LPM rebuilt
  //    from LPE of the diagonal predecessor and two bE cells (mpost). 12 B per slot per row here.
  //  * INPLACE: (float LPM, float LPE) overwrite the bE slot of row t while rows t+1.. are still
  //    being read (lp_out then aliases ws_rd; no address is read after it has been written within one
  //    sweep). 16 B per slot per row make this sweep HBM-bound, but the footprint is 8 instead of
  //    12 B per slot: the host picks it when the separate layout would starve the waves of pages.
  // ws_rd and lp_out are separate __restrict__ parameters on purpose: with a pointer derived from
  // the load pointer hipcc orders every prefetch behind the previous row's stores (s_waitcnt vmcnt(0)
  // at the top of each row = 43 % of the wave's lifetime spent waiting).
  const double* __restrict__ lat = ws_rd;
  float* __restrict__ lat_lp = lp_out;
  RowCursor cur_dma, cur_out;

  // Band edges without per-row masks. Lower edge: the slot of a column that leaves the band is
  // handed to column lo+P. Upper edge: a slot carries the k-mer parameters of its column only from
  // the row in which the column ENTERS the band; before that it carries the "no k-mer" parameters
  // (log_norm = -inf), hence e = -inf, hence fM = fE = LPM = LPE = vM = vE = -inf in that slot with
  // no select in the row loop. Both hand-overs happen in the rare block that looks one row ahead.
  int lo = band_mid(1, ratio) - bw;  // band of row 1 (column 0 of row 0 is inside: band_mid(1) <= 1 <= bw)
  // DYN_SKIP_OOB: lanes whose part of a row is fetched (rows up to RING_D + 1 ahead) / whose posteriors of a row are stored
  // (the row's band, whichever side of the hand-over `lo` is on when the stores are issued)
  constexpr bool MASKED = SKIP_OOB && POST;
  uint64_t dma_mask = ~0ull, dma_half = 0xffffffffull, out_mask = ~0ull;
  auto set_masks = [&]() {
    if constexpr (MASKED) {
      dma_mask = uniform_u64(lanes_of_window(lo, W + OOB_SLACK));
      dma_half = half_mask_of(dma_mask);
      out_mask = uniform_u64(lanes_of_window(lo - 1, W + 1));
    }
  };
  set_masks();
  if constexpr (MASKED) {
    // lanes that fetch nothing of a row read what their ring slots held before: that must be a harmless number (-inf here;
    // later, values of rows they did fetch), never the ballot words the previous read's traceback staged in this ring
    // (a NaN pattern in a slot outside the band would travel into its in-band neighbour through the Viterbi values)
    auto* ringp = (__attribute__((address_space(3))) double*)(size_t)ring_base;
#pragma unroll
    for (int k = 0; k < RING_D * P / 64; ++k) ringp[k * 64 + lane] = NEG_INF;
    wave_lds_sync_local();
  }
  int n[CPL];
  // The loop-carried values of a row. The row loop is unrolled by two and ping-pongs between two of these:
  // with one set, "new" values are computed while the "old" ones are still live and hipcc closes every
  // iteration with a register move per value and cell (19 v_mov_b64 per row in this sweep, 5 % of its VALU
  // instructions -- and the chip's power limit charges per instruction).
  struct RowState {
    double fM[CPL], fE[CPL], e[CPL], vM[CPL], vE[CPL], b[CPL];  // b = bE of the row that is about to be computed
  };
  RowState sa, sb;
  EmisV<CPL> p;
  double p_stdev[STRICT ? CPL : 1];  // STRICT instantiation: the divisor of the reference's emission, beside 1/stdev
  double p_ylo[STRICT ? CPL : 1];    // ... and the low part of 1/stdev (dp_math_strict.hpp: the exact quotient in four operations)
  // POST = false is align(calc_probabilities=false): Z both ways and their agreement, no decision taken -- the cheap
  // arithmetic of train()'s backward sweep (two-operation emission, degree-3 softplus polynomial; Z moves by 1e-12
  // relative, the bar is 1e-9)
  // with_stdev: the certified rows' emission also needs the divisor itself; rows past the strict ones must not keep it
  // alive (14 more registers in a loop that has none to spare: 44 AGPR moves per row pair)
  auto set_p = [&](int j, const Emis& e, auto with_stdev) {
    if constexpr (!POST) set_emis<ARITH_FOLDED>(p, j, e);
    else p.set(j, e);
    if constexpr (STRICT && decltype(with_stdev)::value) {
      p_stdev[j] = e.stdev;
      p_ylo[j] = dynmath::recip_lo(e.stdev, e.inv_stdev);
    }
  };
  // emission of one sample against the lane's cells, in the flavour of the row it belongs to
  auto emission = [&](bool strict_row, double x, double (&out)[CPL]) {
    if constexpr (STRICT) {
      if (strict_row) {
#ifdef DYN_QUOT5
        dynmath::log_normal_pdf_cert_vec<CPL>(x, p, p_stdev, out);
#else
        dynmath::log_normal_pdf_cert4_vec<CPL>(x, p, p_stdev, p_ylo, out);
#endif
        return;
      }
    }
    if constexpr (!POST) emission_vec<ARITH_FOLDED>(x, p, p_stdev, p_ylo, out);
    else log_normal_pdf_vec<CPL>(x, p, out);
  };
  const double x0 = sg[0];
  const size_t r1 = POST ? (size_t)cur_out.at(w, 1) * P : 0;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int slot = lane * CPL + j;
    n[j] = lo + pmod(slot - lo);
    set_p(j, load_emis(pr, n[j], (n[j] <= lo + W - 1) ? N : 0), std::true_type{});
    sa.fE[j] = (n[j] == 0) ? 0.0 : NEG_INF;  // E[bandwidth+1] = 0 (NT_aligner_api.cpp:120)
    sa.fM[j] = NEG_INF;
    sa.vE[j] = sa.fE[j];                     // E[bandwidth+1] = 0 (NT_aligner_api.cpp:336)
    sa.vM[j] = NEG_INF;
    sa.b[j] = POST ? lat[r1 + pos_of(lane, j)] : NEG_INF;
  }
  emission(1 <= strict_rows, x0, sa.e);  // e(1, n)
  // ring prologue: rows 2 .. RING_D+1 (row r lives in ring slot r % RING_D)
  const double* __restrict__ dma_src = ws_rd + lane * 2;
  if (POST) {  // rows past T repeat the all -inf row T (backward sweep): RING_D rows are always in flight
    for (int r = 2; r <= RING_D + 1; ++r) {
      if constexpr (MASKED) ring_dma_row_masked(dma_src + (size_t)cur_dma.at(w, min(r, T)) * P, ring_base + (r % RING_D) * ROW_BYTES, dma_mask, dma_half);
      else ring_dma_row(dma_src + (size_t)cur_dma.at(w, min(r, T)) * P, ring_base + (r % RING_D) * ROW_BYTES);
    }
  }

  // one lattice row: reads the state `in` (row t-1), writes `out` (row t)
  // strict_tag: a row of a STRICT block (certified arithmetic, its look-ahead emission included)
  auto row = [&](auto check_move, auto strict_tag, int t, double xn, const RowState& in, RowState& out) {
    constexpr bool STRICT_ROW = STRICT && decltype(strict_tag)::value;
    double fEl[CPL], vEl[CPL];
    if (POST) {
      // bE(t+1, .) from the ring; its slot is then refilled with row t+1+RING_D (clamped to the
      // -inf row T, so that the same number of memory operations is in flight in every row and
      // one hand-counted s_waitcnt serves the whole loop, tail included)
      wait_vmcnt<RING_WAIT>();
      ring_read_row(ring_base + ((t + 1) % RING_D) * ROW_BYTES, lane, out.b);
      if constexpr (MASKED)
        ring_dma_row_masked(dma_src + (size_t)cur_dma.at(w, min(t + 1 + RING_D, T)) * P, ring_base + ((t + 1) % RING_D) * ROW_BYTES, dma_mask, dma_half);
      else
        ring_dma_row(dma_src + (size_t)cur_dma.at(w, min(t + 1 + RING_D, T)) * P, ring_base + ((t + 1) % RING_D) * ROW_BYTES);
    }
    from_left(in.fE, fEl);
    if (POST) from_left(in.vE, vEl);
    if constexpr (decltype(check_move)::value) {
      const int next_lo = band_mid(t + 1, ratio) - bw;
      if (__builtin_expect(next_lo != lo, 0)) {  // wave-uniform: the window moves up by one column between rows t and t+1
        // uniform addresses -> scalar loads (see backward_sweep)
        const Emis none = load_emis(pr, 0, 0);
        const Emis entering = load_emis(pr, lo + W, N);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          // Column lo is in the band for the last time in this row; its slot becomes column lo+P. n and
          // p are only read by the emission of row t+1 below, which must already be -inf here; e(t, lo)
          // and fE/vE(t-1, lo) stay where this row and the right neighbour (fEl/vEl) still read them.
          const bool leaves = n[j] == lo;
          n[j] = leaves ? lo + P : n[j];
          if (leaves) set_p(j, none, strict_tag);
          if (n[j] == lo + W) set_p(j, entering, strict_tag);  // first band row of this column is t+1
        }
        lo = next_lo;
        set_masks();
      }
    }
    double a1[CPL], a2[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) out.fM[j] = (fEl[j] + in.e[j]) + m1;
#pragma unroll
    for (int j = 0; j < CPL; ++j) a1[j] = in.fM[j] + in.e[j];
#pragma unroll
    for (int j = 0; j < CPL; ++j) a2[j] = (in.fE[j] + in.e[j]) + e2;
    if constexpr (STRICT_ROW) {
      SoftplusLookup<CPL> L;
      log_plus_issue<CPL>(a1, a2, L, s_tab);
      emission(true, xn, out.e);  // e(t+1, .): exact, whichever flavour row t+1 runs in
      log_plus_finish_certified(L, out.fE, strict_tab(s_tab), nfb);
    } else {
      SoftplusLookup<CPL> L;
      log_plus_issue<CPL>(a1, a2, L, s_tab);
      emission(false, xn, out.e);  // e(t+1, n): independent work under the LDS latency
      if constexpr (!POST) dynmath::log_plus_finish3<CPL>(L, out.fE);
      else log_plus_finish<CPL>(L, out.fE);
    }
    if (POST) {
      double LPM[CPL], LPE[CPL], alt[CPL];
      uint64_t bj[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) LPM[j] = (out.fM[j] + (out.b[j] + out.e[j])) - Z;  // bM(t,n) = bE(t+1,n) + e(t+1,n)
#pragma unroll
      for (int j = 0; j < CPL; ++j) LPE[j] = (out.fE[j] + in.b[j]) - Z;
#pragma unroll
      for (int j = 0; j < CPL; ++j) out.vM[j] = vEl[j] + LPM[j];
#pragma unroll
      for (int j = 0; j < CPL; ++j) out.vE[j] = max_f64(in.vM[j], in.vE[j]) + LPE[j];
#pragma unroll
      for (int j = 0; j < CPL; ++j) alt[j] = in.vM[j] + LPE[j];
#pragma unroll
      for (int j = 0; j < CPL; ++j) bj[j] = __ballot(out.vE[j] == alt[j]);
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
      const uint32_t prow = cur_out.at(w, t);
      const size_t rt = (size_t)prow * P;
      // read again only by the traceback, one cell per row: non-temporal; same pairing as the bE rows
      const bool out_live = MASKED ? __builtin_amdgcn_inverse_ballot_w64(out_mask) : true;
      if (!out_live) {
        // (DYN_SKIP_OOB: nothing of this lane lies in the row's band)
      } else if (INPLACE) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          dyn_f4 v4;
          v4.x = (float)LPM[2 * q];
          v4.y = (float)LPE[2 * q];
          v4.z = (float)LPM[2 * q + 1];
          v4.w = (float)LPE[2 * q + 1];
          __builtin_nontemporal_store(v4, reinterpret_cast<dyn_f4*>(&lat_lp[2 * (rt + q * 128 + lane * 2)]));
        }
        dyn_f2 v2;
        v2.x = (float)LPM[6];
        v2.y = (float)LPE[6];
        __builtin_nontemporal_store(v2, reinterpret_cast<dyn_f2*>(&lat_lp[2 * (rt + 384 + lane)]));
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          dyn_f2 v2;
          v2.x = (float)LPE[2 * q];
          v2.y = (float)LPE[2 * q + 1];
          __builtin_nontemporal_store(v2, reinterpret_cast<dyn_f2*>(&lat_lp[rt + q * 128 + lane * 2]));
        }
        __builtin_nontemporal_store((float)LPE[6], &lat_lp[rt + 384 + lane]);
      }
      if (lane < CPL) bits[(size_t)prow * CPL + lane] = mybits;
    }
  };

  // one 64-row block of samples; strict_tag: the block's rows run in the certified arithmetic
  auto block = [&](auto strict_tag, int tb) {
    const int idx = tb + lane;  // sig[t] is the sample of lattice row t+1
    const double xs = (idx < T - 1) ? sg[idx] : 0.0;
    // The block's samples must have ARRIVED before the row loop starts: hipcc otherwise places the
    // s_waitcnt vmcnt(0) for this load in front of its first use INSIDE the loop, where it drains the
    // DMA ring in every row (+7 % on the sweep). A use here pins the wait to once per 64 rows.
    asm volatile("" ::"v"(xs));
    const int iend = min(64, T - tb);
    int i = 0;
#pragma unroll 1
    for (; i + 1 < iend; i += 2) {
      row(std::true_type{}, strict_tag, tb + i, readlane_f64(xs, i), sa, sb);
      row(std::true_type{}, strict_tag, tb + i + 1, readlane_f64(xs, i + 1), sb, sa);
    }
    if (i < iend) {  // odd tail (last block of a read only): one more row, then the roles are swapped back
      row(std::true_type{}, strict_tag, tb + i, readlane_f64(xs, i), sa, sb);
      sa = sb;
    }
  };
  // Whole 64-row blocks are strict or not (rows past strict_rows inside a strict block cost a little and harm nothing), and
  // the strict blocks are a PREFIX: two loops one after the other, each with one arithmetic -- the second one never sees
  // the certified emission's extra operands (p_stdev is dead there). Same-box A/B against one loop that picks the
  // arithmetic per block: launch 51.3 vs 52.8 ms on cfg2 with 26 % tie reads.
  int tb = 1;
  if constexpr (STRICT) {
    for (; tb < T && tb <= strict_rows; tb += 64) block(std::true_type{}, tb);
  }
  for (; tb < T; tb += 64) block(std::false_type{}, tb);
  const double (&fE)[CPL] = sa.fE;
  if (STRICT && fallbacks) *fallbacks += nfb;
  if (POST) wait_vmcnt<0>();  // the clamped tail DMAs still target this wave's LDS ring
  const int nf = band_mid(T - 1, ratio);
  const int sf = pmod(nf);
  double zf = 0.0;
#pragma unroll
  for (int j = 0; j < CPL; ++j)
    if (j == sf % CPL) zf = fE[j];
  return readlane_f64(zf, sf / CPL);
}

// ---------------------------------------------------------------------------------------------
// Z check of align()/train() (NT_aligner_api.cpp:285-291 / 619-625).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool z_ok(const ReadDesc& rd, double Zf, double Zb) {
  const double size = (double)((uint64_t)rd.T * (uint64_t)(2 * rd.bw + 3));
  if (isinf(Zf) || isinf(Zb)) return false;
  return !(fabs(Zf - Zb) / size > 1e-8);
}

// ---------------------------------------------------------------------------------------------
// traceback: decodeMAP (NT_aligner_api.cpp:383-456). Start in state E at (T-1, N-1); walk the
// decision bits. Each lattice row 1..T-1 holds exactly one path cell, so the walk is recorded
// as pathn[row] (column, bit31 = state M) and pp[row] = exp(LP of that cell); a segment is the
// run of rows that share a column, its M cell is the lowest row (segrow).
// 64 rows of bits are staged in LDS per step so the serial walk pays LDS, not HBM, latency.
// Returns true when the walk ended in (0, 0).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool traceback(const ReadDesc& rd, const WaveCtx& w, const float* __restrict__ lp,
                                          const uint64_t* __restrict__ bits, TraceBuffers tb, bool inplace,
                                          lds_u64_t* sb) {
  const int lane = w.lane;
  const int T = (int)rd.T, N = (int)rd.N;
  // separate layout: float LPE rows [.][P]; in-place layout: (float LPM, float LPE) in the 8-byte bE slots
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
#ifndef DYN_EXP_TB_SERIAL
  // The blocks follow each other at a fixed stride (a block is left at its lowest row, whatever the path does inside), so
  // the next block's decision bits are fetched while this one is walked, and a block's path posteriors -- one dependent
  // load per row -- are finished one block later: the walk pays neither HBM latency.
  uint64_t nb[CPL];          // the bits of the block about to be walked (row base + lane)
  uint32_t nprow = 0;
  auto fetch = [&](int base_) {
    const int row_ = base_ + lane;
    nprow = 0;
#pragma unroll
    for (int j = 0; j < CPL; ++j) nb[j] = 0;
    if (row_ >= 1) {
      nprow = pool_row(w, row_);
#pragma unroll
      for (int j = 0; j < CPL; ++j) nb[j] = bits[(size_t)nprow * CPL + j];
    }
  };
  int pend_row = 0, pend_n = 0, pend_st = -1;   // this lane's path cell of the previous block, its posterior in flight
  float pend_lp = 0.0f;
  auto finish = [&]() {
    if (pend_st >= 0) {
      if (inplace || pend_st == 0) pp[pend_row] = exp((double)pend_lp);
      pathn[pend_row] = (uint32_t)pend_n | (pend_st ? 0x80000000u : 0u);
      if (pend_st) segrow[pend_n - 1] = (uint32_t)pend_row;
    }
    pend_st = -1;
  };
  int sj = slot % CPL, sq = slot / CPL;   // the current column's slot, as (cell index within a lane, lane)
  if (t > 0 && n > 0) fetch(t - 63);
  while (t > 0 && n > 0) {
    const int base = t - 63;
    const int row = base + lane;
    const uint32_t prow = nprow;
#pragma unroll
    for (int j = 0; j < CPL; ++j) sb[lane * CPL + j] = nb[j];
    wave_lds_sync_local();
    if (base > 1) fetch(base - 64);   // (base <= 1: this is the last block)
    int my_n = 0, my_st = -1;
    const int block_lo = base < 1 ? 1 : base;  // lowest row of this block that exists
    // The walk is wave-uniform: rows are lanes, sets of rows are 64-bit scalar masks (lane i = row base + i), and a mask
    // reaches the lanes as a select predicate (inverse ballot) -- the per-lane comparisons of rows against t, e_lo and
    // block_lo and the division of the slot by CPL per step were most of the 80 instructions a step took.
    const uint64_t in_block = base >= 1 ? ~0ull : (~0ull << (1 - base));  // lanes whose row exists (>= 1)
    while (t >= block_lo && n > 0) {
      const int tl = t - base;                                        // lane of row t, 0 .. 63
      if (stM) {  // M(t,n) carried over from the block above: segment start; continue with E(t-1, n-1)
        if (__builtin_amdgcn_inverse_ballot_w64(1ull << tl)) {
          my_n = n;
          my_st = 1;
        }
        --n;
        if (sj == 0) {
          sj = CPL - 1;
          sq = sq ? sq - 1 : 63;
        } else {
          --sj;
        }
        stM = 0;
        --t;
        continue;
      }
      // state E in column n at row t: every lane tests its own row's decision bit of this column
      const uint64_t wd = sb[lane * CPL + sj];                       // ballot word of cell index j = slot % CPL
      const uint64_t upto = (2ull << tl) - 1ull;                      // lanes 0 .. tl (tl = 63: the shift leaves 0, minus 1: all)
      const uint64_t m = __ballot((wd >> sq) & 1) & upto & in_block;  // bit of lane slot / CPL; rows block_lo .. t
      if (m) {
        // the path turns at the highest such row r: rows r .. t are E cells of column n, the M cell lies one row below,
        // then E in column n-1 -- taken in the same step when that M cell lies in this block
        const int rl = 63 - __builtin_clzll(m);
        if (__builtin_amdgcn_inverse_ballot_w64(upto & ~((1ull << rl) - 1ull))) {
          my_n = n;
          my_st = 0;
        }
        if (base + rl - 1 >= block_lo) {
          if (__builtin_amdgcn_inverse_ballot_w64(1ull << (rl - 1))) {
            my_n = n;
            my_st = 1;
          }
          --n;
          if (sj == 0) {
            sj = CPL - 1;
            sq = sq ? sq - 1 : 63;
          } else {
            --sj;
          }
          t = base + rl - 2;
        } else {
          t = base + rl - 1;  // the M cell lies in the next block: stM carries over
          stM = 1;
        }
      } else {
        if (__builtin_amdgcn_inverse_ballot_w64(upto & in_block)) {  // rows block_lo .. t are E cells of column n
          my_n = n;
          my_st = 0;
        }
        t = block_lo - 1;
      }
    }
    finish();   // the previous block's cells: their loads have had this block's walk to arrive
    if (my_st >= 0) {
      // E cell of the path: the forward sweep stored its log-posterior. M cells (segment starts, ~1 row in
      // 10) get theirs from mpost, one lane per segment, instead of ~6 dependent loads in this walk.
      const size_t cell = (size_t)prow * P + row_pos(my_n % P);
      if (inplace) pend_lp = lp[2 * cell + (my_st ? 0 : 1)];
      else if (my_st == 0) pend_lp = lp[cell];
      pend_row = row;
      pend_n = my_n;
      pend_st = my_st;
    }
    wave_lds_sync_local();
  }
  finish();
#else
  while (t > 0 && n > 0) {
    const int base = t - 63;
    const int row = base + lane;
    uint32_t prow = 0;
    if (row >= 1) {
      prow = pool_row(w, row);
#pragma unroll
      for (int j = 0; j < CPL; ++j) sb[lane * CPL + j] = bits[(size_t)prow * CPL + j];
    }
    wave_lds_sync();
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
      const uint64_t wd = sb[lane * CPL + (slot % CPL)];  // ballot word of cell index j = slot % CPL
      const bool bit = (row >= block_lo) && (row <= t) && ((wd >> (slot / CPL)) & 1);  // lane = slot / CPL
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
      // E cell of the path: the forward sweep stored its log-posterior. M cells (segment starts, ~1 row in
      // 10) get theirs from mpost, one lane per segment, instead of ~6 dependent loads in this walk.
      const size_t cell = (size_t)prow * P + row_pos(my_slot);
      if (inplace) pp[row] = exp((double)lp[2 * cell + (my_st ? 0 : 1)]);
      else if (my_st == 0) pp[row] = exp((double)lp[cell]);
      pathn[row] = (uint32_t)my_n | (my_st ? 0x80000000u : 0u);
      if (my_st) segrow[my_n - 1] = (uint32_t)row;
    }
    wave_lds_sync();
  }
#endif
  return t == 0 && n == 0;
}

// ---------------------------------------------------------------------------------------------
// mpost: posterior of the M cell (segment start) of every segment. The forward sweep stores only LPE; the
// M cell (row, n) is preceded on the path by the E cell (row-1, n-1), so
//   fE(row-1, n-1) = LPE(row-1, n-1) - bE(row-1, n-1) + Zb      (NT_aligner_api.cpp:222 solved for fE)
//   fM(row, n)     = (fE(row-1, n-1) + e(row, n)) + m1            (:146)
//   bM(row, n)     = bE(row+1, n) + e(row+1, n)                   (:200; row T of the workspace is -inf)
//   LPM(row, n)    = (fM + bM) - Zb                               (:222)
// The float LPE costs <= 6e-8 * |LPE| here; every other term is fp64. One lane per segment.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void mpost(const ReadDesc& rd, const WaveCtx& w, const double* __restrict__ bE,
                                      const float* __restrict__ lp, const double* __restrict__ sig,
                                      const Emis* __restrict__ par, TraceBuffers tb, double Zb, double m1) {
  const int T = (int)rd.T, N = (int)rd.N;
  const double* __restrict__ sg = sig + rd.sig_off;
  for (int n = 1 + w.lane; n < N; n += 64) {  // lattice column of the segment, 1 .. N-1
    const int row = (int)tb.segrow[rd.seg_off + n - 1];
    const int slot = n % P, pslot = (n - 1) % P;
    double fE_prev;
    if (row == 1) {
      fE_prev = n == 1 ? 0.0 : NEG_INF;  // fE(0, 0) = 0, nothing else in row 0 (:120)
    } else {
      const size_t pcell = (size_t)pool_row(w, row - 1) * P + row_pos(pslot);
      fE_prev = ((double)lp[pcell] - bE[pcell]) + Zb;
    }
    const Emis em = par[rd.par_off + n - 1];
    const double e_here = dynmath::log_normal_pdf(sg[row - 1], em);
    const double e_next = row + 1 < T ? dynmath::log_normal_pdf(sg[row], em) : NEG_INF;
    const double fM = (fE_prev + e_here) + m1;
    const double bM = bE[(size_t)pool_row(w, row + 1) * P + row_pos(slot)] + e_next;
    tb.pp[rd.path_off + row] = exp((fM + bM) - Zb);
  }
}

// ---------------------------------------------------------------------------------------------
// Forward sweep of train() as the POSTERIOR CHAIN (round 3). With the backward values B(t, .) of every cell known, the
// forward recursion need not carry forward probabilities at all: the posterior itself satisfies a first-order recursion
//   gamma_E(t,n) = gamma_M(t-1,n) + gamma_E(t-1,n) s(t,n),        gamma_M(t,n) = gamma_E(t-1,n-1) (1 - s(t,n-1)),
//   s(t,n) = P(stay | E(t-1,n), whole signal) = exp( e2 + e(t,n) + bE(t,n) - bE(t-1,n) )
// because bE(t-1,n) = logPlus(move, stay) (NT_aligner_api.cpp:191-203) makes the two ways on from an E cell sum to 1,
// and an M cell has exactly one way on (e1 = 1, :200). The exponent is the stay operand of that logPlus minus its
// result: the backward sweep has both in registers and stores THE DIFFERENCE as row t-1 of the lattice (8 bytes per
// cell, as bE would be). The chain reads a row, takes one exponential per cell and moves mass: every quantity is a
// probability in [0, 1], every row sums to 1 by construction, nothing is scaled and nothing can run out of range,
// whatever the read -- the forward-backward product in the linear domain, tried first, needs e^(several thousand) of
// range WITHIN a row as soon as the basecalls disagree with the signal
// (profiles/r03/linear_domain_on_imperfect_reads.txt). The reference: three exp and two log1p per cell in this sweep.
// What it gives up: a forward value of Z. The reference's check |Zf - Zb| / size <= 1e-8 compares two roundings of the
// same number; here the backward value is THE Z and the check is that the chain delivers all its mass to the end cell
// (T-1, N-1) and every sample has carried weight 1 -- which fails exactly when the backward values are not those of a
// consistent lattice (inf / NaN samples). Statistics as in NT_aligner_api.cpp:462-561; the transition expectations
// (:641-725) are the same for every path (N-1 moves, T-1-2(N-1) extensions) and written as such.
__device__ __forceinline__ double forward_train_chain(const ReadDesc& rd, const WaveCtx& w, const double* __restrict__ sig,
                                                      const double* __restrict__ ws, TrainBuffers tb, double Zb,
                                                      const SoftplusNode* s_tab, unsigned ring_base) {
  const int lane = w.lane;
  const int T = (int)rd.T, N = (int)rd.N, bw = (int)rd.bw;
  const double ratio = rd.ratio;
  const double* __restrict__ sg = sig + rd.sig_off;
  double* __restrict__ cw = tb.col_w + rd.par_off;
  double* __restrict__ cs1 = tb.col_s1 + rd.par_off;
  double* __restrict__ cs2 = tb.col_s2 + rd.par_off;
  const double* __restrict__ etab = exp128_tab(s_tab);
  RowCursor cur_dma;

  int lo = band_mid(1, ratio) - bw;  // band of row 1
  double gE[CPL], gM[CPL];            // posteriors of the previous row, updated in place
  double aw[CPL], a1[CPL], a2[CPL];
  double awsum = 0.0;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int nj = lo + pmod(lane * CPL + j - lo);
    gE[j] = (nj == 0) ? 1.0 : 0.0;  // all mass in E(0, 0)
    gM[j] = 0.0;
    aw[j] = a1[j] = a2[j] = 0.0;
  }
  // ring: row t of the chain reads lattice row t-1 (= ln of the stay probabilities out of row t-1), which lives in ring
  // slot (t-1) % RING_D; rows past T repeat row T
  const double* __restrict__ dma_src = ws + lane * 2;
  for (int r = 0; r < RING_D; ++r)
    ring_dma_row(dma_src + (size_t)cur_dma.at(w, min(r, T)) * P, ring_base + (r % RING_D) * ROW_BYTES);

  const double one = dynmath::vreg_const(1.0);
  auto row = [&](int t, double x) {  // x = the sample row t scores (sig[t-1])
    double ls[CPL], sp[CPL], st[CPL], mv[CPL], mvl[CPL];
    wait_vmcnt<RING_WAIT>();
    const unsigned slot_addr = ring_base + ((t - 1) % RING_D) * ROW_BYTES;
    ring_read_row(slot_addr, lane, ls);
    ring_dma_row(dma_src + (size_t)cur_dma.at(w, min(t - 1 + RING_D, T)) * P, slot_addr);
    dynmath::exp_table128_vec<CPL>(ls, sp, etab);  // NaN, -inf: 0
#pragma unroll
    for (int j = 0; j < CPL; ++j) sp[j] = dynmath::min_hw(sp[j], one);  // <= 1 whatever the last bits say (NaN: 1, times a mass of 0)
#pragma unroll
    for (int j = 0; j < CPL; ++j) st[j] = gE[j] * sp[j];
#pragma unroll
    for (int j = 0; j < CPL; ++j) mv[j] = gE[j] - st[j];
    from_left(mv, mvl);
#pragma unroll
    for (int j = 0; j < CPL; ++j) gE[j] = gM[j] + st[j];
    const double x2 = x * x;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      gM[j] = mvl[j];
      const double wgt = gE[j] + mvl[j];
      aw[j] += wgt;
      a1[j] = dynmath::fma_(wgt, x, a1[j]);
      a2[j] = dynmath::fma_(wgt, x2, a2[j]);
    }
    // the band of row t+1
    const int next_lo = band_mid(t + 1, ratio) - bw;
    if (__builtin_expect(next_lo != lo, 0)) {  // wave-uniform: the window moves up by one column between rows t and t+1
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int nj = lo + pmod(lane * CPL + j - lo);
        // the slot handed on at the PREVIOUS move (column lo-1) has received nothing but zeros since: the column it
        // now stands for is far above the band, no mass can be there
        if (nj == lo - 1 + P) {
          if (lo - 1 >= 1 && lo - 1 < N) {
            cw[lo - 2] = aw[j];
            cs1[lo - 2] = a1[j];
            cs2[lo - 2] = a2[j];
          }
          awsum += aw[j];
          aw[j] = a1[j] = a2[j] = 0.0;
        }
      }
      lo = next_lo;
    }
  };

  for (int tb0 = 1; tb0 < T; tb0 += 64) {
    const int idx = tb0 - 1 + lane;  // row t scores sample t - 1
    const double xs = sg[min(idx, T - 2)];
    asm volatile("" ::"v"(xs));  // the load's wait belongs here, not into the row loop (see forward_sweep)
    const int iend = min(64, T - tb0);
#pragma unroll 1
    for (int i = 0; i < iend; ++i) row(tb0 + i, readlane_f64(xs, i));
  }
  wait_vmcnt<0>();  // the clamped tail DMAs still target this wave's LDS ring
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int nj = lo + pmod(lane * CPL + j - lo);
    const int cidx = (nj == lo - 1 + P) ? lo - 1 : nj;  // the slot handed on at the last move still holds that column's sums
    if (cidx >= 1 && cidx < N) {
      cw[cidx - 1] = aw[j];
      cs1[cidx - 1] = a1[j];
      cs2[cidx - 1] = a2[j];
    }
    awsum += aw[j];
  }
  for (int off = 32; off >= 1; off >>= 1) awsum += __shfl_xor(awsum, off);
  // expected transition counts: EVERY path from (0,0) to (T-1,N-1) takes N-1 moves E->M, each followed by the forced
  // M->E, and spends its other T-1 - 2(N-1) steps on E->E
  if (lane == 0) {
    tb.trans[2 * rd.read] = (double)(N - 1);
    tb.trans[2 * rd.read + 1] = (double)(T - 1) - 2.0 * (double)(N - 1);
  }
  // all of the mass must have arrived in E(T-1, N-1), and every sample must have carried weight 1
  const int nf = band_mid(T - 1, ratio);
  const int sf = pmod(nf);
  double gf = 0.0;
#pragma unroll
  for (int j = 0; j < CPL; ++j)
    if (j == sf % CPL) gf = gE[j];
  gf = readlane_f64(gf, sf / CPL);
  const double S = (double)(T - 1);
  const bool ok = fabs(gf - 1.0) <= 1e-6 && fabs(awsum - S) <= 1e-6 * S;  // (NaN fails)
  return ok ? Zb : NEG_INF;
}

// ---- read queue and page pool -------------------------------------------------------------------------
// Shared control words are only ever touched with agent-scope atomics (sc1 accesses: the per-XCD L2s are
// not coherent with each other, MI355X_MICROARCH.md "inter-workgroup visibility"); a lock serialises the
// queue head and the free-page stack. It is taken twice per read (take pages, give them back), i.e. every
// ~20 us at full speed, by ONE lane of the wave.
__device__ __forceinline__ uint32_t ctl_load(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ctl_store(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every wait below is bounded: a wave that has waited for seconds (a lost lock, a page count that never
// recovers) raises the abort word, which drains the queue -- the grid always terminates, and the host
// reports the launch as failed instead of hanging the device.
constexpr int CTL_LOCK = 0, CTL_HEAD = 1, CTL_FREE = 2, CTL_ABORT = 3, CTL_PROVISIONED = 4, CTL_WAITING = 5;
// (round 5, paged sessions: PagePool::reserve_after) a wave that has waited for pages for ~2 ms RESERVES its request: [6] pages the others must leave on the stack,
// [7] the reserving wave (slot + 1; 0 = nobody). Without it a request for many pages (a 100 k-sample read: ~390) can lose
// against a stream of small ones for as long as reads keep coming -- in a resident session, for ever.
constexpr int CTL_RESERVE = 6, CTL_RESERVE_OWNER = 7;
constexpr int LOCK_SPINS_MAX = 1 << 24;   // x ~0.3 us
constexpr int PAGE_WAITS_MAX = 1 << 18;   // x ~30 us

__device__ __forceinline__ void queue_lock(uint32_t* ctl, int lane) {
  if (lane == 0) {
    for (int spins = 0;; ++spins) {
      uint32_t expect = 0;
      if (__hip_atomic_compare_exchange_strong(&ctl[CTL_LOCK], &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT))
        break;
      if (spins > LOCK_SPINS_MAX) {
        ctl_store(&ctl[CTL_ABORT], 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void queue_unlock(uint32_t* ctl, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every access of the critical section has completed
  if (lane == 0) ctl_store(&ctl[CTL_LOCK], 0u);
}

// A wave KEEPS the pages of the read it has finished: the queue is sorted longest first, so the next
// read it claims never needs more (no lock, no fence in the steady state -- one atomic add on the queue
// head per read). The shared free-page stack (behind the lock) is only touched
//   * by waves that start without an arena because the pool could not serve every wave (they wait
//     for pages while holding none, so waiting can never deadlock),
//   * by waves whose next read needs much less than they hold while some wave is waiting, and
//   * at exit, while a claimed read still lacks its pages.

// pt[off .. off+count) -> free list. Every store of this wave to those pages must have been written
// back from this XCD's L2 before a wave on another XCD fills them again (agent-scope release).
__device__ __forceinline__ void pages_give(const PagePool& pool, const WaveCtx& w, uint32_t off, uint32_t count) {
  uint32_t* ctl = pool.ctl;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  queue_lock(ctl, w.lane);
  const uint32_t fc = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl_load(&ctl[CTL_FREE]));
  for (uint32_t k = w.lane; k < count; k += 64) ctl_store(&pool.free_list[fc + k], w.pt[off + k]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (w.lane == 0) ctl_store(&ctl[CTL_FREE], fc + count);
  queue_unlock(ctl, w.lane);
}

// `count` pages from the free list -> pt[0 .. count); the wave holds no pages while it waits. `me` = the wave's slot + 1.
// Returns false when the launch was aborted.
__device__ __forceinline__ bool pages_take(const PagePool& pool, const WaveCtx& w, uint32_t count, uint32_t me) {
  uint32_t* ctl = pool.ctl;
  auto sload = [&](int word) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl_load(&ctl[word])); };
  if (w.lane == 0) __hip_atomic_fetch_add(&ctl[CTL_WAITING], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool ok = false;
  for (int waits = 0; !ok; ++waits) {
    if (sload(CTL_ABORT) != 0) break;
    if (waits > PAGE_WAITS_MAX) {
      if (w.lane == 0) ctl_store(&ctl[CTL_ABORT], 1u);
      break;
    }
    // look before locking: the lock is only worth taking when the request can be served -- beside what another wave has
    // reserved -- or when it is time to reserve
    const uint32_t owner = sload(CTL_RESERVE_OWNER);
    const uint32_t kept = (owner != 0 && owner != me) ? sload(CTL_RESERVE) : 0u;
    const bool reserve_now = pool.reserve_after != 0 && owner == 0 && (uint32_t)waits >= pool.reserve_after;
    if (!reserve_now && sload(CTL_FREE) < count + kept) {
      for (int k = 0; k < 8; ++k) __builtin_amdgcn_s_sleep(127);  // pages come back every ~20 us at best
      continue;
    }
    queue_lock(ctl, w.lane);
    const uint32_t fc = sload(CTL_FREE);
    const uint32_t owner2 = sload(CTL_RESERVE_OWNER);
    const uint32_t kept2 = (owner2 != 0 && owner2 != me) ? sload(CTL_RESERVE) : 0u;
    if (fc >= count + kept2) {
      for (uint32_t k = w.lane; k < count; k += 64) w.pt[k] = ctl_load(&pool.free_list[fc - count + k]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (w.lane == 0) {
        ctl_store(&ctl[CTL_FREE], fc - count);
        if (owner2 == me) {
          ctl_store(&ctl[CTL_RESERVE], 0u);
          ctl_store(&ctl[CTL_RESERVE_OWNER], 0u);
        }
      }
      ok = true;
    } else if (pool.reserve_after != 0 && owner2 == 0 && (uint32_t)waits >= pool.reserve_after) {
      if (w.lane == 0) {  // from here on the others leave `count` pages on the stack
        ctl_store(&ctl[CTL_RESERVE], count);
        ctl_store(&ctl[CTL_RESERVE_OWNER], me);
      }
    }
    queue_unlock(ctl, w.lane);
    if (!ok)
      for (int k = 0; k < 8; ++k) __builtin_amdgcn_s_sleep(127);
  }
  if (!ok && sload(CTL_RESERVE_OWNER) == me) {  // aborted while holding a reservation
    queue_lock(ctl, w.lane);
    if (w.lane == 0) {
      ctl_store(&ctl[CTL_RESERVE], 0u);
      ctl_store(&ctl[CTL_RESERVE_OWNER], 0u);
    }
    queue_unlock(ctl, w.lane);
  }
  if (w.lane == 0) __hip_atomic_fetch_sub(&ctl[CTL_WAITING], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return ok;
}

// ---- one read's whole pipeline (the body both read queues share) -----------------------------------------
// What a read touches beside the lattice pool: the arrays of its batch (classic launch: the launch's; resident queue: its
// ticket's) and the model's transition constants.
struct ReadIO {
  ReadState* st;
  TraceBuffers tb;
  TrainBuffers tr;
  double m1, e2;
  int z_fail_status;
};

// wave-cycles per phase, accumulated over the reads of a wave
struct WaveStats {
  uint64_t cyc_b = 0, cyc_f = 0, cyc_t = 0, cyc_w = 0, cyc_bs = 0, cyc_fs = 0, rows_cert = 0;
  uint32_t n_fallback = 0;
};

// t1: s_memtime when the read's pages were in place. sig / par are `const __restrict__` kernel parameters (or pointers
// derived from one): see k_read_queue.
template <int JOB, bool MIXED>
__device__ __forceinline__ void run_read(const ReadDesc& rd, const WaveCtx& w, const PagePool& pool, const ReadIO& io,
                                         const double* __restrict__ sig, const Emis* __restrict__ par,
                                         const SoftplusNode* s_tab, unsigned ring_base, lds_u64_t* sb, WaveStats& ws, uint64_t t1) {
  constexpr bool LATTICE = JOB != JOB_Z;
  // READ_STRICT: every row bit for bit. READ_STRICT_START: the backward sweep and the first rd.strict_rows rows of the
  // forward sweep -- every decision up to that row is then the reference's own (see forward_sweep), which is what a
  // read-start structural tie needs, at about half the price
  const bool strict = MIXED && (rd.flags & (READ_STRICT | READ_STRICT_START)) != 0;  // wave-uniform
  double Zb, Zf;
  uint64_t t2;
  if constexpr (MIXED) {
    static_assert(JOB == JOB_ALIGN || JOB == JOB_ALIGN_INPLACE, "strict reads exist for align(calc=true) only");
  }
  if (MIXED && strict) {
    const int strict_rows = (rd.flags & READ_STRICT) ? 0x7fffffff : (int)rd.strict_rows;
    Zb = backward_sweep<LATTICE, ARITH_STRICT>(rd, w, sig, par, pool.ws, io.m1, io.e2, s_tab, &ws.n_fallback);
    t2 = __builtin_amdgcn_s_memtime();
    ws.rows_cert += (uint64_t)rd.T + (uint64_t)min((int)rd.T, strict_rows);
    if (JOB == JOB_ALIGN_INPLACE)
      Zf = forward_sweep<true, true, MIXED>(rd, w, sig, par, pool.ws, reinterpret_cast<float*>(pool.ws), pool.bits, Zb, io.m1, io.e2, s_tab, ring_base, strict_rows, &ws.n_fallback);
    else
      Zf = forward_sweep<true, false, MIXED>(rd, w, sig, par, pool.ws, pool.lpe, pool.bits, Zb, io.m1, io.e2, s_tab, ring_base, strict_rows, &ws.n_fallback);
  } else {
    if constexpr (JOB == JOB_TRAIN || JOB == JOB_TRAIN_ZCHECK) {
      // backward sweep in the log domain (the emission's constant folded), then the posterior chain
      Zb = backward_sweep<LATTICE, ARITH_FOLDED>(rd, w, sig, par, pool.ws, io.m1, io.e2, s_tab);
      t2 = __builtin_amdgcn_s_memtime();
      Zf = forward_train_chain(rd, w, sig, pool.ws, io.tr, Zb, s_tab, ring_base);
      if constexpr (JOB == JOB_TRAIN_ZCHECK) {
        // dyn_aligner_set_train_zcheck: the reference's own refusal rule (NT_aligner_api.cpp:619-625) on top of the
        // chain's -- a forward value of Z (the cheap Z-only sweep: no lattice traffic) must agree with the backward one
        // to 1e-8 per lattice cell, so that `.errors` lists the reads the reference lists (|Z| ~ 1e12 and beyond)
        const double Zf_log = forward_sweep<false, false, false>(rd, w, sig, par, nullptr, nullptr, nullptr, 0.0, io.m1, io.e2, s_tab, ring_base);
        if (!z_ok(rd, Zf_log, Zb)) Zf = NEG_INF;
      }
    } else {
      Zb = backward_sweep<LATTICE, JOB == JOB_Z ? ARITH_FOLDED : ARITH_DEFAULT>(rd, w, sig, par, pool.ws, io.m1, io.e2, s_tab);
      t2 = __builtin_amdgcn_s_memtime();
      if (JOB == JOB_ALIGN) {
        Zf = forward_sweep<true, false, false>(rd, w, sig, par, pool.ws, pool.lpe, pool.bits, Zb, io.m1, io.e2, s_tab, ring_base);
      } else if (JOB == JOB_ALIGN_INPLACE) {
        Zf = forward_sweep<true, true, false>(rd, w, sig, par, pool.ws, reinterpret_cast<float*>(pool.ws), pool.bits, Zb, io.m1, io.e2, s_tab, ring_base);
      } else {
        Zf = forward_sweep<false, false, false>(rd, w, sig, par, nullptr, nullptr, nullptr, 0.0, io.m1, io.e2, s_tab, ring_base);
      }
    }
  }
  ws.cyc_b += t2 - t1;
  const uint64_t t3 = __builtin_amdgcn_s_memtime();
  ws.cyc_f += t3 - t2;
  if (MIXED && strict) {
    ws.cyc_bs += t2 - t1;
    ws.cyc_fs += t3 - t2;
  }

  int status = z_ok(rd, Zf, Zb) ? 0 : io.z_fail_status;
  uint32_t n_seg = 0;
  if ((JOB == JOB_ALIGN || JOB == JOB_ALIGN_INPLACE) && status == 0) {
    const float* lp = JOB == JOB_ALIGN ? pool.lpe : reinterpret_cast<const float*>(pool.ws);
    const bool complete = traceback(rd, w, lp, pool.bits, io.tb, JOB == JOB_ALIGN_INPLACE, sb);
#ifdef DYN_EXP_TRACE_SPLIT  // development: the statistics slots of the certified sweeps carry traceback / mpost cycles instead
    const uint64_t t4 = __builtin_amdgcn_s_memtime();
    ws.cyc_bs += t4 - t3;
#endif
    if (complete) {
      if (JOB == JOB_ALIGN) {
        // segrow was written by other lanes of this wave through global memory
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        mpost(rd, w, pool.ws, pool.lpe, sig, par, io.tb, Zb, io.m1);
      }
#ifdef DYN_EXP_TRACE_SPLIT
      ws.cyc_fs += __builtin_amdgcn_s_memtime() - t4;
#endif
      n_seg = rd.N - 1;
    } else {
      status = 7;  // DYN_READ_INTERNAL
    }
  }
  if (w.lane == 0) {
    ReadState s;
    s.Zb = Zb;
    s.Zf = Zf;
    s.status = status;
    s.n_segments = n_seg;
    io.st[rd.read] = s;
  }
  ws.cyc_t += __builtin_amdgcn_s_memtime() - t3;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// per-column emission table: par[i] = model[kmers[i]]  (aligner.cpp:241-245 scoreKmer lookup)
// ---------------------------------------------------------------------------------------------
__global__ void k_prep_params(const int32_t* __restrict__ kmers, const Emis* __restrict__ model,
                              Emis* __restrict__ par, uint64_t total, uint32_t num_kmers) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  // the code is clamped into the table: the columns of a read whose sequence turned out invalid half-way through the
  // encoding never reach the read queue, but they are part of this range, and whatever a table index is made of must
  // not be able to take the load out of bounds
  for (; i < total; i += stride) par[i] = model[min((uint32_t)kmers[i], num_kmers - 1u)];
}

__global__ void k_pool_init(PagePool pool, uint32_t first_free, uint32_t n_static) {
  const uint32_t n_free = pool.n_pages - first_free;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_free) pool.free_list[i] = first_free + i;
  // queue head and provisioned reads start behind the first round, whose pages the host reserved
  if (i < QUEUE_CTL_WORDS) pool.ctl[i] = (i == CTL_HEAD || i == CTL_PROVISIONED) ? n_static : (i == CTL_FREE) ? n_free : 0u;
}

// ---------------------------------------------------------------------------------------------
// The read queue. JOB_Z: align(calc_probabilities=false), Z check only, no stored lattice.
// JOB_ALIGN / JOB_ALIGN_INPLACE: align(calc_probabilities=true) up to the per-row path arrays
// (k_median / k_final follow). JOB_TRAIN: train() up to the per-column sums.
// ---------------------------------------------------------------------------------------------
// The read-only inputs are separate `const __restrict__` kernel parameters on purpose: as members of the
// by-value argument struct they carry no no-alias information, and hipcc then turns the wave-uniform
// loads from them (emission parameters in the window-move blocks, read descriptors) into VECTOR loads
// guarded by s_waitcnt vmcnt(0) -- which drains the forward sweep's DMA ring every ~13 rows.
// MIXED: the launch holds reads flagged READ_STRICT; such a read takes the sweeps instantiated with the bit-for-bit
// arithmetic (dp_math_strict.hpp). The default launches (MIXED = false) do not contain that code at all.
template <int JOB, bool MIXED>
__global__ DYN_ONE_WAVE_PER_SIMD void k_read_queue(const QueueArgs q, const ReadDesc* __restrict__ descs,
                                                   const double* __restrict__ sig, const Emis* __restrict__ par,
                                                   const SoftplusNode* __restrict__ sp_tab) {
  constexpr bool LATTICE = JOB != JOB_Z;
  // + 2^(i/128) as plain doubles (exp_table128_vec) + glibc's table of the strict exp (dp_math_strict.hpp)
  constexpr int TAB_NODES = SP_NODES + dynmath::EXP128_NODES + dynmath::STRICT_EXP_WORDS / 2;
  __shared__ __attribute__((aligned(16))) SoftplusNode s_tab[TAB_NODES];
  __shared__ __attribute__((aligned(16))) double s_ring[DYN_WAVES_PER_GROUP][RING_D][P];
  __shared__ uint32_t s_pt[DYN_WAVES_PER_GROUP][PT_MAX];
  for (int i = threadIdx.x; i < TAB_NODES; i += 64 * DYN_WAVES_PER_GROUP) s_tab[i] = sp_tab[i];
  __syncthreads();
  // readfirstlane: the wave index is uniform, and the compiler must know it -- otherwise the read
  // descriptor, every pointer and loop bound derived from it live in VGPRs and every table /
  // parameter access becomes a vector load.
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int slot = blockIdx.x * DYN_WAVES_PER_GROUP + wave;
  WaveCtx w;
  w.lane = threadIdx.x & 63;
  w.pt = (lds_u32_t*)&s_pt[wave][0];
  w.log_r = q.pool.log_rows;
  const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)&s_ring[wave][0][0];
  lds_u64_t* sb = (lds_u64_t*)&s_ring[wave][0][0];  // traceback staging: the ring is idle by then

  WaveStats ws;
  const uint64_t t_start = __builtin_amdgcn_s_memtime();
  uint32_t* ctl = q.pool.ctl;
  uint32_t have = 0;  // pages in this wave's table
  bool first = true;
  for (;;) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    int idx;
    if (first && slot < q.n_static) {
      idx = slot;  // first round: read and pages assigned by the host
    } else {
      uint32_t h = 0;
      if (w.lane == 0) h = __hip_atomic_fetch_add(&ctl[CTL_HEAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      idx = __builtin_amdgcn_readfirstlane((int)h);
    }
    if (idx >= q.n_reads || idx < 0) break;
    const ReadDesc rd = descs[idx];
    if (LATTICE) {
      if (first && rd.first_page != NO_PAGE) {
        for (uint32_t k = w.lane; k < rd.n_pages; k += 64) w.pt[k] = rd.first_page + k;
        have = rd.n_pages;
      } else {
        if (have < rd.n_pages) {  // a wave without an arena (or, with another queue order, too small a one)
          if (have) pages_give(q.pool, w, 0, have);
          have = 0;
          if (!pages_take(q.pool, w, rd.n_pages, (uint32_t)slot + 1u)) break;
          have = rd.n_pages;
        } else if (have - rd.n_pages >= 8 && 8 * (have - rd.n_pages) >= have &&
                   __builtin_amdgcn_readfirstlane((int)ctl_load(&ctl[CTL_WAITING])) != 0) {
          pages_give(q.pool, w, rd.n_pages, have - rd.n_pages);  // a wave is waiting: hand over what this read leaves unused
          have = rd.n_pages;
        }
        if (w.lane == 0) __hip_atomic_fetch_add(&ctl[CTL_PROVISIONED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    first = false;
    wave_lds_sync();
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    ws.cyc_w += t1 - t0;
    run_read<JOB, MIXED>(rd, w, q.pool, ReadIO{q.st, q.tb, q.tr, q.m1, q.e2, q.z_fail_status}, sig, par, s_tab, ring_base, sb, ws, t1);
  }
  // leaving: while a claimed read still lacks its pages, somebody may be waiting for these
  if (LATTICE && have && (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl_load(&ctl[CTL_PROVISIONED])) < (uint32_t)q.n_reads)
    pages_give(q.pool, w, 0, have);
  if (w.lane == 0) {
    unsigned long long* stats = reinterpret_cast<unsigned long long*>(q.pool.ctl + QUEUE_STATS);
    atomicAdd(&stats[0], (unsigned long long)ws.cyc_b);
    atomicAdd(&stats[1], (unsigned long long)ws.cyc_f);
    atomicAdd(&stats[2], (unsigned long long)ws.cyc_t);
    atomicAdd(&stats[3], (unsigned long long)ws.cyc_w);
    const unsigned long long life = (unsigned long long)(__builtin_amdgcn_s_memtime() - t_start);
    atomicAdd(&stats[4], life);
    atomicMax(&stats[5], life);
#ifdef DYN_EXP_TRACE_SPLIT
    if (true) {
#else
    if (MIXED) {
#endif
      atomicAdd(&stats[6], (unsigned long long)ws.cyc_bs);
      atomicAdd(&stats[7], (unsigned long long)ws.cyc_fs);
      atomicAdd(&stats[8], (unsigned long long)ws.n_fallback);
      atomicAdd(&stats[9], (unsigned long long)ws.rows_cert);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The RESIDENT read queue (nt_kernels.hpp, SessionArgs): the waves of k_read_queue<JOB_ALIGN>, kept on the chip across
// batches. A wave claims global read indices one at a time; index g belongs to the ticket whose [base, base + n_reads)
// holds it, and a wave that has claimed an index nobody has published yet waits for that ticket to appear -- or for the
// host to close the session.
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ uint32_t sctl_load(const uint32_t* p) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl_load(p)); }

// the record of ticket `index`: 32 lanes load a dword each (agent scope: the record was written by a kernel of another
// stream, possibly on another XCD), the values are handed round as scalars
__device__ __forceinline__ SessionTicket load_ticket(const SessionTicket* ring, uint32_t index, uint32_t ring_size, int lane) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(ring + (index % ring_size));
  const uint32_t v = ctl_load(src + (lane & 31));
  uint32_t wds[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) wds[k] = (uint32_t)__builtin_amdgcn_readlane((int)v, k);
  SessionTicket t;
  __builtin_memcpy(&t, wds, sizeof t);
  return t;
}

}  // namespace

// LAYOUT 0: an arena of sa.arena_pages pages for every wave, separate float LPE (JOB_ALIGN). LAYOUT 1 / 2: the batches are
// PAGE-STARVED (an arena per wave does not fit the memory budget: reads of 100 k samples): the pool's pages are shared through
// the free list exactly as in k_read_queue -- a wave keeps what it holds, takes more only while holding none, gives a large
// surplus back when somebody waits -- with the separate (1, JOB_ALIGN) or the in-place (2, JOB_ALIGN_INPLACE) posterior layout.
template <bool MIXED, int LAYOUT>
__global__ DYN_ONE_WAVE_PER_SIMD void k_session(const SessionArgs sa, const char* __restrict__ in_base, char* out_base,
                                                const SoftplusNode* __restrict__ sp_tab) {
  constexpr int JOB = LAYOUT == 2 ? JOB_ALIGN_INPLACE : JOB_ALIGN;
  constexpr bool PAGED = LAYOUT != 0;
  constexpr int TAB_NODES = SP_NODES + dynmath::EXP128_NODES + dynmath::STRICT_EXP_WORDS / 2;
  __shared__ __attribute__((aligned(16))) SoftplusNode s_tab[TAB_NODES];
  __shared__ __attribute__((aligned(16))) double s_ring[DYN_WAVES_PER_GROUP][RING_D][P];
  __shared__ uint32_t s_pt[DYN_WAVES_PER_GROUP][PT_MAX];
  for (int i = threadIdx.x; i < TAB_NODES; i += 64 * DYN_WAVES_PER_GROUP) s_tab[i] = sp_tab[i];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t slot = blockIdx.x * DYN_WAVES_PER_GROUP + wave;
  WaveCtx w;
  w.lane = threadIdx.x & 63;
  w.pt = (lds_u32_t*)&s_pt[wave][0];
  w.log_r = sa.pool.log_rows;
  const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)&s_ring[wave][0][0];
  lds_u64_t* sb = (lds_u64_t*)&s_ring[wave][0][0];
  // this wave's arena, for the whole session (PAGED: pages come and go through the free list)
  if (!PAGED)
    for (uint32_t k = w.lane; k < sa.arena_pages; k += 64) w.pt[k] = slot * sa.arena_pages + k;
  wave_lds_sync();
  uint32_t held = 0;  // PAGED: pages in this wave's table

  uint32_t* ctl = sa.ctl;
  const uint64_t t_start = __builtin_amdgcn_s_memtime();
  uint64_t cyc_busy = 0, cyc_idle = 0, cyc_pages = 0, n_done = 0;
  uint64_t cyc_head = 0, cyc_head_pages = 0, t0 = t_start;
  uint64_t last_read = 0, last_pages = 0;  // the wave's last read: its queue index and what it waited for pages  // idle before this wave's first read; start of the current turn
  uint32_t cur = 0, tail_seen = 0;   // the ticket this wave looks at; tickets it knows to be published
  bool have = false;
  SessionTicket tk{};
  for (;;) {
    t0 = __builtin_amdgcn_s_memtime();                     // shader clock: phase shares
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz: durations
    // An aborted session takes no more reads: the waves that idled into the watchdog have left, the host publishes what is
    // incomplete again elsewhere (session_recover) -- a few busy waves must not work the queue off on their own meanwhile.
    if (sctl_load(&ctl[S_ABORT])) break;
    uint32_t h = 0;
    if (w.lane == 0) h = __hip_atomic_fetch_add(&ctl[S_HEAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
    // the ticket that holds read g (indices only grow, so it is `cur` or a later one)
    bool leave = false;
    for (;;) {
      if (have && g - tk.base < tk.n_reads) break;
      if (have) {
        ++cur;
        have = false;
      }
      while (cur >= tail_seen) {
        // closed BEFORE tail: the host closes behind its last publish, so a tail read after a set `closed` is final
        const uint32_t closed = sctl_load(&ctl[S_CLOSED]) | sctl_load(&ctl[S_ABORT]);
        tail_seen = sctl_load(&ctl[S_TAIL]);
        if (cur < tail_seen) break;
        if (closed) {
          leave = true;
          break;
        }
        if (__builtin_amdgcn_s_memrealtime() - r0 > sa.idle_limit_ticks) {  // the host is gone, or stuck: never spin for ever
          if (w.lane == 0) ctl_store(&ctl[S_ABORT], 2u);
          leave = true;
          break;
        }
        // PAGED: an idle wave does not sit on pages somebody is waiting for
        if (PAGED && held && sctl_load(&sa.pool.ctl[CTL_WAITING]) != 0) {
          pages_give(sa.pool, w, 0, held);
          held = 0;
        }
        for (int k = 0; k < 4; ++k) __builtin_amdgcn_s_sleep(127);  // ~2 us: one lane's poll per wave, agent scope
      }
      if (leave) break;
      tk = load_ticket(sa.ring, cur, sa.ring_size, w.lane);
      have = true;
    }
    if (leave) break;
    // The ticket's inputs were written while this kernel runs, into buffers earlier tickets used: drop what the scalar
    // cache and this CU's L1 may still hold of them (MI355X_MICROARCH.md, inter-workgroup visibility)
    __builtin_amdgcn_s_dcache_inv();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const ReadDesc* __restrict__ descs = reinterpret_cast<const ReadDesc*>(in_base + tk.descs_off);
    const double* __restrict__ sig = reinterpret_cast<const double*>(in_base + tk.sig_off);
    const Emis* __restrict__ par = reinterpret_cast<const Emis*>(in_base + tk.par_off);
    const ReadDesc rd = descs[g - tk.base];
    if (PAGED) {
      if (held < rd.n_pages) {
        const uint64_t tp = __builtin_amdgcn_s_memtime();
        if (held) pages_give(sa.pool, w, 0, held);
        held = 0;
        const bool got = pages_take(sa.pool, w, rd.n_pages, slot + 1u);
        last_pages = __builtin_amdgcn_s_memtime() - tp;
        cyc_pages += last_pages;   // (part of cyc_idle; apart: what the page pool costs)
        if (!got) {
          // waited for pages for seconds: the session is over; what is incomplete is published again (session_recover)
          if (w.lane == 0) ctl_store(&ctl[S_ABORT], 3u);
          break;
        }
        held = rd.n_pages;
      } else if (held > rd.n_pages && (sa.give_always || (held - rd.n_pages >= 8 && 8 * (held - rd.n_pages) >= held &&
                                                          sctl_load(&sa.pool.ctl[CTL_WAITING]) != 0))) {
        pages_give(sa.pool, w, rd.n_pages, held - rd.n_pages);  // a wave is waiting: hand over what this read leaves unused
        held = rd.n_pages;
      }
      wave_lds_sync();
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    cyc_idle += t1 - t0;
    last_read = g;
    if (!PAGED) last_pages = 0;
    if (n_done == 0) {
      cyc_head = t1 - t_start;
      cyc_head_pages = cyc_pages;
    }
    WaveStats ws;
    ReadIO io{};
    io.st = reinterpret_cast<ReadState*>(out_base + tk.st_off);
    io.tb = TraceBuffers{reinterpret_cast<double*>(out_base + tk.pp_off), reinterpret_cast<uint32_t*>(out_base + tk.pathn_off),
                         reinterpret_cast<uint32_t*>(out_base + tk.segrow_off), reinterpret_cast<double*>(out_base + tk.medhi_off),
                         reinterpret_cast<double*>(out_base + tk.medlo_off)};
    io.m1 = sa.m1;
    io.e2 = sa.e2;
    io.z_fail_status = tk.z_fail_status;
    run_read<JOB, MIXED>(rd, w, sa.pool, io, sig, par, s_tab, ring_base, sb, ws, t1);
    // everything this wave wrote for the read (state, path arrays, segment rows) must have left this XCD's L2 before the
    // ticket's counter says so: the per-segment kernels and the copies that follow run elsewhere
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (w.lane == 0) {
      uint32_t* tctl = reinterpret_cast<uint32_t*>(out_base + tk.tctl_off);
      unsigned long long* ts = reinterpret_cast<unsigned long long*>(tctl + SESSION_TSTATS);
      atomicAdd(&ts[0], (unsigned long long)ws.cyc_b);
      atomicAdd(&ts[1], (unsigned long long)ws.cyc_f);
      atomicAdd(&ts[2], (unsigned long long)ws.cyc_t);
      // the read's duration in 10 ns ticks (s_memtime runs with the shader clock: good for shares, not for times)
      atomicAdd(&ts[3], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - r1));
      if (MIXED) {
        atomicAdd(&ts[6], (unsigned long long)ws.cyc_bs);
        atomicAdd(&ts[7], (unsigned long long)ws.cyc_fs);
        atomicAdd(&ts[8], (unsigned long long)ws.n_fallback);
        atomicAdd(&ts[9], (unsigned long long)ws.rows_cert);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t before = __hip_atomic_fetch_add(&tctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before + 1u == tk.n_reads)  // the ticket is complete: tell the host (fine-grained pinned memory)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(out_base + tk.flag_off), tk.n_reads, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    ++n_done;
    cyc_busy += __builtin_amdgcn_s_memtime() - t1;
  }
  if (PAGED && held) pages_give(sa.pool, w, 0, held);  // (a wave that claimed a read may still be waiting for these)
  if (w.lane == 0) {
    unsigned long long* stats = reinterpret_cast<unsigned long long*>(ctl + SESSION_STATS);
    const uint64_t t_end = __builtin_amdgcn_s_memtime();
    const unsigned long long life = (unsigned long long)(t_end - t_start);
    // where the idle share sits: before a wave's first read (the first ticket's inputs on their way), and in its LAST turn (no
    // read left to claim: the other waves' last reads, the host's close) -- what is left of stats[1] lies between tickets
    atomicAdd(&stats[6], (unsigned long long)cyc_head);
    atomicAdd(&stats[7], (unsigned long long)(t_end - t0));
    if (PAGED) atomicAdd(&stats[8], (unsigned long long)cyc_head_pages);
    atomicAdd(&stats[9], (unsigned long long)(((t_end - t0) >> 10) * ((t_end - t0) >> 10)));  // (shape of the end: E[x^2] / E[x]^2)
    atomicMax(&stats[10], (unsigned long long)(t_end - t0));
    // the read that ended LAST: (end of the wave's work in 1 024-cycle units) << 24 | queue index; its page wait rides on the
    // same key in the next word (both maxima are taken by the same wave unless two waves end within 1 024 cycles)
    if (n_done) {
      const unsigned long long key = (unsigned long long)((t0 - t_start) >> 10) << 24;
      atomicMax(&stats[11], key | (unsigned long long)(last_read & 0xffffffu));
      atomicMax(&stats[12], key | (unsigned long long)(min((unsigned long long)(last_pages >> 10), 0xffffffull)));
    }
    atomicAdd(&stats[0], (unsigned long long)cyc_busy);
    atomicAdd(&stats[1], (unsigned long long)cyc_idle);
    atomicAdd(&stats[2], life);
    atomicMax(&stats[3], life);
    atomicAdd(&stats[4], (unsigned long long)n_done);
    if (PAGED) atomicAdd(&stats[5], (unsigned long long)cyc_pages);
  }
}

// one wave: lanes 0..31 store a dword of the record each, then lane 0 moves the tail
__global__ void k_session_publish(SessionTicket* ring, uint32_t* ctl, const SessionTicket tk, uint32_t index, uint32_t ring_size) {
  const int lane = threadIdx.x;
  uint32_t wds[32];
  __builtin_memcpy(wds, &tk, sizeof tk);
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < 32; ++k) mine = (lane == k) ? wds[k] : mine;
  if (lane < 32) ctl_store(reinterpret_cast<uint32_t*>(ring + (index % ring_size)) + lane, mine);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) ctl_store(&ctl[S_TAIL], index + 1u);
}

__global__ void k_session_close(uint32_t* ctl) {
  if (threadIdx.x == 0) ctl_store(&ctl[S_CLOSED], 1u);
}

// ---------------------------------------------------------------------------------------------
// K_median: formattedMedian (aligner.cpp:247-263) by rank counting. One thread per path row;
// the segment of column n spans rows [segrow[n-1], segrow[n]) (last column: up to T-1).
// Ties are broken by row so ranks are a permutation. Rank counting costs L compares per row, L^2 per
// segment: fine for the usual dwell of ~10 rows, not for a stall (a pore that sits on one k-mer for
// 20 000 samples would cost 4e8 compares); segments longer than MEDIAN_SHORT_MAX rows are left to
// k_median_long.
// ---------------------------------------------------------------------------------------------
constexpr int MEDIAN_SHORT_MAX = 256;

__global__ void k_median(const ReadDesc* __restrict__ descs, int n_reads, uint64_t rows_total, const ReadState* __restrict__ st,
                         TraceBuffers tb) {
  // one thread per path row of the WHOLE batch (rows_total = sum of T): the read is found by bisection over the
  // descriptors' path offsets (ascending in processing order). Rounds 1-3 launched max_T / 256 blocks for every read:
  // in a batch of reads of 10 k .. 100 k samples half of the blocks found nothing to do (3.8 ms per config-3 launch).
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows_total) return;
  int lo_i = 0, hi_i = n_reads - 1;
  while (lo_i < hi_i) {  // last read whose path_off <= g
    const int mid = (lo_i + hi_i + 1) >> 1;
    if (descs[mid].path_off <= g) lo_i = mid;
    else hi_i = mid - 1;
  }
  const ReadDesc rd = descs[lo_i];
  if (st[rd.read].status != 0) return;
  const int T = (int)rd.T, N = (int)rd.N;
  const int t = (int)(g - rd.path_off);
  if (t < 1 || t >= T) return;
  const double* __restrict__ pp = tb.pp + rd.path_off;
  const uint32_t* __restrict__ segrow = tb.segrow + rd.seg_off;
  const int n = (int)(tb.pathn[rd.path_off + t] & 0x7fffffffu);
  const int a = (int)segrow[n - 1];
  const int b = (n < N - 1) ? (int)segrow[n] : T;
  const int L = b - a;
  if (L > MEDIAN_SHORT_MAX) return;
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

// ---------------------------------------------------------------------------------------------
// K_median_long: the same two order statistics for segments longer than MEDIAN_SHORT_MAX rows, by an
// 8-bit-per-pass radix select over the bit patterns (posteriors are non-negative doubles: value order =
// unsigned order of the bits): 8 passes + 1 over the segment, O(L) instead of O(L^2). One 256-thread
// block per read; reads without a long segment leave after one strided look at their segment table.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_median_long(const ReadDesc* __restrict__ descs, const ReadState* __restrict__ st,
                                                     TraceBuffers tb) {
  __shared__ uint32_t s_hist[256];
  __shared__ unsigned long long s_prefix, s_maxless;
  __shared__ uint32_t s_k, s_cntless;
  const ReadDesc rd = descs[blockIdx.x];
  if (st[rd.read].status != 0) return;
  const int T = (int)rd.T, N = (int)rd.N;
  const int tid = threadIdx.x;
  const double* __restrict__ pp = tb.pp + rd.path_off;
  const uint32_t* __restrict__ segrow = tb.segrow + rd.seg_off;
  int any = 0;
  for (int i = tid; i < N - 1; i += 256) {
    const int a = (int)segrow[i], b = (i + 1 < N - 1) ? (int)segrow[i + 1] : T;
    any |= (b - a > MEDIAN_SHORT_MAX);
  }
  if (!__syncthreads_or(any)) return;
  for (int i = 0; i < N - 1; ++i) {  // block-uniform walk over the segments of this read
    const int a = (int)segrow[i], b = (i + 1 < N - 1) ? (int)segrow[i + 1] : T;
    const int L = b - a;
    if (L <= MEDIAN_SHORT_MAX) continue;
    unsigned long long prefix = 0, mask = 0;
    uint32_t k = (uint32_t)(L >> 1);  // rank of the upper middle element
    for (int shift = 56; shift >= 0; shift -= 8) {
      s_hist[tid] = 0;
      __syncthreads();
      for (int u = a + tid; u < b; u += 256) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(pp[u]);
        if ((bits & mask) == prefix) atomicAdd(&s_hist[(bits >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t cum = 0, bin = 0;
        for (; bin < 255; ++bin) {
          if (k < cum + s_hist[bin]) break;
          cum += s_hist[bin];
        }
        s_k = k - cum;
        s_prefix = prefix | ((unsigned long long)bin << shift);
      }
      __syncthreads();
      prefix = s_prefix;
      k = s_k;
      mask |= 0xffull << shift;
    }
    const double hi = __longlong_as_double((long long)prefix);
    double lo = hi;
    if (!(L & 1)) {  // rank mid-1: another copy of hi, or the largest element below it
      if (tid == 0) {
        s_cntless = 0;
        s_maxless = 0;
      }
      __syncthreads();
      uint32_t cnt = 0;
      unsigned long long mx = 0;
      for (int u = a + tid; u < b; u += 256) {
        const double y = pp[u];
        if (y < hi) {
          ++cnt;
          mx = max(mx, (unsigned long long)__double_as_longlong(y));
        }
      }
      if (cnt) {
        atomicAdd(&s_cntless, cnt);
        atomicMax(&s_maxless, mx);
      }
      __syncthreads();
      if ((uint32_t)(L >> 1) - 1u < s_cntless) lo = __longlong_as_double((long long)s_maxless);
      __syncthreads();
    }
    if (tid == 0) {
      tb.med_hi[rd.seg_off + i] = hi;
      tb.med_lo[rd.seg_off + i] = lo;
    }
  }
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
// P1/P2 on the device: the per-read preprocessing of segment.py:146-153 / train.py:163-170.
//   x = REAL(raw[i]); x -= shift; x /= scale            (REAL = double for dynamont-resquiggle,
//   hampel(x, W, n_sigmas)   (utils.py:16-43)             float for dynamont-train, whose signal
// Windows are taken from the UNFILTERED normalised signal; centres W/2 .. W/2 + nwin - 1 with        stays float32)
// nwin = S - W - (W even); a centre is replaced by the window median when
// |x - med| > n_sigmas * (1.4826 * MAD). Every operation is a single IEEE op in the reference's
// order, so the result is bit-identical to the NumPy code (tests/test_gpu_preprocess.py).
// ---------------------------------------------------------------------------------------------
// CAL: the samples are int16 ADC counts with the read's pod5 calibration (pod5_io.py:6-16, `signal_pa`): picoampere
// = (float(adc) + offset) * scale in float32, one IEEE operation each -- the value the reference's reader hands over.
template <class REAL, class RAW, bool CAL>
__global__ void k_normalise(const RAW* __restrict__ raw, const uint64_t* __restrict__ offs,
                            const double* __restrict__ shift, const double* __restrict__ scale,
                            const float* __restrict__ cal_offset, const float* __restrict__ cal_scale,
                            REAL* __restrict__ norm, int n_reads) {
  const int r = blockIdx.y;
  if (r >= n_reads) return;
  const uint64_t a = offs[r], b = offs[r + 1];
  const REAL sh = (REAL)shift[r], sc = (REAL)scale[r];
  const float co = CAL ? cal_offset[r] : 0.0f, cs = CAL ? cal_scale[r] : 1.0f;
  for (uint64_t i = a + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < b; i += (uint64_t)gridDim.x * blockDim.x) {
    REAL x;
    if (CAL) {
      float pa = (float)raw[i] + co;
      pa = pa * cs;
      x = (REAL)pa;
    } else {
      x = (REAL)raw[i];
    }
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

// per-read kernels put the read index in gridDim.y (<= 65 535): larger batches go in slices
constexpr int MAX_GRID_Y = 65535;

template <class REAL, class RAW, bool CAL>
static void preprocess_t(const RAW* raw, const uint64_t* offs, const double* shift, const double* scale,
                         const float* cal_offset, const float* cal_scale, void* norm_tmp, double* out, int n_reads,
                         uint64_t max_len, int W, double ns, hipStream_t s) {
  const int bx = (int)std::min<uint64_t>(1024, (max_len + 255) / 256);
  for (int r0 = 0; r0 < n_reads; r0 += MAX_GRID_Y) {  // offs hold absolute sample positions: slices just shift the read index
    const int nr = std::min(MAX_GRID_Y, n_reads - r0);
    hipLaunchKernelGGL((k_normalise<REAL, RAW, CAL>), dim3(bx ? bx : 1, nr), dim3(256), 0, s, raw, offs + r0, shift + r0,
                       scale + r0, CAL ? cal_offset + r0 : nullptr, CAL ? cal_scale + r0 : nullptr, (REAL*)norm_tmp, nr);
    hipLaunchKernelGGL((k_hampel<REAL>), dim3(bx ? bx : 1, nr), dim3(256), 0, s, (const REAL*)norm_tmp, offs + r0, out,
                       nr, W, ns);
  }
}

void launch_preprocess(const void* raw, int raw_dtype, int compute_f32, const uint64_t* offs,
                       const double* shift, const double* scale, const float* cal_offset, const float* cal_scale,
                       void* norm_tmp, double* out, int n_reads, uint64_t max_len, int W, double n_sigmas, hipStream_t s) {
  if (n_reads <= 0) return;
#define DYN_PRE(REAL, RAW, CAL) \
  preprocess_t<REAL, RAW, CAL>((const RAW*)raw, offs, shift, scale, cal_offset, cal_scale, norm_tmp, out, n_reads, max_len, W, n_sigmas, s)
  if (compute_f32) {
    if (raw_dtype == 0) DYN_PRE(float, float, false);
    else if (raw_dtype == 1) DYN_PRE(float, int16_t, false);
    else if (raw_dtype == 3) DYN_PRE(float, int16_t, true);
    else DYN_PRE(float, double, false);
  } else {
    if (raw_dtype == 0) DYN_PRE(double, float, false);
    else if (raw_dtype == 1) DYN_PRE(double, int16_t, false);
    else if (raw_dtype == 3) DYN_PRE(double, int16_t, true);
    else DYN_PRE(double, double, false);
  }
#undef DYN_PRE
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
void launch_prep_params(const int32_t* kmers, const Emis* model, Emis* par, uint64_t total, uint32_t num_kmers,
                        hipStream_t s) {
  if (!total || !num_kmers) return;
  const int block = 256;
  const uint64_t want = (total + block - 1) / block;
  const int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(k_prep_params, dim3(grid), dim3(block), 0, s, kmers, model, par, total, num_kmers);
}

void launch_pool_init(const PagePool& pool, uint32_t first_free, int n_static, hipStream_t s) {
  const uint32_t n = std::max<uint32_t>(pool.n_pages - first_free, QUEUE_CTL_WORDS);
  hipLaunchKernelGGL(k_pool_init, dim3((n + 255) / 256), dim3(256), 0, s, pool, first_free, (uint32_t)n_static);
}

void launch_read_queue(QueueJob job, bool with_strict, const QueueArgs& q, int n_cus, hipStream_t s) {
  if (q.n_reads <= 0) return;
  const int groups = std::min((q.n_reads + DYN_WAVES_PER_GROUP - 1) / DYN_WAVES_PER_GROUP, std::max(1, n_cus));
  const dim3 grid(groups), block(64 * DYN_WAVES_PER_GROUP);
  if (with_strict) {  // only jobs with integer outputs have strict reads
    if (job == JOB_ALIGN) hipLaunchKernelGGL((k_read_queue<JOB_ALIGN, true>), grid, block, 0, s, q, q.descs, q.sig, q.par, q.sp_tab);
    else hipLaunchKernelGGL((k_read_queue<JOB_ALIGN_INPLACE, true>), grid, block, 0, s, q, q.descs, q.sig, q.par, q.sp_tab);
    return;
  }
  switch (job) {
    case JOB_Z: hipLaunchKernelGGL((k_read_queue<JOB_Z, false>), grid, block, 0, s, q, q.descs, q.sig, q.par, q.sp_tab); break;
    case JOB_ALIGN: hipLaunchKernelGGL((k_read_queue<JOB_ALIGN, false>), grid, block, 0, s, q, q.descs, q.sig, q.par, q.sp_tab); break;
    case JOB_ALIGN_INPLACE: hipLaunchKernelGGL((k_read_queue<JOB_ALIGN_INPLACE, false>), grid, block, 0, s, q, q.descs, q.sig, q.par, q.sp_tab); break;
    case JOB_TRAIN: hipLaunchKernelGGL((k_read_queue<JOB_TRAIN, false>), grid, block, 0, s, q, q.descs, q.sig, q.par, q.sp_tab); break;
    case JOB_TRAIN_ZCHECK: hipLaunchKernelGGL((k_read_queue<JOB_TRAIN_ZCHECK, false>), grid, block, 0, s, q, q.descs, q.sig, q.par, q.sp_tab); break;
  }
}

void launch_session(bool with_strict, int layout, const SessionArgs& a, const void* in_base, void* out_base, const dynmath::SoftplusNode* sp_tab,
                    int n_cus, hipStream_t s) {
  const dim3 grid(std::max(1, n_cus)), block(64 * DYN_WAVES_PER_GROUP);
  const char* in = static_cast<const char*>(in_base);
  char* out = static_cast<char*>(out_base);
#define DYN_SESSION_LAUNCH(M, L) hipLaunchKernelGGL((k_session<M, L>), grid, block, 0, s, a, in, out, sp_tab)
  if (with_strict) {
    if (layout == 0) DYN_SESSION_LAUNCH(true, 0);
    else if (layout == 1) DYN_SESSION_LAUNCH(true, 1);
    else DYN_SESSION_LAUNCH(true, 2);
  } else {
    if (layout == 0) DYN_SESSION_LAUNCH(false, 0);
    else if (layout == 1) DYN_SESSION_LAUNCH(false, 1);
    else DYN_SESSION_LAUNCH(false, 2);
  }
#undef DYN_SESSION_LAUNCH
}

void launch_session_publish(SessionTicket* ring, uint32_t* ctl, const SessionTicket& tk, uint32_t index, uint32_t ring_size, hipStream_t s) {
  hipLaunchKernelGGL(k_session_publish, dim3(1), dim3(64), 0, s, ring, ctl, tk, index, ring_size);
}

void launch_session_close(uint32_t* ctl, hipStream_t s) { hipLaunchKernelGGL(k_session_close, dim3(1), dim3(64), 0, s, ctl); }

void launch_segments(const ReadDesc* descs, int n_reads, uint64_t rows_total, uint32_t max_N, const ReadState* st,
                     TraceBuffers tb, SegRow* rows, int kmer_size, hipStream_t s) {
  if (n_reads > 0 && rows_total)
    hipLaunchKernelGGL(k_median, dim3((unsigned)((rows_total + 255) / 256)), dim3(256), 0, s, descs, n_reads, rows_total, st, tb);
  for (int r0 = 0; r0 < n_reads; r0 += MAX_GRID_Y) {
    const int nr = std::min(MAX_GRID_Y, n_reads - r0);
    hipLaunchKernelGGL(k_median_long, dim3(nr), dim3(256), 0, s, descs + r0, st, tb);
    hipLaunchKernelGGL(k_final, dim3((max_N + 255) / 256, nr), dim3(256), 0, s, descs + r0, st, tb, rows, kmer_size);
  }
}

}  // namespace dynk
