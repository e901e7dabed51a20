// dp_math.hpp -- fp64 arithmetic of one lattice cell, shared by the HIP kernels and by a
// host-side accuracy test (tests/test_dp_math.py compiles it with g++).
//
// Reference arithmetic being reproduced (relative to /root/reference):
//   log_normal_pdf  src/cpp/aligner.cpp:287-292   -0.5*z*z - log(stdev) - 0.5*log(2*pi), z = (x-mean)/stdev
//   logPlus         src/cpp/aligner.cpp:276-285   hi + log1p(exp(lo - hi)), -inf operands pass through
//
// ROCm's ocml log1p(double) alone is ~135 VALU instructions (exp ~42); one logPlus per cell per
// pass is the dominant cost of the whole path, so softplus g(d) = log1p(exp(d)), d <= 0, is
// evaluated here in ~45 fp64 instructions with |error| <= ~1.5e-16 absolute (<= ~1 ulp of the
// values 0.35..0.69 it returns most often), i.e. the accuracy class of glibc's exp+log1p:
//   exp:   d = k ln2 + r, |r| <= ln2/2, exp(r) = 1 + r + r^2 Q9(r)              (Q9: degree 9)
//   log1p: e in [0,1]; e <= sqrt2-1: 2 atanh(e/(2+e)); else ln2 + 2 atanh((e-1)/(e+3));
//          atanh(t) = t + t s F6(s), s = t^2 <= 0.0295                           (F6: degree 6)
// Coefficients: Chebyshev-node interpolation in 60-digit arithmetic (mpmath), rounded to fp64.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DYN_HD __host__ __device__ __forceinline__
#else
#define DYN_HD inline
#endif

namespace dynmath {

constexpr double NEG_INF = -__builtin_huge_val();

DYN_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// max as the hardware does it (a NaN operand loses). __builtin_fmax means the same, but the compiler canonicalises
// every operand it cannot prove NaN-free first (v_max_f64 x, x, x): one wasted instruction per operand in the
// training sweeps, where the values come out of selects.
DYN_HD double max_hw(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return __builtin_fmax(a, b);
#endif
}
DYN_HD double min_hw(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return __builtin_fmin(a, b);
#endif
}

// 1/x to ~1 ulp for x in [2, 4]: hardware seed + two Newton steps (same structure the compiler
// uses for fp64 division, minus scaling/fixup that this range never needs).
DYN_HD double rcp_seed(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcp(x);
#else
  return (double)(1.0f / (float)x);
#endif
}

DYN_HD double div_2to4(double num, double den) {
  double r = rcp_seed(den);
  double e = fma_(-den, r, 1.0);
  r = fma_(r, e, r);
  e = fma_(-den, r, 1.0);
  r = fma_(r, e, r);
  double q = num * r;
  double rem = fma_(-den, q, num);
  return fma_(rem, r, q);
}

// exp(d) for d in [-1000, 0] (callers clamp). Underflows gracefully to denormal/0 via ldexp.
DYN_HD double exp_nonpos(double d) {
  const double LOG2E = 0x1.71547652b82fep+0;
  const double LN2_HI = 0x1.62e42fee00000p-1;
  const double LN2_LO = 0x1.a39ef35793c76p-33;
  double kf = __builtin_rint(d * LOG2E);
  double r = fma_(-kf, LN2_HI, d);
  r = fma_(-kf, LN2_LO, r);
  double q = 0x1.af390ba7e6f47p-26;
  q = fma_(q, r, 0x1.2891d4ffbb0f9p-22);
  q = fma_(q, r, 0x1.71de0d85293b8p-19);
  q = fma_(q, r, 0x1.a019b8ca26fcfp-16);
  q = fma_(q, r, 0x1.a01a01a7cebcdp-13);
  q = fma_(q, r, 0x1.6c16c1789d1d7p-10);
  q = fma_(q, r, 0x1.11111111109a6p-7);
  q = fma_(q, r, 0x1.5555555553d37p-5);
  q = fma_(q, r, 0x1.5555555555556p-3);
  q = fma_(q, r, 0x1.0000000000001p-1);
  double p = fma_(r * r, q, r) + 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ldexp(p, (int)kf);
#else
  return std::ldexp(p, (int)kf);
#endif
}

// log1p(e) for e in [0, 1]. c = 1 selects the upper branch (e > sqrt2 - 1).
DYN_HD double log1p_unit(double e) {
  const double SQRT2M1 = 0x1.a827999fcef32p-2;
  const double LN2 = 0x1.62e42fefa39efp-1;
  const double c = (e > SQRT2M1) ? 1.0 : 0.0;
  const double num = e - c;
  const double den = (e + 2.0) + c;
  const double t = div_2to4(num, den);
  const double s = t * t;
  double f = 0x1.2b6686d1072f3p-4;
  f = fma_(f, s, 0x1.39fd39474ad34p-4);
  f = fma_(f, s, 0x1.7462bd8e53c17p-4);
  f = fma_(f, s, 0x1.c71c62c63e016p-4);
  f = fma_(f, s, 0x1.2492492e09d1ap-3);
  f = fma_(f, s, 0x1.9999999995204p-3);
  f = fma_(f, s, 0x1.5555555555558p-2);
  const double a = fma_(t * s, f, t);  // atanh(t)
  return fma_(c, LN2, a + a);
}

// softplus for d <= 0 (NaN and -inf map to 0 through the clamp).
DYN_HD double softplus_nonpos(double d) {
  d = __builtin_fmax(d, -1000.0);
  return log1p_unit(exp_nonpos(d));
}

// aligner.cpp:276-285. -inf operands: d = -inf or NaN -> clamp -> g = +0 -> hi + 0 == hi, so
// logPlus(x, -inf) == x and logPlus(-inf, -inf) == -inf exactly, as in the reference.
DYN_HD double log_plus(double x, double y) {
  const double hi = __builtin_fmax(x, y);
  const double lo = __builtin_fmin(x, y);
  return hi + softplus_nonpos(lo - hi);
}

// Per-k-mer emission constants kept with each band slot.
struct Emis {
  double mean;
  double inv_stdev;      // 1/stdev rounded to nearest
  double neg_log_stdev;  // -log(stdev); std::log evaluated on the host by the libm the reference calls
  double stdev;          // only for log_normal_pdf_exact (host tests)
};

constexpr double HALF_LOG_2PI = 0x1.d67f1c864beb4p-1;  // 0.5*log(2*pi) = 0.91893853320467274178

// aligner.cpp:287-292 with the reference's evaluation order: z = diff/stdev formed as diff*inv + one
// FMA residual correction (= the correctly rounded quotient, Markstein), then
// (-0.5*z*z - log(stdev)) - 0.5*log(2*pi). Bit-identical to the CPU expression (tests/test_dp_math.py);
// 8 fp64 operations. Kept as the yardstick for the 5-operation form the kernels run.
DYN_HD double log_normal_pdf_exact(double x, double mean, double stdev, double log_stdev) {
  const double inv = 1.0 / stdev;
  const double diff = x - mean;
  double z = diff * inv;
  const double rem = fma_(-z, stdev, diff);
  z = fma_(rem, inv, z);
  double t = -0.5 * z;
  t = t * z;
  t = t - log_stdev;
  return t - HALF_LOG_2PI;
}

// The kernels' emission, 5 fp64 operations instead of 8:
//   z = diff * (1/stdev);  t = z*z;  e = fma(t, -0.5, -log(stdev)) - 0.5*log(2*pi)
// The tail rounds exactly like the reference's (-0.5*z*z - log(stdev)) - 0.5*log(2*pi): halving is exact,
// so fma(t, -0.5, c) = RN(RN(-0.5 z z) + c). Only the quotient is not corrected (z differs from diff/stdev
// in its last bit in a fraction of the cases). Why not fewer: rounds 1-2 pre-added the two constants and
// fused them into one FMA (4 operations; a 3-operation form with 1/(stdev sqrt 2) was tried too). Where two
// neighbouring columns carry the same k-mer the traceback's comparison is a tie in exact arithmetic and
// the reference's choice rests on the last bits of exactly these sums: replaying the
// oracle's control flow with each candidate arithmetic on 1 000 short reads (tests/tie_parity.py), the
// 3-, 4- and 5-operation forms give 17, 11 and 3 reads whose borders differ from the reference's -- 3 is
// also what the fully corrected 8-operation form gives (the rest is the <= 1 ulp of the logPlus).
DYN_HD double log_normal_pdf(double x, const Emis& p) {
  const double z = (x - p.mean) * p.inv_stdev;
  const double t = z * z;
  return fma_(t, -0.5, p.neg_log_stdev) - HALF_LOG_2PI;
}

// Host only (model load).
inline Emis make_emis(double mean, double stdev, double log_stdev) {
  Emis e;
  e.mean = mean;
  e.inv_stdev = 1.0 / stdev;
  e.neg_log_stdev = -log_stdev;
  e.stdev = stdev;
  return e;
}

// ---------------------------------------------------------------------------------------------
// "Vector across cells" forms. A wave owns M independent lattice cells per lane; each stage of
// the arithmetic is written as a loop over the M cells so that the M dependency chains are
// interleaved in program order. A single wave then issues back-to-back fp64 instructions
// (measured: ~4.8 cycles/instr at 8 independent chains vs 8.6 cycles for one dependent chain,
// tools/ubench/issue_rate.hip) instead of M serial Horner chains.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Table-driven softplus. g(d) = log1p(exp(d)) satisfies g1 = s (the logistic function) and every
// higher derivative is a polynomial in s:  with u = s(1-s), w = 1-2s
//   g2 = u,  g3 = u w,  g4 = u (1-6u),  g5 = u w (1-12u),  |g6| <= 1/4      (gk = k-th derivative)
// A table of (g, s) at the nodes d_i = -i/128, i = 0..5120 (81 936 bytes, one copy in the LDS of
// each CU) plus the degree-5 Taylor polynomial about the NEAREST node (|r| <= 1/256), written with the
// common factor u r pulled out of the upper terms,
//   g(d_i + r) = g_i + r ( s + (u r) ( 1/2 + r ( w/6 + r ( (1-6u)/24 + r w (1-12u)/120 ) ) ) ),
// costs 13 fp64 operations after the lookup (the plain Horner form of rounds 1-2 took 16).
// |error| <= 1/4 (1/256)^6 / 720 = 1.2e-18 + the rounding of g_i and of the last FMA, i.e. the same
// <= ~1 ulp class as glibc's log1p(exp(d)). That class is not a luxury: where two neighbouring columns
// carry the same k-mer the traceback's comparison is a tie in exact arithmetic and the reference's
// choice rests on the last bits of its logPlus; a degree-4 polynomial (1.0e-15,
// 4 operations fewer, +6.6 % throughput) was measured and flipped such a decision in the fuzz tests.
// d <= -40 (incl. -inf and NaN, clamped) maps to the last node, which holds (0, 0): g = 0 exactly.
// ---------------------------------------------------------------------------------------------
struct SoftplusNode {
  double g, s;
};
constexpr int SP_STEPS = 128;                   // nodes per unit of d
constexpr int SP_RANGE = 40;                    // table covers d in [-40, 0]
constexpr int SP_NODES = SP_STEPS * SP_RANGE + 1;
// 1.5 * 2^52: adding it to a value in [0, 2^31) rounds that value to the nearest integer (ties to
// even, like rint) and leaves the integer in the low 32 bits of the sum's bit pattern.
constexpr double SP_MAGIC = 0x1.8p52;

// Host-side generator (long double arithmetic, rounded once to double).
inline void softplus_build_table(SoftplusNode* t) {
  for (int i = 0; i < SP_NODES; ++i) {
    const long double d = -(long double)i / SP_STEPS;
    const long double e = expl(d);
    t[i].g = (i == SP_NODES - 1) ? 0.0 : (double)log1pl(e);
    t[i].s = (i == SP_NODES - 1) ? 0.0 : (double)(e / (1.0L + e));
  }
}

// A constant the compiler must keep in ONE vector register pair. gfx9 VALU instructions read at most one
// scalar/literal operand, so fma(x, c1, c2) with two non-inline constants gets c2 through the accumulator of a
// v_fmac_f64 -- which hipcc re-materialises for EVERY cell (v_mov_b32 x 2 per cell and FMA: 14-28 extra
// instructions per row and lookup). Hidden behind an empty asm it is loaded once and used as the third
// operand of a v_fma_f64.
DYN_HD double vreg_const(double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(c));
#endif
  return c;
}

// The multiplier of such an FMA, kept in a scalar register pair: left as a literal it forces the VOP2 encoding
// (v_fmac_f64 with its destructive accumulator, hence a copy of the addend per cell); VOP3 has no literals on gfx9.
DYN_HD double sreg_const(double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+s"(c));
#endif
  return c;
}

DYN_HD int low_word(double m) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __double2loint(m);
#else
  unsigned long long b;
  __builtin_memcpy(&b, &m, 8);
  return (int)(unsigned)(b & 0xffffffffull);
#endif
}

// Two-phase form of the table-driven logPlus so that independent work (the next row's emission)
// can be placed between issuing the LDS lookups and consuming them:
//   SoftplusLookup<M> L;  log_plus_issue(x, y, L, tab);   ...independent code...   log_plus_finish(L, out);
template <int M>
struct SoftplusLookup {
  double hi[M], r[M], g0[M], s[M];
  double diff[M];  // x - y as rounded: only the certified logPlus (dp_math_strict.hpp) reads it again, for its fallback
};

// d = lo - hi = -|x - y| (a correctly rounded difference has the same magnitude either way round);
// node index k = rint(-128 d) by the magic-number addition, r = d + k/128 exactly. 5 fp64 operations
// + the max for hi (the reference form min, sub, clamp, mul, rndne, cvt, fma took 8).
template <int M>
DYN_HD void log_plus_issue(const double (&x)[M], const double (&y)[M], SoftplusLookup<M>& L,
                           const SoftplusNode* __restrict__ tab) {
#if defined(DYN_EXP_MAXPLUS)
  // FLOOR MEASUREMENT ONLY (tools/floor_bench.sh; never a product build): logPlus -> max, i.e. the sweeps in the max-plus
  // semiring: one operation instead of 19 and no table lookup, with the same loads, stores, ballots, band hand-overs and
  // traceback (forward and backward maxima agree, so every read passes its Z check). What is left is what the sweeps
  // cost WITHOUT the softplus: the bound DESIGN.md section 7 quotes.
#pragma unroll
  for (int j = 0; j < M; ++j) L.hi[j] = __builtin_fmax(x[j], y[j]);
  (void)tab;
  return;
#endif
  double d[M], m[M], kf[M];
  const double magic = vreg_const(SP_MAGIC), neg_steps = sreg_const(-(double)SP_STEPS);
#pragma unroll
  for (int j = 0; j < M; ++j) L.hi[j] = __builtin_fmax(x[j], y[j]);
#if defined(__HIP_DEVICE_COMPILE__)
  // Pins the max HERE, next to the additions that produced x and y. Left alone, hipcc sinks it to its use in
  // log_plus_finish; where a branch lies in between (backward sweep) it no longer knows the operands to be
  // canonical there and quiets both with a v_max_f64 v, v, v first: two extra fp64 instructions per cell.
#pragma unroll
  for (int j = 0; j < M; ++j) asm volatile("" : "+v"(L.hi[j]));
#endif
#pragma unroll
  for (int j = 0; j < M; ++j) L.diff[j] = x[j] - y[j];
#pragma unroll
  for (int j = 0; j < M; ++j) d[j] = __builtin_fmax(-__builtin_fabs(L.diff[j]), -(double)SP_RANGE);  // also NaN -> -40
#pragma unroll
  for (int j = 0; j < M; ++j) m[j] = fma_(d[j], neg_steps, magic);
#pragma unroll
  for (int j = 0; j < M; ++j) kf[j] = m[j] - magic;                          // exact
#pragma unroll
  for (int j = 0; j < M; ++j) L.r[j] = fma_(kf[j], 1.0 / SP_STEPS, d[j]);    // exact
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const SoftplusNode nd = tab[low_word(m[j])];
    L.g0[j] = nd.g;
    L.s[j] = nd.s;
  }
}

template <int M>
DYN_HD void log_plus_finish(const SoftplusLookup<M>& L, double (&out)[M]) {
#if defined(DYN_EXP_MAXPLUS)
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = L.hi[j];
  return;
#endif
  double u[M], w[M], p[M], q[M];
  const double c120 = vreg_const(1.0 / 120.0), c24 = vreg_const(1.0 / 24.0);
  const double m10 = sreg_const(-12.0 / 120.0), m4 = sreg_const(-0.25);
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = 1.0 - L.s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) u[j] = L.s[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = w[j] - L.s[j];                          // w = 1 - 2s (exact)
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(u[j], m10, c120);                   // (1 - 12u)/120
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = q[j] * w[j];                             // g5/(5! u)
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(u[j], m4, c24);                     // (1 - 6u)/24 = g4/(4! u)
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(q[j], L.r[j], p[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], w[j] * (1.0 / 6.0));
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], 0.5);
#pragma unroll
  for (int j = 0; j < M; ++j) u[j] = u[j] * L.r[j];
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], u[j], L.s[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = L.hi[j] + fma_(p[j], L.r[j], L.g0[j]);
}

// The same logPlus to degree 3: g_i + r (s + (u r)(1/2 + r w/6)), 9 operations instead of 14. Truncation
// u r^4 |1-6u| / 24 <= 1.2e-12 -- far too coarse where an integer decision hangs on the sum, and what train() wants
// for its backward sweep: the posterior chain (nt_kernels.hip) turns backward values into stay probabilities
// exp(... - bE), where 1e-12 is a relative error of 1e-12 per row, and Z = bE(0,0) moves by 1e-13 relative.
template <int M>
DYN_HD void log_plus_finish3(const SoftplusLookup<M>& L, double (&out)[M]) {
  double u[M], w[M], p[M];
  const double c6 = sreg_const(1.0 / 6.0);
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = 1.0 - L.s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) u[j] = L.s[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = w[j] - L.s[j];                          // w = 1 - 2s (exact)
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(w[j] * c6, L.r[j], 0.5);
#pragma unroll
  for (int j = 0; j < M; ++j) u[j] = u[j] * L.r[j];
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], u[j], L.s[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = L.hi[j] + fma_(p[j], L.r[j], L.g0[j]);
}

// softplus alone (tests): the same lookup and polynomial for d <= 0.
template <int M>
DYN_HD void softplus_table_vec(double (&d)[M], double (&g)[M], const SoftplusNode* __restrict__ tab) {
  double zero[M];
  SoftplusLookup<M> L;
#pragma unroll
  for (int j = 0; j < M; ++j) zero[j] = 0.0;
  log_plus_issue<M>(zero, d, L, tab);  // hi = 0 for d <= 0
  log_plus_finish<M>(L, g);
}
template <int M>
DYN_HD void softplus_table3_vec(double (&d)[M], double (&g)[M], const SoftplusNode* __restrict__ tab) {
  double zero[M];
  SoftplusLookup<M> L;
#pragma unroll
  for (int j = 0; j < M; ++j) zero[j] = 0.0;
  log_plus_issue<M>(zero, d, L, tab);
  log_plus_finish3<M>(L, g);
}

// Structure-of-arrays emission constants of the M cells of a lane.
template <int M>
struct EmisV {
  double mean[M], inv_stdev[M], neg_log_stdev[M];
  DYN_HD void set(int j, const Emis& e) {
    mean[j] = e.mean;
    inv_stdev[j] = e.inv_stdev;
    neg_log_stdev[j] = e.neg_log_stdev;
  }
};

template <int M>
DYN_HD void log_normal_pdf_vec(double x, const EmisV<M>& p, double (&out)[M]) {
  double z[M];
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = x - p.mean[j];
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = z[j] * p.inv_stdev[j];
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = z[j] * z[j];
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = fma_(z[j], -0.5, p.neg_log_stdev[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = z[j] - HALF_LOG_2PI;
}

}  // namespace dynmath
